#!/bin/bash
# builds scratch/lib_tp_ablate_<mask>.so = current sources with -DLDMSEG_TPROJ_ABLATE=<mask> (tproj.hip only; compile-time phase
# ablation: 1 no weight DMA, 2 no global stores, 4 no MFMA, 8 no fragment reads).  usage: tools/build_tp_ablate.sh 1 2 4 12 ...
set -e
mkdir -p /tmp/probe scratch; cd "$(dirname "$0")/.."; C=latent-diffusion-segmentation_amd/csrc; B=$C/build
python __graft_entry__.py build | tail -1
for m in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DLDMSEG_TPROJ_ABLATE=$m -Iinclude -c $C/tproj.hip -o /tmp/probe/tproj_ablate.o \
    -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A9 "proj_ln_qkv_kernel" | grep -E "VGPRs Spill" | sed "s/^.*remark:/mask $m:/"
  hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/lib_tp_ablate_$m.so /tmp/probe/tproj_ablate.o $(ls $B/*.o | grep -v tproj.o)
done
ls scratch/lib_tp_ablate_*.so
