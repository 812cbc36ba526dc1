#!/bin/bash
# builds scratch/lib_tf_ablate_<mask>.so = current sources with -DLDMSEG_TFUSE_ABLATE=<mask> (tfuse.hip only; compile-time phase
# ablation + wall-clock stamps of block 0).  usage: tools/build_tf_ablate.sh 0 1 2 12 ...
set -e
mkdir -p /tmp/probe scratch; cd "$(dirname "$0")/.."; C=latent-diffusion-segmentation_amd/csrc; B=$C/build
python __graft_entry__.py build | tail -1
for m in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DLDMSEG_TFUSE_ABLATE=$m -Iinclude -c $C/tfuse.hip -o /tmp/probe/tfuse_ablate.o \
    -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A9 "mlp_fused_kernelILb1" | grep -E "VGPRs Spill" | sed "s/^.*remark:/mask $m:/"
  hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/lib_tf_ablate_$m.so /tmp/probe/tfuse_ablate.o $(ls $B/*.o | grep -v tfuse.o)
done
ls scratch/lib_tf_ablate_*.so
