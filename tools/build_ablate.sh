#!/bin/bash
# builds scratch/lib_ablate.so = current sources with -DLDMSEG_IGEMM_ABLATE (igemm only)
set -e
mkdir -p /tmp/probe scratch; cd "$(dirname "$0")/.."; C=latent-diffusion-segmentation_amd/csrc; B=$C/build
python __graft_entry__.py build | tail -1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DLDMSEG_IGEMM_ABLATE -Iinclude -c $C/igemm.hip -o /tmp/probe/igemm_ablate.o
hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/lib_ablate.so /tmp/probe/igemm_ablate.o $B/norm.o $B/attention.o $B/attention3.o $B/attention_fp8.o $B/misc.o $B/postproc.o $B/sched.o $B/engine.o $B/ops_api.o
ls -la scratch/lib_ablate.so
