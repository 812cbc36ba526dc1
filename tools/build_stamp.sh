#!/bin/bash
# builds scratch/lib_stamp.so = current sources with -DLDMSEG_IGEMM_STAMP (igemm only: s_memtime stamps of workgroup phases,
# no ablation branches); select it with LDMSEG_HIP_LIB=scratch/lib_stamp.so (tools/stamps2.py)
set -e
mkdir -p /tmp/probe scratch; cd "$(dirname "$0")/.."; C=latent-diffusion-segmentation_amd/csrc; B=$C/build
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DLDMSEG_IGEMM_STAMP -Iinclude -c $C/igemm.hip -o /tmp/probe/igemm_stamp.o
OBJS=$(ls $B/*.o | grep -v "/igemm.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/lib_stamp.so /tmp/probe/igemm_stamp.o $OBJS
ls -la scratch/lib_stamp.so
