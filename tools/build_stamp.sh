#!/bin/bash
# builds tools/ab/lib_stamp.so = current sources with -DLDMSEG_IGEMM_STAMP (igemm only: s_memtime / wall-clock stamps of workgroup phases,
# no ablation branches); select it with LDMSEG_HIP_LIB=tools/ab/lib_stamp.so (tools/stamps2.py, tools/launch_boundary.py)
set -e
mkdir -p /tmp/probe; cd "$(dirname "$0")/.."; mkdir -p tools/ab; C=latent-diffusion-segmentation_amd/csrc; B=$C/build
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DLDMSEG_IGEMM_STAMP -Iinclude -c $C/igemm.hip -o /tmp/probe/igemm_stamp.o
OBJS=$(ls $B/*.o | grep -v "/igemm.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/lib_stamp.so /tmp/probe/igemm_stamp.o $OBJS
ls -la tools/ab/lib_stamp.so
