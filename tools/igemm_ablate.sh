#!/bin/bash
# phase ablation of one launch shape with the -DLDMSEG_IGEMM_ABLATE build (scratch/lib_ablate.so): bash tools/igemm_ablate.sh <shape index> [ln]
# bits: 1 no DMA, 4 no MFMA, 8 no LDS reads, 16 no epilogue
R=$GRAFT_REPO_ROOT
for d in 0 16 20 28 24 29 21; do
  echo -n "dbg=$d: "; LDMSEG_HIP_LIB=$R/scratch/lib_ablate.so DBG=$(( (61<<8) | d )) python $R/tools/kbench.py igemm1 $1 $2 2>&1 | grep "M=" | cut -c60-160
done
