#!/bin/bash
# PMC passes over ONE igemm launch shape (index into tests/test_igemm_shapes_gpu.py SHAPES; "ln" = folded-LayerNorm launch)
# usage on the GPU box: bash tools/igemm_pmc.sh <out-name> <shape index> [ln]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-igemm_pmc}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
CMD="python $R/tools/kbench.py igemm1 $2 $3"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES -d $O/p1 -- $CMD > $O/p1.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/p2 -- $CMD > $O/p2.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE -d $O/p3 -- $CMD > $O/p3.log 2>&1
for d in p1 p2 p3; do python $R/tools/pmc_query.py $O/$d igemm_kernel > $O/$d.txt 2>&1; done
cat $O/p1.txt $O/p2.txt $O/p3.txt
find $O -name "*.db" -size +5M -delete
