#!/bin/bash
# run on the GPU box from the repo root; writes everything under gpurun_out/prof_final
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_final; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-images --profile-steps 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r1 -- $BENCH > $O/bench_under_rocprof.json 2> $O/stats.err
BENCH3="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-images --profile-steps 0"
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -- $BENCH3 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -- $BENCH3 > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_mfma -- $BENCH3 > $O/pmc_mfma.log 2>&1
for d in pmc_fetch pmc_write pmc_mfma; do python $R/tools/pmc_query.py $O/$d igemm_kernel > $O/$d.txt 2>&1; python $R/tools/pmc_query.py $O/$d attention_kernel >> $O/$d.txt 2>&1; done
cd $R; python tools/prof_layers.py bf16 8 > $O/layers_top.txt 2>&1; cp gpurun_out/layers.csv $O/per_launch_events.csv
python bench.py > $O/bench.json 2> $O/bench.err
find $O -name "*.db" -size +20M -delete
du -sh $O
