#!/bin/bash
# on the GPU box: per-layer igemm kernel durations from a rocprofv3 kernel trace -> gpurun_out/trace_layers.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace; mkdir -p $O; DT=${1:-bf16}; B=${2:-8}
cd $R; python tools/prof_layers.py $DT $B > $O/layers_top.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o t -- python $R/tools/trace_fwd.py $DT $B > $O/kt.log 2>&1
python $R/tools/trace_join.py $(find $O/kt -name "*kernel_trace.csv" | head -1) $R/gpurun_out/layers.csv > $R/gpurun_out/trace_layers.txt 2>&1
find $O -name "*.csv" -size +30M -delete
