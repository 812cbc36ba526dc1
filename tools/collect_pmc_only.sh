#!/bin/bash
# The part of tools/collect_profiles.sh that is tied to the hash of the kernel sources (bench.py reports roofline.traffic only while
# csrc still hashes to the value the PMC passes were taken on): the rocprofv3 stats run, the three --pmc passes, the default bench line,
# the same-box A/B against the previous round's library and the per-mode family times.   bash tools/collect_pmc_only.sh r06
set -x
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-images --no-extras --no-torch-reference --profile-steps 0"
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_mfma
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- $BENCH > $O/bench_under_rocprof.json 2> $O/stats.err
BENCH3="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-images --no-extras --no-torch-reference --profile-steps 0"
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -- $BENCH3 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -- $BENCH3 > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_mfma -- $BENCH3 > $O/pmc_mfma.log 2>&1
for d in pmc_fetch pmc_write pmc_mfma; do
  python $R/tools/pmc_query.py $O/$d igemm_kernel > $O/$d.txt 2>&1
  for k in mlp_fused_kernel proj_ln_qkv_kernel conv_out_tail_kernel attn3_kernel attn4_kernel; do python $R/tools/pmc_query.py $O/$d $k >> $O/$d.txt 2>&1; done
done
cd $R
bash tools/trace_layers.sh bf16 8; cp gpurun_out/trace_layers.txt $O/trace_layers_b8_l64_bf16.txt; cp gpurun_out/layers.csv $O/per_launch_events.csv
python bench.py > $O/bench_default.json 2> $O/bench_default.err
[ -f tools/ab/lib_r05.so ] && { LDMSEG_HIP_LIB=tools/ab/lib_r05.so python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras --no-images --no-torch-reference --profile-steps 0 > $O/bench_r05_lib_same_box.json 2>/dev/null; python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras --no-images --no-torch-reference --profile-steps 0 > $O/bench_this_lib_same_box.json 2>/dev/null; }
python tools/fam.py bf16x3 fp32 bf16 2>&1 | grep -v amdgpu.ids > $O/modes_ms_per_forward.txt
python tools/acc_round5.py 2>&1 | grep -v amdgpu.ids > $O/accuracy_round5.txt
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +8M -delete
du -sh $O
