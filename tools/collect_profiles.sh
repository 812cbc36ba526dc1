#!/bin/bash
# Run on the GPU box from the repo root (gpurun): every measurement profiles/ holds for one round, from ONE box.
#   bash tools/collect_profiles.sh r02     -> gpurun_out/prof_r02/...;  then locally: python tools/import_profiles.py r02
set -x
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-images --no-extras --no-torch-reference --profile-steps 0"
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
# BASELINE configs[4] as lines of their own (20 steps): bf16 attention and the fp8 attention path on the 16384-token level
python bench.py --batch 4 --latent 128 --steps 20 --warmup 2 --no-cpu-baseline --no-extras --no-images > $O/bench_config4_b4_l128_bf16.json 2>/dev/null
python bench.py --batch 4 --latent 128 --steps 20 --warmup 2 --attention-fp8 16384 --no-cpu-baseline --no-extras --no-images > $O/bench_config4_b4_l128_fp8attn.json 2>/dev/null
# round 4: same-box comparisons
timeout 1500 python tools/yardstick.py --compile --out $O/yardstick.txt > $O/yardstick.log 2>&1
# QUICK=1: skip the measurements of kernels that did not change since the round's previous collection (their files stay as they are)
[ -z "$QUICK" ] && python tools/ff_bench.py 32768 65536 > $O/ff_bench.txt 2>&1
python tools/tin_bench.py 32768 65536 2>&1 | grep -v amdgpu.ids > $O/tin_bench.txt
[ -z "$QUICK" ] && { python tools/attn8_bench.py > $O/attn8_bench.txt 2>&1; python tools/attn8_bench.py 4 4096 320 >> $O/attn8_bench.txt 2>&1; }
[ -z "$QUICK" ] && python tools/attn8_acc.py 2>&1 | grep "N=" > $O/attn8_acc.txt
python tools/attn4_bench.py 2>&1 | grep -v amdgpu.ids > $O/attn4_bench.txt
python tools/ab_forward.py "12=0,14=0,16=0,2=7" "12=3,14=0,16=0,2=7" "12=3,14=1,16=0,2=7" "12=3,14=1,16=3,2=7" "12=3,14=1,16=3,2=0" --rounds 3 > $O/ab_knobs.txt 2>&1
# same-box A/B against the previous round's kernels (tools/ab/lib_r05.so = the library of the round-5 HEAD, built from that tree)
[ -f tools/ab/lib_r05.so ] && { LDMSEG_HIP_LIB=tools/ab/lib_r05.so python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras --no-images --no-torch-reference --profile-steps 0 > $O/bench_r05_lib_same_box.json 2>/dev/null; python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras --no-images --no-torch-reference --profile-steps 0 > $O/bench_this_lib_same_box.json 2>/dev/null; }
# round 6: K slices finished inside the launch (debug key 23: off | 256-row tiles (shipped) | every tile form that has the instantiation),
# per launch shape and in whole forwards
python tools/cf_bench.py 2>&1 | grep -v amdgpu.ids > $O/cf_bench_per_shape.txt
python tools/ab_forward.py "23=0" "23=1" "23=9" --rounds 3 2>&1 | grep -v amdgpu.ids > $O/ab_round6_knobs.txt
[ -z "$QUICK" ] && python tools/cf_tune.py 2>&1 | grep -v amdgpu.ids > $O/cf_tune_32x32_candidates.txt
python tools/fam.py bf16x3 fp32 bf16 2>&1 | grep -v amdgpu.ids > $O/modes_ms_per_forward.txt
# round 5: the three data-flow restructurings switched on one by one in one process (19: conv2 + conv_shortcut in one launch, 20: ff.net.2 +
# proj_out chained, 21: upsampler convs as four 2x2 phase convs, 22: the 320-channel transformers' GroupNorm folded into the fused entry),
# and the resnet-tail / upsampler launches against what they replace
python tools/ab_forward.py "19=0,20=0,21=0,22=0" "19=1,20=0,21=0,22=0" "19=1,20=1,21=0,22=0" "19=1,20=1,21=1,22=0" "19=1,20=1,21=1,22=1" --rounds 3 2>&1 | grep -v amdgpu.ids > $O/ab_round5_knobs.txt
[ -z "$QUICK" ] && python tools/xt_bench.py 2>&1 | grep -v amdgpu.ids > $O/xt_bench.txt
python tools/acc_round5.py 2>&1 | grep -v amdgpu.ids > $O/accuracy_round5.txt
[ -z "$QUICK" ] && ( for i in 12 29 44; do for v in 0 1; do UP4=$v python tools/kbench.py igemm1 $i 2>/dev/null | grep "M=" | sed "s/^/up4=$v /"; done; done ) > $O/kbench_up4.txt
[ -f tools/ab/lib_stamp.so ] && { LDMSEG_HIP_LIB=tools/ab/lib_stamp.so LDMSEG_OP_TIMING_NHWC=1 python tools/launch_boundary.py 2>&1 | grep -v amdgpu.ids > $O/launch_boundary_raw.txt; }
python tools/kbench.py gn > $O/kbench_groupnorm.txt 2>&1
python tools/kbench.py attn > $O/kbench_attention.txt 2>&1
[ -z "$QUICK" ] && for u in mfma_lds mfma_lds2 buf_lds valu_trans copy_floor launch_floor barrier_cost mx_probe; do [ -x tools/ubench/bin/$u ] && timeout 300 tools/ubench/bin/$u > $O/ubench_$u.txt 2>&1; done
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +8M -delete
du -sh $O
