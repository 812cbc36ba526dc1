#!/usr/bin/env python
"""Copy the summaries of gpurun_out/prof_<tag>/ (tools/collect_profiles.sh) into profiles/<tag>_* and derive
<tag>_pmc_traffic.json (HBM bytes per igemm launch, the figure bench.py reports as roofline.traffic)."""
import glob, json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
src, dst = os.path.join(ROOT, "gpurun_out", f"prof_{tag}"), os.path.join(ROOT, "profiles")


def cp(a, b):
    a = os.path.join(src, a)
    if os.path.exists(a):
        shutil.copy(a, os.path.join(dst, f"{tag}_{b}"))
        print("->", f"{tag}_{b}")


for f in glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, f"{tag}_kernel_stats_bench_b8_l64_bf16.csv"))
for f in glob.glob(os.path.join(src, "stats", "**", "*agent_info.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, f"{tag}_agent_info.csv"))
cp("bench_under_rocprof.json", "bench_under_rocprof.json")
cp("bench_default.json", "bench_default.json")
cp("per_launch_events.csv", "per_launch_events_unet_forward_b8_l64_bf16.csv")
cp("trace_layers_b8_l64_bf16.txt", "kernel_trace_per_layer_b8_l64_bf16.txt")
for n in ("pmc_fetch", "pmc_write", "pmc_mfma"):
    cp(n + ".txt", n + ".txt")
for u in ("mfma_lds", "mfma_lds2", "buf_lds", "valu_trans", "copy_floor", "launch_floor", "barrier_cost", "mx_probe"):
    cp(f"ubench_{u}.txt", f"ubench_{u}.txt")
cp("pytest_gpu.log", "pytest_gpu.log")
cp("bench_config4_b4_l128_bf16.json", "bench_config4_b4_l128_bf16.json")
cp("bench_config4_b4_l128_fp8attn.json", "bench_config4_b4_l128_fp8attn.json")
cp("kbench_groupnorm.txt", "kbench_groupnorm.txt")
cp("kbench_attention.txt", "kbench_attention.txt")
for n in ("yardstick.txt", "ff_bench.txt", "tin_bench.txt", "attn8_bench.txt", "attn8_acc.txt", "attn4_bench.txt", "ab_forward_vs_r03.txt", "bench_r03_lib_same_box.json", "ab_knobs.txt",
          "bench_r04_lib_same_box.json", "bench_r05_lib_same_box.json", "bench_this_lib_same_box.json", "cf_bench_per_shape.txt", "ab_round6_knobs.txt",
          "cf_tune_32x32_candidates.txt", "launch_boundary_raw.txt", "modes_ms_per_forward.txt", "igemm_stamps.txt",
          "ab_round5_knobs.txt", "xt_bench.txt", "kbench_up4.txt", "accuracy_round5.txt"):
    cp(n, n)


def per_dispatch(path, counter, kernel):
    """sum over instantiations of (dispatches x per-dispatch total) for `counter` on kernels whose name contains `kernel`"""
    names, vals, tot_n, tot_v = [], [], 0, 0.0
    for ln in open(path):
        m = re.match(r"dispatches (\S+) (\d+) avg_ns", ln)
        if m:
            names.append((m.group(1), int(m.group(2))))
        m = re.match(rf"{counter}\s+per-dispatch total\s+([\d.]+)", ln)
        if m:
            vals.append(float(m.group(1)))
    ig = [(n, c) for n, c in names if any(k in n for k in kernel.split("|"))]
    for (n, c), v in zip(ig, vals[:len(ig)]):
        tot_n += c
        tot_v += c * v
    return tot_n, tot_v


try:
    n_f, kib_f = per_dispatch(os.path.join(dst, f"{tag}_pmc_fetch.txt"), "FETCH_SIZE", "igemm_kernel|mlp_fused_kernel|proj_ln_qkv_kernel|conv_out_tail_kernel")
    n_w, kib_w = per_dispatch(os.path.join(dst, f"{tag}_pmc_write.txt"), "WRITE_SIZE", "igemm_kernel|mlp_fused_kernel|proj_ln_qkv_kernel|conv_out_tail_kernel")
    import bench
    # the hash of the sources the passes ran on = the one the collection's own bench line reports (the local tree may have moved on)
    try:
        sha = json.loads(open(os.path.join(dst, f"{tag}_bench_default.json")).read().strip().splitlines()[-1])["roofline"]["csrc_sha"]
    except Exception:  # noqa: BLE001
        sha = bench.csrc_hash()
    out = {"kernel": "igemm_kernel (all instantiations) + mlp_fused_kernel + proj_ln_qkv_kernel + conv_out_tail_kernel", "launches": n_f,
           "fetch_bytes_per_launch": 2 * 1024 * kib_f / n_f, "write_bytes_per_launch": 1024 * kib_w / n_w,
           "hbm_bytes_per_launch": 2 * 1024 * kib_f / n_f + 1024 * kib_w / n_w, "csrc_sha": sha,
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 3 --warmup 1 "
                     "--no-cpu-baseline --no-images --no-extras --profile-steps 0`; FETCH_SIZE (KiB) doubled per MI355X_MICROARCH.md "
                     "(HBM section: gfx950 tallies 128-B requests at 64 B); WRITE_SIZE uncalibrated; Infinity-Cache hits are counted",
           "source": [f"profiles/{tag}_pmc_fetch.txt", f"profiles/{tag}_pmc_write.txt"]}
    json.dump(out, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))
except Exception as e:  # noqa: BLE001
    print("pmc_traffic not derived:", e)
