"""Join a rocprofv3 kernel trace of tools/trace_fwd.py (5 forwards) with the per-launch labels in gpurun_out/layers.csv
(tools/prof_layers.py, same B / dtype): true kernel durations per layer shape, without the per-launch event brackets.
usage: trace_join.py <kernel_trace.csv> <layers.csv> [forwards_in_trace=5]"""
import csv, sys, collections
trace, layers = sys.argv[1], sys.argv[2]
NF = int(sys.argv[3]) if len(sys.argv) > 3 else 5
rows = list(csv.DictReader(open(trace)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
FAM0 = ("igemm_kernel", "mlp_fused_kernel", "proj_ln_qkv_kernel", "conv_out_tail_kernel")      # the launches ldmseg_profile_dump labels as family 0, in order
ig = [r for r in rows if any(k in r["Kernel_Name"] for k in FAM0)]
lab = [ln.split(",") for ln in open(layers).read().splitlines()[1:]]
lab0 = [l for l in lab if l[0] == "0"]
per_fwd = len(ig) // NF
assert per_fwd * NF == len(ig), (len(ig), NF)
R = len(lab0) // per_fwd
assert R * per_fwd == len(lab0), (len(lab0), per_fwd)
agg = collections.OrderedDict()
for f in range(2, NF):                              # skip the first two forwards (warm-up)
    for i in range(per_fwd):
        r = ig[f * per_fwd + i]; l = lab0[i]
        name = r["Kernel_Name"]
        name = (name[name.find("<"):name.find(">") + 1] if "igemm_kernel" in name else next(k for k in FAM0 if k in name)).replace("__hip_bfloat16", "bf16")
        a = agg.setdefault((l[1], name), [0, 0.0, 0.0]); a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3; a[2] += float(l[3])
n = NF - 2
tot = sum(a[1] for a in agg.values()) / n; fl = sum(a[2] for a in agg.values()) / n
print(f"GEMM-family kernels (igemm + fused transformer kernels + conv_out tail): {per_fwd} launches/forward, {tot:.1f} us/forward, {fl/tot/1e6:.1f} TFLOP/s")
oth = collections.Counter(); 
t0 = int(ig[2 * per_fwd]["Start_Timestamp"])
for r in rows:
    if int(r["Start_Timestamp"]) >= t0 and not any(k in r["Kernel_Name"] for k in FAM0):
        oth[r["Kernel_Name"][:60]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3 / n
for k, v in oth.most_common(14): print(f"   other {k:60s} {v:8.1f} us/forward")
last = [r for r in rows if int(r["Start_Timestamp"]) >= t0]
print(f"wall per forward (first launch of forward 3 to last end): {(int(last[-1]['End_Timestamp']) - t0) * 1e-3 / n:.1f} us; sum of kernel durations {sum((int(r['End_Timestamp'])-int(r['Start_Timestamp'])) for r in last)*1e-3/n:.1f} us")
for (label, name), (c, us, f) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{label:62s} {name:34s} n={c/n:3.0f} us/fwd={us/n:7.1f} us={us/c:7.1f} TF={f/us/1e6:7.1f}")
