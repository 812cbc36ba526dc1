#!/usr/bin/env python
"""Same-box yardstick (measurement only - nothing here is linked into or called by the product).

What the reference executes this path with is PyTorch on the GPU (tools/main_ldm.py:146-192,
trainers_ldm_cond.py:1140-1141).  On an MI355X that is torch-ROCm: MIOpen / hipBLASLt / its SDPA and norm kernels.
This script puts that stack's time next to this library's, in one process on one box:

  (i)  per launch shape of the B=8, L=64 bf16 forward: F.conv2d / F.linear / F.scaled_dot_product_attention /
       F.group_norm in bf16 (best of NCHW and channels_last for convs; weights rotated over copies so that they come from
       HBM as in a forward) against this library's kernel time for the same launch inside a real forward (HIP events of
       ldmseg_profile_dump);
  (ii) the whole forward: oracle/unet.py moved to the GPU in bf16 and in fp32, eager, 3 timed forwards, against
       ldmseg_unet_forward.

    python tools/yardstick.py [--batch 8] [--latent 64] [--out gpurun_out/yardstick.txt] [--json]
"""
import argparse
import collections
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "latent-diffusion-segmentation_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def ev_time(fn, iters=10, warm=3):
    for _ in range(warm):
        fn(0)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / iters      # us


def ncopies(nbytes):
    return max(1, min(16, (256 << 20) // max(1, nbytes)))


def torch_gemm_us(M, N, K, taps, stride, up, epi, B, L):
    """torch-ROCm bf16 time of one igemm launch shape.  taps=9: conv3x3 on [B, Cin, H, W]; taps=1: linear on [M, K]."""
    dev, dt = "cuda", torch.bfloat16
    if taps == 9:
        cin = K // 9
        Ho = int(round((M // B) ** 0.5))
        Hin = Ho * stride // (2 if up else 1) if stride == 1 else Ho * 2
        if up:
            Hin = Ho // 2
        n = ncopies(N * K * 2)
        ws = [torch.randn(N, cin, 3, 3, device=dev, dtype=dt) * 0.02 for _ in range(n)]
        bias = torch.randn(N, device=dev, dtype=dt)
        best = None
        for fmt in (torch.contiguous_format, torch.channels_last):
            x = torch.randn(B, cin, Hin, Hin, device=dev, dtype=dt).contiguous(memory_format=fmt)
            wf = [w.contiguous(memory_format=fmt) for w in ws]

            def run(i):
                xx = F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x
                return F.conv2d(xx, wf[i % n], bias, stride=stride, padding=1)
            try:
                t = ev_time(run)
            except RuntimeError:
                continue
            best = t if best is None else min(best, t)
        return best
    n = ncopies(N * K * 2)
    ws = [torch.randn(N, K, device=dev, dtype=dt) * 0.02 for _ in range(n)]
    bias = torch.randn(N, device=dev, dtype=dt)
    x = torch.randn(M, K, device=dev, dtype=dt)
    if epi == 1:            # GEGLU: N counts value + gate columns
        def run(i):
            a, g = F.linear(x, ws[i % n], bias).chunk(2, dim=-1)
            return a * F.gelu(g)
    else:
        def run(i):
            return F.linear(x, ws[i % n], bias)
    return ev_time(run)


def torch_fused_us(kind, M, C=320):
    """torch-ROCm bf16 time of what one fused row-local launch covers: 'ff' = LayerNorm -> GEGLU -> ff.net.2 (+h) -> proj_out (+x),
    'in' = proj_in -> LayerNorm -> q|k|v (three Linears as one [3C, C] GEMM)."""
    dev, dt = "cuda", torch.bfloat16
    h, x = torch.randn(M, C, device=dev, dtype=dt), torch.randn(M, C, device=dev, dtype=dt)
    g, b = torch.ones(C, device=dev, dtype=dt), torch.zeros(C, device=dev, dtype=dt)
    n = 8
    if kind == "ff":
        w1 = [torch.randn(8 * C, C, device=dev, dtype=dt) * 0.02 for _ in range(n)]
        w2 = [torch.randn(C, 4 * C, device=dev, dtype=dt) * 0.02 for _ in range(n)]
        wp = [torch.randn(C, C, device=dev, dtype=dt) * 0.02 for _ in range(n)]
        b1, b2 = torch.randn(8 * C, device=dev, dtype=dt), torch.randn(C, device=dev, dtype=dt)

        def run(i):
            a, gate = F.linear(F.layer_norm(h, (C,), g, b), w1[i % n], b1).chunk(2, dim=-1)
            h2 = F.linear(a * F.gelu(gate), w2[i % n], b2) + h
            return F.linear(h2, wp[i % n], b2) + x
        return ev_time(run)
    wp = [torch.randn(C, C, device=dev, dtype=dt) * 0.02 for _ in range(n)]
    wq = [torch.randn(3 * C, C, device=dev, dtype=dt) * 0.02 for _ in range(n)]
    bp = torch.randn(C, device=dev, dtype=dt)

    def run(i):
        hh = F.linear(x, wp[i % n], bp)
        return hh, F.linear(F.layer_norm(hh, (C,), g, b), wq[i % n])
    return ev_time(run)


def torch_attn_us(B, N, C):
    d = C // 8
    q, k, v = (torch.randn(B, 8, N, d, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    return ev_time(lambda i: F.scaled_dot_product_attention(q, k, v))


def torch_gn_us(B, C, HW):
    H = int(round(HW ** 0.5))
    g, b = torch.ones(C, device="cuda", dtype=torch.bfloat16), torch.zeros(C, device="cuda", dtype=torch.bfloat16)
    best = None
    for fmt in (torch.contiguous_format, torch.channels_last):
        x = torch.randn(B, C, H, H, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=fmt)
        t = ev_time(lambda i: F.silu(F.group_norm(x, 32, g, b, 1e-5)))
        best = t if best is None else min(best, t)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "yardstick.txt"))
    ap.add_argument("--compile", action="store_true", help="add a torch.compile(mode='max-autotune') leg to the whole forward (minutes)")
    ap.add_argument("--fast", action="store_true", help="fp32: the eager leg only (bench.py's subprocess call; the tuned legs run in bf16)")
    ap.add_argument("--whole-only", action="store_true", help="only part (ii), print one JSON object (bench.py's torch_rocm_reference)")
    args = ap.parse_args()
    B, L = args.batch, args.latent
    from ldmseg_amd import _lib, weights
    from ldmseg_amd.models import UNet
    from oracle import unet as o_unet
    usd = weights.generate(weights.unet_schema(12, False), seed=0)
    res = whole_forward(usd, B, L, UNet, o_unet, compile_leg=args.compile, fast=args.fast)
    if args.whole_only:
        print(json.dumps(res))
        return
    lines = []
    out = lambda s: (lines.append(s), print(s, flush=True))
    out(f"# yardstick: torch {torch.__version__} on {torch.cuda.get_device_name(0)}, B={B}, L={L}, bf16")
    out("## (ii) whole UNet forward, ms (eager, 3 timed forwards)")
    for k, v in res.items():
        out(f"  {k:34s} {v}")
    # ---- (i) per launch shape (MIOpen in find mode: the fastest solver per conv shape, round 5)
    torch.backends.cudnn.benchmark = True
    u = UNet(usd, 12, "cuda:0", "bf16")
    x = torch.randn(B, 12, L, L, device="cuda")
    for _ in range(2):
        u(x, 500)
    torch.cuda.synchronize()
    lib = _lib.lib()
    lib.ldmseg_profile_reset()
    lib.ldmseg_profile_enable(1)
    R = 3
    for _ in range(R):
        u(x, 500)
    torch.cuda.synchronize()
    lib.ldmseg_profile_enable(0)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    csv = args.out + ".layers.csv"
    lib.ldmseg_profile_dump(csv.encode())
    agg = collections.OrderedDict()
    for line in open(csv).read().splitlines()[1:]:
        fam, label, ms, fl = line.split(",")
        a = agg.setdefault((int(fam), label), [0, 0.0, 0.0])
        a[0] += 1
        a[1] += float(ms)
        a[2] += float(fl)
    lib.ldmseg_profile_reset()
    del u
    torch.cuda.empty_cache()
    out("## (i) per launch shape: this library (HIP events inside a forward) vs torch-ROCm bf16 (standalone, rotated weights)")
    out(f"{'launch':66s} {'n/fwd':>5s} {'ours us':>9s} {'torch us':>9s} {'torch/ours':>10s}  verdict")
    tot_o = tot_t = 0.0
    behind = []
    for (fam, label), (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        ours = 1e3 * ms / n
        tt = None
        if fam == 0:
            m = re.match(r"M=(\d+) N=(\d+) K=(\d+) taps=(\d+) stride=(\d+) up=(\d+) epi=(\d+)", label)
            if m:
                M, N, K, taps, stride, up, epi = map(int, m.groups())
                if epi in (0, 1) and N >= 32:
                    tt = torch_gemm_us(M, N, K, taps, stride, up, epi, B, L)
            else:
                m = re.match(r"M=(\d+) (mlp_fused|proj_ln_qkv) C=(\d+)", label)
                if m:
                    tt = torch_fused_us("ff" if m.group(2) == "mlp_fused" else "in", int(m.group(1)), int(m.group(3)))
        elif fam == 1:
            m = re.match(r"N=(\d+) C=(\d+)", label)
            if m:
                tt = torch_attn_us(B, int(m.group(1)), int(m.group(2)))
        elif fam == 2:
            m = re.match(r"HW=(\d+) C=(\d+)", label)
            if m:
                tt = torch_gn_us(B, int(m.group(2)), int(m.group(1)))
        if tt is None:
            continue
        per = n / R
        tot_o += ours * per
        tot_t += tt * per
        ratio = tt / ours
        verdict = "ahead" if ratio > 1.10 else ("behind" if ratio < 0.90 else "level")
        if ratio < 0.90:
            behind.append((label, ours, tt, per))
        out(f"{('fam%d ' % fam + label)[:66]:66s} {per:5.0f} {ours:9.1f} {tt:9.1f} {ratio:10.2f}  {verdict}")
    out(f"## sum over the compared launches per forward: ours {tot_o / 1e3:.3f} ms, torch-ROCm {tot_t / 1e3:.3f} ms "
        f"(ratio {tot_t / max(tot_o, 1e-9):.2f})")
    out("## shapes where the vendor stack is >10 % faster (work-list):")
    for label, ours, tt, per in behind:
        out(f"   {label}: ours {ours:.1f} us, torch {tt:.1f} us, x{per:.0f} per forward = {(ours - tt) * per:.0f} us per forward")
    if not behind:
        out("   none")
    with open(args.out, "w") as fh:
        fh.write("\n".join(lines) + "\n")


def whole_forward(usd, B, L, UNet, o_unet, compile_leg=False, fast=False):
    """ms per UNet forward: this library (bf16, fp32) and the oracle graph run by torch-ROCm eager on the same GPU."""
    dev = "cuda:0"
    x = torch.randn(B, 12, L, L, generator=torch.Generator().manual_seed(0))
    res = {}

    def t_ms(fn, n=3):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return round(a.elapsed_time(b) / n, 3)
    # diffusers 0.16.1 routes attention through F.scaled_dot_product_attention on torch >= 2.0 (AttnProcessor2_0): the
    # vendor stack is timed with it; the oracle's explicit softmax(QK^T)V (what the parity tests compare with) is timed too
    math_attention = o_unet.attention

    def sdpa_attention(sd, p, xx, ctx=None):
        Bq, N, C = xx.shape
        d = C // o_unet.HEADS
        q = F.linear(xx, sd[p + "to_q.weight"]).view(Bq, N, o_unet.HEADS, d).transpose(1, 2)
        k = F.linear(xx, sd[p + "to_k.weight"]).view(Bq, N, o_unet.HEADS, d).transpose(1, 2)
        v = F.linear(xx, sd[p + "to_v.weight"]).view(Bq, N, o_unet.HEADS, d).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(Bq, N, C)
        return F.linear(o, sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])
    for mode, tdt in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
        u = UNet(usd, 12, dev, mode)
        xd = x.to(dev)
        res[f"ldmseg_hip_{mode}_ms"] = t_ms(lambda: u(xd, 499))
        y_ours = u(xd, 499).sample.float().cpu()
        del u
        torch.cuda.empty_cache()
        sd = {k: v.to(dev, tdt) for k, v in usd.items()}
        xt = x.to(dev, tdt)
        tt = torch.tensor(499, device=dev)
        legs = {}
        with torch.no_grad():
            try:
                o_unet.attention = sdpa_attention
                torch.backends.cudnn.benchmark = False
                res[f"torch_rocm_eager_{mode}_ms"] = legs["eager NCHW, SDPA"] = t_ms(lambda: o_unet.unet_forward(sd, xt, tt))
                y_t = o_unet.unet_forward(sd, xt, tt).float().cpu()
                res[f"rel_l2_ours_vs_torch_{mode}"] = float((y_ours - y_t).norm() / y_t.norm())
                o_unet.attention = math_attention
                res[f"torch_rocm_eager_{mode}_math_attention_ms"] = t_ms(lambda: o_unet.unet_forward(sd, xt, tt))
                o_unet.attention = sdpa_attention
                if fast and mode == "fp32":        # bench.py's call: the tuned legs only for the headline dtype
                    raise StopIteration
                # round 5: the comparator at its best, not at its defaults.
                # (a) MIOpen find mode: every conv shape benchmarked once, the fastest solver kept
                torch.backends.cudnn.benchmark = True
                for _ in range(2):
                    o_unet.unet_forward(sd, xt, tt)
                legs["eager NCHW, SDPA, cudnn.benchmark (MIOpen find)"] = t_ms(lambda: o_unet.unet_forward(sd, xt, tt))
                # (b) channels_last activations and conv weights (NHWC kernels), find mode on
                sd_cl = {k: (v.contiguous(memory_format=torch.channels_last) if v.dim() == 4 else v) for k, v in sd.items()}
                xt_cl = xt.contiguous(memory_format=torch.channels_last)
                for _ in range(2):
                    o_unet.unet_forward(sd_cl, xt_cl, tt)
                legs["eager channels_last, SDPA, cudnn.benchmark"] = t_ms(lambda: o_unet.unet_forward(sd_cl, xt_cl, tt))
                # (c) torch.compile(max-autotune) of the same graph (Inductor + Triton: a comparator, never product) - minutes of
                # compilation, so only on request (tools/collect_profiles.sh passes --compile; bench.py does not)
                if compile_leg and mode == "bf16":
                    import time as _t
                    t0 = _t.time()
                    best_in = (sd_cl, xt_cl) if legs["eager channels_last, SDPA, cudnn.benchmark"] < legs["eager NCHW, SDPA, cudnn.benchmark (MIOpen find)"] else (sd, xt)
                    try:
                        cf = torch.compile(lambda a, b: o_unet.unet_forward(best_in[0], a, b), mode="max-autotune")
                        for _ in range(3):
                            cf(best_in[1], tt)
                        legs["torch.compile(max-autotune), best layout"] = t_ms(lambda: cf(best_in[1], tt))
                        res["torch_compile_s"] = round(_t.time() - t0, 1)
                    except Exception as e:            # Inductor failures must not take the yardstick down
                        res["torch_compile_error"] = repr(e)[:300]
                del sd_cl
            except StopIteration:
                pass
            except RuntimeError as e:          # e.g. an op without a bf16 kernel: say so instead of dying
                res.setdefault(f"torch_rocm_eager_{mode}_ms", None)
                res[f"torch_rocm_eager_{mode}_error"] = str(e)[:200]
            finally:
                o_unet.attention = math_attention
                torch.backends.cudnn.benchmark = False
        if legs:
            bk = min(legs, key=legs.get)
            res[f"legs_{mode}_ms"] = legs
            res[f"best_{mode}_ms"] = legs[bk]
            res[f"best_{mode}_recipe"] = bk
        del sd
        torch.cuda.empty_cache()
    if res.get("best_bf16_ms"):
        res["best_ms"] = res["best_bf16_ms"]
        res["best_recipe"] = res["best_bf16_recipe"]
        res["speedup_bf16_vs_best"] = round(res["best_bf16_ms"] / res["ldmseg_hip_bf16_ms"], 3)
    if res.get("best_fp32_ms"):
        res["speedup_fp32_vs_best"] = round(res["best_fp32_ms"] / res["ldmseg_hip_fp32_ms"], 3)
    if res.get("torch_rocm_eager_bf16_ms"):
        res["speedup_bf16"] = round(res["torch_rocm_eager_bf16_ms"] / res["ldmseg_hip_bf16_ms"], 3)
    if res.get("torch_rocm_eager_fp32_ms"):
        res["speedup_fp32"] = round(res["torch_rocm_eager_fp32_ms"] / res["ldmseg_hip_fp32_ms"], 3)
    res["torch"] = torch.__version__
    res["what"] = (f"oracle/unet.py graph on the GPU, eager, B={B}, L={L}: the stack the reference itself runs on "
                   f"(tools/main_ldm.py:146-192); 3 timed forwards after one warm-up, HIP events")
    return res


if __name__ == "__main__":
    main()
