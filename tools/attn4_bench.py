#!/usr/bin/env python
"""bf16 attention, head dim 40: attention3.hip (16x16x32 score blocks) vs attention4.hip (32x32x16): timing + agreement.
Variants are timed in interleaved rounds (boxes drift by several % within seconds).  python tools/attn4_bench.py [B N C]"""
import ctypes as C
import os
import statistics
import sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion-segmentation_amd"))
import torch  # noqa: E402
from ldmseg_amd import _lib  # noqa: E402
L = _lib.lib()
shapes = [tuple(int(a) for a in sys.argv[1:4])] if len(sys.argv) > 3 else [(8, 4096, 320), (4, 16384, 320), (16, 4096, 320), (2, 1000, 320), (1, 256, 320)]
VARIANTS = (("attention3 (round-3 rule)", 7), ("attention3 4w lazy16", 10), ("attention4 8w lazy16", 11), ("attention4 8w every", 12), ("attention4 4w lazy16", 13),
            ("attention4 4w every", 14), ("shipped rule", 100))
P = lambda t: C.c_void_p(t.data_ptr())
for B, N, Cc in shapes:
    torch.manual_seed(0)
    qkv = torch.randn(B, N, 3 * Cc, device="cuda")
    fl = 4.0 * B * 8 * N * N * (Cc / 8)
    outs, times = {}, {n: [] for n, _ in VARIANTS}
    for rnd in range(5):
        for name, key in VARIANTS:
            L.ldmseg_debug_set(2, 0 if key >= 100 else key)     # (key 0 = the shipped choice)
            if rnd == 0:
                out = torch.empty(B, N, Cc, device="cuda")
                rc = L.ldmseg_op_attention(P(qkv), B, N, Cc, 8, 1, P(out), None)
                assert rc == 0, (name, rc)
                outs[name] = out
            us = C.c_float(0)
            assert L.ldmseg_bench_attention(P(qkv), B, N, Cc, 8, 1, 10, C.byref(us), None) == 0
            times[name].append(us.value)
    L.ldmseg_debug_set(2, 0)
    ref = outs[VARIANTS[0][0]]
    for name, _ in VARIANTS:
        t = times[name]
        d = (outs[name] - ref).abs().max().item()
        print(f"{name:26s} B={B} N={N}: min {min(t):8.1f} median {statistics.median(t):8.1f} us ({fl / min(t) / 1e6:6.1f} TF/s)  max|d| vs shipped {d:.3e} "
              f"finite={bool(torch.isfinite(outs[name]).all())}", flush=True)
    if N <= 4096 and B <= 8:
        q, k, v = (t.reshape(B, N, 8, Cc // 8).transpose(1, 2).to(torch.bfloat16).double() for t in qkv.chunk(3, dim=-1))
        refd = (torch.softmax(q @ k.transpose(-1, -2) / (Cc // 8) ** 0.5, -1) @ v).transpose(1, 2).reshape(B, N, Cc)
        print("   max|d| vs fp64: " + ", ".join(f"{n.split()[0][-1]}/{' '.join(n.split()[1:])} {(o.double() - refd).abs().max().item():.2e}" for n, o in outs.items()))
