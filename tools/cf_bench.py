#!/usr/bin/env python
"""K-sliced launches of the B = 8 / L = 64 forward: slabs + finish launch (debug key 23 = 0) against the in-launch cooperative
finish (23 = 1), per launch shape, cold weights (rotated over copies), HIP events around back-to-back launches.
    python tools/cf_bench.py [iters]"""
import ctypes as C
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion-segmentation_amd"))
import torch  # noqa: E402
from ldmseg_amd import _lib  # noqa: E402

L = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
# (B, Ci, Ci2, H, Co, k, stride, up, residual, rowbias, launches per forward)
SHAPES = [
    (8, 1280, 0, 8, 1280, 3, 1, 0, 0, 1, 8),      # M = 512, K = 11520, 8 slices
    (8, 2560, 0, 8, 1280, 3, 1, 0, 0, 1, 3),      # M = 512, K = 23040
    (8, 5120, 1280, 8, 1280, 1, 1, 0, 1, 0, 1),   # chained Linear at 8x8: K = 6400
    (8, 1280, 0, 16, 1280, 3, 1, 0, 0, 1, 2),     # M = 2048, K = 11520, 4 slices on 256-row tiles
    (8, 2560, 0, 16, 1280, 3, 1, 0, 0, 1, 2),     # M = 2048, K = 23040
    (8, 5120, 1280, 16, 1280, 1, 1, 0, 1, 0, 5),  # chained Linear at 16x16: K = 6400, 2 slices
    (8, 640, 0, 16, 1280, 3, 1, 0, 0, 1, 1),      # M = 2048, K = 5760, 2 slices
    (8, 1280, 0, 8, 1280, 3, 1, 1, 0, 0, 1),      # 8x8 -> 16x16 upsampler (four phase convs, 4 slices)
    (8, 1920, 0, 32, 640, 3, 1, 0, 0, 1, 1),      # M = 8192, K = 17280, 2 slices
]
L.ldmseg_debug_set(6, 6)
saved = L.ldmseg_debug_get(23)
tot = {0: 0.0, 1: 0.0}
try:
    for (B, Ci, Ci2, H, Co, k, stride, up, use_res, use_rb, n) in SHAPES:
        x = torch.randn(B, Ci, H, H, device="cuda")
        x2 = torch.randn(B, Ci2, H, H, device="cuda") if Ci2 else None
        w = torch.randn(Co, Ci + Ci2, k, k, device="cuda") / ((Ci + Ci2) * k * k) ** 0.5
        b = torch.randn(Co, device="cuda")
        Ho = H * (2 if up else 1) // stride
        res = torch.randn(B, Co, Ho, Ho, device="cuda") if use_res else None
        rb = torch.randn(B, Co, device="cuda") if use_rb else None
        row = {}
        for rnd in range(3):
            for mode in (0, 1):
                L.ldmseg_debug_set(23, mode)
                us = C.c_float()
                _lib.check(L.ldmseg_bench_igemm(P(x), P(x2), P(w), P(b), P(res), P(rb), B, Ci, Ci2, H, H, Co, k, stride, up, 0, 0, 0, 1,
                                                ITERS, C.byref(us), None), "bench")
                row.setdefault(mode, []).append(us.value)
                row[("name", mode)] = _lib.igemm_last_kernel()
        a, c = min(row[0]), min(row[1])
        tot[0] += n * a
        tot[1] += n * c
        print(f"M={B * Ho * Ho:6d} N={Co:5d} K={(Ci + Ci2) * k * k:6d} up={up}: two launches {a:7.2f} us   in-launch {c:7.2f} us   x{n}   {row[('name', 1)]}", flush=True)
finally:
    L.ldmseg_debug_set(23, saved)
    L.ldmseg_debug_set(6, 1)
print(f"per forward over these shapes: two launches {tot[0]:.1f} us, in-launch finish {tot[1]:.1f} us")
