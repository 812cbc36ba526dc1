#!/usr/bin/env python
"""Launch-table candidates that the in-launch K-slice finish makes worth a look (round 6): the 32x32-level convs and resnet tails of the
B = 8 / L = 64 forward on 256-row tiles with two K slices finished inside the launch, against the shipped entries (128-row tiles, one
slice).  Cold weights (rotated over copies), HIP events around back-to-back launches, interleaved rounds, min of 3.
    python tools/cf_tune.py [iters]"""
import ctypes as C
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion-segmentation_amd"))
import torch  # noqa: E402
from ldmseg_amd import _lib  # noqa: E402

L = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 30
# plain convs: (H, Ci, Ci2, Co, k, rowbias, launches per forward)   |   resnet tails: (H, C, Cs, Cs2, launches)
CONVS = [(32, 640, 0, 640, 3, 1, 2), (32, 640, 640, 640, 3, 1, 1), (32, 640, 320, 640, 3, 1, 1), (32, 320, 0, 640, 3, 1, 1),
         (32, 1280, 640, 640, 3, 1, 1), (16, 640, 0, 1280, 3, 1, 1), (16, 1280, 0, 1280, 3, 1, 2)]
TAILS = [(32, 640, 1280, 640, 1), (32, 640, 640, 640, 1), (32, 640, 640, 320, 1), (32, 640, 320, 0, 1),
         (16, 1280, 1280, 1280, 2), (16, 1280, 1280, 640, 1), (16, 1280, 640, 0, 1)]
# settings: (label, key 24 override or -1, key 23)
SETTINGS = [("shipped", -1, 1), ("e0 x2 cf", 0 | (2 << 8), 1), ("e0 x2", 0 | (2 << 8), 0), ("e0 x4 cf", 0 | (4 << 8), 1), ("e3 x2 cf", 3 | (2 << 8), 9)]
B = 8
L.ldmseg_debug_set(6, 6)
tot = {s[0]: 0.0 for s in SETTINGS}
try:
    for kind, shapes in (("conv", CONVS), ("tail", TAILS)):
        for sh in shapes:
            if kind == "conv":
                H, Ci, Ci2, Co, k, use_rb, n = sh
                x = torch.randn(B, Ci, H, H, device="cuda")
                x2 = torch.randn(B, Ci2, H, H, device="cuda") if Ci2 else None
                w = torch.randn(Co, Ci + Ci2, k, k, device="cuda") / ((Ci + Ci2) * k * k) ** 0.5
                b = torch.randn(Co, device="cuda")
                rb = torch.randn(B, Co, device="cuda") if use_rb else None
                label = f"conv M={B * H * H:5d} N={Co:4d} K={(Ci + Ci2) * k * k:5d}"
            else:
                H, Cc, Cs, Cs2, n = sh
                h = torch.randn(B, Cc, H, H, device="cuda")
                xs = torch.randn(B, Cs, H, H, device="cuda")
                xs2 = torch.randn(B, Cs2, H, H, device="cuda") if Cs2 else None
                w2 = torch.randn(Cc, Cc, 3, 3, device="cuda") / (9 * Cc) ** 0.5
                ws = torch.randn(Cc, Cs + Cs2, 1, 1, device="cuda") / (Cs + Cs2) ** 0.5
                b = torch.randn(Cc, device="cuda")
                label = f"tail M={B * H * H:5d} N={Cc:4d} K={9 * Cc:5d}+{Cs + Cs2:4d}"
            res = {s[0]: [] for s in SETTINGS}
            names = {}
            for rnd in range(3):
                for (lab, ov, cf) in SETTINGS:
                    L.ldmseg_debug_set(24, ov)
                    L.ldmseg_debug_set(23, cf)
                    us = C.c_float()
                    if kind == "conv":
                        r = L.ldmseg_bench_igemm(P(x), P(x2), P(w), P(b), None, P(rb), B, Ci, Ci2, H, H, Co, k, 1, 0, 0, 0, 0, 1, ITERS, C.byref(us), None)
                    else:
                        r = L.ldmseg_op_conv3x3_plus_1x1(P(h), P(w2), P(b), P(xs), P(xs2), P(ws), P(b), B, Cc, Cs, Cs2, H, H, Cc, 0, 1, None, ITERS,
                                                         C.byref(us), None)
                    if r == 0:
                        res[lab].append(us.value)
                        names[lab] = _lib.igemm_last_kernel()
            line = label + f" x{n}: "
            for (lab, ov, cf) in SETTINGS:
                v = min(res[lab]) if res[lab] else float("nan")
                tot[lab] += n * v
                line += f" {lab} {v:6.1f}"
            print(line, "|", names.get("shipped"), "|", names.get("e0 x2 cf"), flush=True)
finally:
    L.ldmseg_debug_set(24, -1)
    L.ldmseg_debug_set(23, 1)
    L.ldmseg_debug_set(6, 1)
print("per forward:", {k: round(v, 1) for k, v in tot.items()})
