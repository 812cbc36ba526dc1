#!/usr/bin/env python
"""Resnet tail (conv2 + conv_shortcut) at the UNet's launch shapes: the one extra-tap launch (igemm.hip XT, round 5) against the two
launches it replaces (conv_shortcut, then conv2 with the shortcut as its residual).  HIP events around back-to-back launches,
warm weights on both sides; B = 8 unless given.      python tools/xt_bench.py [B] [L]"""
import ctypes as C
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion-segmentation_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from ldmseg_amd import _lib  # noqa: E402
from test_igemm_shapes_gpu import XT_SHAPES  # noqa: E402

L = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
LAT = int(sys.argv[2]) if len(sys.argv) > 2 else 64
COUNT = {(64, 320, 640, 320): 1, (64, 320, 320, 320): 2, (32, 640, 1280, 640): 1, (32, 640, 640, 640): 1, (32, 640, 640, 320): 1,
         (32, 640, 320, 0): 1, (16, 1280, 1280, 1280): 2, (16, 1280, 1280, 640): 1, (16, 1280, 640, 0): 1, (8, 1280, 1280, 1280): 3}
tot_f = tot_u = 0.0
for case in XT_SHAPES:
    H0, Cc, Cs, Cs2 = case
    H = H0 * LAT // 64
    h = torch.randn(B, Cc, H, H, device="cuda")
    xs = torch.randn(B, Cs, H, H, device="cuda")
    xs2 = torch.randn(B, Cs2, H, H, device="cuda") if Cs2 else None
    w2 = torch.randn(Cc, Cc, 3, 3, device="cuda") / (9 * Cc) ** 0.5
    ws = torch.randn(Cc, Cs + Cs2, 1, 1, device="cuda") / (Cs + Cs2) ** 0.5
    b = torch.randn(Cc, device="cuda")
    res = torch.randn(B, Cc, H, H, device="cuda")
    us = C.c_float()
    _lib.check(L.ldmseg_op_conv3x3_plus_1x1(P(h), P(w2), P(b), P(xs), P(xs2), P(ws), P(b), B, Cc, Cs, Cs2, H, H, Cc, 0, 1, None, 20,
                                            C.byref(us), None), "conv3x3_plus_1x1")
    fused, kf = us.value, _lib.igemm_last_kernel()
    _lib.check(L.ldmseg_bench_igemm(P(h), None, P(w2), P(b), P(res), None, B, Cc, 0, H, H, Cc, 3, 1, 0, 0, 0, 0, 1, 20, C.byref(us), None), "conv2")
    c2 = us.value
    _lib.check(L.ldmseg_bench_igemm(P(xs), P(xs2), P(ws), P(b), None, None, B, Cs, Cs2, H, H, Cc, 1, 1, 0, 0, 0, 0, 1, 20, C.byref(us), None), "shortcut")
    sc = us.value
    n = COUNT.get(case, 1) if (B, LAT) == (8, 64) else 1
    tot_f += n * fused
    tot_u += n * (c2 + sc)
    print(f"M={B * H * H:6d} C={Cc:4d} shortcut K={Cs + Cs2:4d}: one launch {fused:7.1f} us   conv2 {c2:7.1f} + shortcut {sc:6.1f} = {c2 + sc:7.1f} us   x{n}  {kf}")
print(f"per forward (launch counts of the B=8 / L=64 UNet): one launch {tot_f:.1f} us, two launches {tot_u:.1f} us, saved {tot_u - tot_f:.1f} us")
