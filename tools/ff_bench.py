#!/usr/bin/env python
"""Times the tail of a 320-channel transformer block (LayerNorm_3 -> GEGLU -> ff.net.2 -> proj_out) as unfused launches
(mode 0), with the fused feed-forward (1) and fully fused (3) in one process: python tools/ff_bench.py [M ...]"""
import ctypes as C
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion-segmentation_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from ldmseg_amd import _lib  # noqa: E402
from test_ops_gpu import _ff_case, dev, P  # noqa: E402

L = _lib.lib()
for kv in os.environ.get("KEYS", "").split(","):          # e.g. KEYS=13=0x10000 (debug key = value) before timing
    if kv:
        L.ldmseg_debug_set(int(kv.split("=")[0]), int(kv.split("=")[1], 0))
for M in [int(a) for a in sys.argv[1:]] or [32768, 65536]:
    case = [dev(t) for t in _ff_case(M, 320, 1)]
    out = torch.empty(M, 320, device="cuda")
    fl = 2.0 * M * (8 * 320 * 320 + 4 * 320 * 320 + 320 * 320)
    for mode in [int(m) for m in os.environ.get("MODES", "0,1,3").split(",")]:
        us = C.c_float(0)
        r = L.ldmseg_op_transformer_ff(*[P(t) for t in case], M, 320, 1e-5, 1, mode, P(out), 20, C.byref(us), None)
        assert r == 0, r
        print(f"M={M} mode={mode}: {us.value:8.1f} us per tail  ({fl / us.value / 1e6:6.1f} TF/s incl. the h copy)", flush=True)
