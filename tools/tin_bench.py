#!/usr/bin/env python
"""proj_in -> LayerNorm_1 -> q|k|v of a 320-channel transformer: unfused launches vs the row-local fused kernel (tproj.hip).
python tools/tin_bench.py [M ...]"""
import ctypes as C
import os
import sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion-segmentation_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from ldmseg_amd import _lib  # noqa: E402
L = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr())
Ms = [int(a) for a in sys.argv[1:]] or [32768, 65536]
for M in Ms:
    g = torch.Generator().manual_seed(M)
    Cc = 320
    case = [torch.randn(M, Cc, generator=g), torch.randn(Cc, Cc, generator=g) / Cc ** 0.5, torch.randn(Cc, generator=g), 1 + 0.2 * torch.randn(Cc, generator=g),
            0.2 * torch.randn(Cc, generator=g)] + [torch.randn(Cc, Cc, generator=g) / Cc ** 0.5 for _ in range(3)]
    keep = [t.cuda() for t in case]
    res = {}
    for rnd in range(3):
        for mode in (0, 1):
            h = torch.empty(M, Cc, device="cuda"); qkv = torch.empty(M, 3 * Cc, device="cuda")
            us = C.c_float(0)
            r = L.ldmseg_op_transformer_in(*[P(t) for t in keep], M, Cc, 1e-5, 1, mode, P(h), P(qkv), 20, C.byref(us), None)
            assert r == 0, r
            res.setdefault(mode, []).append(us.value)
            res[("o", mode)] = (h, qkv)
    d = (res[("o", 1)][1].double() - res[("o", 0)][1].double()).norm() / res[("o", 0)][1].double().norm()
    print(f"M={M}: unfused (proj_in + rowstats + qkv) min {min(res[0]):7.1f} us, fused min {min(res[1]):7.1f} us  ratio {min(res[1]) / min(res[0]):.3f}  "
          f"rel-L2 qkv fused vs unfused {float(d):.2e}", flush=True)
