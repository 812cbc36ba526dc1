#!/usr/bin/env python
"""Accuracy of the fp8 attention variants against fp64 softmax attention on the SAME quantised operands (tools only)."""
import ctypes as C
import os
import sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion-segmentation_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from ldmseg_amd import _lib  # noqa: E402
from test_ops_gpu import attention_ref, bf16_round, fp8_e4m3_round, dev, P  # noqa: E402
from conftest import rel_err  # noqa: E402
L = _lib.lib()
for (B, N, Cc, qs) in [(1, 1024, 320, 1.5), (1, 4096, 320, 1.5), (1, 4096, 320, 4.0)]:
    g = torch.Generator().manual_seed(N + Cc)
    qkv = torch.randn(B, N, 3 * Cc, generator=g)
    qkv[:, :, :Cc] *= qs
    d = Cc // 8
    src = bf16_round(qkv)
    sc = torch.tensor(d ** -0.5, dtype=torch.float32) * torch.tensor(1.4426950408889634, dtype=torch.float32)
    q8 = fp8_e4m3_round(src[..., :Cc] * sc) / sc
    k8, v8 = fp8_e4m3_round(src[..., Cc:2 * Cc]), fp8_e4m3_round(src[..., 2 * Cc:])
    ref8 = attention_ref(torch.cat([q8, k8, v8], -1), B, N, Cc)
    ref = attention_ref(src, B, N, Cc)
    dq = dev(qkv)
    out = torch.empty(B, N, Cc, device="cuda")
    settings = [("unscaled fp8", 0), ("scaled, exact exp", 0x111)] + \
               [(f"scaled, direct byte, pshift {13 + k / 1000:.4f}", 0x331 | (k << 16)) for k in (900, 950, 962, 975, 1000, 1025, 1050)]
    for name, key in settings:
        L.ldmseg_debug_set(15, key)
        assert L.ldmseg_op_attention_fp8(P(dq), B, N, Cc, 8, P(out), 0, None, None) == 0
        torch.cuda.synchronize()
        o = out.cpu()
        l2 = float((o - ref8).norm() / ref8.norm())
        print(f"N={N} q x{qs}: {name:42s} vs fp64 on quantised operands: max {rel_err(o, ref8):.4f} rel-L2 {l2:.4f}   vs unquantised rel-L2 "
              f"{float((o - ref).norm() / ref.norm()):.4f}", flush=True)
L.ldmseg_debug_set(15, 0x111)
