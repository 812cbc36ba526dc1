#!/usr/bin/env python
"""fp8 attention timing: python tools/attn8_bench.py [B N C]  (scaled-MFMA path vs unscaled vs the bf16 kernel)"""
import ctypes as C
import os
import sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion-segmentation_amd"))
import torch  # noqa: E402
from ldmseg_amd import _lib  # noqa: E402
L = _lib.lib()
B, N, Cc = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (4, 16384, 320)
qkv = torch.randn(B, N, 3 * Cc, device="cuda")
P = lambda t: C.c_void_p(t.data_ptr())
fl = 4.0 * B * 8 * N * N * (Cc / 8)
for name, key in (("scaled, 4 waves, plain", 0x101), ("scaled, 8 waves, plain", 0x111), ("scaled, 8 waves, direct byte", 0x131), ("unscaled fp8 (attention_fp8)", 0)):
    L.ldmseg_debug_set(15, key)
    us = C.c_float(0)
    assert L.ldmseg_op_attention_fp8(P(qkv), B, N, Cc, 8, None, 10, C.byref(us), None) == 0
    print(f"{name:30s} B={B} N={N} C={Cc}: {us.value:9.1f} us incl. the pre-pass  ({fl / us.value / 1e6:7.1f} TF/s)", flush=True)
L.ldmseg_debug_set(15, 0x111)
us = C.c_float(0)
assert L.ldmseg_bench_attention(P(qkv), B, N, Cc, 8, 1, 10, C.byref(us), None) == 0
print(f"{'bf16 (shipped rule: attention4)':30s} B={B} N={N} C={Cc}: {us.value:9.1f} us  ({fl / us.value / 1e6:7.1f} TF/s)")
