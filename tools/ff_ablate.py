#!/usr/bin/env python
"""Phase ablation of the fused feed-forward kernel (LDMSEG_HIP_LIB=scratch/lib_tf_ablate.so python tools/ff_ablate.py [M] [mode]).
Flags (debug key 13): 1 no weight DMA, 2 GEGLU without the erf polynomial, 4 no MFMA in GEMM1, 8 no MFMA in GEMM2,
16 no fragment reads in GEMM1, 32 none in GEMM2, 64 no hidden-chunk store, 256 no start-chunk rotation."""
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
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 3
case = [dev(t) for t in _ff_case(M, 320, 1)]
out = torch.empty(M, 320, device="cuda")
flags = [int(v, 0) for v in os.environ.get("DBGS", "0,256,1,2,4,8,12,16,32,48,64,28,60,61,63").split(",")]
for f in flags:
    L.ldmseg_debug_set(13, f)
    us = C.c_float(0)
    r = L.ldmseg_op_transformer_ff(*[P(t) for t in case], M, 320, 1e-5, 1, mode, P(out), 20, C.byref(us), None)
    line = f"M={M} mode={mode} dbg={f:3d}: {us.value:8.1f} us"
    if hasattr(L, "ldmseg_debug_tf_stamps"):
        ts = (C.c_ulonglong * 48)()
        L.ldmseg_debug_tf_stamps(ts, 48)
        t = [v / 100.0 for v in ts]     # wall_clock64: 100 MHz -> us
        names = ["A-wait", "LN", "B-wait", "chunk0", "chunks1-9", "chunks10-19", "tile-write", "sync", "proj-loop", "proj-epi"]
        seg = [t[i + 1] - t[i] for i in range(10)]
        line += "  | block 0 wave 0: " + " ".join(f"{n}={d:.1f}" for n, d in zip(names, seg)) + \
                f" | chunk5: gemm1={t[12] - t[11]:.2f} gemm2={t[13] - t[12]:.2f} | total {t[10] - t[0]:.1f}"
        base = t[11]
        line += "\n      chunk 5 compute: start 0 | after barrier kt0..4: " + " ".join(f"{t[16 + k] - base:.2f}" for k in range(5)) + \
                f" | gemm2 end {t[13] - base:.2f}"
        line += "\n      chunk 5 loader (step: landed/barrier/issued): " + "  ".join(
            f"[{t[24 + 3 * j] - base:.2f} {t[25 + 3 * j] - base:.2f} {t[26 + 3 * j] - base:.2f}]" for j in range(6))
    print(line, flush=True)
L.ldmseg_debug_set(13, 0)
