#!/usr/bin/env python
"""Where the microseconds of one igemm launch go that no workgroup's instruction stream accounts for.
Needs the stamp build:   bash tools/build_stamp.sh
    LDMSEG_HIP_LIB=tools/ab/lib_stamp.so LDMSEG_OP_TIMING_NHWC=1 python tools/launch_boundary.py
Every workgroup stamps the device-wide 100 MHz wall clock (s_memrealtime: comparable across XCDs) when it starts and when it has
finished its last item, its XCD, and s_memtime at the phase boundaries.  Per launch shape the script prints
  * launch-to-launch time T of back-to-back launches (HIP events; what a forward pays),
  * span = first workgroup start -> last workgroup end (wall clock), and T - span = time outside every workgroup (dispatch of the
    first wave, end-of-kernel cache write-back, the dependent-launch boundary),
  * start skew (first -> last workgroup start) and the distribution of workgroup durations: span - median duration - skew is what
    the slowest workgroups add (tail), per XCD as well,
  * median phase times of a workgroup (prologue / K loop / epilogue)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion-segmentation_amd"))
import torch  # noqa: E402
from ldmseg_amd import _lib  # noqa: E402

L = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
B = 8
# (label, Ci, H, Co, k)
SHAPES = [("conv3x3 64x64 320->320  (M=32768 N=320 K=2880)", 320, 64, 320, 3),
          ("conv3x3 64x64 640->320  (M=32768 N=320 K=5760)", 640, 64, 320, 3),
          ("conv3x3 32x32 640->640  (M=8192 N=640 K=5760)", 640, 32, 640, 3),
          ("conv1x1 32x32 640->640  (M=8192 N=640 K=640)", 640, 32, 640, 1),
          ("conv1x1 16x16 1280->1280 (M=2048 N=1280 K=1280)", 1280, 16, 1280, 1)]
# (K-sliced launches are not in the list: ldmseg_op_conv2d, which carries the stamps, does not plan K slices)
ts = torch.zeros(4096 * 16, dtype=torch.int64, device="cuda")
ptr = ts.data_ptr()
lo = ptr & 0xffffffff
L.ldmseg_debug_set(3, lo if lo < 2 ** 31 else lo - 2 ** 32)
L.ldmseg_debug_set(4, ptr >> 32)
L.ldmseg_debug_set(6, 6)           # the timing loop rotates over weight copies (cold weights, as in a forward)
for label, Ci, H, Co, k in SHAPES:
    x = torch.randn(B, Ci, H, H, device="cuda")
    w = torch.randn(Co, Ci, k, k, device="cuda") * 0.02
    b = torch.zeros(Co, device="cuda")
    out = torch.empty(B, Co, H, H, device="cuda")
    us = C.c_float()
    _lib.check(L.ldmseg_bench_igemm(P(x), None, P(w), P(b), None, None, B, Ci, 0, H, H, Co, k, 1, 0, 0, 0, 0, 1, 40, C.byref(us), None), "bench")
    T = us.value
    rows = []
    for rep in range(7):
        ts.zero_()
        torch.cuda.synchronize()
        L.ldmseg_op_conv2d(P(x), None, P(w), P(b), B, Ci, 0, H, H, Co, k, 1, 0, 1, P(out), None)
        torch.cuda.synchronize()
        if rep < 2:
            continue
        t = ts.cpu().numpy().reshape(-1, 16)
        t = t[(t[:, 14] != 0) & (t[:, 15] != 0)]
        w0, w1 = t[:, 14].astype(np.float64) * 0.01, t[:, 15].astype(np.float64) * 0.01      # microseconds
        dur = w1 - w0
        span = w1.max() - w0.min()
        skew = w0.max() - w0.min()
        xcd = t[:, 5] - 1
        per_xcd = [(w1[xcd == q].max() - w0[xcd == q].min()) for q in range(8) if (xcd == q).any()]
        clk = (t[:, 4] - t[:, 0]).astype(np.float64) / np.maximum(dur, 1e-3)                   # s_memtime ticks per microsecond
        ph = [np.median((t[:, j] - t[:, i]).astype(np.float64) / clk) for i, j in ((0, 1), (1, 2), (2, 3))]
        rows.append((len(t), span, skew, np.percentile(dur, 5), np.median(dur), np.percentile(dur, 95), dur.max(), min(per_xcd), max(per_xcd),
                     ph[0], ph[1], ph[2], int(len(set(xcd.tolist()))), float(np.mean(np.abs((np.arange(len(xcd)) % 8) - xcd) < 0.5))))
    r = np.median(np.array(rows, dtype=np.float64), axis=0)
    print(label, "|", _lib.igemm_last_kernel())
    print(f"  launch to launch (back to back, cold weights) T = {T:6.1f} us;  workgroups {int(r[0])} on {int(r[12])} XCDs (block b on XCD b % 8: {100 * r[13]:.0f} %)")
    print(f"  span first start -> last end {r[1]:6.1f} us   => outside every workgroup (dispatch, write-back, boundary): T - span = {T - r[1]:5.1f} us")
    print(f"  start skew {r[2]:5.1f} us;  workgroup duration p5 {r[3]:5.1f}  p50 {r[4]:5.1f}  p95 {r[5]:5.1f}  max {r[6]:5.1f} us"
          f"  => tail (span - skew - p50) {r[1] - r[2] - r[4]:5.1f} us;  per-XCD span min {r[7]:5.1f} max {r[8]:5.1f}")
    print(f"  median workgroup: prologue {r[9]:5.1f}  K loop {r[10]:5.1f}  epilogue {r[11]:5.1f} us", flush=True)
L.ldmseg_debug_set(6, 1)
