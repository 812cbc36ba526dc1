"""Phase stamps of the GroupNorm kernels (scratch/lib_gnstamp.so = norm.hip built with -DLDMSEG_GN_STAMP):
LDMSEG_HIP_LIB=scratch/lib_gnstamp.so python tools/gn_stamps.py [B C HW]"""
import ctypes as C, os, sys, numpy as np, torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, os.path.join(R, "latent-diffusion-segmentation_amd"))
from ldmseg_amd import _lib
L = _lib.lib()
B, Cc, HW = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (8, 320, 4096)
x = torch.randn(B, Cc, HW, device="cuda"); g = torch.ones(Cc, device="cuda"); b = torch.zeros(Cc, device="cuda")
out = torch.empty_like(x)
P = lambda t: C.c_void_p(t.data_ptr())
for _ in range(3):
    L.ldmseg_op_groupnorm(P(x), None, P(g), P(b), B, Cc, 0, HW, C.c_float(1e-5), 1, 1, P(out), None)
torch.cuda.synchronize()
h = C.CDLL(os.environ["LDMSEG_HIP_LIB"])
buf = (C.c_ulonglong * (8192 * 8))()
assert h.ldmseg_debug_gn_stamps(buf, 8192 * 8) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8).astype(np.int64)
def rep(name, rows, s0, s1):
    d = (rows[:, s1] - rows[:, s0]); d = d[(rows[:, s0] > 0) & (rows[:, s1] > rows[:, s0])]
    print(f"{name:34s} n={len(d):5d} mean {d.mean():9.0f} ticks  p50 {np.median(d):9.0f}  max {d.max():9.0f}")
pa = a[a[:, 0] > 0]
rep("partial: load+accumulate", pa, 0, 1); rep("partial: syncthreads", pa, 1, 2); rep("partial: group combine+store", pa, 2, 3); rep("partial: whole block", pa, 0, 3)
print("partial: kernel span (first start -> last end)", pa[:, 3].max() - pa[:, 0].min(), "ticks")
ap = a[a[:, 4] > 0]
rep("apply: reduce_stats prologue", ap, 4, 5); rep("apply: stream", ap, 5, 6); rep("apply: whole block", ap, 4, 6)
print("apply: kernel span", ap[:, 6].max() - ap[:, 4].min(), "ticks;  partial start -> apply end", ap[:, 6].max() - pa[:, 0].min())
