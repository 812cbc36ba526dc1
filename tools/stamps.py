import sys, os, ctypes as C, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "latent-diffusion-segmentation_amd"))
from ldmseg_amd import _lib
L = _lib.lib()
B, Ci, H, Co = 8, int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x = torch.randn(B, Ci, H, H, device="cuda"); w = torch.randn(Co, Ci, 3, 3, device="cuda") * 0.02; b = torch.zeros(Co, device="cuda")
out = torch.empty(B, Co, H, H, device="cuda")
P = lambda t: C.c_void_p(t.data_ptr())
ts = torch.zeros(1024 * 16, dtype=torch.int64, device="cuda")
ptr = ts.data_ptr()
L.ldmseg_debug_set(3, C.c_int(ptr & 0xffffffff).value if (ptr & 0xffffffff) < 2**31 else (ptr & 0xffffffff) - 2**32)
L.ldmseg_debug_set(4, ptr >> 32)
for _ in range(3):
    ts.zero_()
    L.ldmseg_op_conv2d(P(x), None, P(w), P(b), B, Ci, 0, H, H, Co, 3, 1, 0, 1, P(out), None)
    torch.cuda.synchronize()
t = ts.cpu().numpy().reshape(-1, 16)
t = t[t[:, 0] != 0]
print("workgroups", len(t))
names = ["start", "prologue done", "k-loop end", "epilogue end", "after barrier", "-", "b0 start", "b0 staged", "b1 start", "b1 staged", "b2 start", "b2 staged", "b3 start", "b3 staged"]
rel = (t - t[:, :1]).astype(np.float64)
tot = rel[:, 4].mean()
print("ticks start->end mean", tot)
for i, n in enumerate(names):
    ok = t[:, i] != 0
    if ok.any(): print(f"{n:15s} mean {rel[ok, i].mean():8.1f} ticks ({100*rel[ok, i].mean()/tot:5.1f}%)  min {rel[ok, i].min():8.1f} max {rel[ok, i].max():8.1f}")
