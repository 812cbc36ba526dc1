"""Per-workgroup s_memtime stamps of one igemm launch (stamp-only build: tools/build_stamp.sh, LDMSEG_HIP_LIB=tools/ab/lib_stamp.so).
    python tools/stamps2.py Ci H Co [k] [B]      e.g. 320 64 320   (M = 32768, N = 320, K = 2880)
Prints, relative to the EARLIEST workgroup start of the launch: when workgroups start, finish their prologue (first tile landed),
their K loop, their epilogue - i.e. dispatch ramp, prologue, K loop, epilogue, tail of one launch."""
import sys, os, ctypes as C, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "latent-diffusion-segmentation_amd"))
from ldmseg_amd import _lib
L = _lib.lib()
Ci, H, Co = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
k = int(sys.argv[4]) if len(sys.argv) > 4 else 3
B = int(sys.argv[5]) if len(sys.argv) > 5 else 8
x = torch.randn(B, Ci, H, H, device="cuda"); w = torch.randn(Co, Ci, k, k, device="cuda") * 0.02; b = torch.zeros(Co, device="cuda")
out = torch.empty(B, Co, H, H, device="cuda")
P = lambda t: C.c_void_p(t.data_ptr())
ts = torch.zeros(2048 * 16, dtype=torch.int64, device="cuda")
ptr = ts.data_ptr()
lo = ptr & 0xffffffff
L.ldmseg_debug_set(3, lo if lo < 2**31 else lo - 2**32)
L.ldmseg_debug_set(4, ptr >> 32)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(4):
    ts.zero_()
    torch.cuda.synchronize()
    e0.record()
    L.ldmseg_op_conv2d(P(x), None, P(w), P(b), B, Ci, 0, H, H, Co, k, 1, 0, 1, P(out), None)
    e1.record()
    torch.cuda.synchronize()
print("kernel:", _lib.igemm_last_kernel(), " op wall (incl. pack/unpack launches) us:", 1e3 * e0.elapsed_time(e1))
t = ts.cpu().numpy().reshape(-1, 16)
t = t[t[:, 0] != 0]
g0 = t[:, 0].min()
print("workgroups", len(t), " (ticks = shader clocks; divide by ~2100 for us)")
names = {0: "start", 1: "prologue done", 2: "k-loop end", 3: "epilogue end", 4: "after barrier", 6: "b0 start", 7: "b0 staged", 8: "b1 start",
         9: "b1 staged", 10: "b2 start", 11: "b2 staged", 12: "b3 start", 13: "b3 staged"}
for i, n in names.items():
    ok = t[:, i] != 0
    if ok.any():
        r = (t[ok, i] - g0).astype(np.float64)
        print(f"{n:15s} mean {r.mean():9.0f}  min {r.min():9.0f}  p50 {np.median(r):9.0f}  max {r.max():9.0f}")
