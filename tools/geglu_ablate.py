import sys, ctypes as C, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "latent-diffusion-segmentation_amd"))
from ldmseg_amd import _lib
L = _lib.lib()
M, K, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])   # N = GEMM columns (2x outputs)
x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.05; b = torch.zeros(N, device="cuda")
out = torch.empty(M, N // 2, device="cuda")
P = lambda t: C.c_void_p(t.data_ptr())
for dbg in [int(v, 0) for v in os.environ.get("DBGS", "0,32,16,4,1,8,20,17").split(",")]:
    L.ldmseg_debug_set(1, (int(os.environ.get("POLICY", "61")) << 8) | dbg)
    for _ in range(2): L.ldmseg_op_linear(P(x), P(w), P(b), None, None, M, M, K, N, 1, 0, 1, 1, P(out), None)
    torch.cuda.synchronize()
