import sys, os, ctypes as C, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "latent-diffusion-segmentation_amd"))
from ldmseg_amd import _lib
L = _lib.lib()
B, Ci, H, Co = 8, int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x = torch.randn(B, Ci, H, H, device="cuda"); w = torch.randn(Co, Ci, 3, 3, device="cuda") * 0.02; b = torch.zeros(Co, device="cuda")
out = torch.empty(B, Co, H, H, device="cuda")
P = lambda t: C.c_void_p(t.data_ptr())
for dbg in (0, 1, 4, 16, 1 | 16, 4 | 16, 1 | 4 | 16):
    L.ldmseg_debug_set(1, (29 << 8) | dbg)
    for _ in range(2): L.ldmseg_op_conv2d(P(x), None, P(w), P(b), B, Ci, 0, H, H, Co, 3, 1, 0, 1, P(out), None)
    torch.cuda.synchronize()
