"""Phase ablation of one igemm launch shape under the shipped dispatch (needs scratch/lib_ablate.so = -DLDMSEG_IGEMM_ABLATE):
    LDMSEG_HIP_LIB=scratch/lib_ablate.so bash tools/run_ablate.sh igemm_ablate_shape.py <index into SHAPES> [B]
prints (through run_ablate.sh) the igemm kernel durations for: full, no DMA (1), no MFMA (4), no LDS reads + MFMA (8), no epilogue (16),
no DMA + no epilogue (17), no MFMA + no epilogue (20), nothing but the skeleton (1|8|16 = 25); two launches each."""
import ctypes as C
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion-segmentation_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from ldmseg_amd import _lib  # noqa: E402
from test_igemm_shapes_gpu import SHAPES  # noqa: E402

L = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
H, Ci, Ci2, Co, k, stride, up, geglu, use_res, use_rb = SHAPES[int(sys.argv[1])]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ct = Ci + Ci2
x = torch.randn(B, Ci, H, H, device="cuda")
x2 = torch.randn(B, Ci2, H, H, device="cuda") if Ci2 else None
w = torch.randn(Co, ct, k, k, device="cuda") / (ct * k * k) ** 0.5
b = torch.randn(Co, device="cuda")
Hl = 2 * H if up else H
Ho = (Hl - 1) // 2 + 1 if (k == 3 and stride == 2) else Hl
cout = Co // 2 if geglu else Co
res = torch.randn(B, cout, Ho, Ho, device="cuda") if use_res else None
rb = torch.randn(B, Co, device="cuda") if use_rb else None
us = C.c_float()
pol = L.ldmseg_debug_get(-1)
for dbg in (0, 1, 4, 8, 16, 17, 20, 25):
    L.ldmseg_debug_set(1, pol | dbg)
    L.ldmseg_bench_igemm(P(x), P(x2), P(w), P(b), P(res), P(rb), B, Ci, Ci2, H, H, Co, k, stride, up, geglu, 0, 0, 1, 1, C.byref(us), None)
    torch.cuda.synchronize()
L.ldmseg_debug_set(1, pol)
print(_lib.igemm_last_kernel())
