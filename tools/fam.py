"""ms per UNet forward (B=8, L=64) and HIP-event time per kernel family for the given compute dtypes:  python tools/fam.py bf16x3 fp32 bf16"""
import sys, time, ctypes as C, torch
sys.path.insert(0, "latent-diffusion-segmentation_amd")
from ldmseg_amd import weights, _lib
from ldmseg_amd.models import UNet
usd = weights.generate(weights.unet_schema(12, False), seed=0)
x = torch.randn(8, 12, 64, 64, device="cuda")
lib = _lib.lib()
for mode in sys.argv[1:]:
    u = UNet(usd, 12, "cuda:0", mode)
    for _ in range(2): u(x, 499)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 5
    for _ in range(n): u(x, 499)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    lib.ldmseg_profile_reset(); lib.ldmseg_profile_enable(1)
    for _ in range(2): u(x, 499)
    torch.cuda.synchronize(); lib.ldmseg_profile_enable(0)
    fam = {}
    for i, nme in enumerate(["igemm", "attention", "groupnorm", "layernorm", "other"]):
        n_, m_, fl, by = C.c_int64(), C.c_double(), C.c_double(), C.c_double()
        lib.ldmseg_profile_read(i, C.byref(n_), C.byref(m_), C.byref(fl), C.byref(by))
        fam[nme] = round(m_.value / 2, 3)
    lib.ldmseg_profile_reset()
    print(mode, "ms per forward B=8 L=64:", round(ms, 3), fam, flush=True)
    del u
