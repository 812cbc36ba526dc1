#!/usr/bin/env python
"""Kernel-level timing of the UNet's launch shapes on the MI355X (tuning tool; product code never calls this).

    python tools/kbench.py attn [N C ...]        attention at B=8
    python tools/kbench.py igemm                 the B=8, L=64 layer shapes (same list as tests/test_igemm_shapes_gpu.py)

Prints microseconds per launch (HIP events around back-to-back launches), achieved TFLOP/s and the instantiation."""
import ctypes as C
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion-segmentation_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from ldmseg_amd import _lib  # noqa: E402

L = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
DT = {"bf16": 1, "fp32": 0}


def attn(B, N, Cc, dt="bf16", iters=20):
    qkv = torch.randn(B, N, 3 * Cc, device="cuda")
    us = C.c_float()
    _lib.check(L.ldmseg_bench_attention(P(qkv), B, N, Cc, 8, DT[dt], iters, C.byref(us), None), "bench_attention")
    fl = 4.0 * B * 8 * N * N * (Cc / 8)
    print(f"attention B={B} N={N} C={Cc} {dt}: {us.value:9.1f} us  {fl / us.value / 1e6:7.1f} TF/s")
    return us.value


def igemm(case, dt="bf16", iters=20, B=8):
    H, Ci, Ci2, Co, k, stride, up, geglu, use_res, use_rb = case
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
    if os.environ.get("ROT"):        # weights rotated over copies so that they come from HBM like in the real forward
        L.ldmseg_debug_set(6, max(1, min(24, (400 << 20) // (Co * ct * k * k * 2))))
    _lib.check(L.ldmseg_bench_igemm(P(x), P(x2), P(w), P(b), P(res), P(rb), B, Ci, Ci2, H, H, Co, k, stride, up, geglu, 0, 0,
                                    DT[dt], iters, C.byref(us), None), "bench_igemm")
    fl = 2.0 * B * Ho * Ho * Co * ct * k * k
    name = _lib.igemm_last_kernel()
    print(f"M={B * Ho * Ho:6d} N={Co:5d} K={ct * k * k:6d} k={k} s={stride} up={up} geglu={geglu} res={use_res} rb={use_rb}: "
          f"{us.value:8.1f} us {fl / us.value / 1e6:7.1f} TF/s  {name}")
    return us.value, fl


def gn(B, Cc, C2, HW, dt="bf16", iters=30, variant=0):
    ga = torch.ones(Cc + C2, device="cuda")
    be = torch.zeros(Cc + C2, device="cuda")
    us = C.c_float()
    L.ldmseg_debug_set(8, variant)
    _lib.check(L.ldmseg_bench_groupnorm(P(ga), P(be), B, Cc, C2, HW, 1, DT[dt], iters, C.byref(us), None), "bench_groupnorm")
    L.ldmseg_debug_set(8, 0)
    byts = 2.0 * B * HW * (Cc + C2) * (2 if dt == "bf16" else 4)
    print(f"groupnorm B={B} C={Cc}+{C2} HW={HW} {dt} variant={variant}: {us.value:8.1f} us  {byts / us.value / 1e6:6.2f} TB/s (1 read + 1 write)")
    return us.value


# the GroupNorm launch shapes of a B=8, L=64 forward: (C, C2, HW, launches per forward)
GN_SHAPES = [(320, 0, 4096, 13), (320, 320, 4096, 2), (640, 320, 4096, 1), (320, 0, 1024, 1), (640, 0, 1024, 8), (640, 640, 1024, 1),
             (640, 320, 1024, 1), (1280, 640, 1024, 1), (640, 0, 256, 1), (1280, 0, 256, 8), (1280, 1280, 256, 2), (1280, 640, 256, 1),
             (1280, 0, 64, 11), (1280, 1280, 64, 3)]


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "attn"
    dt = os.environ.get("DT", "bf16")
    if what == "attn":
        shapes = [(8, 4096, 320), (8, 1024, 640), (8, 256, 1280), (8, 64, 1280), (4, 16384, 320), (4, 4096, 640)]
        if len(sys.argv) > 3:
            shapes = [(int(os.environ.get("B", 8)), int(sys.argv[2]), int(sys.argv[3]))]
        for s in shapes:
            attn(*s, dt=dt)
    elif what == "gn":
        Bq = int(os.environ.get("B", 8))
        for variant in [int(v) for v in os.environ.get("VARIANTS", "0").split(",")]:
            tot = 0.0
            for (c, c2, hw, n) in GN_SHAPES:
                tot += n * gn(Bq, c, c2, hw, dt=dt, variant=variant)
            print(f"variant {variant}: {tot:.1f} us per forward over {sum(s[3] for s in GN_SHAPES)} norms")
    elif what == "igemm1":            # one shape of the list (PMC passes, ablations): kbench.py igemm1 <index> [ln]
        from test_igemm_shapes_gpu import SHAPES
        if os.environ.get("DBG"):
            L.ldmseg_debug_set(1, int(os.environ["DBG"], 0))
        if len(sys.argv) > 3:
            L.ldmseg_debug_set(7, 1)
        if os.environ.get("CFG"):
            L.ldmseg_debug_set(5, int(os.environ["CFG"]))
        if os.environ.get("UP4"):            # upsampler convs as four 2x2 phase convs (debug key 21)
            L.ldmseg_debug_set(21, int(os.environ["UP4"]))
        if os.environ.get("WS"):             # weight-streaming kernel: mode | (max M / 4) << 8 | (min K tiles) << 20
            L.ldmseg_debug_set(17, int(os.environ["WS"], 0))
        igemm(SHAPES[int(sys.argv[2])], dt, iters=int(os.environ.get("ITERS", 20)))
    else:
        if os.environ.get("POLICY"):
            L.ldmseg_debug_set(1, int(os.environ["POLICY"]) << 8)
        from test_igemm_shapes_gpu import SHAPES
        tot_us = tot_fl = 0.0
        for c in SHAPES:
            u, f = igemm(c, dt, B=int(os.environ.get("B", 8)))
            tot_us += u
            tot_fl += f
        print(f"sum over distinct shapes: {tot_us:.1f} us, {tot_fl / tot_us / 1e6:.1f} TF/s")
