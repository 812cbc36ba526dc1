"""GPU parity, per kernel: every gfx950 kernel is driven through the C ABI (ldmseg_op_*) and compared
with the torch-CPU fp32 op the reference executes at that point (the oracle's arithmetic primitives).
fp32 kernels must meet the north-star 1e-3 max-norm bound; bf16 kernels a bf16-rounding bound."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu

F32, BF16 = 0, 1
TOL = {F32: 1e-3, BF16: 3e-2}


@pytest.fixture(scope="module")
def L():
    from ldmseg_amd import _lib
    return _lib


def dev(t):
    return t.to("cuda", torch.float32).contiguous() if t is not None else None


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("case", [
    # B, Ci, Ci2, H, W, Co, k, stride, up
    (2, 64, 0, 16, 16, 320, 3, 1, 0),
    (1, 320, 0, 16, 16, 320, 3, 1, 0),
    (2, 128, 0, 16, 16, 64, 3, 2, 0),
    (1, 64, 0, 8, 8, 128, 3, 1, 1),
    (1, 128, 64, 8, 8, 160, 3, 1, 0),      # channel concat
    (1, 128, 64, 8, 8, 96, 1, 1, 0),       # conv_shortcut on a concat
    (3, 12, 0, 8, 8, 320, 3, 1, 0),        # conv_in style (tiny Cin, padded)
    (1, 320, 0, 8, 8, 4, 3, 1, 0),         # conv_out style (tiny Cout)
    (1, 64, 0, 5, 7, 32, 3, 1, 0),         # ragged M (35 rows)
    (2, 192, 0, 12, 12, 640, 1, 1, 0),
])
def test_conv2d(L, dt, case):
    B, Ci, Ci2, H, W, Co, k, stride, up = case
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    x = torch.randn(B, Ci, H, W, generator=g)
    x2 = torch.randn(B, Ci2, H, W, generator=g) if Ci2 else None
    w = torch.randn(Co, Ci + Ci2, k, k, generator=g) / ((Ci + Ci2) * k * k) ** 0.5
    b = torch.randn(Co, generator=g)
    xin = torch.cat([x, x2], 1) if Ci2 else x
    if dt == BF16:
        xin_r, w_r = bf16_round(xin), bf16_round(w)
    else:
        xin_r, w_r = xin, w
    if up:
        xin_r = F.interpolate(xin_r, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xin_r, w_r, b, stride=stride, padding=k // 2)
    out = torch.empty(ref.shape, device="cuda")
    dx, dx2, dw, db = dev(x), dev(x2), dev(w), dev(b)
    r = L.lib().ldmseg_op_conv2d(P(dx), P(dx2), P(dw), P(db), B, Ci, Ci2, H, W, Co, k, stride, up, dt, P(out), None)
    assert r == 0
    torch.cuda.synchronize()
    # inputs were rounded identically, so even bf16 only differs by accumulation order
    assert rel_err(out, ref) < (1e-3 if dt == BF16 else 2e-5), case


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("case", [
    # M, K, N, geglu, silu, resid, rowbias(rows_per_image), splits
    (256, 320, 960, 0, 0, False, 0, 1),
    (300, 128, 320, 0, 0, True, 0, 1),
    (256, 64, 640, 0, 1, False, 64, 1),
    (128, 320, 2560, 1, 0, False, 0, 1),     # GEGLU: [M, 8C] -> [M, 4C]
    (96, 1280, 320, 0, 0, True, 0, 1),
    (64, 2304, 1280, 0, 0, True, 16, 3),     # split-K
    (1, 64, 32, 0, 0, False, 0, 1),
])
def test_linear(L, dt, case):
    M, K, N, geglu, silu, use_res, rpi, splits = case
    g = torch.Generator().manual_seed(M * 7 + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    nout = N // 2 if geglu else N
    res = torch.randn(M, nout, generator=g) if use_res else None
    rb = torch.randn(M // rpi, N, generator=g) if rpi else None
    xr, wr = (bf16_round(x), bf16_round(w)) if dt == BF16 else (x, w)
    y = xr @ wr.t() + b
    if rb is not None:
        y = y + rb.repeat_interleave(rpi, 0)
    if geglu:
        a, gate = y.chunk(2, -1)
        y = a * F.gelu(gate)
    if res is not None:
        y = y + (bf16_round(res) if dt == BF16 else res)
    if silu:
        y = F.silu(y)
    out = torch.empty(M, nout, device="cuda")
    dx, dw, db, dres, drb = dev(x), dev(w), dev(b), dev(res), dev(rb)
    r = L.lib().ldmseg_op_linear(P(dx), P(dw), P(db), P(dres), P(drb), rpi if rpi else M, M, K, N,
                                 geglu, silu, splits, dt, P(out), None)
    assert r == 0
    torch.cuda.synchronize()
    # the kernel output itself is rounded to the storage type
    assert rel_err(out, y) < (8e-3 if dt == BF16 else 2e-5), case


@pytest.mark.parametrize("dt", [F32, BF16])
def test_geglu_gate_range(L, dt):
    """GEGLU with gates spread over [-12, 12]: the bf16 epilogue evaluates erf by a polynomial on |x| <= 3*sqrt(2) and
    saturates beyond (common.h gelu_erf_bf16_f4, |error| <= 5e-5 absolute); the fp32 one keeps Abramowitz-Stegun 7.1.26
    (1.5e-7).  An identity weight makes value and gate columns exactly the (storage-rounded) inputs."""
    M, K, N = 512, 256, 256                              # (GEGLU launches take N in multiples of 128 GEMM columns)
    g = torch.Generator().manual_seed(5)
    gate = torch.linspace(-12.0, 12.0, M * 128).reshape(M, 128)[torch.randperm(M, generator=g)]
    val = torch.randn(M, 128, generator=g)
    x = torch.cat([val, gate], 1)                       # ff.net.0.proj rows are [value | gate]
    w = torch.eye(N, K)
    b = torch.zeros(N)
    xr = bf16_round(x) if dt == BF16 else x
    a, gt = xr.chunk(2, -1)
    ref = a * F.gelu(gt)
    out = torch.empty(M, N // 2, device="cuda")
    assert L.lib().ldmseg_op_linear(P(dev(x)), P(dev(w)), P(dev(b)), None, None, M, M, K, N, 1, 0, 1, dt, P(out), None) == 0
    torch.cuda.synchronize()
    err = (out.cpu() - ref).abs()
    if dt == BF16:
        # output rounding to bf16 (2^-9 relative) + the 5e-5 absolute bound of the polynomial (scaled by |value|)
        assert bool((err <= ref.abs() * 2.0 ** -8 + 1e-4 * (1 + a.abs())).all()), float(err.max())
    else:
        assert float(err.max()) < 2e-5 * float(ref.abs().max())


@pytest.mark.parametrize("case", [
    # B, C, HW: the two-launch scheme, the single-launch register-resident kernel, the small-map kernel
    (2, 320, 4096), (8, 1280, 256), (8, 1280, 16), (8, 320, 4096),
])
def test_groupnorm_large_mean(L, case):
    """Channel means 1000x the standard deviation (fp32 mode): E[x^2] - E[x]^2 in fp32 would lose the variance entirely
    (1e6 against 1 at 6e-8 relative precision); the kernels combine (count, mean, M2) triples instead (norm.hip chan_add)."""
    B, Cc, HW = case
    g = torch.Generator().manual_seed(HW)
    x = (torch.randn(B, Cc, HW, generator=g, dtype=torch.float64) + 1000.0).float()
    gamma = 1 + 0.1 * torch.randn(Cc, generator=g)
    beta = 0.1 * torch.randn(Cc, generator=g)
    ref = F.group_norm(x.double(), 32, gamma.double(), beta.double(), 1e-5).float()
    out = torch.empty(B, Cc, HW, device="cuda")
    assert L.lib().ldmseg_op_groupnorm(P(dev(x)), None, P(dev(gamma)), P(dev(beta)), B, Cc, 0, HW, 1e-5, 0, F32, P(out), None) == 0
    torch.cuda.synchronize()
    # the inputs themselves carry 1000 * 6e-8 = 6e-5 of rounding relative to a unit deviation
    assert rel_err(out, ref) < 1e-3, case


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("case", [(8, 1280, 64), (8, 1280, 256), (8, 2560, 64), (8, 640, 1024)])
def test_groupnorm_outlier_shift_sample(L, dt, case):
    """The single-launch kernel of the small maps shifts its one-pass sums by the group's first element; when that element
    is an outlier (here 3e3 standard deviations away, and in a second group a plain large offset) the second moment is
    taken again around the mean - the result must not depend on what the sample happened to be."""
    B, Cc, HW = case
    g = torch.Generator().manual_seed(HW + Cc)
    x = torch.randn(B, Cc, HW, generator=g)
    cpg = Cc // 32
    x[:, 0 * cpg, 0] = 3000.0            # group 0: the shift sample itself is the outlier
    x[:, 5 * cpg:6 * cpg, :] += 200.0    # group 5: large common offset, ordinary sample
    x[:, 7 * cpg, 0] = -2500.0
    gamma = 1 + 0.1 * torch.randn(Cc, generator=g)
    beta = 0.1 * torch.randn(Cc, generator=g)
    xin = bf16_round(x) if dt == BF16 else x
    ref = F.group_norm(xin.double(), 32, gamma.double(), beta.double(), 1e-5).float()
    out = torch.empty(B, Cc, HW, device="cuda")
    assert L.lib().ldmseg_op_groupnorm(P(dev(x)), None, P(dev(gamma)), P(dev(beta)), B, Cc, 0, HW, 1e-5, 0, dt, P(out), None) == 0
    torch.cuda.synchronize()
    # groups WITHOUT the planted outliers carry the usual tolerance; the planted ones have outputs of ~50 at the outlier
    # pixel and ~0 elsewhere - compare them against their own scale
    assert rel_err(out, ref) < (8e-3 if dt == BF16 else 1e-4), case


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("case", [
    # B, C, C2, HW, eps, silu
    (2, 320, 0, 256, 1e-5, 1),
    (1, 640, 320, 64, 1e-5, 1),       # concat, group straddles the source boundary (cpg = 30)
    (1, 1280, 640, 16, 1e-5, 1),      # cpg = 60
    (3, 256, 0, 1024, 1e-6, 1),       # seg-VAE GN (cpg = 8)
    (1, 2560, 0, 4, 1e-5, 0),         # 2x2 map, widest concat
    (2, 1280, 0, 1, 1e-6, 0),         # single pixel
    # enough (image, group block) workgroups for the single-launch register-resident kernel:
    (8, 1280, 0, 256, 1e-5, 1),       # one group per workgroup (80-B runs in bf16)
    (8, 640, 0, 1024, 1e-5, 1),       # two groups per workgroup in bf16, 21 vectors per thread
    (8, 1280, 640, 256, 1e-5, 1),     # concat, cpg = 60: group pairs straddle the source boundary
    (4, 2560, 0, 64, 1e-6, 0),
    (8, 256, 0, 100, 1e-6, 1),        # one vector per pixel, ragged pixel count
    (8, 1280, 0, 64, 1e-5, 1),        # 8x8 level: 2 vectors per thread
    (8, 2560, 0, 64, 1e-5, 1),
])
@pytest.mark.parametrize("variant", [0, 4])        # one-pass single-barrier kernel of the small maps / the two-pass one it replaced
def test_groupnorm(L, dt, case, variant):
    B, Cc, C2, HW, eps, silu = case
    g = torch.Generator().manual_seed(Cc + HW)
    x = torch.randn(B, Cc, HW, generator=g) * 2 + 0.5
    x2 = torch.randn(B, C2, HW, generator=g) - 1.0 if C2 else None
    gamma = 1 + 0.1 * torch.randn(Cc + C2, generator=g)
    beta = 0.1 * torch.randn(Cc + C2, generator=g)
    xin = torch.cat([x, x2], 1) if C2 else x
    if dt == BF16:
        xin = bf16_round(xin)
    ref = F.group_norm(xin, 32, gamma, beta, eps)
    if silu:
        ref = F.silu(ref)
    out = torch.empty(B, Cc + C2, HW, device="cuda")
    dx, dx2, dg, db = dev(x), dev(x2), dev(gamma), dev(beta)
    L.lib().ldmseg_debug_set(8, variant)
    try:
        r = L.lib().ldmseg_op_groupnorm(P(dx), P(dx2), P(dg), P(db), B, Cc, C2, HW, eps, silu, dt, P(out), None)
    finally:
        L.lib().ldmseg_debug_set(8, 0)
    assert r == 0
    torch.cuda.synchronize()
    assert rel_err(out, ref) < (8e-3 if dt == BF16 else 5e-5), case


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("case", [
    # B, Ci, H, Co, splits, temb row, silu: resnet conv1 -> norm2 where the conv runs as K slices
    (8, 1280, 8, 1280, 8, 1, 1),      # 8x8 level: 2 vectors per thread
    (8, 640, 16, 1280, 2, 1, 1),      # 16x16 level, first resnet of the level (640 -> 1280 channels)
    (8, 1280, 16, 1280, 4, 1, 1),
    (8, 2560, 8, 1280, 8, 1, 1),      # up-path conv1 on the widest concat
    (8, 320, 16, 640, 3, 0, 0),       # 640 channels: two groups per workgroup in bf16; no time-embedding row, no SiLU
    (8, 256, 8, 320, 2, 1, 1),        # cpg 10: 4-group blocks are not available -> -4, the engine keeps finish and norm apart
])
def test_conv_groupnorm_fused_finish(L, dt, case):
    """launch_finish_groupnorm: K-slice sum + bias + time-embedding row + GroupNorm (+SiLU) in one launch, against the torch
    ops; the fused path keeps fp32 from the accumulators to the normalisation, so it is at least as close as conv-then-norm."""
    B, Ci, H, Co, splits, use_rb, silu = case
    g = torch.Generator().manual_seed(Ci + H + Co)
    x = torch.randn(B, Ci, H, H, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5
    b = torch.randn(Co, generator=g)
    rb = 2.0 * torch.randn(B, Co, generator=g) if use_rb else None
    gamma = 1 + 0.1 * torch.randn(Co, generator=g)
    beta = 0.1 * torch.randn(Co, generator=g)
    rnd = bf16_round if dt == BF16 else (lambda t: t)
    torch.set_num_threads(32)
    h = F.conv2d(rnd(x), rnd(w), b, padding=1)
    if rb is not None:
        h = h + rb[:, :, None, None]
    ref = F.group_norm(h, 32, gamma, beta, 1e-5)
    if silu:
        ref = F.silu(ref)
    out = torch.empty(ref.shape, device="cuda")
    dx, dw, db, drb, dg, dbe = dev(x), dev(w), dev(b), dev(rb), dev(gamma), dev(beta)      # (kept alive until the sync below)
    r = L.lib().ldmseg_op_conv_groupnorm(P(dx), P(dw), P(db), P(drb), P(dg), P(dbe), B, Ci, H, H, Co, 1e-5, silu, splits, dt,
                                         P(out), None)
    if Co == 320 and dt == BF16:
        assert r == -4
        return
    assert r == 0, (r, L.lib().ldmseg_last_error())
    torch.cuda.synchronize()
    assert rel_err(out, ref) < (8e-3 if dt == BF16 else 2e-4), case


@pytest.mark.parametrize("case", [
    # B, C, C2, HW, eps, silu, channel offset: maps whose (image, group) slice fits 20 dwords per thread of a 512-thread workgroup
    (8, 640, 0, 1024, 1e-5, 1, 0.5),       # the 32x32 level of the UNet: 11 norms of a B = 8 forward
    (16, 640, 0, 1024, 1e-6, 0, 0.5),      # batch 16: two workgroups per CU
    (4, 640, 0, 1024, 1e-5, 1, 0.5),       # the fewest workgroups the rule takes
    (8, 512, 0, 1156, 1e-5, 1, 0.5),       # 34 x 34, cpg = 16: ragged last access per thread
    (8, 256, 0, 2500, 1e-6, 0, 0.5),       # 50 x 50, cpg = 8: two accesses per pixel
    (8, 640, 0, 1024, 1e-5, 0, 1000.0),    # channel mean 1000x the deviation: the statistics are two passes over the registers
    (8, 1280, 0, 256, 1e-5, 1, 0.5),       # the 16x16 level
    (8, 1280, 1280, 256, 1e-5, 1, 0.5),    # torch.cat([h, skip]): cpg = 80, every group inside one source
    (8, 1280, 640, 256, 1e-5, 1, 0.5),     # cpg = 60: group 21 straddles the source boundary
    (8, 640, 320, 400, 1e-6, 0, 0.5),      # cpg = 30 is not a multiple of 4: the rule declines, the older kernels run (same bounds)
])
def test_groupnorm_one_workgroup_per_group(L, case):
    """Round 6: bf16 maps whose (image, group) slice is small enough run with ONE workgroup per (image, group) - the slice in
    registers, no hand-off between workgroups (gn_group_kernel, norm.hip), also over torch.cat([x, x2], 1).  Against F.group_norm on
    the rounded input, against the kernels it replaces on these shapes (tuning bit 5 of debug key 8), deterministic launch after launch."""
    B, Cc, C2, HW, eps, silu, off = case
    g = torch.Generator().manual_seed(Cc + HW + B + C2)
    x = torch.randn(B, Cc, HW, generator=g) * 2 + off
    x[:, :, : HW // 3] += 3.0
    x2 = torch.randn(B, C2, HW, generator=g) - 1.0 if C2 else None
    gamma = 1 + 0.1 * torch.randn(Cc + C2, generator=g)
    beta = 0.1 * torch.randn(Cc + C2, generator=g)
    xin = torch.cat([x, x2], 1) if C2 else x
    ref = F.group_norm(bf16_round(xin).double(), 32, gamma.double(), beta.double(), eps)
    if silu:
        ref = F.silu(ref)
    out = torch.empty(B, Cc + C2, HW, device="cuda")
    dx, dx2, dg, db = dev(x), dev(x2), dev(gamma), dev(beta)
    lib = L.lib()
    outs = {}
    try:
        for variant in (0, 32):
            lib.ldmseg_debug_set(8, variant)
            for rep in range(2):
                assert lib.ldmseg_op_groupnorm(P(dx), P(dx2), P(dg), P(db), B, Cc, C2, HW, eps, silu, BF16, P(out), None) == 0
                torch.cuda.synchronize()
                outs[variant, rep] = out.cpu()
    finally:
        lib.ldmseg_debug_set(8, 0)
    tol = 8e-3 if off < 10 else 2e-2          # (bf16 inputs around 1000 carry 4 units of rounding each: the reference sees them too)
    assert rel_err(outs[0, 0], ref) < tol and rel_err(outs[32, 0], ref) < tol, case
    assert torch.equal(outs[0, 0], outs[0, 1])
    assert float((outs[0, 0].double() - outs[32, 0].double()).abs().max()) <= 2 ** -7 * float(ref.abs().max()), case   # one bf16 ulp of the largest output


@pytest.mark.parametrize("variant", [0, 1])        # cooperative one-pass kernel, and the two-launch path it replaces
@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("case", [
    # B, C, C2, HW, eps, silu: the 64x64 maps of the UNet and the 128x128 ones of the 1024x1024 configuration
    (8, 320, 0, 4096, 1e-5, 1),       # cpg 10: 4 groups per block in bf16 (5 vectors per pixel), 4 splits -> 256 workgroups
    (4, 320, 320, 4096, 1e-5, 1),     # concat 640, cpg 20: vectors on either side of the source boundary
    (16, 320, 0, 4096, 1e-6, 0),      # batch 16: 2 splits
    (8, 320, 0, 3969, 1e-6, 0),       # 63 x 63: ragged splits and a ragged last trip
    (4, 320, 0, 16384, 1e-5, 1),      # 128 x 128: 8 splits
    (1, 640, 0, 4096, 1e-5, 1),       # one image
])
def test_groupnorm_cooperative(L, dt, case, variant):
    B, Cc, C2, HW, eps, silu = case
    g = torch.Generator().manual_seed(Cc + HW + B)
    x = torch.randn(B, Cc, HW, generator=g) * 2 + 0.5
    x[:, :, : HW // 3] += 3.0                      # the splits of a slab see different means: exercises the Chan combination
    x2 = torch.randn(B, C2, HW, generator=g) - 1.0 if C2 else None
    gamma = 1 + 0.1 * torch.randn(Cc + C2, generator=g)
    beta = 0.1 * torch.randn(Cc + C2, generator=g)
    xin = torch.cat([x, x2], 1) if C2 else x
    if dt == BF16:
        xin = bf16_round(xin)
    ref = F.group_norm(xin, 32, gamma, beta, eps)
    if silu:
        ref = F.silu(ref)
    out = torch.empty(B, Cc + C2, HW, device="cuda")
    dx, dx2, dg, db = dev(x), dev(x2), dev(gamma), dev(beta)
    L.lib().ldmseg_debug_set(8, variant)
    try:
        outs = []
        for rep in range(3):                        # consecutive launches reuse the hand-off records under new tags
            assert L.lib().ldmseg_op_groupnorm(P(dx), P(dx2), P(dg), P(db), B, Cc, C2, HW, eps, silu, dt, P(out), None) == 0
            torch.cuda.synchronize()
            outs.append(out.clone())
    finally:
        L.lib().ldmseg_debug_set(8, 0)
    assert rel_err(outs[0], ref) < (8e-3 if dt == BF16 else 5e-5), case
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])       # deterministic, launch after launch


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("case", [(8, 320, 0, 4096, 1e-5, 1), (4, 320, 320, 4096, 1e-5, 1), (8, 320, 0, 3969, 1e-6, 0),
                                  (4, 320, 0, 16384, 1e-5, 1), (8, 640, 0, 1024, 1e-5, 1)])
def test_groupnorm_cooperative_without_partners(L, dt, case):
    """The cooperative kernel must not depend on its partners being co-resident: a
    workgroup whose partners' records do not arrive in time computes them itself from the partners' pixels.  Forcing that
    path for EVERY workgroup (debug key 10), and a zero-length poll (key 11: whoever is late is recomputed), must give
    bit-identical results to the hand-off path - and never NaNs."""
    B, Cc, C2, HW, eps, silu = case
    g = torch.Generator().manual_seed(Cc + HW + B + 1)
    x = torch.randn(B, Cc, HW, generator=g) * 2 + 0.5
    x[:, :, : HW // 3] += 3.0
    x2 = torch.randn(B, C2, HW, generator=g) - 1.0 if C2 else None
    gamma = 1 + 0.1 * torch.randn(Cc + C2, generator=g)
    beta = 0.1 * torch.randn(Cc + C2, generator=g)
    out = torch.empty(B, Cc + C2, HW, device="cuda")
    dx, dx2, dg, db = dev(x), dev(x2), dev(gamma), dev(beta)
    lib = L.lib()

    def run():
        out.fill_(float("nan"))
        assert lib.ldmseg_op_groupnorm(P(dx), P(dx2), P(dg), P(db), B, Cc, C2, HW, eps, silu, dt, P(out), None) == 0
        torch.cuda.synchronize()
        return out.clone()
    lib.ldmseg_debug_set(8, 32)                          # (round 6: the 32x32 x 640 map would otherwise take the one-workgroup-per-group kernel)
    base = run()
    assert torch.isfinite(base).all()
    n0 = lib.ldmseg_debug_get(10)
    try:
        lib.ldmseg_debug_set(10, 1)                      # nobody waits for anybody
        forced = [run() for _ in range(3)]
        n1 = lib.ldmseg_debug_get(10)
        lib.ldmseg_debug_set(10, 0)
        lib.ldmseg_debug_set(11, 0)                      # one look at the partners' records, then self-compute what is missing
        hurried = [run() for _ in range(3)]
    finally:
        lib.ldmseg_debug_set(10, 0)
        lib.ldmseg_debug_set(11, 100)
        lib.ldmseg_debug_set(8, 0)
    # (fp32 at 128 x 128: 64 slabs x 8 splits exceed one workgroup per CU - that shape stays on the two-launch scheme)
    assert n1 > n0 or dt == F32, "the self-computing path did not run"     # (several fp32 shapes are not cooperative)
    for o in forced + hurried:
        assert torch.equal(o, base), case
    lib.ldmseg_debug_set(8, 32)
    try:
        assert torch.equal(run(), base)
    finally:
        lib.ldmseg_debug_set(8, 0)


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("M,Cc,eps,silu", [(77, 320, 1e-5, 0), (64, 640, 1e-5, 0), (33, 1280, 1e-5, 0), (50, 256, 1e-6, 1)])
def test_layernorm(L, dt, M, Cc, eps, silu):
    g = torch.Generator().manual_seed(M + Cc)
    x = torch.randn(M, Cc, generator=g) * 3 + 1
    gamma = 1 + 0.1 * torch.randn(Cc, generator=g)
    beta = 0.1 * torch.randn(Cc, generator=g)
    xin = bf16_round(x) if dt == BF16 else x
    ref = F.layer_norm(xin, (Cc,), gamma, beta, eps)
    if silu:
        ref = F.silu(ref)
    out = torch.empty(M, Cc, device="cuda")
    dx, dg, db = dev(x), dev(gamma), dev(beta)
    assert L.lib().ldmseg_op_layernorm(P(dx), P(dg), P(db), M, Cc, eps, silu, dt, P(out), None) == 0
    torch.cuda.synchronize()
    assert rel_err(out, ref) < (8e-3 if dt == BF16 else 2e-5)


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("M,K,N,geglu", [
    (77, 320, 960, 0), (100, 640, 5120, 1), (300, 1280, 3840, 0), (1, 320, 2560, 1),   # ragged M (the full-size shapes: test_igemm_shapes_gpu.py)
])
def test_layernorm_folded_into_linear(L, dt, M, K, N, geglu):
    """LayerNorm -> Linear / GEGLU as the engine runs it (statistics pass + GEMM on the raw input, gamma folded into the
    weights, rstd*(acc - mean*c1) + c2 epilogue) against F.layer_norm + F.linear; the inputs have a large common offset
    (mean 3, std 1.5) so that the mean cancellation in the epilogue is exercised."""
    torch.set_num_threads(32)
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g) * 1.5 + 3.0
    gamma = 1 + 0.2 * torch.randn(K, generator=g)
    beta = 0.2 * torch.randn(K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    xr = bf16_round(x) if dt == BF16 else x
    y = F.linear(F.layer_norm(xr, (K,), gamma, beta, 1e-5), w, b)
    if geglu:
        a, gate = y.chunk(2, -1)
        y = a * F.gelu(gate)
    out = torch.empty(y.shape, device="cuda")
    dx, dg, db, dw, dbias = dev(x), dev(gamma), dev(beta), dev(w), dev(b)
    assert L.lib().ldmseg_op_ln_linear(P(dx), P(dg), P(db), P(dw), P(dbias), M, K, N, 1e-5, geglu, dt, P(out), None) == 0
    torch.cuda.synchronize()
    # bf16: gamma*W and the output are rounded to bf16 (the unfolded form rounds LN(x) and W instead)
    assert rel_err(out, y) < (1.5e-2 if dt == BF16 else 1e-4), L.igemm_last_kernel()


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("B,N,Cc", [(1, 256, 320), (2, 64, 320), (1, 1024, 320), (1, 256, 640), (2, 4, 1280),
                                    (1, 64, 1280), (1, 100, 640), (1, 320, 1280), (1, 4096, 320)])
def test_attention(L, dt, B, N, Cc):
    g = torch.Generator().manual_seed(N + Cc)
    qkv = torch.randn(B, N, 3 * Cc, generator=g)
    qkv[:, :, :Cc] *= 2.0                  # sharper softmax
    qkv[0, N // 2, Cc:Cc + 40] += 6.0      # one dominant key (forces the running-max rescale path)
    src = bf16_round(qkv) if dt == BF16 else qkv
    q, k, v = src.chunk(3, -1)
    d = Cc // 8
    q = q.view(B, N, 8, d).transpose(1, 2)
    k = k.view(B, N, 8, d).transpose(1, 2)
    v = v.view(B, N, 8, d).transpose(1, 2)
    ref = (torch.softmax((q.double() @ k.double().transpose(-1, -2)) * d ** -0.5, -1) @ v.double())
    ref = ref.transpose(1, 2).reshape(B, N, Cc).float()
    out = torch.empty(B, N, Cc, device="cuda")
    dq = dev(qkv)
    assert L.lib().ldmseg_op_attention(P(dq), B, N, Cc, 8, dt, P(out), None) == 0
    torch.cuda.synchronize()
    assert rel_err(out, ref) < (2e-2 if dt == BF16 else 2e-5)   # bf16: scaled Q and P are rounded to bf16 on top of the operands


def attention_ref(src, B, N, Cc, model_q_rounding=False):
    """fp64 softmax(q k^T d^-1/2) v per head.  model_q_rounding: the kernels fold d^-1/2 log2(e) into Q and round the
    product to bf16 once more; with scores of magnitude in the hundreds that rounding alone moves the probabilities by
    tens of percent, so tests that probe such scores model it (fp32 multiply, RNE to bf16, base-2 softmax)."""
    q, k, v = src.chunk(3, -1)
    d = Cc // 8
    if model_q_rounding:
        sc = torch.tensor(d ** -0.5, dtype=torch.float32) * torch.tensor(1.4426950408889634, dtype=torch.float32)
        q = bf16_round(q.float() * sc)
    q = q.view(B, N, 8, d).transpose(1, 2).double()
    k = k.view(B, N, 8, d).transpose(1, 2).double()
    v = v.view(B, N, 8, d).transpose(1, 2).double()
    z = q @ k.transpose(-1, -2)
    if model_q_rounding:
        z = z - z.max(-1, keepdim=True)[0]
        p = torch.exp2(z)
        ref = (p / p.sum(-1, keepdim=True)) @ v
    else:
        ref = torch.softmax(z * d ** -0.5, -1) @ v
    return ref.transpose(1, 2).reshape(B, N, Cc).float()


@pytest.fixture
def attn_variant(L):
    """Select an attention kernel variant (debug key 2) for one test; restored to the shipped choice (0) afterwards."""
    lib = L.lib()
    yield lambda v: lib.ldmseg_debug_set(2, v)
    lib.ldmseg_debug_set(2, 0)


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14])
@pytest.mark.parametrize("B,N,Cc", [(2, 1024, 320), (1, 200, 640), (1, 4096, 320), (8, 1024, 640), (8, 1024, 320), (1, 200, 320), (3, 1000, 320),
                                    (1, 33, 320)])
def test_attention_variants(L, attn_variant, variant, B, N, Cc):
    """Every selectable bf16 attention kernel (attention3.hip variants 1/4/5/6/7/8/9/10 - 4- and 8-wave workgroups, row
    maxima looked at on every tile or every 4th / 16th -, attention4.hip 11..14 - head dim 40 on 32x32x16 score blocks, 8 / 4
    waves, lazy / every tile -, attention.hip 2/3, and the shipped rule 0) vs the fp64 reference.  The batch-8 shapes have
    enough workgroups for the shipped choice to take its 8-wave form; 200 / 1000 / 33 tokens end in ragged key tiles and
    partly empty query blocks."""
    g = torch.Generator().manual_seed(N + Cc + variant)
    qkv = torch.randn(B, N, 3 * Cc, generator=g)
    qkv[:, :, :Cc] *= 2.0
    ref = attention_ref(bf16_round(qkv), B, N, Cc)
    out = torch.empty(B, N, Cc, device="cuda")
    dq = dev(qkv)
    attn_variant(variant)
    assert L.lib().ldmseg_op_attention(P(dq), B, N, Cc, 8, BF16, P(out), None) == 0
    torch.cuda.synchronize()
    assert rel_err(out, ref) < 1.5e-2, variant


@pytest.mark.parametrize("case", ["late_spike", "negative_first_tile", "growing", "huge"])
@pytest.mark.parametrize("Cc", [320, 640])
def test_attention_running_max_paths(L, case, Cc):
    """Inputs that force the rare paths of the running-maximum logic (attention3.hip folds the maximum into the matrix
    product and moves it only when a row outgrows it by 2^6): a dominant key that appears in a late tile, scores that
    are all very negative in the first tile, row maxima that keep growing tile after tile, and scores of magnitude
    > 256 (where the bf16 grid the folded maximum lives on has a spacing of 2 and more)."""
    B, N, d = 1, 640, Cc // 8
    g = torch.Generator().manual_seed(len(case) + Cc)
    qkv = torch.randn(B, N, 3 * Cc, generator=g)
    q, k = qkv[:, :, :Cc], qkv[:, :, Cc:2 * Cc]
    if case == "late_spike":
        k[0, 500] = 0
        k[0, 500, :d] = 12.0 * torch.sign(q[0, 17, :d])          # head 0: query 17 sees a huge score in tile 7
        q[0, 17, :d] *= 4.0
    elif case == "negative_first_tile":
        q[0, :, :d] = 3.0
        k[0, :64, :d] = -6.0                                      # scores ~ -18*sqrt(d) in tile 0, ~0 afterwards
    elif case == "growing":
        q[0, :, :d] = 1.0
        k[0, :, :d] = (torch.arange(N).float() / N * 4.0)[:, None]   # every tile raises every row's maximum
    else:
        q[0, :, :d] *= 40.0
        k[0, :, :d] *= 4.0
    src = bf16_round(qkv)
    ref = attention_ref(src, B, N, Cc, model_q_rounding=True)
    out = torch.empty(B, N, Cc, device="cuda")
    dq = dev(qkv)
    assert L.lib().ldmseg_op_attention(P(dq), B, N, Cc, 8, BF16, P(out), None) == 0
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    assert rel_err(out, ref) < 1.5e-2, case


@pytest.mark.parametrize("variant", [0, 7, 9, 10, 13])
@pytest.mark.parametrize("case", ["overflow_between_looks", "growth_between_looks"])
@pytest.mark.parametrize("Cc", [320, 640])
def test_attention_lazy_maxima(L, attn_variant, variant, case, Cc):
    """The shipped kernels (0: attention4.hip at head dim 40, attention3.hip at 80; 7 / 10 / 13: the round-3 rule and the 4-wave
    forms) look at the row maxima only on tile 0 and every 16th (variant 9: 4th) key tile.  A key in a tile
    that is NOT looked at (tile 5) whose score exceeds everything seen before by (a) far more than fp32 can hold as
    exp2 - the row sum turns inf, the workgroup must notice and redo its rows with the maxima tracked on every tile -
    and (b) by 2^60 - no overflow, no redo: exp2 against the stale maximum must still give the right softmax."""
    B, N, d = 4, 2048, Cc // 8                  # 4 * 8 heads * 8 query blocks: the 8-wave form of variant 0 engages
    g = torch.Generator().manual_seed(len(case) + Cc + variant)
    qkv = torch.randn(B, N, 3 * Cc, generator=g)
    q, k = qkv[:, :, :Cc], qkv[:, :, Cc:2 * Cc]
    key = 5 * 64 + 11
    rows = [3, 700, 1999]
    sc = d ** -0.5 * 1.4426950408889634
    want = 200.0 if case == "overflow_between_looks" else 60.0          # log2-domain score of the spike
    for b, h in ((0, 0), (3, 7)):
        for r in rows:
            qr = torch.sign(torch.randn(d, generator=g)) * 2.0
            q[b, r, h * d:(h + 1) * d] = qr
        # one key aligned with the LAST of those rows (the others see a random +-; the aligned one sees |q|^2 * c)
        c = want / (4.0 * d * sc)
        k[b, key, h * d:(h + 1) * d] = c * q[b, rows[-1], h * d:(h + 1) * d]
    src = bf16_round(qkv)
    ref = attention_ref(src, B, N, Cc, model_q_rounding=True)
    out = torch.empty(B, N, Cc, device="cuda")
    dq = dev(qkv)
    attn_variant(variant)
    assert L.lib().ldmseg_op_attention(P(dq), B, N, Cc, 8, BF16, P(out), None) == 0
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    assert rel_err(out, ref) < 1.5e-2, (variant, case)
    # the spiked row is (numerically) a copy of the spike key's value row
    v = src[..., 2 * Cc:]
    assert float((out[0, rows[-1], :d].cpu() - v[0, key, :d]).abs().max()) < 2e-2 * float(v.abs().max())


def fp8_e4m3_round(t):
    return t.clamp(-448, 448).to(torch.float8_e4m3fn).to(torch.float32)


# 1: block-scaled 2x-rate MFMAs where the shape allows (attention_mx.hip), probabilities as direct e4m3 bytes (the shipped form);
# 0x111: the same with exp + convert; 0: unscaled fp8 MFMAs (attention_fp8.hip)
@pytest.mark.parametrize("mx", [1, 0x111, 0])
@pytest.mark.parametrize("B,N,Cc", [(1, 256, 320), (2, 1024, 320), (1, 200, 640), (1, 4096, 640), (1, 4096, 320), (1, 100, 320),
                                    (2, 128, 320), (1, 384, 320)])
def test_attention_fp8_path(L, B, N, Cc, mx):
    """fp8 (e4m3) operand path of the bf16 attention (BASELINE configs[4]) against (a) the fp64 softmax attention of the
    SAME quantised operands - what the kernel computes up to P's 3 mantissa bits - and (b) the unquantised reference,
    i.e. the total cost of the fp8 path.  Bounds are ~2x the measured errors (e4m3: 2^-4 relative rounding per operand)."""
    g = torch.Generator().manual_seed(N + Cc)
    qkv = torch.randn(B, N, 3 * Cc, generator=g)
    qkv[:, :, :Cc] *= 1.5
    d = Cc // 8
    src = bf16_round(qkv)
    ref = attention_ref(src, B, N, Cc)
    sc = torch.tensor(d ** -0.5, dtype=torch.float32) * torch.tensor(1.4426950408889634, dtype=torch.float32)
    q8 = fp8_e4m3_round(src[..., :Cc] * sc) / sc                      # the kernel quantises q * d^-1/2 log2 e
    k8, v8 = fp8_e4m3_round(src[..., Cc:2 * Cc]), fp8_e4m3_round(src[..., 2 * Cc:])
    ref8 = attention_ref(torch.cat([q8, k8, v8], -1), B, N, Cc)
    out = torch.empty(B, N, Cc, device="cuda")
    dq = dev(qkv)
    if mx == 0 and (N % 128 or Cc != 320):
        pytest.skip("the shape runs on the unscaled kernel in either mode")
    L.lib().ldmseg_debug_set(15, mx)
    try:
        assert L.lib().ldmseg_op_attention_fp8(P(dq), B, N, Cc, 8, P(out), 0, None, None) == 0
        torch.cuda.synchronize()
    finally:
        L.lib().ldmseg_debug_set(15, 1)
    assert torch.isfinite(out).all()
    e_same, e_total = rel_err(out, ref8), rel_err(out, ref)
    l2 = float((out.cpu() - ref).norm() / ref.norm())
    print(f"fp8 attention B={B} N={N} C={Cc}: vs fp64-on-quantised-operands {e_same:.3e}, vs unquantised {e_total:.3e} (rel-L2 {l2:.3e})")
    # measured on MI355X: 0.7-2.0e-2 against the same operands (P's e4m3 rounding; independent of N since the probabilities
    # are shifted to the top of the e4m3 range), 4-12e-2 max-norm / 3-5e-2 rel-L2 against the unquantised tensors (e4m3 Q, K, V
    # on unit-variance random data, where the outputs are averages of noise).  The direct-byte form interpolates the e4m3
    # mantissa linearly: 1.8-2.5e-2 against the same operands (1.2 x the exact form in rel-L2), +2 % against the unquantised ones.
    assert e_same < (4e-2 if mx == 1 else 3e-2) and e_total < 0.2 and l2 < 0.1


def test_attention_fp8_long_context_vs_bf16_kernel(L):
    """N = 16384 (128x128 latents), head dim 40: fp8 path vs the bf16 kernel on the full tensor and vs fp64 rows."""
    B, N, Cc, d = 1, 16384, 320, 40
    g = torch.Generator().manual_seed(17)
    qkv = torch.randn(B, N, 3 * Cc, generator=g)
    qkv[:, :, :Cc] *= 1.5
    dq = dev(qkv)
    out8 = torch.empty(B, N, Cc, device="cuda")
    out16 = torch.empty(B, N, Cc, device="cuda")
    assert L.lib().ldmseg_op_attention_fp8(P(dq), B, N, Cc, 8, P(out8), 0, None, None) == 0
    assert L.lib().ldmseg_op_attention(P(dq), B, N, Cc, 8, BF16, P(out16), None) == 0
    torch.cuda.synchronize()
    assert torch.isfinite(out8).all()
    src = bf16_round(qkv)
    rows = torch.randperm(N, generator=g)[:256]
    q, k, v = src.chunk(3, -1)
    qs = q[0, rows].view(256, 8, d).transpose(0, 1).double()
    kk = k[0].view(N, 8, d).transpose(0, 1).double()
    vv = v[0].view(N, 8, d).transpose(0, 1).double()
    ref = (torch.softmax(qs @ kk.transpose(-1, -2) * d ** -0.5, -1) @ vv).transpose(0, 1).reshape(256, Cc).float()
    e_rows, e_kern = rel_err(out8[0].cpu()[rows], ref), rel_err(out8, out16)
    print(f"fp8 attention N=16384: vs fp64 rows {e_rows:.3e}, vs bf16 kernel {e_kern:.3e}")
    # long rows average the per-element e4m3 noise: relative to the row maximum the error is far below the 2^-4 step
    assert e_rows < 0.25 and e_kern < 0.2        # measured 0.148 / 0.093 (max-norm, relative to the largest output)


@pytest.mark.parametrize("dt", [F32, BF16])
def test_attention_long_context(L, dt):
    """N = 16384 tokens (128x128 latents, BASELINE's 1024x1024 config): 512 sampled query rows against all keys."""
    B, N, Cc, d = 1, 16384, 320, 40
    g = torch.Generator().manual_seed(17)
    qkv = torch.randn(B, N, 3 * Cc, generator=g)
    qkv[:, :, :Cc] *= 1.5
    src = bf16_round(qkv) if dt == BF16 else qkv
    rows = torch.randperm(N, generator=g)[:512]
    q, k, v = src.chunk(3, -1)
    qs = q[0, rows].view(512, 8, d).transpose(0, 1).double()           # [8, 512, d]
    kk = k[0].view(N, 8, d).transpose(0, 1).double()
    vv = v[0].view(N, 8, d).transpose(0, 1).double()
    ref = (torch.softmax(qs @ kk.transpose(-1, -2) * d ** -0.5, -1) @ vv).transpose(0, 1).reshape(512, Cc).float()
    out = torch.empty(B, N, Cc, device="cuda")
    dq = dev(qkv)
    assert L.lib().ldmseg_op_attention(P(dq), B, N, Cc, 8, dt, P(out), None) == 0
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    assert rel_err(out[0].cpu()[rows], ref) < (1.5e-2 if dt == BF16 else 2e-5)


@pytest.mark.parametrize("dt", [F32, BF16])
def test_convt2_and_bilinear(L, dt):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 256, 6, 6, generator=g)
    w = torch.randn(256, 256, 2, 2, generator=g) / 16
    b = torch.randn(256, generator=g)
    xr, wr = (bf16_round(x), bf16_round(w)) if dt == BF16 else (x, w)
    ref = F.conv_transpose2d(xr, wr, b, stride=2)
    out = torch.empty(ref.shape, device="cuda")
    dx, dw, db = dev(x), dev(w), dev(b)
    assert L.lib().ldmseg_op_convt2(P(dx), P(dw), P(db), 2, 256, 6, 6, 256, dt, P(out), None) == 0
    torch.cuda.synchronize()
    assert rel_err(out, ref) < (8e-3 if dt == BF16 else 2e-5)
    y = torch.randn(2, 128, 5, 9, generator=g)
    yr = bf16_round(y) if dt == BF16 else y
    ref = F.interpolate(yr, scale_factor=2, mode="bilinear", align_corners=False)
    out = torch.empty(ref.shape, device="cuda")
    dy = dev(y)
    assert L.lib().ldmseg_op_bilinear2x(P(dy), 2, 128, 5, 9, dt, P(out), None) == 0
    torch.cuda.synchronize()
    assert rel_err(out, ref) < 1e-5


@pytest.fixture
def tile_policy(L):
    """Select an igemm tile policy for one test; the value that was active before (read back from the library, never a
    literal) is restored by the finalizer even when the test fails."""
    lib = L.lib()
    saved = lib.ldmseg_debug_get(1)

    def set_policy(bits):
        assert lib.ldmseg_debug_set(1, (bits & 63) << 8) == 0
    yield set_policy
    lib.ldmseg_debug_set(1, saved)
    assert lib.ldmseg_debug_get(1) == saved


# policy bits: 1 = 256-row 8-wave tiles, 2 = 4-stage ring for mid-size grids (and no loader waves), 4 = lone 64-row
# 4-stage tiles, 8 = pipelined K loop on the 256-row tiles, 16 = 8-wave 128-row tiles (+ loader waves on long K),
# 32 = loader waves on the 256-row tiles (long K slices / GEGLU)
POLICY_SHAPES = {
    "big": (8, 128, 64, 320, 3),      # M = 32768, 256 tiles of 256x160, 18 K tiles
    "big_longK": (8, 320, 64, 320, 3),   # same grid, 45 K tiles (12-wave tile with loader waves under bit 32)
    "mid_longK": (8, 320, 32, 640, 3),   # M = 8192: one 128-row item per CU, 45 K tiles (loader-wave variant)
    "mid_shortK": (8, 128, 32, 640, 3),  # same grid, 18 K tiles
    "small": (8, 128, 16, 1280, 1),      # M = 2048: 256 64-row tiles
}


@pytest.mark.parametrize("shape", sorted(POLICY_SHAPES))
@pytest.mark.parametrize("policy", [0, 1, 2, 3, 4, 5, 8, 9, 13, 16, 17, 18, 20, 21, 25, 29, 31, 33, 41, 61, 63])
def test_igemm_tile_policies_agree(L, tile_policy, policy, shape):
    """Every selectable K-loop structure (tile policy bits) must give the reference result on grids where it engages;
    the shipped policy is whatever ldmseg_debug_get(-1) reports and is covered here like the rest."""
    B, Ci, H, Co, k = POLICY_SHAPES[shape]
    g = torch.Generator().manual_seed(policy * 7 + len(shape))
    x = torch.randn(B, Ci, H, H, generator=g)
    w = torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5
    b = torch.randn(Co, generator=g)
    ref = F.conv2d(bf16_round(x), bf16_round(w), b, padding=k // 2)
    out = torch.empty(ref.shape, device="cuda")
    dx, dw, db = dev(x), dev(w), dev(b)
    tile_policy(policy)
    assert L.lib().ldmseg_op_igemm(P(dx), None, P(dw), P(db), None, None, B, Ci, 0, H, H, Co, k, 1, 0, 0, 0, 0, BF16,
                                   P(out), None) == 0
    torch.cuda.synchronize()
    assert rel_err(out, ref) < 8e-3, (policy, shape, L.igemm_last_kernel())


@pytest.fixture
def k_order(L):
    """K order in which op_igemm packs 3x3 weights (debug key 9); the previous value is restored afterwards."""
    lib = L.lib()
    saved = lib.ldmseg_debug_get(9)

    def set_mode(mode):
        assert lib.ldmseg_debug_set(9, mode) == 0
    yield set_mode
    lib.ldmseg_debug_set(9, saved)
    assert lib.ldmseg_debug_get(9) == saved == -1


# (B, Ci, Ci2, H, Co, splits, residual, time-embedding row)
K_ORDER_CASES = {
    "big": (8, 320, 0, 64, 320, 0, 0, 1),            # 256x160 loader-wave tiles, 45 K tiles
    "concat": (2, 640, 320, 64, 320, 0, 1, 0),       # torch.cat partner: the source switches at a channel-tile boundary
    "mid": (8, 320, 0, 32, 640, 0, 0, 1),            # 128-row loader-wave tiles
    "ragged": (3, 128, 64, 20, 160, 0, 1, 0),        # M = 1200: last tile partly past M, images end inside tiles
    "tiny_maps": (8, 256, 0, 8, 1280, 0, 0, 0),      # 8x8 maps: a 64-row tile spans several images, every row has padding taps
    "splitk": (2, 1280, 0, 16, 1280, 4, 0, 1),       # K slices start in the middle of a channel tile (180 K tiles / 4)
    "splitk3": (1, 640, 0, 16, 320, 7, 0, 0),        # 90 K tiles over 7 slices: every slice starts at a different tap
}


@pytest.mark.parametrize("policy", [None, 0])
@pytest.mark.parametrize("case", sorted(K_ORDER_CASES))
def test_conv3x3_k_orders_agree(L, tile_policy, k_order, case, policy):
    """3x3 conv weights are packed (tap, channel) or (channel tile, tap, channel) - the order large maps with many input
    channels use so that a row's nine gathers stay inside the L2 (bf16 only; one dedicated instantiation, whatever the tile
    policy).  Both orders against F.conv2d; the two HIP results differ only by fp32 summation order."""
    dt = BF16
    B, Ci, Ci2, H, Co, splits, use_res, use_rb = K_ORDER_CASES[case]
    g = torch.Generator().manual_seed(len(case) * 13 + Ci)
    ct = Ci + Ci2
    x = torch.randn(B, Ci, H, H, generator=g)
    x2 = torch.randn(B, Ci2, H, H, generator=g) if Ci2 else None
    w = torch.randn(Co, ct, 3, 3, generator=g) / (ct * 9) ** 0.5
    b = torch.randn(Co, generator=g)
    rb = torch.randn(B, Co, generator=g) if use_rb else None
    xin = torch.cat([x, x2], 1) if Ci2 else x
    rnd = bf16_round if dt == BF16 else (lambda t: t)
    ref = F.conv2d(rnd(xin), rnd(w), b, padding=1)
    if rb is not None:
        ref = ref + rb[:, :, None, None]
    res = torch.randn(ref.shape, generator=g) if use_res else None
    if res is not None:
        ref = ref + rnd(res)
    dx, dx2, dw, db, dres, drb = dev(x), dev(x2), dev(w), dev(b), dev(res), dev(rb)
    if policy is not None:
        tile_policy(policy)
    outs = []
    for mode in (0, 1):
        k_order(mode)
        out = torch.empty(ref.shape, device="cuda")
        assert L.lib().ldmseg_op_igemm(P(dx), P(dx2), P(dw), P(db), P(dres), P(drb), B, Ci, Ci2, H, H, Co, 3, 1, 0, 0, 0,
                                       splits, dt, P(out), None) == 0, L.lib().ldmseg_last_error()
        torch.cuda.synchronize()
        outs.append(out.cpu())
        tol = 8e-3 if dt == BF16 else 2e-4
        assert rel_err(outs[-1], ref) < tol, (case, mode, policy, L.igemm_last_kernel())
        assert (",cm" in L.igemm_last_kernel()) == (mode == 1), L.igemm_last_kernel()
    assert rel_err(outs[1], outs[0]) < 8e-3


def test_shipped_policy_is_active_after_the_policy_tests(L):
    assert L.lib().ldmseg_debug_get(1) == L.lib().ldmseg_debug_get(-1)


def test_fastdiv_on_device(L):
    """The kernels' division (host-prepared multiply-shift pair, one v_mul_hi_u32 + shift on the device) against integer
    division, for the divisors the UNet launches use and for awkward ones, numerators up to 2^31 - 1."""
    import ctypes as C
    g = torch.Generator().manual_seed(0)
    top = 2 ** 31 - 1
    for d in (1, 2, 3, 7, 10, 45, 64, 160, 180, 320, 1024, 2880, 4096, 40960, 65535, 65537, 999983, 2 ** 30 - 1, 2 ** 30 + 1, top):
        base = torch.randint(0, top, (4096,), generator=g, dtype=torch.int64)
        k = torch.randint(0, top // d + 1, (4096,), generator=g, dtype=torch.int64) * d
        n = torch.cat([base, k, (k - 1).clamp(min=0), (k + 1).clamp(max=top), torch.tensor([0, 1, d - 1, d, top, top - 1])]).clamp(0, top)
        nd = n.to(torch.int32).cuda()
        q = torch.empty_like(nd)
        assert L.lib().ldmseg_op_fastdiv(C.c_void_p(nd.data_ptr()), nd.numel(), d, C.c_void_p(q.data_ptr()), None) == 0
        torch.cuda.synchronize()
        assert torch.equal(q.cpu().to(torch.int64), n // d), d


# ------------------------------------------------------------------ row-local fused transformer feed-forward (tfuse.hip)
def _ff_case(M, Cc, seed):
    g = torch.Generator().manual_seed(seed)
    h = torch.randn(M, Cc, generator=g) * 1.5 + 0.3
    h[:, ::7] += 2.0                                     # rows with a mean: the LayerNorm has something to remove
    x = torch.randn(M, Cc, generator=g)
    gamma = 1 + 0.2 * torch.randn(Cc, generator=g)
    beta = 0.2 * torch.randn(Cc, generator=g)
    w1 = torch.randn(8 * Cc, Cc, generator=g) / Cc ** 0.5
    b1 = 0.2 * torch.randn(8 * Cc, generator=g)
    w2 = torch.randn(Cc, 4 * Cc, generator=g) / (4 * Cc) ** 0.5
    b2 = 0.2 * torch.randn(Cc, generator=g)
    wp = torch.randn(Cc, Cc, generator=g) / Cc ** 0.5
    bp = 0.2 * torch.randn(Cc, generator=g)
    return h, x, gamma, beta, w1, b1, w2, b2, wp, bp


def _ff_ref(h, x, gamma, beta, w1, b1, w2, b2, wp, bp, eps=1e-5):
    """torch fp32 on the bf16-rounded operands, the arithmetic of oracle/unet.py::transformer (ff + proj_out)."""
    hr, xr = bf16_round(h), bf16_round(x)
    n = F.layer_norm(hr, (h.shape[1],), gamma, beta, eps)
    a, gate = F.linear(n, w1, b1).chunk(2, dim=-1)
    h2 = F.linear(a * F.gelu(gate), bf16_round(w2), b2) + hr
    return F.linear(h2, bf16_round(wp), bp) + xr


def _ff_run(L, case, M, Cc, mode, dt=BF16, iters=0):
    out = torch.empty(M, Cc, device="cuda")
    keep = [dev(t) for t in case]
    us = C.c_float(0)
    r = L.lib().ldmseg_op_transformer_ff(*[P(t) for t in keep], M, Cc, 1e-5, dt, mode, P(out), iters, C.byref(us), None)
    assert r == 0, (r, L.lib().ldmseg_last_error())
    torch.cuda.synchronize()
    return out.cpu(), us.value


@pytest.mark.parametrize("M", [128, 300, 1000, 4096, 32768])
def test_transformer_ff_fused_vs_torch_and_unfused(L, M):
    """LayerNorm_3 -> GEGLU -> ff.net.2 (+h) -> proj_out (+x) of the 320-channel transformers in ONE launch (modes 1 / 3)
    against torch on the same bf16-rounded operands and against the unfused launches (mode 0): one tile, ragged tiles, and
    the configs[1] size (M = 8 x 64 x 64).  The hidden tensor is rounded to bf16 in every variant; the fused kernel rounds
    the LayerNorm output to bf16 as well (the unfused path folds the norm into the GEMM), hence a bf16-level bound."""
    Cc = 320
    case = _ff_case(M, Cc, 1000 + M)
    ref = _ff_ref(*case)
    outs = {m: _ff_run(L, case, M, Cc, m)[0] for m in (0, 1, 3)}
    for m, o in outs.items():
        assert torch.isfinite(o).all(), m
        e = rel_err(o, ref)
        l2 = float((o.double() - ref.double()).norm() / ref.double().norm())
        assert e < 3e-2 and l2 < 6e-3, (M, m, e, l2)
    # fused against unfused: both are bf16 pipelines of the same math
    l2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    assert l2(outs[1], outs[0]) < 6e-3 and l2(outs[3], outs[0]) < 6e-3
    # deterministic
    assert torch.equal(_ff_run(L, case, M, Cc, 3)[0], outs[3])


def test_transformer_ff_fused_large_mean_rows(L):
    """Rows with a mean 100x their deviation: the in-tile LayerNorm is two-pass (centred variance), so nothing cancels."""
    M, Cc = 256, 320
    case = list(_ff_case(M, Cc, 5))
    case[0] = case[0] * 0.05 + 8.0
    ref = _ff_ref(*case)
    for m in (1, 3):
        o = _ff_run(L, case, M, Cc, m)[0]
        assert float((o.double() - ref.double()).norm() / ref.double().norm()) < 1.5e-2, m


# ------------------------------------------------------------------ row-local fused transformer entry (tproj.hip)
def _tin_case(M, Cc, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, Cc, generator=g)
    wp = torch.randn(Cc, Cc, generator=g) / Cc ** 0.5
    bp = 0.5 * torch.randn(Cc, generator=g) + 0.4            # rows of h get a mean for the LayerNorm to remove
    gamma = 1 + 0.2 * torch.randn(Cc, generator=g)
    beta = 0.2 * torch.randn(Cc, generator=g)
    wq, wk, wv = (torch.randn(Cc, Cc, generator=g) / Cc ** 0.5 for _ in range(3))
    return x, wp, bp, gamma, beta, wq, wk, wv


def _tin_ref(x, wp, bp, gamma, beta, wq, wk, wv, eps=1e-5):
    """torch fp32 on the bf16-rounded operands: proj_in, LayerNorm_1, to_q | to_k | to_v (oracle/unet.py::transformer)."""
    h = F.linear(bf16_round(x), bf16_round(wp), bp)
    n = F.layer_norm(bf16_round(h), (x.shape[1],), gamma, beta, eps)     # (both paths keep h in bf16)
    return h, torch.cat([F.linear(n, w) for w in (wq, wk, wv)], dim=-1)


def _tin_run(L, case, M, Cc, mode, dt=BF16, iters=0):
    h = torch.empty(M, Cc, device="cuda")
    qkv = torch.empty(M, 3 * Cc, device="cuda")
    keep = [dev(t) for t in case]
    us = C.c_float(0)
    r = L.lib().ldmseg_op_transformer_in(*[P(t) for t in keep], M, Cc, 1e-5, dt, mode, P(h), P(qkv), iters, C.byref(us), None)
    assert r == 0, (r, L.lib().ldmseg_last_error())
    torch.cuda.synchronize()
    return h.cpu(), qkv.cpu(), us.value


@pytest.mark.parametrize("M", [128, 1024, 4096, 32768])
def test_transformer_in_fused_vs_torch_and_unfused(L, M):
    """proj_in -> LayerNorm_1 -> q|k|v of the 320-channel transformers in ONE launch (mode 1) against torch on the same
    bf16-rounded operands and against the unfused launches (mode 0: GEMM, row statistics, folded-LayerNorm GEMM): one tile
    up to the configs[1] size (M = 8 x 64 x 64).  h must agree to bf16 rounding (same products, fp32 accumulation in a
    different order); q|k|v is a bf16-level comparison (the fused kernel rounds the normalised tile to bf16, the unfused
    path folds the norm into the GEMM epilogue)."""
    Cc = 320
    case = _tin_case(M, Cc, 2000 + M)
    href, qref = _tin_ref(*case)
    l2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    outs = {m: _tin_run(L, case, M, Cc, m) for m in (0, 1)}
    for m, (h, qkv, _) in outs.items():
        assert torch.isfinite(h).all() and torch.isfinite(qkv).all(), m
        assert rel_err(h, href) < 1e-2 and l2(h, href) < 3e-3, (M, m, rel_err(h, href), l2(h, href))
        assert rel_err(qkv, qref) < 3e-2 and l2(qkv, qref) < 6e-3, (M, m, rel_err(qkv, qref), l2(qkv, qref))
    assert l2(outs[1][0], outs[0][0]) < 3e-3 and l2(outs[1][1], outs[0][1]) < 6e-3
    again = _tin_run(L, case, M, Cc, 1)
    assert torch.equal(again[0], outs[1][0]) and torch.equal(again[1], outs[1][1])      # deterministic


def _gtin_run(L, x, gg, gb, images, gn_mode, case, M, Cc, mode, iters=0):
    h = torch.empty(M, Cc, device="cuda")
    qkv = torch.empty(M, 3 * Cc, device="cuda")
    keep = [dev(t) for t in (x, gg, gb)] + [dev(t) for t in case[1:]]
    us = C.c_float(0)
    r = L.lib().ldmseg_op_gn_transformer_in(P(keep[0]), P(keep[1]), P(keep[2]), 1e-6, images, gn_mode, *[P(t) for t in keep[3:]],
                                            M, Cc, 1e-5, BF16, mode, P(h), P(qkv), iters, C.byref(us), None)
    assert r == 0, (r, L.lib().ldmseg_last_error())
    torch.cuda.synchronize()
    return h.cpu(), qkv.cpu(), us.value


@pytest.mark.parametrize("images,HW", [(1, 128), (3, 256), (2, 1024), (8, 4096), (1, 16384)])
def test_transformer_in_with_groupnorm_folded(L, images, HW):
    """The transformer's GroupNorm as a statistics pass + a sweep over the fused entry's LDS tile (round 5, tproj.hip `gn`):
    against the same kernel behind a GroupNorm launch (same fused multiply-add on the same bf16 input, statistics combined
    in another order: bf16-rounding-level agreement) and against torch on the bf16-rounded operands.  Channel means of
    +-3 and per-image offsets make the statistics matter; 1 x 128 is a single tile, 8 x 4096 the configs[1] size,
    1 x 16384 the 128 x 128 map of configs[4]."""
    Cc, M = 320, images * HW
    g = torch.Generator().manual_seed(7 * images + HW)
    case = _tin_case(M, Cc, 3000 + M)
    x = torch.randn(images, HW, Cc, generator=g) * (0.5 + torch.rand(1, 1, Cc, generator=g)) + 3.0 * torch.randn(1, 1, Cc, generator=g)
    x = x + torch.randn(images, 1, 1, generator=g)
    gg = 1 + 0.3 * torch.randn(Cc, generator=g)
    gb = 0.3 * torch.randn(Cc, generator=g)
    xb = bf16_round(x)
    xn = F.group_norm(xb.permute(0, 2, 1).reshape(images, Cc, HW, 1), 32, gg, gb, 1e-6).reshape(images, Cc, HW).permute(0, 2, 1)
    href, qref = _tin_ref(xn.reshape(M, Cc), *case[1:])
    l2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    sep = _gtin_run(L, x.reshape(M, Cc), gg, gb, images, 0, case, M, Cc, 1)
    fold = _gtin_run(L, x.reshape(M, Cc), gg, gb, images, 1, case, M, Cc, 1)
    for nm, (h, qkv, _) in (("sep", sep), ("fold", fold)):
        assert torch.isfinite(h).all() and torch.isfinite(qkv).all(), nm
        assert l2(h, href) < 4e-3 and l2(qkv, qref) < 7e-3, (nm, l2(h, href), l2(qkv, qref))
    assert l2(fold[0], sep[0]) < 2e-3 and l2(fold[1], sep[1]) < 4e-3, (l2(fold[0], sep[0]), l2(fold[1], sep[1]))
    again = _gtin_run(L, x.reshape(M, Cc), gg, gb, images, 1, case, M, Cc, 1)
    assert torch.equal(again[0], fold[0]) and torch.equal(again[1], fold[1])      # deterministic


def test_transformer_in_groupnorm_fold_rejects_partial_tiles(L):
    """A map whose pixel count is not a multiple of the 128-row tile keeps its GroupNorm launch (the engine asks proj_qkv_gn_fold_ok)."""
    case = _tin_case(384, 320, 5)
    x = torch.randn(384, 320)
    keep = [dev(t) for t in (x, torch.ones(320), torch.zeros(320))] + [dev(t) for t in case[1:]]
    h = torch.empty(384, 320, device="cuda")
    qkv = torch.empty(384, 960, device="cuda")
    us = C.c_float(0)
    call = lambda images, gn_mode: L.lib().ldmseg_op_gn_transformer_in(
        P(keep[0]), P(keep[1]), P(keep[2]), 1e-6, images, gn_mode, *[P(t) for t in keep[3:]], 384, 320, 1e-5, BF16, 1, P(h), P(qkv), 0,
        C.byref(us), None)
    assert call(2, 1) != 0          # 192 pixels per image
    assert call(2, 0) == 0
    assert call(3, 1) == 0          # 128 pixels per image


def test_transformer_in_fused_rejects_what_it_does_not_cover(L):
    """Ragged row counts, other channel counts and fp32 are not the fused kernel's: the operator says so (the engine routes
    such shapes to the unfused launches)."""
    case = _tin_case(192, 320, 1)
    h = torch.empty(192, 320, device="cuda")
    qkv = torch.empty(192, 960, device="cuda")
    keep = [dev(t) for t in case]
    us = C.c_float(0)
    assert L.lib().ldmseg_op_transformer_in(*[P(t) for t in keep], 192, 320, 1e-5, BF16, 1, P(h), P(qkv), 0, C.byref(us), None) != 0
    assert L.lib().ldmseg_op_transformer_in(*[P(t) for t in keep], 128, 320, 1e-5, F32, 1, P(h), P(qkv), 0, C.byref(us), None) != 0
    assert L.lib().ldmseg_op_transformer_in(*[P(t) for t in keep], 192, 320, 1e-5, BF16, 0, P(h), P(qkv), 0, C.byref(us), None) == 0


# ------------------------------------------------------------------ step tail (tail.hip): conv_out + DDIM + paste + pack
@pytest.mark.parametrize("case", [
    # B, H, W, pred_type, clip, last, self-condition, inpainting
    (2, 16, 16, 0, 0, 0, 1, 0),
    (1, 8, 32, 0, 0, 0, 1, 1),            # a single tile row; inpainting paste
    (2, 32, 32, 2, 1, 0, 0, 0),           # v-prediction + clipping, 8-channel variant (no self-condition)
    (2, 16, 32, 1, 0, 1, 1, 1),           # last step: latents <- pred_original_sample, then the paste
    (8, 64, 64, 0, 0, 0, 1, 0),           # configs[1] size
])
def test_step_tail_kernel(L, case):
    """conv_out as the halo-resident stencil (vs F.conv2d on the bf16-rounded operands) and, fused into its epilogue, the
    scheduler step: latents / self-condition must equal ldmseg_ddim_step (+ ldmseg's paste arithmetic) on the SAME eps bit
    for bit, and the packed next input must be bf16([latents | rgb | cond | 0]) exactly."""
    B, H, W, pt, clip, last, selfc, inp = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, 320, H, W, generator=g)
    w = torch.randn(4, 320, 3, 3, generator=g) / (320 * 9) ** 0.5
    b = torch.randn(4, generator=g)
    lat = torch.randn(B, 4, H, W, generator=g)
    rgb = torch.randn(B, 4, H, W, generator=g)
    z0 = torch.randn(B, 4, H, W, generator=g)
    noise = torch.randn(B, 4, H, W, generator=g)
    known = (torch.rand(B, 1, H, W, generator=g) < 0.5).to(torch.uint8)
    coef = (C.c_float * 4)(0.41, 0.912, 0.53, 0.848)
    sa, sb = 0.6, 0.8
    lib = L.lib()
    dx, dw, db = dev(x), dev(w), dev(b)
    eps = torch.empty(B, 4, H, W, device="cuda")
    assert lib.ldmseg_op_conv_out_tail(P(dx), P(dw), P(db), B, H, W, P(eps), 0, 0, None, 0, 0, 1.0, None, None, None, None, None, None,
                                       0.0, 0.0, None, None) == 0
    torch.cuda.synchronize()
    ref = F.conv2d(bf16_round(x), bf16_round(w), b, padding=1)
    assert rel_err(eps, ref) < 1e-3, case
    # fused step
    lat_f, cond_f = lat.cuda().clone(), torch.full((B, 4, H, W), 7.0, device="cuda")
    drgb, dz0, dnoise, dknown = dev(rgb), dev(z0), dev(noise), known.cuda()
    xin = torch.empty(B, H * W, 64, device="cuda")
    eps2 = torch.empty_like(eps)
    assert lib.ldmseg_op_conv_out_tail(P(dx), P(dw), P(db), B, H, W, P(eps2), 1, last, coef, pt, clip, 1.0, P(lat_f),
                                       P(cond_f) if selfc else None, P(drgb), P(dknown) if inp else None, P(dz0), P(dnoise), sa, sb,
                                       P(xin), None) == 0
    torch.cuda.synchronize()
    assert torch.equal(eps2, eps)
    # the unfused sequence on the same eps
    prev, x0 = torch.empty_like(eps), torch.empty_like(eps)
    dlat = lat.cuda()
    assert lib.ldmseg_ddim_step(P(eps), P(dlat), coef[0], coef[1], coef[2], coef[3], pt, clip, 1.0, 0, P(prev), P(x0), B * 4 * H * W, None) == 0
    torch.cuda.synchronize()
    want = (x0 if last else prev).clone()
    if inp:
        paste = (torch.tensor(sa) * z0.cuda()) + (torch.tensor(sb) * noise.cuda())      # fp32, each op rounded (torch eager)
        want = torch.where(dknown.bool().expand_as(want), paste, want)
    assert torch.equal(lat_f, want), case
    if selfc and not last:
        assert torch.equal(cond_f, x0)
    elif selfc:
        assert bool((cond_f == 7.0).all())          # last step: the self-condition is not written
    if not last:
        exp = torch.zeros(B, H * W, 64)
        exp[:, :, 0:4] = want.cpu().reshape(B, 4, H * W).permute(0, 2, 1)
        exp[:, :, 4:8] = rgb.reshape(B, 4, H * W).permute(0, 2, 1)
        if selfc:
            exp[:, :, 8:12] = x0.cpu().reshape(B, 4, H * W).permute(0, 2, 1)
        assert torch.equal(xin.cpu(), bf16_round(exp)), case


# ---- K slices finished inside the igemm launch (round 6, debug key 23) ----
# (B, Ci, Ci2, H, Co, k, stride, up, splits, residual, time-embedding row)
CF_CASES = {
    "m512_3x3": (8, 1280, 0, 8, 1280, 3, 1, 0, 0, 0, 1),          # the 8x8 level's conv: M = 512, the launch table's 8 slices on 128-row tiles
    "m512_res": (8, 1280, 0, 8, 1280, 3, 1, 0, 8, 1, 0),
    "m2048_s4": (8, 1280, 0, 16, 1280, 3, 1, 0, 0, 0, 1),         # 256-row tiles with loader waves, the table's 4 slices: shares of 64 rows
    "ragged_m": (3, 320, 0, 8, 320, 3, 1, 0, 4, 1, 1),            # M = 192: the second m tile is half empty, images end inside tiles
    "ragged_map": (2, 320, 0, 7, 320, 3, 1, 0, 3, 0, 0),          # 7x7 maps: M = 98; three slices (the run-time slice loop)
    "stride2": (2, 320, 0, 16, 640, 3, 2, 0, 4, 0, 0),            # downsampler
    "up": (2, 320, 0, 4, 320, 3, 1, 1, 2, 0, 0),                  # nearest x2 folded into the gather (nine-tap form: odd map for up4)
    "up4": (8, 1280, 0, 8, 1280, 3, 1, 1, 0, 0, 0),               # the 8x8 -> 16x16 upsampler as four phase convs, K slices by the plan (scatter in the finish)
    "concat": (2, 320, 192, 8, 320, 3, 1, 0, 5, 0, 1),            # torch.cat partner; five slices
    "lin_2560": (2, 2560, 0, 8, 320, 1, 1, 0, 8, 1, 0),           # 1x1: 40 K tiles
    "slices7": (2, 640, 0, 8, 320, 3, 1, 0, 7, 0, 0),             # 90 K tiles over 7 slices
    "narrow_n": (2, 640, 0, 8, 128, 3, 1, 0, 4, 0, 0),            # N = 128 tiles
    "slices45": (1, 320, 0, 8, 320, 3, 1, 0, 45, 0, 0),           # more slices than the ticket word's mask holds: two launches
}


@pytest.mark.parametrize("dt", [BF16, F32])
@pytest.mark.parametrize("case", sorted(CF_CASES))
def test_splitk_finished_inside_the_launch(L, case, dt):
    """K-sliced launches whose items all fit on the chip at once reduce their slabs inside the launch: every slice workgroup
    publishes its slab write-through, takes a ticket, and reduces 1 / S of the tile in the finish kernel's slice order.  Against
    F.conv2d on the rounded operands, and BIT FOR BIT against the two-launch path (key 23 = 0) - with the partner poll at its
    shipped bound and at zero length (bit 1: every workgroup but the last arriver gives up at once and the last arriver reduces their
    shares: the path that makes progress without co-residency).  Each mode runs three times on fresh inputs so that a stale
    cached slab line of the previous launch would show."""
    B, Ci, Ci2, H, Co, k, stride, up, splits, use_res, use_rb = CF_CASES[case]
    lib = L.lib()
    saved = lib.ldmseg_debug_get(23)
    assert saved == 5
    ct = Ci + Ci2
    outs = {}
    names = {}
    try:
        for mode in (0, 9, 11):                              # off | on for every tile form that has it | + zero-length poll
            assert lib.ldmseg_debug_set(23, mode) == 0
            for rep in range(3):
                g = torch.Generator().manual_seed(len(case) * 131 + rep)
                x = torch.randn(B, Ci, H, H, generator=g)
                x2 = torch.randn(B, Ci2, H, H, generator=g) if Ci2 else None
                w = torch.randn(Co, ct, k, k, generator=g) / (ct * k * k) ** 0.5
                b = torch.randn(Co, generator=g)
                rb = torch.randn(B, Co, generator=g) if use_rb else None
                rnd = bf16_round if dt == BF16 else (lambda t: t)
                xin = rnd(torch.cat([x, x2], 1) if Ci2 else x)
                if up:
                    xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
                ref = F.conv2d(xin, rnd(w), b, stride=stride, padding=k // 2)
                if rb is not None:
                    ref = ref + rb[:, :, None, None]
                res = torch.randn(ref.shape, generator=g) if use_res else None
                if res is not None:
                    ref = ref + rnd(res)
                out = torch.empty(ref.shape, device="cuda")
                dx, dx2, dw, db, dres, drb = dev(x), dev(x2), dev(w), dev(b), dev(res), dev(rb)
                r = lib.ldmseg_op_igemm(P(dx), P(dx2), P(dw), P(db), P(dres), P(drb), B, Ci, Ci2, H, H, Co, k, stride, up, 0,
                                        0, splits, dt, P(out), None)
                assert r == 0, lib.ldmseg_last_error()
                torch.cuda.synchronize()
                names[mode] = L.igemm_last_kernel()
                assert rel_err(out, ref) < (8e-3 if dt == BF16 else 2e-5), (case, mode, rep, names[mode])
                outs[mode, rep] = out.cpu()
    finally:
        lib.ldmseg_debug_set(23, saved)
    assert "/splitk " in names[0] + " ", names
    if case in ("m512_3x3", "m512_res", "m2048_s4", "up4") and dt == BF16:   # (instantiated for the bf16 tile forms K-sliced launches of the UNet use)
        assert "/splitk-cf" in names[9] and "/splitk-cf" in names[11], names
    elif case == "slices45" or dt == F32:
        assert "/splitk-cf" not in names[9], names
    for rep in range(3):
        assert torch.equal(outs[9, rep], outs[0, rep]), (case, rep, names)
        assert torch.equal(outs[11, rep], outs[0, rep]), (case, rep, names)
    assert lib.ldmseg_debug_get(23) == saved


# ---- ff.net.2 and proj_out as one chained Linear (round 5's launch, round 6's operator-level test) ----
@pytest.mark.parametrize("M", [512, 2048, 8192])
@pytest.mark.parametrize("Cc", [640, 1280])
def test_chained_ff_out_vs_oracle(L, Cc, M):
    """proj_out(h + ff.net.2(g)) + x (diffusers BasicTransformerBlock / Transformer2DModel, unet.py:401-425) in fp64 on the bf16-rounded
    operands and the SEPARATELY rounded matrices W2 and Wp - what the two-GEMM form of the oracle computes - against the engines' one
    launch over [g | h] with the chained matrix [Wp W2 | Wp] (formed in fp32, rounded to bf16 once) at the 640- / 1280-channel levels'
    token counts.  Then the bias-only case g = h = 0: bp + Wp b2 + x must come out to fp32 rounding (fp32 launch) / one bf16 rounding
    (bf16 launch), which pins the chained bias and the residual path on their own."""
    g_ = torch.Generator().manual_seed(Cc + M)
    g = torch.randn(M, 4 * Cc, generator=g_) * 0.5
    h = torch.randn(M, Cc, generator=g_)
    x = torch.randn(M, Cc, generator=g_)
    w2 = torch.randn(Cc, 4 * Cc, generator=g_) / (4 * Cc) ** 0.5
    wp = torch.randn(Cc, Cc, generator=g_) / Cc ** 0.5
    b2, bp = torch.randn(Cc, generator=g_), torch.randn(Cc, generator=g_)
    lib = L.lib()
    dg, dh, dx, dw2, db2, dwp, dbp = dev(g), dev(h), dev(x), dev(w2), dev(b2), dev(wp), dev(bp)
    D = lambda t: bf16_round(t).double().cuda()
    ref = (D(h) + D(g) @ D(w2).t() + b2.double().cuda()) @ D(wp).t() + bp.double().cuda() + D(x)
    out = torch.empty(M, Cc, device="cuda")
    assert lib.ldmseg_op_chained_ff_out(P(dg), P(dh), P(dx), P(dw2), P(db2), P(dwp), P(dbp), M, Cc, M // 8, BF16, P(out), None) == 0, lib.ldmseg_last_error()
    torch.cuda.synchronize()
    name = L.igemm_last_kernel()
    l2 = float((out.double() - ref).norm() / ref.norm())
    print(f"chained Linear C={Cc} M={M}: rel-L2 {l2:.2e} max-norm {rel_err(out, ref):.2e}  {name}")
    assert torch.isfinite(out).all() and l2 <= 6e-3 and rel_err(out, ref) < 3e-2, (Cc, M, l2, name)
    # fp32 launch on unrounded operands: the chained form is the same function
    ref32 = (h.double().cuda() + g.double().cuda() @ w2.double().cuda().t() + b2.double().cuda()) @ wp.double().cuda().t() + bp.double().cuda() + x.double().cuda()
    assert lib.ldmseg_op_chained_ff_out(P(dg), P(dh), P(dx), P(dw2), P(db2), P(dwp), P(dbp), M, Cc, M // 8, F32, P(out), None) == 0
    torch.cuda.synchronize()
    assert rel_err(out, ref32) < 2e-5, (Cc, M)
    # bias only
    z4, z1 = torch.zeros(M, 4 * Cc, device="cuda"), torch.zeros(M, Cc, device="cuda")
    bias_ref = bp.double() + wp.double() @ b2.double()
    assert lib.ldmseg_op_chained_ff_out(P(z4), P(z1), P(dx), P(dw2), P(db2), P(dwp), P(dbp), M, Cc, M // 8, F32, P(out), None) == 0
    torch.cuda.synchronize()
    assert rel_err(out, bias_ref.cuda() + x.double().cuda()) < 2e-6, (Cc, M)
    assert lib.ldmseg_op_chained_ff_out(P(z4), P(z1), P(dx), P(dw2), P(db2), P(dwp), P(dbp), M, Cc, M // 8, BF16, P(out), None) == 0
    torch.cuda.synchronize()
    want = bias_ref.float().cuda() + bf16_round(x).cuda()            # fp32 bias + bf16 residual, stored as bf16
    assert (out - want).abs().max() <= want.abs().max() * 2 ** -8, (Cc, M)
    assert torch.equal(out, bf16_round(out.cpu()).cuda())


# ---- split-bf16 GEMM arithmetic on fp32 operands (compute_dtype "bf16x3", round 5) ----
X3_CASES = [
    # B, Ci, Ci2, H, Co, k, stride, up, geglu, residual, time-embedding row
    (8, 320, 0, 32, 320, 3, 1, 0, 0, 0, 1),      # 256-row tiles
    (2, 640, 320, 16, 640, 3, 1, 0, 0, 1, 0),    # concat + residual, 64-row tiles
    (8, 1280, 0, 8, 1280, 3, 1, 0, 0, 0, 1),     # K slices + finish
    (2, 320, 0, 32, 2560, 1, 1, 0, 1, 0, 0),     # GEGLU
    (2, 320, 0, 32, 320, 3, 2, 0, 0, 0, 0),      # stride 2
    (1, 320, 0, 16, 4, 3, 1, 0, 0, 0, 0),        # narrow N (conv_out style)
    (3, 96, 0, 5, 160, 3, 1, 0, 0, 0, 0),        # ragged everything
]


@pytest.mark.parametrize("case", X3_CASES)
def test_split_bf16_gemm_vs_fp32_reference(L, case):
    """dtype 2 of the operator ABI = LDMSEG_BF16X3: fp32 operands in HBM and LDS, hi/lo split in registers, three
    v_mfma_f32_16x16x32_bf16 per product block.  Against F.conv2d in fp32 on the UNROUNDED operands: the bound of the exact fp32
    kernels (1e-4 max-norm), i.e. an order of magnitude inside the north-star 1e-3."""
    B, Ci, Ci2, H, Co, k, stride, up, geglu, use_res, use_rb = case
    g = torch.Generator().manual_seed(sum(case))
    ct = Ci + Ci2
    x = torch.randn(B, Ci, H, H, generator=g)
    x2 = torch.randn(B, Ci2, H, H, generator=g) if Ci2 else None
    w = torch.randn(Co, ct, k, k, generator=g) / (ct * k * k) ** 0.5
    b = torch.randn(Co, generator=g)
    rb = torch.randn(B, Co, generator=g) if use_rb else None
    xin = torch.cat([x, x2], 1) if Ci2 else x
    if up:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xin, w, b, stride=stride, padding=k // 2)
    if rb is not None:
        ref = ref + rb[:, :, None, None]
    if geglu:
        a, gate = ref.chunk(2, 1)
        ref = a * F.gelu(gate)
    res = torch.randn(ref.shape, generator=g) if use_res else None
    if res is not None:
        ref = ref + res
    out = torch.empty(ref.shape, device="cuda")
    dx, dx2, dw, db, dres, drb = dev(x), dev(x2), dev(w), dev(b), dev(res), dev(rb)
    r = L.lib().ldmseg_op_igemm(P(dx), P(dx2), P(dw), P(db), P(dres), P(drb), B, Ci, Ci2, H, H, Co, k, stride, up, geglu,
                                0, 0, 2, P(out), None)
    assert r == 0, L.lib().ldmseg_last_error()
    torch.cuda.synchronize()
    name = L.igemm_last_kernel()
    assert name.startswith("igemm<f32,") and ",0,0" in name, name          # a plain-K-loop fp32 instantiation
    e = rel_err(out, ref)
    assert e < 1e-4, (case, name, e)
    # round 6: dtype 3 = the same arithmetic with the weights split into hi | lo planes once, up front (what a bf16x3 handle holds):
    # the planes are exactly the values the K loop would have computed, so the result must not move a bit
    out3 = torch.empty(ref.shape, device="cuda")
    r = L.lib().ldmseg_op_igemm(P(dx), P(dx2), P(dw), P(db), P(dres), P(drb), B, Ci, Ci2, H, H, Co, k, stride, up, geglu,
                                0, 0, 3, P(out3), None)
    assert r == 0, L.lib().ldmseg_last_error()
    torch.cuda.synchronize()
    assert L.igemm_last_kernel() == name
    assert torch.equal(out3, out), case


@pytest.mark.parametrize("M,K,N,geglu", [(2048, 640, 1920, 0), (512, 1280, 10240, 1), (4096, 320, 960, 0)])
def test_split_bf16_layernorm_folded_gemm(L, M, K, N, geglu):
    """LayerNorm -> Linear / GEGLU with the split-bf16 products (the ',ln' plain-loop instantiations)."""
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g) * 1.5 + 3.0
    gamma = 1 + 0.2 * torch.randn(K, generator=g)
    beta = 0.2 * torch.randn(K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    y = F.linear(F.layer_norm(x, (K,), gamma, beta, 1e-5), w, b)
    if geglu:
        a, gate = y.chunk(2, -1)
        y = a * F.gelu(gate)
    out = torch.empty(y.shape, device="cuda")
    dx, dg, db, dw, dbias = dev(x), dev(gamma), dev(beta), dev(w), dev(b)
    assert L.lib().ldmseg_op_ln_linear(P(dx), P(dg), P(db), P(dw), P(dbias), M, K, N, 1e-5, geglu, 2, P(out), None) == 0
    torch.cuda.synchronize()
    name = L.igemm_last_kernel()
    assert ",ln" in name and name.startswith("igemm<f32,"), name
    assert rel_err(out, y) < 2e-4, name
    out3 = torch.empty(y.shape, device="cuda")             # weights pre-split into planes (dtype 3): bit-identical
    assert L.lib().ldmseg_op_ln_linear(P(dx), P(dg), P(db), P(dw), P(dbias), M, K, N, 1e-5, geglu, 3, P(out3), None) == 0
    torch.cuda.synchronize()
    assert torch.equal(out3, out), name


@pytest.mark.parametrize("B,N,Cc", [(2, 4096, 320), (2, 1024, 640), (8, 256, 1280), (3, 64, 1280), (1, 200, 320), (1, 33, 640)])
def test_split_bf16_attention(L, B, N, Cc):
    """dtype 2 of ldmseg_op_attention: q, k, v and the probabilities split into bf16 hi + lo, every score / output block
    as three bf16 MFMAs (attention.hip X3), tensors fp32 in HBM.  Against the fp64 reference on the UNROUNDED inputs, with a
    dominant key that moves the running maximum; bound 1e-4 (the exact-fp32 kernel sits at 2e-5, bf16 at 2e-2)."""
    g = torch.Generator().manual_seed(N + Cc + 2)
    qkv = torch.randn(B, N, 3 * Cc, generator=g)
    qkv[:, :, :Cc] *= 2.0
    qkv[0, N // 2, Cc:Cc + 40] += 6.0
    ref = attention_ref(qkv, B, N, Cc)
    out = torch.empty(B, N, Cc, device="cuda")
    dq = dev(qkv)
    assert L.lib().ldmseg_op_attention(P(dq), B, N, Cc, 8, 2, P(out), None) == 0, L.lib().ldmseg_last_error()
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    e = rel_err(out, ref)
    assert e < 1e-4, e


# ---- resnet tail as one launch: conv2 + conv_shortcut through an extra centre tap (igemm.hip XT, round 5) ----
# (B, H, W, C = cout, Cs, Cs2, splits)
XT_CASES = {
    "plain": (2, 16, 16, 320, 640, 0, 0),
    "concat": (2, 16, 16, 320, 640, 320, 0),             # shortcut over torch.cat([h, skip]): the extra tap switches tensors
    "ragged": (3, 5, 7, 320, 64, 128, 0),                # M = 105: rows past M, images end inside tiles, every row has padding taps
    "slices3": (1, 8, 8, 320, 192, 0, 3),                # 45 + 3 K tiles over 3 slices of 16: the extra tap is the tail of the last slice
    "slices7": (2, 8, 8, 640, 640, 640, 7),              # 110 K tiles over 7 slices: one slice starts inside the extra tap
    "big_tile": (8, 64, 64, 320, 320, 320, 0),           # the 256 x 160 loader-wave form of the 64 x 64 level
}


@pytest.mark.parametrize("case", sorted(XT_CASES))
def test_resnet_tail_extra_tap(L, case):
    """conv2(h) + conv_shortcut(cat([x, skip])) as ONE implicit GEMM whose K range runs on past the nine taps (IgemmParams::src2):
    against the two torch convs on the bf16-rounded operands, on shapes that hit the edges of the extra segment."""
    B, H, W, Cc, Cs, Cs2, splits = XT_CASES[case]
    g = torch.Generator().manual_seed(len(case) * 17 + Cs)
    h = torch.randn(B, Cc, H, W, generator=g)
    xs = torch.randn(B, Cs, H, W, generator=g)
    xs2 = torch.randn(B, Cs2, H, W, generator=g) if Cs2 else None
    w2 = torch.randn(Cc, Cc, 3, 3, generator=g) / (9 * Cc) ** 0.5
    ws = torch.randn(Cc, Cs + Cs2, 1, 1, generator=g) / (Cs + Cs2) ** 0.5
    b2, bs = torch.randn(Cc, generator=g), torch.randn(Cc, generator=g)
    xin = torch.cat([xs, xs2], 1) if Cs2 else xs
    ref = F.conv2d(bf16_round(h), bf16_round(w2), b2, padding=1) + F.conv2d(bf16_round(xin), bf16_round(ws), bs)
    out = torch.empty(ref.shape, device="cuda")
    dh, dxs, dxs2, dw2, dws, db2, dbs = dev(h), dev(xs), dev(xs2), dev(w2), dev(ws), dev(b2), dev(bs)
    r = L.lib().ldmseg_op_conv3x3_plus_1x1(P(dh), P(dw2), P(db2), P(dxs), P(dxs2), P(dws), P(dbs), B, Cc, Cs, Cs2, H, W, Cc, splits, BF16,
                                           P(out), 0, None, None)
    assert r == 0, (r, L.lib().ldmseg_last_error())
    torch.cuda.synchronize()
    name = L.igemm_last_kernel()
    assert ",xt" in name, name
    assert rel_err(out, ref) < 8e-3, (case, name)
    # the switch (debug key 19) makes the engine keep the two launches: the operator then says so
    lib = L.lib()
    try:
        lib.ldmseg_debug_set(19, 0)
        assert lib.ldmseg_op_conv3x3_plus_1x1(P(dh), P(dw2), P(db2), P(dxs), P(dxs2), P(dws), P(dbs), B, Cc, Cs, Cs2, H, W, Cc, splits, BF16,
                                              P(out), 0, None, None) == -4
    finally:
        lib.ldmseg_debug_set(19, 1)


# ---- upsampler conv as four 2x2 phase convs on the low-resolution map (igemm.hip UP4, round 5) ----
@pytest.mark.parametrize("case", [(8, 320, 8, 320), (2, 640, 16, 640), (4, 64, 8, 160), (1, 128, 16, 320), (16, 1280, 8, 1280)])
def test_upsampler_conv_as_phase_convs(L, case):
    """F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, b, padding=1) - diffusers' Upsample2D - as the engine runs it in
    bf16: 4 B H W virtual rows (phase, image, i, j), K = 4 C with the 3x3 taps that fall on one source pixel summed beforehand.
    Reference on the bf16-rounded INPUT and the fp32 weights rounded per tap (the kernel rounds the per-phase sums instead: the
    bound allows for that), plus an exact check of the weight sums against a torch restatement; debug key 21 = 0 gives the nine-tap
    launch on the same operands."""
    B, Ci, H, Co = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Ci, H, H, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (9 * Ci) ** 0.5
    b = torch.randn(Co, generator=g)
    ref = F.conv2d(F.interpolate(bf16_round(x), scale_factor=2.0, mode="nearest"), w, b, padding=1)       # unrounded weights
    out = torch.empty(ref.shape, device="cuda")
    dx, dw, db = dev(x), dev(w), dev(b)
    lib = L.lib()
    assert lib.ldmseg_op_igemm(P(dx), None, P(dw), P(db), None, None, B, Ci, 0, H, H, Co, 3, 1, 1, 0, 0, 0, BF16, P(out), None) == 0
    torch.cuda.synchronize()
    name = L.igemm_last_kernel()
    assert ",up4" in name, name
    assert rel_err(out, ref) < 8e-3, (case, name)
    try:
        lib.ldmseg_debug_set(21, 0)
        out9 = torch.empty(ref.shape, device="cuda")
        assert lib.ldmseg_op_igemm(P(dx), None, P(dw), P(db), None, None, B, Ci, 0, H, H, Co, 3, 1, 1, 0, 0, 0, BF16, P(out9), None) == 0
        torch.cuda.synchronize()
        assert ",up4" not in L.igemm_last_kernel()
    finally:
        lib.ldmseg_debug_set(21, 1)
    assert rel_err(out9, ref) < 8e-3
    assert float((out - out9).norm() / out9.norm()) < 6e-3          # two roundings of the same convolution
