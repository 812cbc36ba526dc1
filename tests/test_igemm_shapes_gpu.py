"""Parity of the SHIPPED implicit-GEMM instantiations at the real UNet launch shapes of the three single-GPU
configurations of BASELINE.json: configs[1] (B = 8, L = 64), configs[3] (B = 16, L = 64), configs[4] (B = 4, L = 128).

Every distinct conv / Linear / GEGLU launch shape of a UNet forward of each configuration is launched through
``ldmseg_op_igemm`` - the engine's own launch path: NHWC operands, the engine's split-K plan, the row-major store /
GEGLU epilogues, residual and time-embedding bias rows - in bf16 and fp32 under the shipped tile policy, and compared
with the torch-CPU op the reference executes there (F.conv2d / F.linear / GEGLU of diffusers 0.16.1).
Each test records which template instantiation ran; the last test runs full-size forwards with the dispatch log on and
fails if the forward used an instantiation that no per-layer oracle comparison above has exercised.
"""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu

F32, BF16 = 0, 1
# (batch, latent size): the measured launch table (csrc/igemm_tuned.inc) has entries - instantiation x K-slice count - for
# exactly these three; every other shape goes through the rules that the same lists exercise
CONFIGS = [(8, 64), (16, 64), (4, 128)]
CFG_IDS = [f"b{b}l{l}" for b, l in CONFIGS]
SEEN = {(c, dt): set() for c in CONFIGS for dt in (F32, BF16)}


@pytest.fixture(scope="module")
def L():
    from ldmseg_amd import _lib
    assert _lib.lib().ldmseg_debug_get(1) == _lib.lib().ldmseg_debug_get(-1), "a previous test leaked a tile policy"
    assert _lib.lib().ldmseg_debug_get(12) == 3 and _lib.lib().ldmseg_debug_get(14) == 3, "a previous test leaked a fused-kernel switch"
    assert _lib.lib().ldmseg_debug_get(19) == 1 and _lib.lib().ldmseg_debug_get(17) == 0 and _lib.lib().ldmseg_debug_get(20) == 1 and \
        _lib.lib().ldmseg_debug_get(21) == 1 and _lib.lib().ldmseg_debug_get(22) == 1, \
        "a previous test leaked a GEMM-path switch"
    return _lib


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def dev(t):
    return t.to("cuda", torch.float32).contiguous() if t is not None else None


def P(t):
    import ctypes as C
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


# (H, Ci, Ci2, Co, k, stride, up, geglu, resid, rowbias)  -  H = input side AT L = 64 (scaled by L / 64 for the other latent size)
SHAPES = [
    # ---- 64x64 maps (M = 32768)
    (64, 320, 0, 320, 3, 1, 0, 0, 0, 1),      # resnet conv1 (+ time-embedding row)
    (64, 320, 0, 320, 3, 1, 0, 0, 1, 0),      # resnet conv2 (+ residual)
    (64, 320, 320, 320, 3, 1, 0, 0, 0, 1),    # up-path conv1 on cat([h, skip])
    (64, 640, 320, 320, 3, 1, 0, 0, 0, 1),
    (64, 12, 0, 320, 3, 1, 0, 0, 0, 0),       # conv_in
    (64, 320, 0, 320, 1, 1, 0, 0, 1, 0),      # proj_out / to_out (+ residual)
    (64, 320, 0, 320, 1, 1, 0, 0, 0, 0),      # proj_in
    (64, 320, 0, 960, 1, 1, 0, 0, 0, 0),      # fused q|k|v
    (64, 320, 0, 2560, 1, 1, 0, 1, 0, 0),     # GEGLU
    (64, 1280, 0, 320, 1, 1, 0, 0, 1, 0),     # ff.net.2 (+ residual)
    (64, 320, 320, 320, 1, 1, 0, 0, 0, 0),    # conv_shortcut on a concat
    (64, 640, 320, 320, 1, 1, 0, 0, 0, 0),
    (32, 640, 0, 640, 3, 1, 1, 0, 0, 0),      # upsampler conv: nearest x2 folded into the gather (M = 32768)
    (64, 320, 0, 320, 3, 2, 0, 0, 0, 0),      # downsampler conv, stride 2 (M = 8192)
    # ---- 32x32 maps (M = 8192)
    (32, 640, 0, 640, 3, 1, 0, 0, 0, 1),
    (32, 640, 0, 640, 3, 1, 0, 0, 1, 0),
    (32, 320, 0, 640, 3, 1, 0, 0, 0, 1),
    (32, 640, 640, 640, 3, 1, 0, 0, 0, 1),
    (32, 1280, 640, 640, 3, 1, 0, 0, 0, 1),
    (32, 640, 320, 640, 3, 1, 0, 0, 0, 1),
    (32, 640, 0, 640, 1, 1, 0, 0, 1, 0),
    (32, 640, 0, 1920, 1, 1, 0, 0, 0, 0),
    (32, 640, 0, 5120, 1, 1, 0, 1, 0, 0),
    (32, 2560, 0, 640, 1, 1, 0, 0, 1, 0),
    (32, 2560, 640, 640, 1, 1, 0, 0, 1, 0),   # ff.net.2 + proj_out chained into one Linear over cat([g, h]) (+ x) - round 5
    (32, 320, 0, 640, 1, 1, 0, 0, 0, 0),
    (32, 1280, 640, 640, 1, 1, 0, 0, 0, 0),
    (32, 640, 640, 640, 1, 1, 0, 0, 0, 0),
    (32, 640, 320, 640, 1, 1, 0, 0, 0, 0),
    (16, 1280, 0, 1280, 3, 1, 1, 0, 0, 0),    # upsampler conv 16 -> 32 (M = 8192)
    (32, 640, 0, 640, 3, 2, 0, 0, 0, 0),      # downsampler (M = 2048)
    # ---- 16x16 maps (M = 2048)
    (16, 1280, 0, 1280, 3, 1, 0, 0, 0, 1),
    (16, 1280, 0, 1280, 3, 1, 0, 0, 1, 0),
    (16, 640, 0, 1280, 3, 1, 0, 0, 0, 1),
    (16, 1280, 1280, 1280, 3, 1, 0, 0, 0, 1),
    (16, 1280, 640, 1280, 3, 1, 0, 0, 0, 1),
    (16, 1280, 0, 1280, 1, 1, 0, 0, 1, 0),
    (16, 1280, 0, 3840, 1, 1, 0, 0, 0, 0),
    (16, 1280, 0, 10240, 1, 1, 0, 1, 0, 0),
    (16, 5120, 0, 1280, 1, 1, 0, 0, 1, 0),
    (16, 5120, 1280, 1280, 1, 1, 0, 0, 1, 0),  # chained ff.net.2 + proj_out
    (16, 640, 0, 1280, 1, 1, 0, 0, 0, 0),
    (16, 1280, 1280, 1280, 1, 1, 0, 0, 0, 0),
    (16, 1280, 640, 1280, 1, 1, 0, 0, 0, 0),
    (8, 1280, 0, 1280, 3, 1, 1, 0, 0, 0),     # upsampler conv 8 -> 16 (M = 2048)
    (16, 1280, 0, 1280, 3, 2, 0, 0, 0, 0),    # downsampler (M = 512)
    # ---- 8x8 maps (M = 512)
    (8, 1280, 0, 1280, 3, 1, 0, 0, 0, 1),
    (8, 1280, 0, 1280, 3, 1, 0, 0, 1, 0),
    (8, 1280, 1280, 1280, 3, 1, 0, 0, 0, 1),
    (8, 1280, 0, 1280, 1, 1, 0, 0, 1, 0),
    (8, 1280, 0, 3840, 1, 1, 0, 0, 0, 0),
    (8, 1280, 0, 10240, 1, 1, 0, 1, 0, 0),
    (8, 5120, 0, 1280, 1, 1, 0, 0, 1, 0),
    (8, 5120, 1280, 1280, 1, 1, 0, 0, 1, 0),   # chained ff.net.2 + proj_out
    (8, 1280, 1280, 1280, 1, 1, 0, 0, 0, 0),
]


@pytest.mark.parametrize("dt", [BF16, F32])
@pytest.mark.parametrize("cfg", CONFIGS, ids=CFG_IDS)
@pytest.mark.parametrize("case", SHAPES)
def test_unet_layer_shape_vs_oracle(L, dt, cfg, case):
    H, Ci, Ci2, Co, k, stride, up, geglu, use_res, use_rb = case
    B, lat = cfg
    H = H * lat // 64
    torch.set_num_threads(64)
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    ct = Ci + Ci2
    x = torch.randn(B, Ci, H, H, generator=g)
    x2 = torch.randn(B, Ci2, H, H, generator=g) if Ci2 else None
    w = torch.randn(Co, ct, k, k, generator=g) / (ct * k * k) ** 0.5
    b = torch.randn(Co, generator=g)
    rb = torch.randn(B, Co, generator=g) if use_rb else None
    xin = torch.cat([x, x2], 1) if Ci2 else x
    xin_r, w_r = (bf16_round(xin), bf16_round(w)) if dt == BF16 else (xin, w)
    if up:
        xin_r = F.interpolate(xin_r, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xin_r, w_r, b, stride=stride, padding=k // 2)
    if rb is not None:
        ref = ref + rb[:, :, None, None]
    if geglu:
        a, gate = ref.chunk(2, 1)
        ref = a * F.gelu(gate)
    res = torch.randn(ref.shape, generator=g) if use_res else None
    if res is not None:
        ref = ref + (bf16_round(res) if dt == BF16 else res)
    out = torch.empty(ref.shape, device="cuda")
    dx, dx2, dw, db, dres, drb = dev(x), dev(x2), dev(w), dev(b), dev(res), dev(rb)
    r = L.lib().ldmseg_op_igemm(P(dx), P(dx2), P(dw), P(db), P(dres), P(drb), B, Ci, Ci2, H, H, Co, k, stride, up, geglu,
                                0, 0, dt, P(out), None)
    assert r == 0, L.lib().ldmseg_last_error()
    torch.cuda.synchronize()
    name = L.igemm_last_kernel()
    SEEN[(cfg, dt)].add(name.split(" ")[0])
    # bf16: operands rounded identically, the difference is accumulation order + the bf16 rounding of the stored output
    assert rel_err(out, ref) < (8e-3 if dt == BF16 else 1e-4), (cfg, case, name)
    if "/splitk-cf" in name:
        # the same instantiation writes plain slabs where the consumer is not the launch itself (resnet conv1 -> norm2 on the small
        # maps: the fused finish + GroupNorm kernel reads them): compare that form here too - bit for bit with the in-launch finish
        lib = L.lib()
        saved = lib.ldmseg_debug_get(23)
        out2 = torch.empty(ref.shape, device="cuda")
        try:
            assert lib.ldmseg_debug_set(23, 0) == 0
            assert lib.ldmseg_op_igemm(P(dx), P(dx2), P(dw), P(db), P(dres), P(drb), B, Ci, Ci2, H, H, Co, k, stride, up, geglu,
                                       0, 0, dt, P(out2), None) == 0
            torch.cuda.synchronize()
            name2 = L.igemm_last_kernel()
        finally:
            lib.ldmseg_debug_set(23, saved)
        assert name2.split(" ")[0] == name.split(" ")[0].replace("/splitk-cf", "/splitk"), (name, name2)
        SEEN[(cfg, dt)].add(name2.split(" ")[0])
        assert torch.equal(out2, out), (cfg, case, name2)


LN_SHAPES = [   # (M at B = 8 / L = 64, K = C, N, geglu): norm1 -> q|k|v and norm3 -> ff.net.0.proj of every transformer level
    (32768, 320, 960, 0), (32768, 320, 2560, 1), (8192, 640, 1920, 0), (8192, 640, 5120, 1),
    (2048, 1280, 3840, 0), (2048, 1280, 10240, 1), (512, 1280, 3840, 0), (512, 1280, 10240, 1),
]


@pytest.mark.parametrize("dt", [BF16, F32])
@pytest.mark.parametrize("cfg", CONFIGS, ids=CFG_IDS)
@pytest.mark.parametrize("M,K,N,geglu", LN_SHAPES)
def test_unet_layernorm_folded_gemm_vs_oracle(L, dt, cfg, M, K, N, geglu):
    """LayerNorm -> Linear / GEGLU of the transformer blocks as the engine runs them: one statistics pass, then the GEMM
    on the un-normalised tokens with gamma folded into the weights and rstd*(acc - mean*c1) + c2 in the epilogue (its own
    template instantiations, ',ln').  Reference: F.layer_norm + F.linear (+ GEGLU); the tokens carry a large common offset
    (mean 3, std 1.5) so the mean cancellation of the epilogue is exercised."""
    torch.set_num_threads(64)
    M = M * cfg[0] * cfg[1] * cfg[1] // (8 * 64 * 64)
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
    name = L.igemm_last_kernel()
    assert ",ln" in name
    SEEN[(cfg, dt)].add(name.split(" ")[0])
    # bf16: gamma*W and the output are rounded to bf16 (the unfolded form rounds LN(x) and W instead)
    assert rel_err(out, y) < (1.5e-2 if dt == BF16 else 1e-4), name


@pytest.mark.parametrize("dt", [BF16, F32])
@pytest.mark.parametrize("cfg", CONFIGS, ids=CFG_IDS)
def test_conv_out_shape_vs_oracle(L, dt, cfg):
    """conv_out: 320 -> 4 channels at the full latent resolution, written straight to fp32 NCHW (EPI_NCHW_F32, the narrow-N tile)."""
    B, lat = cfg
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, 320, lat, lat, generator=g)
    w = torch.randn(4, 320, 3, 3, generator=g) / 2880 ** 0.5
    b = torch.randn(4, generator=g)
    xr, wr = (bf16_round(x), bf16_round(w)) if dt == BF16 else (x, w)
    ref = F.conv2d(xr, wr, b, padding=1)
    out = torch.empty(ref.shape, device="cuda")
    dx, dw, db = dev(x), dev(w), dev(b)
    assert L.lib().ldmseg_op_conv2d(P(dx), None, P(dw), P(db), B, 320, 0, lat, lat, 4, 3, 1, 0, dt, P(out), None) == 0
    torch.cuda.synchronize()
    SEEN[(cfg, dt)].add(L.igemm_last_kernel().split(" ")[0])
    assert rel_err(out, ref) < (1e-3 if dt == BF16 else 1e-4)
    if dt == BF16:          # the bf16 forward runs conv_out as the halo-resident stencil of tail.hip
        out2 = torch.empty(ref.shape, device="cuda")
        assert L.lib().ldmseg_op_conv_out_tail(P(dx), P(dw), P(db), B, lat, lat, P(out2), 0, 0, None, 0, 0, 1.0, None, None, None, None,
                                               None, None, 0.0, 0.0, None, None) == 0
        torch.cuda.synchronize()
        assert rel_err(out2, ref) < 1e-3
        SEEN[(cfg, dt)].add("conv_out_tail<bf16>")


# conv2 + conv_shortcut of the resnets whose input and output channel counts differ, as the bf16 forward runs them (round 5):
# ONE launch, K = 9 * C + Cs + Cs2.  (H at L = 64, C = cout, Cs = hidden channels, Cs2 = skip channels of torch.cat([h, skip], 1))
XT_SHAPES = [
    (64, 320, 640, 320), (64, 320, 320, 320),                                           # up_blocks.3
    (32, 640, 1280, 640), (32, 640, 640, 640), (32, 640, 640, 320), (32, 640, 320, 0),  # up_blocks.2, down_blocks.1.resnets.0
    (16, 1280, 1280, 1280), (16, 1280, 1280, 640), (16, 1280, 640, 0),                  # up_blocks.1, down_blocks.2.resnets.0
    (8, 1280, 1280, 1280),                                                              # up_blocks.0
]


@pytest.mark.parametrize("cfg", CONFIGS, ids=CFG_IDS)
@pytest.mark.parametrize("case", XT_SHAPES)
def test_resnet_tail_one_launch_vs_oracle(L, cfg, case):
    """F.conv2d(h, w2, b2, padding=1) + F.conv2d(cat([x, skip]), ws, bs) (diffusers ResnetBlock2D: conv2 + conv_shortcut) against the
    engine's single extra-tap launch at the configuration's shapes; records the ',xt' instantiation that ran."""
    import ctypes as C
    H, Cc, Cs, Cs2 = case
    B, lat = cfg
    H = H * lat // 64
    torch.set_num_threads(64)
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    h = torch.randn(B, Cc, H, H, generator=g)
    xs = torch.randn(B, Cs, H, H, generator=g)
    xs2 = torch.randn(B, Cs2, H, H, generator=g) if Cs2 else None
    w2 = torch.randn(Cc, Cc, 3, 3, generator=g) / (9 * Cc) ** 0.5
    ws = torch.randn(Cc, Cs + Cs2, 1, 1, generator=g) / (Cs + Cs2) ** 0.5
    b2, bs = torch.randn(Cc, generator=g), torch.randn(Cc, generator=g)
    xin = torch.cat([xs, xs2], 1) if Cs2 else xs
    ref = F.conv2d(bf16_round(h), bf16_round(w2), b2, padding=1) + F.conv2d(bf16_round(xin), bf16_round(ws), bs)
    out = torch.empty(ref.shape, device="cuda")
    dh, dxs, dxs2, dw2, dws, db2, dbs = dev(h), dev(xs), dev(xs2), dev(w2), dev(ws), dev(b2), dev(bs)
    r = L.lib().ldmseg_op_conv3x3_plus_1x1(P(dh), P(dw2), P(db2), P(dxs), P(dxs2), P(dws), P(dbs), B, Cc, Cs, Cs2, H, H, Cc, 0, BF16,
                                           P(out), 0, None, None)
    assert r == 0, (r, L.lib().ldmseg_last_error())
    torch.cuda.synchronize()
    name = L.igemm_last_kernel()
    assert ",xt" in name, name
    SEEN[(cfg, BF16)].add(name.split(" ")[0])
    assert rel_err(out, ref) < 8e-3, (cfg, case, name)


@pytest.mark.parametrize("cfg", CONFIGS, ids=CFG_IDS)
def test_fused_feed_forward_at_config_shape(L, cfg):
    """The row-local fused feed-forward kernel (tfuse.hip: LayerNorm_3 -> GEGLU -> ff.net.2 (+h) -> proj_out (+x)) that the
    bf16 forward runs on the 320-channel level, at this configuration's token count M = B * L * L, against torch on the same
    bf16-rounded operands (the arithmetic of oracle/unet.py::transformer)."""
    from test_ops_gpu import _ff_case, _ff_ref, _ff_run
    B, lat = cfg
    M = B * lat * lat
    case = _ff_case(M, 320, 7 + M)
    ref = _ff_ref(*case)
    for mode, name in ((3, "mlp_fused<bf16,proj=1>"), (1, "mlp_fused<bf16,proj=0>")):
        out, _ = _ff_run(L, case, M, 320, mode)
        l2 = float((out.double() - ref.double()).norm() / ref.double().norm())
        assert torch.isfinite(out).all() and l2 < 6e-3 and rel_err(out, ref) < 3e-2, (cfg, mode, l2)
        SEEN[(cfg, BF16)].add(name)


@pytest.mark.parametrize("cfg", CONFIGS, ids=CFG_IDS)
def test_fused_transformer_entry_at_config_shape(L, cfg):
    """The row-local fused entry kernel (tproj.hip: proj_in -> LayerNorm_1 -> q|k|v) that the bf16 forward runs on the
    320-channel level, at this configuration's token count, against torch on the same bf16-rounded operands."""
    from test_ops_gpu import _tin_case, _tin_ref, _tin_run
    B, lat = cfg
    M = B * lat * lat
    case = _tin_case(M, 320, 11 + M)
    href, qref = _tin_ref(*case)
    h, qkv, _ = _tin_run(L, case, M, 320, 1)
    l2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    assert torch.isfinite(qkv).all() and l2(h, href) < 3e-3 and l2(qkv, qref) < 6e-3, (cfg, l2(h, href), l2(qkv, qref))
    SEEN[(cfg, BF16)].add("proj_ln_qkv<bf16>")


@pytest.mark.parametrize("cfg", CONFIGS, ids=CFG_IDS)
def test_fused_transformer_entry_with_groupnorm_at_config_shape(L, cfg):
    """The same kernel with the transformer's GroupNorm folded in (statistics pass + apply sweep on the LDS tile: what the bf16
    forward launches since round 5), at this configuration's image count and map size, against torch GroupNorm -> proj_in ->
    LayerNorm_1 -> q|k|v on the bf16-rounded operands."""
    import torch.nn.functional as F
    from test_ops_gpu import _gtin_run, _tin_case, _tin_ref, bf16_round
    B, lat = cfg
    HW, M = lat * lat, B * lat * lat
    g = torch.Generator().manual_seed(13 + M)
    case = _tin_case(M, 320, 17 + M)
    x = torch.randn(B, HW, 320, generator=g) * (0.5 + torch.rand(1, 1, 320, generator=g)) + 2.0 * torch.randn(1, 1, 320, generator=g)
    gg = 1 + 0.3 * torch.randn(320, generator=g)
    gb = 0.3 * torch.randn(320, generator=g)
    xn = F.group_norm(bf16_round(x).permute(0, 2, 1).reshape(B, 320, HW, 1), 32, gg, gb, 1e-6).reshape(B, 320, HW).permute(0, 2, 1)
    href, qref = _tin_ref(xn.reshape(M, 320), *case[1:])
    h, qkv, _ = _gtin_run(L, x.reshape(M, 320), gg, gb, B, 1, case, M, 320, 1)
    l2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    assert torch.isfinite(qkv).all() and l2(h, href) < 4e-3 and l2(qkv, qref) < 7e-3, (cfg, l2(h, href), l2(qkv, qref))
    SEEN[(cfg, BF16)].add("proj_ln_qkv<bf16,gn>")


@pytest.mark.parametrize("mode,dt", [("bf16", BF16), ("fp32", F32)])
@pytest.mark.parametrize("cfg", CONFIGS, ids=CFG_IDS)
def test_every_forward_instantiation_is_oracle_tested(L, unet_sd, cfg, mode, dt):
    """Run a forward of the configuration (BASELINE configs[1] / [3] / [4]) with the dispatch log on: every igemm
    instantiation - tile shape, wave layout, ring depth, K-sliced or not - it launches must be one that a per-layer test
    above has just compared with the oracle AT THIS CONFIGURATION'S shapes."""
    from ldmseg_amd.models import UNet
    B, lat = cfg
    u = UNet(unet_sd, in_channels=12, device="cuda:0", compute_dtype=mode)
    x = torch.randn(B, 12, lat, lat, generator=torch.Generator().manual_seed(1)).cuda()
    L.igemm_log(True)
    try:
        y = u(x, 499).sample
        torch.cuda.synchronize()
        used = L.igemm_log_read()
    finally:
        L.igemm_log(False)
    assert torch.isfinite(y).all()
    assert len(used) >= 4, used
    missing = used - SEEN[(cfg, dt)]
    assert not missing, (f"{cfg}: forward instantiations without a per-layer oracle test: {sorted(missing)}; "
                         f"tested: {sorted(SEEN[(cfg, dt)])}")
