"""GPU parity of the image-VAE encoder against oracle/vae_image.py (parity unpinned: the
AutoencoderKL arithmetic is diffusers', restated).  fp32 path <= 1e-3 max-norm relative (north-star tolerance),
bf16 path within the bf16 budget."""
import pytest
import torch

from conftest import rel_err
from ldmseg_amd import weights
from oracle import vae_image as o_vi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vi_sd():
    return weights.generate(weights.vae_image_schema(), seed=11, norm_keys=weights.VAE_IMAGE_NORM_KEYS)


@pytest.fixture(scope="module")
def encoders(vi_sd):
    from ldmseg_amd.models import GeneralVAEImage
    return {dt: GeneralVAEImage(vi_sd, scaling_factor=0.18215, device="cuda:0", compute_dtype=dt) for dt in ("fp32", "bf16", "bf16x3")}


def test_structure(encoders):
    assert encoders["fp32"].num_parameters == 34_163_592 + 72
    assert encoders["bf16"].num_parameters == 34_163_592 + 72


@pytest.mark.parametrize("B,H,W", [(1, 64, 64), (2, 128, 64), (1, 64, 192)])
def test_fp32_parity_vs_oracle(encoders, vi_sd, B, H, W):
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(H + W))
    ref = o_vi.encode_moments(vi_sd, 2 * x - 1)
    got = encoders["fp32"].encode_moments(x.cuda(), in_mul=2.0, in_add=-1.0).cpu()
    assert got.shape == ref.shape
    assert rel_err(got, ref) <= 1e-3                          # north-star tolerance, fp32 vs torch-CPU


@pytest.mark.parametrize("B,H,W", [(1, 64, 64), (2, 128, 64)])
def test_bf16x3_parity_vs_oracle(encoders, vi_sd, B, H, W):
    """compute_dtype="bf16x3" (fp32 storage, every GEMM - the single-head attention's S = Q K^T and O = P V included - as three bf16
    MFMAs on hi + lo operands): the exact mode's bound."""
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(H + W + 1))
    ref = o_vi.encode_moments(vi_sd, 2 * x - 1)
    got = encoders["bf16x3"].encode_moments(x.cuda(), in_mul=2.0, in_add=-1.0).cpu()
    assert got.shape == ref.shape
    assert rel_err(got, ref) <= 1e-3


def test_bf16_close_to_oracle(encoders, vi_sd):
    x = torch.rand(2, 3, 128, 128, generator=torch.Generator().manual_seed(5))
    ref = o_vi.encode_moments(vi_sd, 2 * x - 1)
    got = encoders["bf16"].encode_moments(x.cuda(), in_mul=2.0, in_add=-1.0).cpu()
    assert rel_err(got, ref) <= 6e-2                          # bf16 storage through 23 conv/norm layers


def test_encode_inputs_surface(encoders, vi_sd, vae_sd):
    """TrainerDiffusion.encode_inputs (trainers_ldm_cond.py:335-394): default = image VAE; seg VAE via encode_func."""
    from ldmseg_amd.models import GeneralVAESeg
    from ldmseg_amd.trainers import TrainerDiffusion
    from oracle import vae as o_vae
    enc = encoders["fp32"]
    seg = GeneralVAESeg(vae_sd, scaling_factor=0.2, device="cuda:0", compute_dtype="fp32")
    tr = TrainerDiffusion.__new__(TrainerDiffusion)
    tr.vae_image, tr.vae_semseg, tr.latent_size = enc, seg, 8
    x = torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(9))
    lat, mean = tr.encode_inputs(x.cuda())
    ref = o_vi.encode_mode(vi_sd, x)
    assert torch.equal(lat, mean)
    assert torch.allclose(lat.cpu(), ref, rtol=0, atol=1e-3 * ref.abs().max().item())
    g = torch.Generator(device="cuda").manual_seed(1)
    lat_s, mean_s = tr.encode_inputs(x.cuda(), sample_posterior=True, generator=g)
    assert torch.equal(mean_s, mean) and not torch.equal(lat_s, mean_s) and torch.isfinite(lat_s).all()
    # resize path (:365-366, :377-389)
    lat_r, _ = tr.encode_inputs(torch.rand(1, 3, 96, 96).cuda(), resize=64)
    assert lat_r.shape == (1, 4, 8, 8)
    # segmentation maps through the seg VAE with its own scaling factor
    bits = torch.rand(1, 7, 64, 64, generator=torch.Generator().manual_seed(2)).round()
    lat_seg, _ = tr.encode_inputs(bits.cuda(), encode_func=seg.encode, scaling_factor=seg.scaling_factor)
    ref_seg = o_vae.encode_moments(vae_sd, 2 * bits - 1)[:, :4] * 0.2
    assert torch.allclose(lat_seg.cpu(), ref_seg, rtol=0, atol=1e-3 * ref_seg.abs().max().item())
    enc.set_scaling_factor(0.5)
    assert enc.scaling_factor == 0.5
    enc.set_scaling_factor(0.18215)


def test_full_size_properties(encoders):
    """512x512 (BASELINE size): finite, deterministic, batch-permutation equivariant."""
    enc = encoders["bf16"]
    x = torch.rand(2, 3, 512, 512, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    a = enc.encode_moments(x, 2.0, -1.0)
    b = enc.encode_moments(x, 2.0, -1.0)
    assert a.shape == (2, 8, 64, 64) and torch.isfinite(a).all() and torch.equal(a, b)
    c = enc.encode_moments(x.flip(0).contiguous(), 2.0, -1.0)
    assert torch.equal(c.flip(0), a)


def test_rejects_bad_inputs(encoders):
    with pytest.raises(RuntimeError):
        encoders["fp32"].encode_moments(torch.zeros(1, 3, 64, 64))
    with pytest.raises(RuntimeError):
        encoders["fp32"].encode_moments(torch.zeros(1, 3, 64, 96, device="cuda"))   # 8*12 = 96 tokens, not /64
