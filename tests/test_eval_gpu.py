"""End-to-end evaluation flows on the MI355X against the oracle chain (fp32 parity mode):

* BASELINE configs[0]: one 256x256 image, 10-step DDIM, B = 1 - pixels -> RGB image encoder -> sampling loop -> seg-VAE
  decode -> bilinear to the original size -> panoptic post-processing (`TrainerDiffusion.predict_panoptic`, the body of
  the reference's compute_pq, trainers_ldm_cond.py:1218-1313, as driven by tools/main_ldm.py:219-232);
* BASELINE configs[3], the whole inpainting flow instead of random known latents: segment ids -> bit maps
  (coco.py:377-382) -> 2x-1 -> seg-VAE encode -> mode * scaling_factor -> mask-inpainting sampler -> decode.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from oracle import bitcodec as o_bits, ddim as o_ddim, postprocess as o_post, sample as o_sample, unet as o_unet
from oracle import vae as o_vae, vae_image as o_img

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def models(unet_sd, vae_sd):
    from ldmseg_amd import weights
    from ldmseg_amd.models import UNet, GeneralVAESeg, GeneralVAEImage
    isd = weights.generate(weights.vae_image_schema(), seed=11, norm_keys=weights.VAE_IMAGE_NORM_KEYS)
    unet = UNet(unet_sd, in_channels=12, device=DEV, compute_dtype="fp32")
    vae = GeneralVAESeg(vae_sd, scaling_factor=0.2, device=DEV, compute_dtype="fp32")
    enc = GeneralVAEImage(isd, scaling_factor=0.18215, device=DEV, compute_dtype="fp32")
    return unet, vae, enc, isd


def smooth_image(h, w, seed):
    g = torch.Generator().manual_seed(seed)
    low = torch.rand(1, 3, h // 16 + 2, w // 16 + 2, generator=g)
    return F.interpolate(low, size=(h, w), mode="bicubic", align_corners=False).clamp(0, 1)[0]


def test_config0_pixels_to_panoptic_256px_10_steps(models, unet_sd, vae_sd, sched_kw):
    from ldmseg_amd.schedulers import DDIMNoiseScheduler
    from ldmseg_amd.trainers import TrainerDiffusion
    torch.set_num_threads(32)
    unet, vae, enc, isd = models
    tr = TrainerDiffusion(vae, unet, DDIMNoiseScheduler(**sched_kw), vae_image=enc, latent_size=32)
    img = smooth_image(256, 256, 5)[None]                         # what load_rgb() hands over: [1,3,256,256] in [0,1]
    hw = (240, 320)                                               # original size of the photo
    res, mid = tr.predict_panoptic(img.to(DEV), [hw], None, num_inference_steps=10, seed=42, threshold_output=True,
                                   mask_th=0.02, count_th=64, overlap_th=0.5, return_intermediates=True)
    # ---- the same chain on the oracle ----
    with torch.no_grad():
        rgb_lat = o_img.encode_mode(isd, img, 0.18215)
        so = o_ddim.OracleDDIM(**sched_kw)
        so.set_timesteps_inference(10)
        lat = o_sample.sample(lambda inp, t: o_unet.unet_forward(unet_sd, inp, t), so, rgb_lat, seed=42)
        logits = o_sample.decode_latents(lambda z: o_vae.decode(vae_sd, z), lat, 0.2)
        logits = F.interpolate(logits, size=(256, 256), mode="bilinear", align_corners=False)
        final = F.interpolate(logits, size=hw, mode="bilinear", align_corners=False)[0]
    assert rel_err(mid["rgb_latents"], rgb_lat) < 1e-3
    assert rel_err(mid["latents"], lat) < 3e-3                    # 10 UNet evaluations deep
    assert rel_err(mid["logits"], logits) < 3e-3
    pan_ref, info_ref, raw_ref, _ = o_post.panoptic_postprocess(final, threshold_output=True, mask_th=0.02, count_th=64,
                                                                overlap_th=0.5, ignore_label=0)
    pan, info = res[0]["panoptic_seg"]
    assert pan.shape == hw and pan.dtype == torch.int32
    top2 = final.topk(2, dim=0)[0]
    clear = (top2[0] - top2[1]) > 1e-2 * float(final.abs().max())
    agree = (pan.cpu().numpy() == pan_ref)
    assert clear.float().mean() > 0.5
    assert agree[clear.numpy()].mean() > 0.995 and agree.mean() > 0.97
    # the kept segments agree except for labels whose pixel count sits at a threshold
    ids, ids_ref = {s["id"] for s in info}, {s["id"] for s in info_ref}
    assert len(ids ^ ids_ref) <= max(2, len(ids_ref) // 10), (sorted(ids ^ ids_ref), len(ids_ref))


def test_inpainting_full_flow_vs_oracle(models, unet_sd, vae_sd, sched_kw):
    from ldmseg_amd.data import encode_bitmap
    from ldmseg_amd.schedulers import DDIMNoiseScheduler
    from ldmseg_amd.trainers import TrainerDiffusion
    torch.set_num_threads(32)
    unet, vae, enc, _ = models
    tr = TrainerDiffusion(vae, unet, DDIMNoiseScheduler(**sched_kw), vae_image=enc, latent_size=32)
    g = torch.Generator().manual_seed(11)
    B, S, L = 2, 256, 32
    # partial segmentation: a few rectangles of segment ids on a void (0) background
    ids = torch.zeros(B, S, S, dtype=torch.int64)
    for b in range(B):
        for k in range(6):
            y, x = torch.randint(0, S - 64, (2,), generator=g).tolist()
            ids[b, y:y + 64, x:x + 48] = int(torch.randint(1, 128, (1,), generator=g))
    rgb = 0.18215 * torch.randn(B, 4, L, L, generator=g)
    known = torch.rand(B, 1, L, L, generator=g) < 0.5
    # ---- HIP path: bit maps -> 2x-1 (fused) -> seg-VAE encoder -> mode * scaling -> inpainting sampler -> decode
    bits, _ = encode_bitmap(ids.to(DEV))                                  # [B,7,S,S], void -> 0.5
    z0, z0_mean = tr.encode_inputs(bits, encode_func=vae.encode, scaling_factor=vae.scaling_factor)
    out = tr.sample_inpaint([""] * B, known, z0, num_inference_steps=6, seed=42, rgb_latents=rgb.to(DEV))
    logits = tr.decode_latents(out, return_logits=True)
    # ---- oracle
    with torch.no_grad():
        ob = torch.stack([torch.from_numpy(o_bits.encode_bitmap(ids[b].numpy())[0]) for b in range(B)])
        assert torch.equal(bits.cpu(), ob)                                 # integer/bit work: exact
        mom = o_vae.encode_moments(vae_sd, 2.0 * ob - 1.0)
        z0_ref = mom[:, :4] * 0.2
        so = o_ddim.OracleDDIM(**sched_kw)
        so.set_timesteps_inference(6)
        ref = o_sample.sample_inpaint(lambda inp, t: o_unet.unet_forward(unet_sd, inp, t), so, rgb, z0_ref, known, seed=42)
        ref_logits = o_sample.decode_latents(lambda z: o_vae.decode(vae_sd, z), ref, 0.2)
    assert rel_err(z0, z0_ref) < 1e-3 and torch.equal(z0, z0_mean)
    assert rel_err(out, ref) < 3e-3
    m = known.expand_as(ref)
    assert torch.equal(out.cpu()[m], z0.cpu()[m])                         # the known latents come back exactly
    assert rel_err(logits, ref_logits) < 3e-3
