"""Generate golden vectors by IMPORTING the reference (runs only where /root/reference exists).

    python tests/golden/make_golden.py

Writes small .npz fixtures next to this file.  Nothing from the reference is copied: the
script stubs the packages the reference imports at module level but that are absent here
(detectron2, easydict, diffusers - SURVEY.md App. C), imports
``ldmseg.schedulers.ddim_scheduler.DDIMNoiseScheduler`` and ``ldmseg.models.vae.GeneralVAESeg``
and records their outputs on seeded inputs.  Weights for the seg-VAE come from this
repository's deterministic generator (ldmseg_amd.weights) loaded into the reference module
with ``load_state_dict(strict=True)``.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"

BASE_SCHED = dict(prediction_type="epsilon", beta_schedule="scaled_linear", num_train_timesteps=1000,
                  beta_start=0.00085, beta_end=0.012, steps_offset=1, clip_sample=False,
                  set_alpha_to_one=False, thresholding=False, dynamic_thresholding_ratio=0.995,
                  clip_sample_range=1.0, sample_max_value=1.0, weight="none", max_snr=5.0)   # base.yaml:48-62


def stub_modules():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    dummy = type("Dummy", (), {})
    import transformers  # noqa: F401  (must be imported before any torchvision stub)
    mod("detectron2"); mod("detectron2.utils")
    mod("detectron2.utils.visualizer", Visualizer=dummy, _PanopticPrediction=dummy, ColorMode=dummy,
        _OFF_WHITE=None, _create_text_labels=None)
    mod("easydict", EasyDict=dict)
    mod("diffusers", AutoencoderKL=type("AutoencoderKL", (torch.nn.Module,), {}),
        UNet2DConditionModel=type("UNet2DConditionModel", (torch.nn.Module,), {}))
    mod("diffusers.models")
    mod("diffusers.models.unet_2d_blocks", UNetMidBlock2D=dummy)
    mod("diffusers.training_utils", EMAModel=dummy)


def scheduler_golden(DDIM):
    out = {}
    s = DDIM(**BASE_SCHED)
    out["betas"] = s.betas.numpy()
    out["alphas_cumprod"] = s.alphas_cumprod.numpy()
    out["final_alpha_cumprod"] = np.float32(s.final_alpha_cumprod)
    out["timesteps_default"] = s.timesteps.numpy()
    for n in (10, 50, 30, 7, 1000):
        s.set_timesteps_inference(n)
        out[f"timesteps_{n}"] = s.timesteps.numpy()
    s.set_timesteps_inference(50, tmin=500)
    out["timesteps_50_tmin500"] = s.timesteps.numpy()
    for mode in ("none", "max_clamp_snr", "linear", "fixed"):
        out[f"weights_{mode}"] = DDIM(**{**BASE_SCHED, "weight": mode}).weights.numpy().astype(np.float32)
    for sched in ("linear", "squaredcos_cap_v2", "sigmoid"):
        out[f"alphas_cumprod_{sched}"] = DDIM(**{**BASE_SCHED, "beta_schedule": sched}).alphas_cumprod.numpy()
    out["alphas_cumprod_one"] = np.float32(DDIM(**{**BASE_SCHED, "set_alpha_to_one": True}).final_alpha_cumprod)

    g = torch.Generator().manual_seed(0)
    eps = torch.randn((1, 4, 8, 8), generator=g)
    x = torch.randn((1, 4, 8, 8), generator=g)
    out["step_eps"] = eps.numpy()
    out["step_x"] = x.numpy()
    for pt in ("epsilon", "sample", "v_prediction"):
        for clip in (False, True):
            for ucmo in (False, True):
                s = DDIM(**{**BASE_SCHED, "prediction_type": pt, "clip_sample": clip})
                s.set_timesteps_inference(50)
                prev, x0 = [], []
                for t in s.timesteps:
                    o = s.step(eps, t, x, use_clipped_model_output=ucmo)
                    prev.append(o.prev_sample.numpy())
                    x0.append(o.pred_original_sample.numpy())
                key = f"{pt}_clip{int(clip)}_ucmo{int(ucmo)}"
                out[f"step_prev_{key}"] = np.stack(prev)
                out[f"step_x0_{key}"] = np.stack(x0)
    # per-step coefficients exactly as step() forms them (ddim_scheduler.py:231-267)
    s = DDIM(**BASE_SCHED)
    s.set_timesteps_inference(50)
    coef = []
    for t in s.timesteps:
        t = int(t)
        pt_ = t - 1000 // 50
        a = s.alphas_cumprod[t]
        ap = s.alphas_cumprod[pt_] if pt_ >= 0 else s.final_alpha_cumprod
        coef.append([float(a ** 0.5), float((1 - a) ** 0.5), float(ap ** 0.5), float((1 - ap) ** 0.5)])
    out["coef_50"] = np.asarray(coef, dtype=np.float32)
    # add_noise / remove_noise with per-sample timesteps
    x0 = torch.arange(48, dtype=torch.float32).reshape(3, 4, 2, 2) / 10
    noise = torch.randn((3, 4, 2, 2), generator=g)
    tt = torch.tensor([0, 500, 999])
    out["an_x0"] = x0.numpy(); out["an_noise"] = noise.numpy(); out["an_t"] = tt.numpy()
    noisy = s.add_noise(x0, noise.clone(), tt)
    out["an_out"] = noisy.numpy()
    out["an_out_scale"] = s.add_noise(x0, noise.clone(), tt, scale=0.5).numpy()
    out["rn_out"] = s.remove_noise(noisy, noise, tt).numpy()
    np.savez_compressed(os.path.join(HERE, "scheduler.npz"), **out)
    print("scheduler.npz:", len(out), "arrays")


def vae_golden(GeneralVAESeg):
    sys.path.insert(0, os.path.join(ROOT, "latent-diffusion-segmentation_amd"))
    from ldmseg_amd import weights
    kw = dict(in_channels=7, int_channels=256, out_channels=128, block_out_channels=[32, 64, 128, 256],
              latent_channels=4, num_latents=2, num_upscalers=2, upscale_channels=256, norm_num_groups=32,
              scaling_factor=0.2, parametrization="gaussian", act_fn="none", clamp_output=False,
              freeze_codebook=False, num_mid_blocks=0, fuse_rgb=False, resize_input=False, skip_encoder=False)
    ref = GeneralVAESeg(**kw).eval()
    sd = weights.generate(weights.vae_schema(), seed=7, norm_keys=weights.VAE_NORM_KEYS)
    ref.load_state_dict(sd, strict=True)
    out = {"n_params": np.int64(sum(p.numel() for p in ref.parameters()))}
    g = torch.Generator().manual_seed(11)
    ids = torch.randint(0, 128, (2, 32, 32), generator=g)
    bits = torch.stack([(ids >> i) % 2 for i in range(7)], dim=1).float()
    bits[:, :, :4, :4] = 0.5                       # a void patch (coco.py:377-382 fill value)
    x = 2.0 * bits - 1.0
    with torch.no_grad():
        post = ref.encode(x).latent_dist
        out["enc_x"] = x.numpy()
        out["enc_moments"] = post.parameters.numpy()
        out["enc_mode"] = post.mode().numpy()
        gz = torch.Generator().manual_seed(5)
        noise = torch.randn((2, 4, 4, 4), generator=gz)
        out["enc_noise"] = noise.numpy()
        out["enc_sample"] = (post.mean + post.std * noise).numpy()
        z = torch.randn((1, 4, 4, 4), generator=g) * 1.5
        out["dec_z"] = z.numpy()
        out["dec_logits_4L"] = ref.decode(z, interpolate=False).numpy()
        out["dec_logits_8L"] = ref.decode(z, interpolate=True).numpy()
        h = ref.decoder[0](z)
        out["dec_after_conv_in"] = h.numpy()
        h = ref.decoder[2](h)
        out["dec_after_convt2"] = h.numpy()
        h = ref.decoder[4](ref.decoder[3](h))
        out["dec_after_ln_silu"] = h.numpy()
        fw = ref(x, sample_posterior=False)
        out["fwd_sample"] = fw.sample.numpy()
    out["attrs"] = np.asarray([ref.downsample_factor, ref.interpolation_factor, ref.num_latents], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "vae.npz"), **out)
    print("vae.npz:", {k: v.shape for k, v in out.items() if hasattr(v, "shape")})


def loop_golden(DDIM):
    """TrainerDiffusion.sample cannot be imported (wandb/CUDA); pin the loop by running the
    REFERENCE scheduler object inside a transcription of its control flow
    (trainers_ldm_cond.py:1121-1159) with a fixed stand-in epsilon network."""
    g = torch.Generator().manual_seed(3)
    wmix = torch.randn((4, 12, 3, 3), generator=g) * 0.2

    def eps_net(inp, t):
        return torch.tanh(torch.nn.functional.conv2d(inp, wmix, padding=1)) * (1.0 + float(t) / 1000.0)

    out = {"wmix": wmix.numpy()}
    for n in (10, 50):
        s = DDIM(**BASE_SCHED)
        s.set_timesteps_inference(n)
        rgb = 0.18215 * torch.randn((2, 4, 8, 8), generator=torch.Generator().manual_seed(1234))
        latents = torch.randn((2, 4, 8, 8), generator=torch.Generator().manual_seed(42)) * s.init_noise_sigma
        cond = torch.zeros_like(rgb)
        for i, t in enumerate(s.timesteps):
            eps = eps_net(torch.cat([latents, rgb, cond], dim=1), t)
            cond = s.step(eps, t, latents).pred_original_sample
            if i == len(s.timesteps) - 1:
                latents = s.step(eps, t, latents).pred_original_sample
            else:
                latents = s.step(eps, t, latents).prev_sample
        out[f"rgb_{n}"] = rgb.numpy()
        out[f"final_{n}"] = latents.numpy()
    np.savez_compressed(os.path.join(HERE, "sample_loop.npz"), **out)
    print("sample_loop.npz ok")


def bitcodec_golden():
    """COCO.encode_bitmap / decode_bitmap (ldmseg/data/coco.py:377-390), called unbound with a dummy
    self (ignore_label = 0) as SURVEY App. C describes; torchvision is stubbed AFTER transformers."""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
    mod("torchvision"); mod("torchvision.transforms", Compose=object, InterpolationMode=object)
    mod("torchvision.transforms.functional")
    for name in ("termcolor", "wandb"):
        if name not in sys.modules:
            mod(name, colored=lambda s, *a, **k: s)
    from ldmseg.data.coco import COCO
    dummy = types.SimpleNamespace(ignore_label=0)
    g = torch.Generator().manual_seed(17)
    ids = torch.randint(0, 128, (24, 40), generator=g)
    ids[:3, :5] = 0                                   # void region
    ids[5, 7] = 127
    bits, ignore = COCO.encode_bitmap(dummy, ids.clone())
    dec = COCO.decode_bitmap(dummy, 2.0 * bits - 1.0)
    out = {"ids": ids.numpy(), "bits": bits.numpy(), "ignore": ignore.numpy(), "decoded": dec.numpy()}
    logits = torch.randn((7, 24, 40), generator=g)
    out["dec_in"] = logits.numpy()
    out["dec_out"] = COCO.decode_bitmap(dummy, logits).numpy()
    np.savez_compressed(os.path.join(HERE, "bitcodec.npz"), **out)
    print("bitcodec.npz ok", bits.shape, dec.dtype)


def colormap_golden():
    """color_map() (ldmseg/utils/utils.py:240-258) and TrainerDiffusion.encode_seg (trainers_ldm_cond.py:324-332; the
    method body only needs numpy + color_map, so it is exercised through the same lookup the method performs)."""
    from ldmseg.utils.utils import color_map
    cmap = color_map()
    g = np.random.RandomState(3)
    ids = g.randint(0, 256, size=(2, 9, 11)).astype(np.int64)
    seg_t = ids.astype(np.uint8)
    painted = np.empty(seg_t.shape + (3,), dtype=cmap.dtype)
    for c in np.unique(seg_t):
        painted[seg_t == c] = cmap[c]
    np.savez_compressed(os.path.join(HERE, "colormap.npz"), cmap=cmap, cmap_norm=color_map(normalized=True).astype(np.float32),
                        ids=ids, painted=painted)
    print("colormap.npz ok", cmap.shape, cmap.dtype)


def real_coco_golden():
    """Two of the reference's own example pairs (data/examples/coco/{rgb_images,panoptic_images}: one portrait 480x640, one
    landscape 640x427) through the reference's data path: panoptic PNG -> segment ids (coco.py:500-501, the two numpy lines of
    _load_semseg - the rest of that method needs the COCO annotation json, which the examples do not ship), then the
    reference's OWN functions, called unbound on a dummy self as in bitcodec_golden(): COCO._remap_labels_fn (coco.py:320-351,
    np.random seeded), COCO.encode_bitmap / decode_bitmap (:377-390), and pil_transforms.CropResize.crop_and_resize
    (:105-138; bicubic for the image, nearest for the id map, to 512 x 512 as base.yaml's eval transforms do).
    The fixture holds the four data files' bytes (inputs) and the reference's outputs."""
    import hashlib
    import io
    from PIL import Image
    from ldmseg.data.coco import COCO
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
    if "torchvision.transforms.functional" not in sys.modules:
        mod("torchvision.transforms.functional")
    from ldmseg.data.util.pil_transforms import CropResize
    out = {}
    names = ["000000012280", "000000084752"]
    ex = os.path.join(REF, "data", "examples", "coco")
    for k, name in enumerate(names):
        jpg = open(os.path.join(ex, "rgb_images", name + ".jpg"), "rb").read()
        png = open(os.path.join(ex, "panoptic_images", name + ".png"), "rb").read()
        out[f"jpg_{k}"] = np.frombuffer(jpg, dtype=np.uint8)
        out[f"png_{k}"] = np.frombuffer(png, dtype=np.uint8)
        img = Image.open(io.BytesIO(jpg)).convert("RGB")
        sem = np.array(Image.open(io.BytesIO(png)).convert("RGB"))
        # coco.py:500-501: R + 256 G + 256^2 B.  (Under the reference's numpy 1.x the scalar factors promote the uint8 planes to
        # uint16 / uint32; numpy 2 refuses the literal line with an OverflowError, so the planes are widened explicitly.)
        ids_true = sem[:, :, 0].astype(np.int64) + 256 * sem[:, :, 1].astype(np.int64) + 65536 * sem[:, :, 2].astype(np.int64)
        out[f"ids_{k}"] = ids_true.astype(np.int32)
        dummy = types.SimpleNamespace(ignore_label=0, num_classes=128)
        np.random.seed(100 + k)
        remapped, mapping = COCO._remap_labels_fn(dummy, ids_true.copy(), max_val=128, keep_background_fixed=True)
        out[f"remapped_{k}"] = remapped.astype(np.uint8)
        out[f"mapping_{k}"] = np.array(sorted((int(a), int(b)) for a, b in mapping.items()), dtype=np.int64)
        bits, ignore = COCO.encode_bitmap(dummy, torch.from_numpy(remapped.astype(np.int64)), n=7, fill_value=0.5)
        out[f"bits_{k}"] = bits.numpy().astype(np.float16)                               # 0 / 0.5 / 1: exact in fp16
        out[f"ignore_{k}"] = ignore.numpy()
        out[f"decoded_{k}"] = COCO.decode_bitmap(dummy, 2.0 * bits - 1.0).numpy().astype(np.uint8)
        cr = CropResize(512)
        r_img = np.asarray(cr.crop_and_resize(img, 512, 512, mode="bicubic"))
        r_ids = np.asarray(cr.crop_and_resize(Image.fromarray(remapped.astype(np.uint8)), 512, 512, mode="nearest"))
        out[f"resized_sample_{k}"] = r_img[::8, ::8].copy()                              # every 8th pixel of the 512 x 512 image
        out[f"resized_sha_{k}"] = np.frombuffer(hashlib.sha256(r_img.tobytes()).digest(), dtype=np.uint8)
        out[f"resized_ids_{k}"] = r_ids
        print(name, img.size, "segments", len(mapping), "void px", int(ignore.sum()))
    import PIL
    out["pillow_version"] = np.array(PIL.__version__)     # the resize / JPEG bytes above are this Pillow's; the test is bit-exact only under it
    np.savez_compressed(os.path.join(HERE, "real_coco.npz"), **out)
    print("real_coco.npz ok", os.path.getsize(os.path.join(HERE, "real_coco.npz")), "bytes")


def main():
    if not os.path.isdir(REF):
        print("reference not present; nothing to do")
        return
    stub_modules()
    sys.path.insert(0, REF)
    from ldmseg.schedulers.ddim_scheduler import DDIMNoiseScheduler
    from ldmseg.models.vae import GeneralVAESeg
    torch.manual_seed(0)
    scheduler_golden(DDIMNoiseScheduler)
    vae_golden(GeneralVAESeg)
    loop_golden(DDIMNoiseScheduler)
    colormap_golden()
    try:
        bitcodec_golden()
        real_coco_golden()
    except Exception as e:   # the dataset module drags in many optional deps
        print('bitcodec / real-data golden skipped:', repr(e))


if __name__ == "__main__":
    main()
