"""Segmentation VAE with the reference's call surface, executed by the gfx950 library.

Stands behind /root/reference/ldmseg/models/vae.py::GeneralVAESeg for the
default (gaussian, num_mid_blocks=0) configuration of base.yaml:14-33:

    logits = vae_semseg.decode(z)                           # trainers_ldm_cond.py:422
    z = vae_semseg.encode(x).latent_dist.mode() / .sample() # trainers_ldm_cond.py:371-375
"""
import ctypes as C
from typing import Optional, Tuple, Union

import torch

from .. import _lib
from ..utils import EncoderOutput, VAEOutput
from ..weights import vae_schema


class DiagonalGaussianDistribution(object):
    """vae.py:370-424 on device moments [B, 8, l, l] (mean | logvar)."""

    def __init__(self, parameters: torch.Tensor, clamp_output: bool = False, act_fn: str = 'none'):
        if clamp_output or act_fn != 'none':
            raise NotImplementedError("only clamp_output=False / act_fn='none' (base.yaml:25-26)")
        self.parameters = parameters
        self.clamp_output = clamp_output
        self.act_fn = act_fn

    def _posterior(self, noise):
        p = self.parameters
        B, _, l, _ = p.shape
        out = torch.empty((B, 4, l, l), device=p.device, dtype=torch.float32)
        with torch.cuda.device(p.device):
            _lib.check(_lib.lib().ldmseg_vae_posterior(_lib.ptr(p), _lib.ptr(noise), 1.0, B, l, _lib.ptr(out),
                                                       _lib.stream_ptr(p.device)), "ldmseg_vae_posterior")
        return out

    @property
    def mean(self):
        return self.parameters[:, :4]

    @property
    def logvar(self):
        return torch.clamp(self.parameters[:, 4:], -30.0, 20.0)

    def mode(self):
        return self._posterior(None)

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        p = self.parameters
        shape = (p.shape[0], 4, p.shape[2], p.shape[3])
        noise = torch.randn(shape, generator=generator, device=p.device, dtype=p.dtype)
        return self._posterior(noise)


class GeneralVAESeg(object):
    def __init__(self, state_dict, in_channels: int = 7, int_channels: int = 256, out_channels: int = 128,
                 block_out_channels: Tuple[int] = (32, 64, 128, 256), latent_channels: int = 4,
                 norm_num_groups: int = 32, scaling_factor: float = 0.18215, num_mid_blocks: int = 0,
                 num_latents: int = 2, num_upscalers: int = 2, upscale_channels: int = 256,
                 parametrization: str = 'gaussian', act_fn: str = 'none', clamp_output: bool = False,
                 device: Union[str, torch.device] = "cuda:0", compute_dtype="bf16", **unused):
        if parametrization != 'gaussian' or num_mid_blocks != 0 or len(block_out_channels) != 4:
            raise NotImplementedError("only the default gaussian seg-VAE without mid blocks is built")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("GeneralVAESeg needs an MI355X device (no CPU fallback)")
        self.scaling_factor = scaling_factor
        self.downsample_factor = 2 ** (len(block_out_channels) - 1)
        self.interpolation_factor = self.downsample_factor // (2 ** num_upscalers)
        if self.interpolation_factor not in (1, 2):
            raise NotImplementedError("interpolation factor must be 1 or 2")
        self.parametrization = parametrization
        self.num_latents = num_latents
        self.act_fn = act_fn
        self.clamp_output = clamp_output
        self.out_channels = out_channels
        self.dtype = torch.float32
        schema = vae_schema(in_channels, int_channels, out_channels, block_out_channels, latent_channels,
                            num_latents, num_upscalers, upscale_channels)
        state_dict = {k.replace('module.', ''): v for k, v in state_dict.items()}   # vae.py:119
        missing = [k for k in schema if k not in state_dict]
        if missing:
            raise KeyError(f"state dict lacks seg-VAE tensors, e.g. {missing[:3]}")
        cd = {"bf16": _lib.BF16, torch.bfloat16: _lib.BF16, "fp32": _lib.F32, torch.float32: _lib.F32,
              "float32": _lib.F32, "bfloat16": _lib.BF16, "bf16x3": _lib.BF16X3}[compute_dtype]
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        cfg = _lib.VAECfg(in_channels, int_channels, out_channels, latent_channels, num_latents, num_upscalers,
                          upscale_channels, norm_num_groups, (C.c_int32 * 4)(*block_out_channels), cd, idx)
        n, names, ptrs, numels, keep = _lib.weight_arrays(state_dict, self.device)
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            torch.cuda.synchronize(self.device)
            _lib.check(_lib.lib().ldmseg_vae_create(C.byref(cfg), n, names, ptrs, numels, C.byref(handle)),
                       "ldmseg_vae_create")
        del keep
        self._h = handle
        self._in_channels = in_channels
        self._upscale = 2 ** num_upscalers

    @property
    def num_parameters(self) -> int:
        return int(_lib.lib().ldmseg_vae_num_params(self._h))

    def eval(self):
        return self

    def to(self, *_a, **_k):
        return self

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().ldmseg_vae_destroy(h)
            except Exception:
                pass
            self._h = None

    def encode_moments(self, semseg: torch.Tensor, in_mul: float = 1.0, in_add: float = 0.0) -> torch.Tensor:
        x = _lib.require_cuda_f32(semseg, "semseg")
        B, Cin, H, W = x.shape
        if Cin != self._in_channels or H != W:
            raise ValueError(f"expected [B,{self._in_channels},H,H], got {tuple(x.shape)}")
        l = H // self.downsample_factor
        mom = torch.empty((B, 8, l, l), device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().ldmseg_vae_encode(self._h, _lib.ptr(x), in_mul, in_add, B, H, _lib.ptr(mom),
                                                    _lib.stream_ptr(x.device)), "ldmseg_vae_encode")
        return mom

    def encode(self, semseg: torch.Tensor) -> EncoderOutput:
        return EncoderOutput(latent_dist=DiagonalGaussianDistribution(self.encode_moments(semseg)))

    def decode(self, z: torch.Tensor, interpolate: bool = True, z_scale: float = 1.0) -> torch.Tensor:
        z = _lib.require_cuda_f32(z, "z")
        B, _, L, _ = z.shape
        up = self._upscale * L * (self.interpolation_factor if interpolate else 1)
        interp = 1 if (interpolate and self.interpolation_factor == 2) else 0
        out = torch.empty((B, self.out_channels, up, up), device=z.device, dtype=torch.float32)
        with torch.cuda.device(z.device):
            _lib.check(_lib.lib().ldmseg_vae_decode(self._h, _lib.ptr(z), float(z_scale), B, L, interp, _lib.ptr(out),
                                                    _lib.stream_ptr(z.device)), "ldmseg_vae_decode")
        return out

    def decode_argmax(self, z: torch.Tensor, z_scale: float = 1.0, mask_th: Optional[float] = None,
                      ignore_label: int = 0, return_prob: bool = False):
        """Fused decode -> bilinear x2 -> argmax (+ max-softmax threshold): predictions [B,8L,8L] int64."""
        z = _lib.require_cuda_f32(z, "z")
        B, _, L, _ = z.shape
        up = self._upscale * L * self.interpolation_factor
        ids = torch.empty((B, up, up), device=z.device, dtype=torch.int64)
        prob = torch.empty((B, up, up), device=z.device, dtype=torch.float32) if return_prob else None
        with torch.cuda.device(z.device):
            _lib.check(_lib.lib().ldmseg_vae_decode_argmax(self._h, _lib.ptr(z), float(z_scale), B, L,
                                                           -1.0 if mask_th is None else float(mask_th), int(ignore_label),
                                                           _lib.ptr(ids), _lib.ptr(prob), _lib.stream_ptr(z.device)),
                       "ldmseg_vae_decode_argmax")
        return (ids, prob) if return_prob else ids

    def decode_panoptic(self, z: torch.Tensor, in_size, out_sizes, crop_boxes=None, z_scale: float = 1.0,
                        threshold_output: bool = True, threshold_mode: str = "max", mask_th: float = 0.5,
                        count_th: int = 512, overlap_th: float = 0.5, ignore_label: int = 0, return_stats: bool = False):
        """The evaluation tail of `compute_pq` (trainers_ldm_cond.py:1243-1313) fused behind the decoder: decode ->
        bilinear x2 -> bilinear to `in_size` (H, W) -> crop to `crop_boxes[b]` = (y0, x0, height, width) -> bilinear to
        `out_sizes[b]` = (h, w) -> argmax / thresholds / segment filtering, without the [B,128,H,W] logits.
        Returns a list of (panoptic [h,w] int32 tensor on the GPU (label + 1, 0 = void), kept label list)."""
        import numpy as np
        z = _lib.require_cuda_f32(z, "z")
        if threshold_mode not in ("max", "topk_diff"):
            raise ValueError(f"unknown threshold_mode {threshold_mode!r}")
        B, _, L, _ = z.shape
        Cn = self.out_channels
        sizes = np.ascontiguousarray(np.asarray(out_sizes, dtype=np.int32).reshape(B, 2))
        boxes = None if crop_boxes is None else np.ascontiguousarray(np.asarray(crop_boxes, dtype=np.int32).reshape(B, 4))
        npix = sizes[:, 0].astype(np.int64) * sizes[:, 1].astype(np.int64)
        offs = np.ascontiguousarray(np.concatenate([[0], np.cumsum(npix)[:-1]]).astype(np.int64))
        total = int(npix.sum())
        dev = z.device
        labels = torch.empty(total, dtype=torch.int32, device=dev)
        pan = torch.empty(total, dtype=torch.int32, device=dev)
        keep = torch.empty(B, Cn, dtype=torch.uint8, device=dev)
        counts = torch.empty(B, Cn, dtype=torch.int32, device=dev)
        mcounts = torch.empty(B, Cn, dtype=torch.int32, device=dev)
        vp = lambda a: None if a is None else C.c_void_p(a.ctypes.data)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().ldmseg_vae_decode_panoptic(
                self._h, _lib.ptr(z), float(z_scale), B, L, int(in_size[0]), int(in_size[1]), vp(boxes), vp(sizes), vp(offs),
                int(bool(threshold_output)), 1 if threshold_mode == "topk_diff" else 0, float(mask_th), int(count_th),
                float(overlap_th), int(ignore_label), _lib.ptr(labels), _lib.ptr(pan), _lib.ptr(keep), _lib.ptr(counts),
                _lib.ptr(mcounts), _lib.stream_ptr(dev)), "ldmseg_vae_decode_panoptic")
        keep_h = keep.cpu()
        out = []
        for b in range(B):
            o, n = int(offs[b]), int(npix[b])
            out.append((pan[o:o + n].view(int(sizes[b, 0]), int(sizes[b, 1])), torch.nonzero(keep_h[b]).flatten().tolist()))
        if return_stats:
            lab = [labels[int(offs[b]):int(offs[b]) + int(npix[b])].view(int(sizes[b, 0]), int(sizes[b, 1])) for b in range(B)]
            return out, {"labels": lab, "counts": counts, "mask_counts": mcounts, "keep": keep}
        return out

    def forward(self, sample, sample_posterior: bool = True, return_dict: bool = True,
                generator: Optional[torch.Generator] = None, rgb_sample=None, valid_mask=None):
        if rgb_sample is not None:
            raise NotImplementedError("fuse_rgb is off in base.yaml:30")
        posterior = self.encode(sample).latent_dist
        z = posterior.sample(generator=generator) if sample_posterior else posterior.mode()
        if valid_mask is not None:
            z = z * valid_mask[:, None]
        dec = self.decode(z, interpolate=False)
        if not return_dict:
            return (dec,)
        return VAEOutput(sample=dec, posterior=posterior)

    __call__ = forward



class GeneralVAEImage(object):
    """ldmseg/models/vae.py:36-39 `GeneralVAEImage(AutoencoderKL)` with the decoder removed (tools/main_ldm.py:137-139):
    the RGB encoder used as `encode_func` in `TrainerDiffusion.encode_inputs` (trainers_ldm_cond.py:360-375).

    state_dict: `AutoencoderKL.state_dict()` / the 'vae_image' entry of ldmseg.pt (decoder keys are ignored).
    """

    def __init__(self, state_dict, scaling_factor: float = 0.18215, device: Union[str, torch.device] = "cuda:0",
                 compute_dtype="bf16", **unused):
        from ..weights import vae_image_schema
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("GeneralVAEImage needs an MI355X device (no CPU fallback)")
        self.scaling_factor = scaling_factor
        self.dtype = torch.float32
        old = "encoder.mid_block.attentions.0.query.weight" in state_dict
        schema = vae_image_schema(old_attention_names=old)
        missing = [k for k in schema if k not in state_dict]
        if missing:
            raise KeyError(f"state dict lacks AutoencoderKL encoder tensors, e.g. {missing[:3]}")
        sd = {}
        for k in schema:
            t = state_dict[k]
            # newer diffusers store the attention projections as 1x1 convs or Linear; both flatten to [512, 512]
            sd[k] = t.reshape(schema[k]) if tuple(t.shape) != tuple(schema[k]) else t
        cd = {"bf16": _lib.BF16, torch.bfloat16: _lib.BF16, "fp32": _lib.F32, torch.float32: _lib.F32,
              "float32": _lib.F32, "bfloat16": _lib.BF16, "bf16x3": _lib.BF16X3}[compute_dtype]
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        cfg = _lib.VAEImageCfg(cd, idx)
        n, names, ptrs, numels, keep = _lib.weight_arrays(sd, self.device)
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            torch.cuda.synchronize(self.device)
            _lib.check(_lib.lib().ldmseg_vae_image_create(C.byref(cfg), n, names, ptrs, numels, C.byref(handle)),
                       "ldmseg_vae_image_create")
        del keep
        self._h = handle

    def set_scaling_factor(self, scaling_factor):          # vae.py:38-39
        self.scaling_factor = scaling_factor

    @property
    def num_parameters(self) -> int:
        return int(_lib.lib().ldmseg_vae_image_num_params(self._h))

    def eval(self):
        return self

    def to(self, *_a, **_k):
        return self

    def requires_grad_(self, *_a, **_k):
        return self

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().ldmseg_vae_image_destroy(h)
            except Exception:
                pass
            self._h = None

    def encode_moments(self, images: torch.Tensor, in_mul: float = 1.0, in_add: float = 0.0) -> torch.Tensor:
        x = _lib.require_cuda_f32(images, "images")
        B, Cin, H, W = x.shape
        if Cin != 3 or H % 8 or W % 8:
            raise ValueError(f"expected [B,3,H,W] with H, W multiples of 8, got {tuple(x.shape)}")
        mom = torch.empty((B, 8, H // 8, W // 8), device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().ldmseg_vae_image_encode(self._h, _lib.ptr(x), in_mul, in_add, B, H, W, _lib.ptr(mom),
                                                          _lib.stream_ptr(x.device)), "ldmseg_vae_image_encode")
        return mom

    def encode(self, x: torch.Tensor) -> EncoderOutput:
        """AutoencoderKL.encode(x).latent_dist (.mode() / .sample())."""
        return EncoderOutput(latent_dist=DiagonalGaussianDistribution(self.encode_moments(x)))
