"""Sampling entry points with the reference's call surface.

Stands behind the hot-path methods of
/root/reference/ldmseg/trainers/trainers_ldm_cond.py::TrainerDiffusion:

    sample(prompts, num_inference_steps, guidance_scale, seed, rgb_latents, ...)   # :1045-1170
    decode_latents(latents, return_logits=True)                                   # :397-442
    encode_inputs(images, encode_func=vae_semseg.encode, scaling_factor=...)      # :335-394 (seg side)

plus two things the reference does elsewhere or not at all:
  * ``sample_sharded`` - the data-parallel eval decomposition the reference gets from
    DistributedSampler (:245): independent images are split over ranks, each rank runs
    the whole loop locally and ONE RCCL all-gather of the final latents replaces the
    detectron2 ``comm.gather`` of PNGs (evaluations/panoptic_evaluation_agnostic.py:129-131);
  * ``sample_inpaint`` - the build-defined mask-inpainting sampler.

Training, logging, datasets and PQ evaluation are out of scope.
"""
import ctypes as C
from typing import List, Optional

import numpy as np
import torch

from .. import _lib


class TrainerDiffusion(object):
    def __init__(self, vae_semseg, unet, noise_scheduler, self_condition: Optional[bool] = None,
                 device=None, latent_size: int = 64, vae_image=None):
        self.vae_image = vae_image            # RGB encoder (trainers_ldm_cond.py:58,111); optional here
        self.vae_semseg = vae_semseg
        self.unet_model = unet
        self.noise_scheduler = noise_scheduler
        self.self_condition = (unet.in_channels == 12) if self_condition is None else bool(self_condition)
        self.device = torch.device(device) if device is not None else unet.device
        self.args = {'gpu': self.device}
        self.latent_size = latent_size
        self.unet_dtype = torch.float32
        self.image_descriptor_model = None
        self.textencoder = None
        self.tokenizer = None

    # ------------------------------------------------------------------ noise
    @staticmethod
    def draw_noise(batch_size: int, latent_size: int, seed: Optional[int]) -> torch.Tensor:
        """CPU generator draw, identical for every batch of a given size (:1088-1091)."""
        g = torch.Generator().manual_seed(seed) if seed is not None else None
        return torch.randn((batch_size, 4, latent_size, latent_size), generator=g)

    # ------------------------------------------------------------------ sample
    @torch.no_grad()
    def sample(self, prompts: List[str], num_inference_steps: int = 50, guidance_scale: float = 7.5,
               seed: Optional[int] = None, rgb_latents: Optional[torch.Tensor] = None,
               return_all_latents: bool = False, disable_progress_bar: bool = False,
               rgb_images: Optional[torch.Tensor] = None, scheduler=None, repeat_noise: Optional[bool] = None,
               python_loop: bool = False, latents: Optional[torch.Tensor] = None) -> torch.Tensor:
        """DDIM sampling loop.  ``python_loop=True`` walks ``scheduler.timesteps`` in Python calling
        ``unet(...)`` and ``scheduler.step(...)`` exactly like the reference; the default hands the
        whole loop to ldmseg_sample_loop (same kernels, no per-step Python)."""
        if rgb_latents is None:
            raise ValueError("rgb_latents is required (the reference dereferences it at :1126)")
        if scheduler is None:
            scheduler = self.noise_scheduler
            scheduler.set_timesteps_inference(num_inference_steps)
        repeat_noise = bool(repeat_noise)
        batch_size = len(prompts)
        rgb = _lib.require_cuda_f32(rgb_latents, "rgb_latents")
        L = rgb.shape[-1]
        if latents is None:
            latents = self.draw_noise(batch_size, L, seed)
        latents = latents.to(device=rgb.device, dtype=torch.float32)
        if repeat_noise:
            latents = latents[0:1].repeat(batch_size, 1, 1, 1)
            original_noise = latents.clone()
        latents = (latents * scheduler.init_noise_sigma).contiguous()

        if python_loop:
            out = self._sample_python(scheduler, latents, rgb, return_all_latents)
        else:
            out = self._sample_native(scheduler, latents, rgb, return_all_latents)
        if return_all_latents:
            return out
        if repeat_noise:
            return out, original_noise
        return out

    def _sample_python(self, scheduler, latents, rgb, return_all):
        all_latents = []
        condition = torch.zeros_like(rgb)
        n = len(scheduler.timesteps)
        for idx, t in enumerate(scheduler.timesteps):
            parts = [latents, rgb] + ([condition] if self.self_condition else [])
            inputs = torch.cat(parts, dim=1).to(self.unet_dtype)
            noise_pred = self.unet_model(inputs, t, encoder_hidden_states=None).sample
            if self.self_condition:
                condition = scheduler.step(noise_pred, t, latents).pred_original_sample
            if idx == n - 1:
                latents = scheduler.step(noise_pred, t, latents).pred_original_sample
            else:
                latents = scheduler.step(noise_pred, t, latents).prev_sample
            if return_all:
                all_latents.append(latents)
        return torch.cat(all_latents, dim=0) if return_all else latents

    def _loop_cfg(self, scheduler):
        ts = scheduler.timesteps_host()
        n = len(ts)
        coef = np.ascontiguousarray(scheduler.coefficient_table(), dtype=np.float32)
        c_ts = (C.c_int64 * n)(*ts)
        cfg = _lib.SampleCfg()
        cfg.n_steps = n
        cfg.timesteps = c_ts
        cfg.coef = coef.ctypes.data_as(C.POINTER(C.c_float))
        cfg.prediction_type = _lib.PRED[scheduler.prediction_type]
        cfg.clip_sample = int(bool(scheduler.clip_sample))
        cfg.clip_sample_range = float(scheduler.clip_sample_range)
        cfg.self_condition = int(self.self_condition)
        return cfg, (c_ts, coef)

    def _sample_native(self, scheduler, latents, rgb, return_all, inpaint=None):
        if scheduler.thresholding:
            raise NotImplementedError
        cfg, keep = self._loop_cfg(scheduler)
        B, _, L, _ = rgb.shape
        lat = latents.clone()
        allv = torch.empty((cfg.n_steps * B, 4, L, L), device=rgb.device) if return_all else None
        if inpaint is not None:
            known, z0, noise, paste = inpaint
            cfg.known_dev = known.data_ptr()
            cfg.z0_dev = z0.data_ptr()
            cfg.noise_dev = noise.data_ptr()
            cfg.paste_coef = paste.ctypes.data_as(C.POINTER(C.c_float))
        with torch.cuda.device(rgb.device):
            _lib.check(_lib.lib().ldmseg_sample_loop(self.unet_model._h, C.byref(cfg), _lib.ptr(lat), _lib.ptr(rgb), B, L,
                                                     _lib.ptr(allv), _lib.stream_ptr(rgb.device)), "ldmseg_sample_loop")
        del keep
        return allv if return_all else lat

    # ------------------------------------------------------------------ inpainting (build-defined)
    @torch.no_grad()
    def sample_inpaint(self, prompts, known_mask: torch.Tensor, known_latents: torch.Tensor,
                       num_inference_steps: int = 50, seed: Optional[int] = None,
                       rgb_latents: Optional[torch.Tensor] = None, scheduler=None) -> torch.Tensor:
        """Mask-inpainting DDIM sampling (absent from the reference).  ``known_mask``
        bool [B,1,L,L], True = latent given (convention of trainers_ldm_cond.py:613-615);
        ``known_latents`` = scaling_factor * vae_semseg.encode(partial).mode().  After every step
        the known region is replaced by the known latents re-noised to the next timestep with the
        same fixed noise; the final step pastes them clean."""
        if scheduler is None:
            scheduler = self.noise_scheduler
            scheduler.set_timesteps_inference(num_inference_steps)
        rgb = _lib.require_cuda_f32(rgb_latents, "rgb_latents")
        B, _, L, _ = rgb.shape
        noise = self.draw_noise(len(prompts), L, seed).to(rgb.device).contiguous()
        z0 = _lib.require_cuda_f32(known_latents, "known_latents")
        known = known_mask.to(device=rgb.device).reshape(B, 1, L, L).to(torch.uint8).contiguous()
        ts = scheduler.timesteps_host()
        paste = np.zeros((len(ts), 2), dtype=np.float32)
        for i in range(len(ts)):
            if i == len(ts) - 1:
                paste[i] = (1.0, 0.0)
            else:
                a = scheduler.alphas_cumprod[ts[i + 1]]
                paste[i] = (float(a ** 0.5), float((1 - a) ** 0.5))
        latents = (noise * scheduler.init_noise_sigma).contiguous()
        return self._sample_native(scheduler, latents, rgb, False, inpaint=(known, z0, noise, paste))

    # ------------------------------------------------------------------ multi-GPU
    @torch.no_grad()
    def sample_sharded(self, prompts, num_inference_steps: int = 50, seed: Optional[int] = None,
                       rgb_latents: Optional[torch.Tensor] = None, scheduler=None, noise_mode: str = "reference",
                       group=None) -> torch.Tensor:
        """Data-parallel sampling of a GLOBAL batch: rank r owns images [r*B/W, (r+1)*B/W).
        ``rgb_latents`` is this rank's shard [B/W,4,L,L] (or the global batch on any device, from
        which the shard is sliced).  noise_mode 'reference': every rank seeds identically and draws
        randn(B/W) (what the reference's per-rank sample() does); 'global': one randn(B) draw,
        sliced.  Returns the all-gathered latents [B,4,L,L] on every rank (RCCL over xGMI)."""
        import torch.distributed as dist
        W = dist.get_world_size(group) if dist.is_initialized() else 1
        r = dist.get_rank(group) if dist.is_initialized() else 0
        B = len(prompts)
        if B % W:
            raise ValueError("global batch must divide by world size")
        per = B // W
        if rgb_latents.shape[0] == B and W > 1:
            rgb_latents = rgb_latents[r * per:(r + 1) * per]
        L = rgb_latents.shape[-1]
        if noise_mode == "reference":
            noise = self.draw_noise(per, L, seed)
        elif noise_mode == "global":
            noise = self.draw_noise(B, L, seed)[r * per:(r + 1) * per]
        else:
            raise ValueError(noise_mode)
        local = self.sample(prompts[r * per:(r + 1) * per], num_inference_steps, seed=seed,
                            rgb_latents=rgb_latents.to(self.device), scheduler=scheduler, latents=noise)
        if W == 1:
            return local
        out = torch.empty((B,) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out

    # ------------------------------------------------------------------ VAE side
    @torch.no_grad()
    def decode_latents(self, latents: torch.Tensor, return_logits: bool = False, threshold_output: bool = False,
                       rgb_latents=None, weight_dtype: torch.dtype = torch.float32, mask_th: float = 0.5,
                       ignore_label: int = 0, return_ids: bool = False):
        """:397-442.  return_logits=True -> fp32 logits [B,128,8L,8L] on the GPU; otherwise, like the reference, the
        colour-encoded uint8 numpy image [B,8L,8L,3] of the argmax ids (`encode_seg`, :436).  `return_ids=True`
        (extension) returns the int64 id map on the GPU instead of painting it.  The 1/scaling_factor multiply is
        fused into the decoder's input packing, and argmax (+ max-softmax threshold) with the decoder tail: the
        logits are never materialised on that path."""
        zs = 1.0 / self.vae_semseg.scaling_factor
        if return_logits:
            return self.vae_semseg.decode(latents, z_scale=zs).float()
        ids = self.vae_semseg.decode_argmax(latents, z_scale=zs, mask_th=mask_th if threshold_output else None,
                                            ignore_label=ignore_label)
        if return_ids:
            return ids
        from ..utils import encode_seg
        return encode_seg(ids.cpu().numpy()).astype(np.uint8)

    @torch.no_grad()
    def postprocess_panoptic(self, masks_logits: torch.Tensor, threshold_output: bool = False,
                             threshold_mode: str = "max", mask_th: float = 0.5, count_th: int = 512,
                             overlap_th: float = 0.5, ignore_label: int = 0, return_stats: bool = False):
        """The per-image post-processing loop of the evaluation (:1277-1313) in one pass on the GPU.

        masks_logits [B,C,H,W] fp32 on the GPU, already at the output size.  Returns `processed_results`
        like the reference: a list of {"panoptic_seg": (panoptic [H,W] int32 tensor (label+1, 0 = void),
        segments_info)} with segments_info = [{"id": label+1, "category_id": 1, "isthing": True}, ...] in
        ascending label order (np.unique order).  Only the [B,C] keep table is copied to the host.
        """
        from .. import _lib
        import ctypes as C
        x = _lib.require_cuda_f32(masks_logits, "masks_logits")
        if x.dim() != 4:
            raise ValueError("masks_logits must be [B,C,H,W]")
        if threshold_mode not in ("max", "topk_diff"):
            raise ValueError(f"unknown threshold_mode {threshold_mode!r}")
        B, Cn, H, W = x.shape
        dev = x.device
        labels = torch.empty(B, H, W, dtype=torch.int32, device=dev)
        pan = torch.empty(B, H, W, dtype=torch.int32, device=dev)
        keep = torch.empty(B, Cn, dtype=torch.uint8, device=dev)
        counts = torch.empty(B, Cn, dtype=torch.int32, device=dev)
        mcounts = torch.empty(B, Cn, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().ldmseg_panoptic_postprocess(
                _lib.ptr(x), B, Cn, H, W, int(bool(threshold_output)), 1 if threshold_mode == "topk_diff" else 0,
                float(mask_th), int(count_th), float(overlap_th), int(ignore_label), _lib.ptr(labels), _lib.ptr(pan),
                _lib.ptr(keep), _lib.ptr(counts), _lib.ptr(mcounts), _lib.stream_ptr(dev)), "panoptic_postprocess")
        keep_h = keep.cpu()
        results = []
        for b in range(B):
            info = [{"id": int(c) + 1, "category_id": 1, "isthing": True} for c in torch.nonzero(keep_h[b]).flatten().tolist()]
            results.append({"panoptic_seg": (pan[b], info)})
        if return_stats:
            return results, {"labels": labels, "counts": counts, "mask_counts": mcounts, "keep": keep}
        return results

    # ------------------------------------------------------------------ evaluation loop body (compute_pq)
    @staticmethod
    def crop_padding(prediction: torch.Tensor, padding_mask: torch.Tensor) -> torch.Tensor:
        """:1172-1178 - bounding box of the non-padded region."""
        co = padding_mask.nonzero()
        y0, y1 = int(co[:, 0].min()), int(co[:, 0].max())
        x0, x1 = int(co[:, 1].min()), int(co[:, 1].max())
        return prediction[:, y0:y1 + 1, x0:x1 + 1]

    @staticmethod
    def padding_boxes(padding_masks: torch.Tensor):
        """The crop boxes of `crop_padding` for a whole batch, one D2H copy: [B,4] int32 numpy (y0, x0, height, width)."""
        m = padding_masks != 0
        rows, cols = m.any(dim=2), m.any(dim=1)                                   # [B,H], [B,W]
        if not bool(rows.any(dim=1).all()):
            raise ValueError("a padding mask without a single valid pixel")        # (the reference's .min() raises too)
        H, W = rows.shape[1], cols.shape[1]
        y0 = rows.int().argmax(dim=1)
        y1 = H - 1 - rows.flip(1).int().argmax(dim=1)
        x0 = cols.int().argmax(dim=1)
        x1 = W - 1 - cols.flip(1).int().argmax(dim=1)
        return torch.stack([y0, x0, y1 - y0 + 1, x1 - x0 + 1], dim=1).to(torch.int32).cpu().numpy()

    @torch.no_grad()
    def predict_panoptic(self, rgb_images: torch.Tensor, im_sizes, padding_masks: Optional[torch.Tensor] = None,
                         num_inference_steps: int = 50, guidance_scale: float = 7.5, seed: Optional[int] = None,
                         threshold_output: bool = True, threshold_mode: str = "max", scheduler=None, rgb_size: Optional[int] = None,
                         mask_th: float = 0.5, count_th: int = 512, overlap_th: float = 0.5, ignore_label: int = 0,
                         return_intermediates: bool = False, fused: bool = True):
        """One batch of `compute_pq` (:1218-1313): RGB images [B,3,S,S] in [0,1] on the GPU -> `processed_results`
        (per image {"panoptic_seg": (panoptic map at the original size (h, w), segments_info)}).
        Pixels -> image-VAE latents -> DDIM sampling -> seg-VAE decoder -> [bilinear x2 -> bilinear to the input size ->
        crop the padding -> bilinear to (h, w) -> argmax / thresholds / segment filtering].  The bracket runs as ONE fused
        tail behind the decoder (`ldmseg_vae_decode_panoptic`: no [B,128,H,W] logits, no per-image torch interpolation);
        `fused=False` (and `return_intermediates`) materialises the logits and walks the reference's steps one by one."""
        import torch.nn.functional as F
        if self.vae_image is None:
            raise ValueError("predict_panoptic needs the image VAE (TrainerDiffusion(..., vae_image=...))")
        rgb_images = _lib.require_cuda_f32(rgb_images, "rgb_images")
        B = rgb_images.shape[0]
        if scheduler is None:
            scheduler = self.noise_scheduler
            scheduler.set_timesteps_inference(num_inference_steps)
        rgb_latents, _ = self.encode_inputs(rgb_images, encode_func=self.vae_image.encode,
                                            scaling_factor=self.vae_image.scaling_factor, resize=rgb_size)
        latents = self.sample([""] * B, num_inference_steps, guidance_scale, seed, rgb_latents=rgb_latents,
                              scheduler=scheduler, disable_progress_bar=True)
        sizes = [(int(s[0]), int(s[1])) for s in im_sizes]
        if fused and not return_intermediates:
            boxes = self.padding_boxes(padding_masks) if padding_masks is not None else None
            outs = self.vae_semseg.decode_panoptic(
                latents, (rgb_images.shape[-2], rgb_images.shape[-1]), sizes, boxes, z_scale=1.0 / self.vae_semseg.scaling_factor,
                threshold_output=threshold_output, threshold_mode=threshold_mode, mask_th=mask_th, count_th=count_th,
                overlap_th=overlap_th, ignore_label=ignore_label)
            return [{"panoptic_seg": (pan, [{"id": int(c) + 1, "category_id": 1, "isthing": True} for c in kept])}
                    for pan, kept in outs]
        logits = self.decode_latents(latents, return_logits=True)
        logits = F.interpolate(logits, size=(rgb_images.shape[-2], rgb_images.shape[-1]), mode="bilinear",
                               align_corners=False)                                           # :1252-1257
        results = []
        for i in range(B):
            m = logits[i]
            if padding_masks is not None:
                m = self.crop_padding(m, padding_masks[i])                                    # :1263
            h, w = sizes[i]
            m = F.interpolate(m[None].float(), size=(h, w), mode="bilinear", align_corners=False)   # :1266-1271
            results += self.postprocess_panoptic(m.contiguous(), threshold_output=threshold_output,
                                                 threshold_mode=threshold_mode, mask_th=mask_th, count_th=count_th,
                                                 overlap_th=overlap_th, ignore_label=ignore_label)
        if return_intermediates:
            return results, {"rgb_latents": rgb_latents, "latents": latents, "logits": logits}
        return results

    @torch.no_grad()
    def compute_pq(self, dataloader, evaluator, num_inference_steps: int = 50, guidance_scale: float = 7.5,
                   seed: Optional[int] = None, threshold_output: bool = True, max_iter: Optional[int] = None,
                   threshold_mode: str = "max", **post_kw):
        """`compute_pq` (:1181-1346) over an iterable of batches shaped like the reference's `collate_fn` output:
        {'image': [B,3,S,S] in [0,1], 'mask': [B,S,S] padding masks or None, 'meta': [{'image_file', 'image_id',
        'im_size': (h, w)}, ...]}.  `evaluator` is a PanopticEvaluatorAgnostic; all ranks must call this (the
        evaluator gathers).  Returns evaluator.evaluate() (rank 0) / None."""
        evaluator.reset()
        scheduler = self.noise_scheduler
        scheduler.set_timesteps_inference(num_inference_steps=num_inference_steps)
        scheduler.move_timesteps_to(self.device)                                              # :1211-1212
        for batch_idx, data in enumerate(dataloader):
            meta = data["meta"]
            file_names = [x["image_file"] for x in meta]
            image_ids = [x["image_id"] for x in meta]
            sizes = [x["im_size"] for x in meta]
            rgb = data["image"].to(self.device, non_blocking=True)
            masks = data.get("mask")
            masks = masks.to(self.device) if masks is not None else None
            processed = self.predict_panoptic(rgb, sizes, masks, num_inference_steps, guidance_scale, seed,
                                              threshold_output, threshold_mode, scheduler=scheduler, **post_kw)
            evaluator.process(file_names, image_ids, processed)
            if max_iter is not None and batch_idx > max_iter:                                 # (sic, :1332)
                break
        return evaluator.evaluate()

    @torch.no_grad()
    def encode_inputs(self, images: torch.Tensor, sample_posterior: bool = False, encode_func=None,
                      scaling_factor: Optional[float] = None, resize: Optional[int] = None, weight_dtype=None,
                      generator=None):
        """:335-394.  images in [0,1] (RGB, or bit maps for the segmentation VAE) -> 2x-1 -> encoder ->
        (latents, latents_mean) * scaling_factor.  As in the reference the default encoder is the image VAE
        (`self.vae_image.encode`); pass `encode_func=self.vae_semseg.encode` with its scaling factor for
        segmentation maps.  The 2x-1 is fused into the encoder's input packing."""
        import torch.nn.functional as F
        from ..models.vae import DiagonalGaussianDistribution
        if encode_func is None:
            if self.vae_image is None:
                raise ValueError("no encode_func given and the trainer was built without vae_image")
            encode_func = self.vae_image.encode
        owner = getattr(encode_func, "__self__", None)
        if scaling_factor is None:
            scaling_factor = (self.vae_image if self.vae_image is not None else owner).scaling_factor
        if resize is not None:
            images = F.interpolate(images, size=(resize, resize), mode='bilinear', align_corners=False)
        if owner is not None and hasattr(owner, "encode_moments"):
            dist_ = DiagonalGaussianDistribution(owner.encode_moments(images, in_mul=2.0, in_add=-1.0))   # :369 fused
        else:
            dist_ = encode_func(2. * images - 1.).latent_dist
        mean = dist_.mode()
        latents = dist_.sample(generator) if sample_posterior else mean.clone()
        if resize is not None:
            size = (self.latent_size, self.latent_size)
            latents = F.interpolate(latents, size=size, mode='bilinear', align_corners=False)
            mean = F.interpolate(mean, size=size, mode='bilinear', align_corners=False)
        return latents * scaling_factor, mean * scaling_factor
