"""UNet with the reference's call surface, executed by the gfx950 library.

Stands behind /root/reference/ldmseg/models/unet.py::UNet
(UNet2DConditionModel, SD-1.x topology) as TrainerDiffusion uses it:

    noise_pred = unet(inputs, t, encoder_hidden_states=None).sample      # trainers_ldm_cond.py:1141

What the reference assembles in steps - ``UNet.from_pretrained`` (main_ldm.py:146),
``remove_cross_attention`` (unet.py:83-105), ``modify_encoder`` (unet.py:178-233),
``load_state_dict`` (trainers_ldm_cond.py:1863-1891) - collapses into one
constructor that takes the final state dict (the ``unet`` entry of ldmseg.pt).
All arithmetic runs in libldmseg_hip.so; a missing library or a CPU tensor is an
error, never a fallback.
"""
import ctypes as C
from types import SimpleNamespace
from typing import Optional, Union

import torch

from .. import _lib
from ..utils import UNetOutput
from ..weights import BLOCK_OUT, unet_schema


class UNet(object):
    def __init__(self, state_dict, in_channels: int = 12, device: Union[str, torch.device] = "cuda:0",
                 compute_dtype: Union[str, torch.dtype] = "bf16", cross_attention: bool = False):
        if cross_attention:
            raise NotImplementedError("cross-attention conditioning is removed in the reference default "
                                      "(image_descriptors: remove, base.yaml:71) and is not built here")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("UNet needs an MI355X device (no CPU fallback)")
        self.compute_dtype = {"bf16": _lib.BF16, torch.bfloat16: _lib.BF16, "bfloat16": _lib.BF16,
                              "fp32": _lib.F32, "float32": _lib.F32, torch.float32: _lib.F32,
                              "bf16x3": _lib.BF16X3}[compute_dtype]   # bf16x3: fp32 storage, GEMMs as three bf16 MFMAs (hi + lo)
        # what callers see as `unet.dtype` is the boundary dtype (the reference keeps the UNet fp32, main_ldm.py:168)
        self.dtype = torch.float32
        self.in_channels = int(in_channels)
        self.config = SimpleNamespace(block_out_channels=list(BLOCK_OUT), in_channels=self.in_channels,
                                      out_channels=4, attention_head_dim=8, cross_attention_dim=None,
                                      sample_size=64, layers_per_block=2)
        missing = [k for k in unet_schema(self.in_channels, False) if k not in state_dict]
        if missing:
            raise KeyError(f"state dict lacks {len(missing)} UNet tensors, e.g. {missing[:3]}")
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        cfg = _lib.UNetCfg(self.in_channels, 0, self.compute_dtype, idx)
        n, names, ptrs, numels, keep = _lib.weight_arrays(state_dict, self.device)
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            torch.cuda.synchronize(self.device)
            _lib.check(_lib.lib().ldmseg_unet_create(C.byref(cfg), n, names, ptrs, numels, C.byref(handle)),
                       "ldmseg_unet_create")
        del keep                      # the library repacked its own copy
        self._h = handle
        self.module = self            # DDP-style `.module` access (trainers_ldm_cond.py uses both)

    # --- reference-surface helpers -------------------------------------------------
    @property
    def num_parameters(self) -> int:
        return int(_lib.lib().ldmseg_unet_num_params(self._h))

    def workspace_bytes(self, batch: int, latent_size: int) -> int:
        return int(_lib.lib().ldmseg_unet_workspace_bytes(self._h, batch, latent_size))

    def reserve(self, batch: int, latent_size: int):
        """Allocate the forward / sampling workspaces for (batch, latent_size) now, so that later calls at that size
        never allocate or synchronise (ldmseg_unet_reserve)."""
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ldmseg_unet_reserve(self._h, int(batch), int(latent_size)), "ldmseg_unet_reserve")
        return self

    def set_attention_fp8(self, min_tokens: int = 4096):
        """bf16 mode: run every self-attention level with at least `min_tokens` tokens on the fp8 (e4m3) operand path -
        BASELINE configs[4], the 1024x1024 / 128x128-latent configuration.  0 switches it off."""
        _lib.check(_lib.lib().ldmseg_unet_set_attention_fp8(self._h, int(min_tokens)), "ldmseg_unet_set_attention_fp8")
        return self

    def gn_fallbacks(self) -> int:
        """Workgroups of this handle's cooperative GroupNorm launches that computed a missing partner's statistics themselves
        (full poll bound only; ldmseg_unet_gn_fallbacks).  Synchronises."""
        n = C.c_int64(0)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ldmseg_unet_gn_fallbacks(self._h, C.byref(n)), "ldmseg_unet_gn_fallbacks")
        return int(n.value)

    def cf_fallbacks(self) -> int:
        """Workgroups of this handle's K-sliced GEMM launches that gave up waiting for their tile's other slices and left their share
        to the last arriver (ldmseg_unet_cf_fallbacks; 0 on an undisturbed device).  Synchronises."""
        n = C.c_int64(0)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ldmseg_unet_cf_fallbacks(self._h, C.byref(n)), "ldmseg_unet_cf_fallbacks")
        return int(n.value)

    def gn_backoff(self) -> int:
        """Sampling-loop calls this handle will still run with the short partner poll (ldmseg_unet_gn_backoff)."""
        n = C.c_int32(0)
        _lib.check(_lib.lib().ldmseg_unet_gn_backoff(self._h, C.byref(n)), "ldmseg_unet_gn_backoff")
        return int(n.value)

    def eval(self):
        return self

    def to(self, *_a, **_k):
        return self

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().ldmseg_unet_destroy(h)
            except Exception:
                pass
            self._h = None

    # --- forward -------------------------------------------------------------------
    def _timestep_args(self, timestep, B):
        if isinstance(timestep, torch.Tensor):
            if timestep.is_cuda:
                t = timestep.to(torch.int64).reshape(-1).contiguous()
                if t.numel() not in (1, B):
                    raise ValueError("timestep must be a scalar or hold one entry per sample")
                return t, _lib.ptr(t), t.numel(), 0
            vals = timestep.reshape(-1).tolist()
            if len(vals) == 1:
                return None, C.c_void_p(0), 1, int(vals[0])
            t = timestep.to(device=self.device, dtype=torch.int64).contiguous()
            return t, _lib.ptr(t), t.numel(), 0
        return None, C.c_void_p(0), 1, int(timestep)

    def forward(self, sample: torch.Tensor, timestep, encoder_hidden_states: Optional[torch.Tensor] = None,
                timestep_img=None, return_dict: bool = True, **unused):
        if encoder_hidden_states is not None:
            raise NotImplementedError("encoder_hidden_states must be None (cross-attention removed)")
        x = _lib.require_cuda_f32(sample, "sample")
        B, Cin, H, W = x.shape
        if Cin != self.in_channels or H != W:
            raise ValueError(f"expected [B,{self.in_channels},L,L], got {tuple(x.shape)}")
        out = torch.empty((B, 4, H, W), device=x.device, dtype=torch.float32)
        keep, tptr, tcount, thost = self._timestep_args(timestep, B)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().ldmseg_unet_forward(self._h, _lib.ptr(x), tptr, tcount, thost, B, H, _lib.ptr(out),
                                                      _lib.stream_ptr(x.device)), "ldmseg_unet_forward")
        del keep
        if not return_dict:
            return (out,)
        return UNetOutput(sample=out)

    __call__ = forward

    def forward_parts(self, latents, rgb_latents, condition, timestep):
        """Forward on the un-concatenated inputs of the sampler (skips torch.cat, trainers_ldm_cond.py:1128-1138)."""
        lat = _lib.require_cuda_f32(latents, "latents")
        rgb = _lib.require_cuda_f32(rgb_latents, "rgb_latents")
        cond = _lib.require_cuda_f32(condition, "condition") if condition is not None else None
        B, _, H, _ = lat.shape
        out = torch.empty((B, 4, H, H), device=lat.device, dtype=torch.float32)
        keep, tptr, tcount, thost = self._timestep_args(timestep, B)
        with torch.cuda.device(lat.device):
            _lib.check(_lib.lib().ldmseg_unet_forward_parts(self._h, _lib.ptr(lat), _lib.ptr(rgb), _lib.ptr(cond), tptr,
                                                            tcount, thost, B, H, _lib.ptr(out),
                                                            _lib.stream_ptr(lat.device)), "ldmseg_unet_forward_parts")
        del keep
        return UNetOutput(sample=out)
