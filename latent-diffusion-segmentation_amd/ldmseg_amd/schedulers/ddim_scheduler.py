"""DDIM noise scheduler with the reference's call surface.

Mirrors /root/reference/ldmseg/schedulers/ddim_scheduler.py::DDIMNoiseScheduler
(same constructor kwargs, attributes, method names and error behaviour).  Host
side: fp32 tables, the integer inference grid (bit-exact) and the per-step 0-d
coefficients are computed here exactly as the reference computes them (torch CPU
fp32 scalar math).  Device side: `step`, `add_noise` and `remove_noise` run as
HIP kernels (libldmseg_hip.so: ldmseg_ddim_step / ldmseg_add_noise) whose op
order reproduces the reference's chain of torch ops bit for bit - and because
the coefficients live on the host, `step` needs no D2H sync even when the
timestep arrives as a GPU tensor out of `scheduler.timesteps` (the reference
indexes a CPU table with it, ddim_scheduler.py:234).
"""
import math
from typing import Optional, Union

import numpy as np
import torch

from ..utils import DDIMNoiseSchedulerOutput  # noqa: F401  (re-exported)
from .. import _lib


class DDIMNoiseScheduler(object):
    def __init__(
        self,
        num_train_timesteps: int = 1000,
        beta_start: float = 0.0001,
        beta_end: float = 0.02,
        beta_schedule: str = "linear",
        clip_sample: bool = True,
        set_alpha_to_one: bool = True,
        steps_offset: int = 0,
        prediction_type: str = "epsilon",
        thresholding: bool = False,
        dynamic_thresholding_ratio: float = 0.995,
        clip_sample_range: float = 1.0,
        sample_max_value: float = 1.0,
        weight: str = 'none',
        max_snr: float = 5.0,
        device: Union[str, torch.device] = None,
        verbose: bool = True,
    ):
        T = int(num_train_timesteps)
        self.betas = self._make_betas(beta_schedule, beta_start, beta_end, T)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)           # fp32 table, host
        # alpha-bar "before step 0": 1.0, or alphas_cumprod[0] when set_alpha_to_one=False
        self.final_alpha_cumprod = self.alphas_cumprod[0] if not set_alpha_to_one else torch.tensor(1.0)
        self.compute_loss_weights(mode=weight, max_snr=max_snr)
        self.weights = self.weights.to(device)
        self._install_timesteps(torch.arange(T - 1, -1, -1, dtype=torch.int64), list(range(T - 1, -1, -1)))   # 999..0
        self.num_train_timesteps = T
        self.num_inference_steps = None
        self.init_noise_sigma = 1.0
        for name, value in (("clip_sample", clip_sample), ("clip_sample_range", clip_sample_range),
                            ("prediction_type", prediction_type), ("thresholding", thresholding),
                            ("dynamic_thresholding_ratio", dynamic_thresholding_ratio),
                            ("steps_offset", steps_offset), ("beta_schedule", beta_schedule),
                            ("beta_start", beta_start), ("beta_end", beta_end), ("verbose", verbose)):
            setattr(self, name, value)
        self._ac_dev = {}               # device -> alphas_cumprod copy for add/remove_noise

    # `timesteps` is a property: the host copy (python ints, so that `step` needs no D2H sync) is only trusted while
    # the tensor it was built from is still the one installed.  Assigning `scheduler.timesteps = something` (e.g. a
    # slice for partial denoising) drops the cache; it is rebuilt from the tensor on first use.
    @property
    def timesteps(self):
        return self._timesteps

    @timesteps.setter
    def timesteps(self, value):
        self._install_timesteps(value, None)

    @staticmethod
    def _tensor_key(t):
        return (t.untyped_storage().data_ptr(), t.storage_offset(), tuple(t.shape), str(t.device))

    def _install_timesteps(self, tensor, host_list):
        self._timesteps = tensor
        self._timesteps_host = host_list
        self._timesteps_key = self._tensor_key(tensor) if isinstance(tensor, torch.Tensor) else None

    # ------------------------------------------------------------------ host logic
    def compute_loss_weights(self, mode='max_clamp_snr', max_snr=5.0):
        assert mode in ['inverse_log_snr', 'max_clamp_snr', 'linear', 'fixed', 'none']
        self.weight_mode = mode
        snr = self.alphas_cumprod / (1 - self.alphas_cumprod)
        if mode == 'inverse_log_snr':
            w = torch.log(1. / snr).clamp(min=1)
            self.weights = w / w[-1]
        elif mode == 'max_clamp_snr':
            self.weights = snr.clamp(max=max_snr) / snr
        elif mode == 'fixed':
            self.weights = snr.clone()
            self.weights[:len(self.weights) // 4] = 0.1
        elif mode == 'linear':
            self.weights = torch.arange(1, len(snr) + 1) / len(snr)
        else:
            self.weights = torch.ones_like(snr)

    def set_timesteps_inference(self, num_inference_steps: int, device: Union[str, torch.device] = None, tmin: int = 0):
        """Integer inference grid, bit-exact with ddim_scheduler.py:119-131."""
        self.num_inference_steps = num_inference_steps
        step_ratio = self.num_train_timesteps // self.num_inference_steps
        self.steps_offset = step_ratio - 1
        timesteps = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.int64)
        timesteps = timesteps + self.steps_offset
        timesteps = timesteps[timesteps >= tmin]
        self._install_timesteps(torch.from_numpy(timesteps).to(device), [int(t) for t in timesteps])

    def move_timesteps_to(self, device: Union[str, torch.device]):
        host = self._timesteps_host if self._cache_valid() else None
        self._install_timesteps(self._timesteps.to(device), host)

    def _cache_valid(self) -> bool:
        t = self._timesteps
        return (self._timesteps_host is not None and isinstance(t, torch.Tensor)
                and self._timesteps_key == self._tensor_key(t) and len(self._timesteps_host) == t.numel())

    def get_betas_for_alpha_bar(self, num_diffusion_timesteps, max_beta=0.999) -> torch.Tensor:
        """Glide cosine schedule: beta_i = min(1 - abar((i+1)/T) / abar(i/T), max_beta)."""
        T = num_diffusion_timesteps
        abar = [math.cos((i / T + 0.008) / 1.008 * math.pi / 2) ** 2 for i in range(T + 1)]
        return torch.tensor([min(1 - abar[i + 1] / abar[i], max_beta) for i in range(T)], dtype=torch.float32)

    def _make_betas(self, schedule, b0, b1, T):
        if schedule == "linear":
            return torch.linspace(b0, b1, T, dtype=torch.float32)
        if schedule == "scaled_linear":          # the latent-diffusion schedule (base.yaml:50)
            return torch.linspace(b0 ** 0.5, b1 ** 0.5, T, dtype=torch.float32) ** 2
        if schedule == "squaredcos_cap_v2":
            return self.get_betas_for_alpha_bar(T)
        if schedule == "sigmoid":
            return torch.sigmoid(torch.linspace(-6, 6, T)) * (b1 - b0) + b0
        raise NotImplementedError(f"{schedule} does is not implemented for {self.__class__}")

    def timesteps_host(self):
        """`self.timesteps` as python ints without touching the device."""
        if not self._cache_valid():
            t = self._timesteps
            self._install_timesteps(t, [int(v) for v in torch.as_tensor(t).reshape(-1).cpu()])
        return self._timesteps_host

    def _timestep_int(self, timestep) -> int:
        if isinstance(timestep, torch.Tensor):
            if timestep.is_cuda and self._cache_valid():
                # the common case: `for t in scheduler.timesteps` - an element view of the installed tensor is
                # resolved through the host copy by its storage offset (valid only while that tensor is installed)
                base = self._timesteps
                if (base.is_cuda and base.dim() == 1 and base.is_contiguous() and timestep.dim() == 0
                        and timestep.untyped_storage().data_ptr() == base.untyped_storage().data_ptr()):
                    idx = timestep.storage_offset() - base.storage_offset()
                    if 0 <= idx < len(self._timesteps_host):
                        return self._timesteps_host[idx]
            return int(timestep.item())
        return int(timestep)

    def step_coefficients(self, timestep: int):
        """The four 0-d fp32 scalars `step` uses (ddim_scheduler.py:231-267), as python floats
        holding exactly-representable fp32 values: sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev)."""
        prev_timestep = timestep - self.num_train_timesteps // self.num_inference_steps
        alpha_prod_t = self.alphas_cumprod[timestep]
        alpha_prod_t_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        beta_prod_t = 1 - alpha_prod_t
        return (float(alpha_prod_t ** 0.5), float(beta_prod_t ** 0.5),
                float(alpha_prod_t_prev ** 0.5), float((1 - alpha_prod_t_prev) ** 0.5))

    def coefficient_table(self):
        """[n_steps, 4] fp32 numpy table for the whole inference grid."""
        return np.asarray([self.step_coefficients(t) for t in self.timesteps_host()], dtype=np.float32)

    # ------------------------------------------------------------------ device ops
    def _ac_on(self, device):
        key = str(device)
        if key not in self._ac_dev:
            self._ac_dev[key] = self.alphas_cumprod.to(device=device, dtype=torch.float32).contiguous()
        return self._ac_dev[key]

    def _noise_op(self, fn_name, a, noise, timesteps, scale):
        a = _lib.require_cuda_f32(a, "samples")
        noise = _lib.require_cuda_f32(noise, "noise")
        timesteps = torch.as_tensor(timesteps)
        if not timesteps.is_cuda and timesteps.numel() and (
                int(timesteps.min()) < -self.num_train_timesteps or int(timesteps.max()) >= self.num_train_timesteps):
            raise IndexError(f"timestep out of range for a table of {self.num_train_timesteps} entries")   # as the reference's table lookup
        t = timesteps.to(device=a.device, dtype=torch.int64).contiguous().flatten()
        if not timesteps.is_cuda:
            t = torch.where(t < 0, t + self.num_train_timesteps, t)      # python-style negative indices, like tensor[t]
        B = a.shape[0]
        if t.numel() == 1 and B > 1:
            t = t.expand(B).contiguous()
        if t.numel() != B:
            raise ValueError("timesteps must hold one entry per sample")
        out = torch.empty_like(a)
        with torch.cuda.device(a.device):
            fn = getattr(_lib.lib(), fn_name)
            _lib.check(fn(_lib.ptr(a), _lib.ptr(noise), _lib.ptr(t), _lib.ptr(self._ac_on(a.device)),
                          int(self.num_train_timesteps), float(scale), _lib.ptr(out), B, a.numel() // B,
                          _lib.stream_ptr(a.device)), fn_name)
        return out

    def add_noise(self, original_samples, noise, timesteps, scale: float = 1.0, mask_noise_perc: Optional[float] = None):
        if mask_noise_perc is not None:
            mask = torch.rand_like(original_samples) < mask_noise_perc
            noise = noise * mask
        return self._noise_op("ldmseg_add_noise", original_samples, noise, timesteps, scale)

    @torch.no_grad()
    def remove_noise(self, noisy_samples, noise, timesteps, scale: float = 1.0):
        return self._noise_op("ldmseg_remove_noise", noisy_samples, noise, timesteps, scale)

    def step(self, model_output, timestep, sample, use_clipped_model_output: bool = False):
        if self.prediction_type not in _lib.PRED:
            raise NotImplementedError
        if self.thresholding:
            raise NotImplementedError
        t = self._timestep_int(timestep)
        sa_t, sb_t, sa_p, sb_p = self.step_coefficients(t)
        mo = _lib.require_cuda_f32(model_output, "model_output")
        x = _lib.require_cuda_f32(sample, "sample")
        prev = torch.empty_like(x)
        x0 = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().ldmseg_ddim_step(
                _lib.ptr(mo), _lib.ptr(x), sa_t, sb_t, sa_p, sb_p, _lib.PRED[self.prediction_type],
                int(bool(self.clip_sample)), float(self.clip_sample_range), int(bool(use_clipped_model_output)),
                _lib.ptr(prev), _lib.ptr(x0), x.numel(), _lib.stream_ptr(x.device)), "ldmseg_ddim_step")
        return DDIMNoiseSchedulerOutput(prev_sample=prev, pred_original_sample=x0)

    def __str__(self) -> str:
        fields = ["num_inference_steps", "num_train_timesteps", "prediction_type", "beta_start", "beta_end",
                  "beta_schedule", "clip_sample", "clip_sample_range", "thresholding",
                  "dynamic_thresholding_ratio", "steps_offset", "weight_mode"]
        body = ", ".join(f"{k}={getattr(self, k)}" for k in fields)
        return f"DDIMScheduler({body}, weights={self.weights if self.verbose else 'VerboseDisabled'})"

    def __repr__(self) -> str:
        return self.__str__()

    def __len__(self) -> int:
        return self.num_train_timesteps
