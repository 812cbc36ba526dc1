"""Oracle: DDIM noise schedule (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates /root/reference/ldmseg/schedulers/ddim_scheduler.py:
  tables          <- DDIMNoiseScheduler.__init__            (:32-95)
  loss weights    <- compute_loss_weights                   (:97-117)
  inference grid  <- set_timesteps_inference                (:119-131)
  glide betas     <- get_betas_for_alpha_bar                (:138-153)
  add/remove      <- add_noise / remove_noise               (:155-216)
  step            <- step                                   (:218-269)

Arithmetic is done with torch CPU fp32 tensors so that 0-d scalar math
(``x ** 0.5`` on fp32) rounds exactly as the reference does.
"""
import math

import numpy as np
import torch


def make_betas(schedule, beta_start, beta_end, T):
    if schedule == "linear":
        return torch.linspace(beta_start, beta_end, T, dtype=torch.float32)
    if schedule == "scaled_linear":
        return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, T, dtype=torch.float32) ** 2
    if schedule == "squaredcos_cap_v2":
        def abar(s):
            return math.cos((s + 0.008) / 1.008 * math.pi / 2) ** 2
        out = [min(1 - abar((i + 1) / T) / abar(i / T), 0.999) for i in range(T)]
        return torch.tensor(out, dtype=torch.float32)
    if schedule == "sigmoid":
        return torch.sigmoid(torch.linspace(-6, 6, T)) * (beta_end - beta_start) + beta_start
    raise NotImplementedError(schedule)


def loss_weights(alphas_cumprod, mode, max_snr):
    snr = alphas_cumprod / (1 - alphas_cumprod)
    if mode == "max_clamp_snr":
        return snr.clamp(max=max_snr) / snr
    if mode == "fixed":
        w = snr.clone()
        w[: len(w) // 4] = 0.1
        return w
    if mode == "linear":
        return torch.arange(1, len(snr) + 1) / len(snr)
    if mode == "none":
        return torch.ones_like(snr)
    # 'inverse_log_snr' raises inside the reference under torch 2.10
    raise NotImplementedError(mode)


class OracleDDIM:
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02,
                 beta_schedule="linear", clip_sample=True, set_alpha_to_one=True,
                 prediction_type="epsilon", clip_sample_range=1.0,
                 weight="none", max_snr=5.0, **_ignored):
        self.T = num_train_timesteps
        self.betas = make_betas(beta_schedule, beta_start, beta_end, num_train_timesteps)
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.weights = loss_weights(self.alphas_cumprod, weight, max_snr)
        self.timesteps = torch.from_numpy(np.arange(num_train_timesteps)[::-1].copy().astype(np.int64))
        self.num_inference_steps = None
        self.clip_sample = clip_sample
        self.clip_sample_range = clip_sample_range
        self.prediction_type = prediction_type
        self.init_noise_sigma = 1.0

    def set_timesteps_inference(self, n, tmin=0):
        self.num_inference_steps = n
        ratio = self.T // n
        grid = (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64)
        t = torch.from_numpy(grid) + (ratio - 1)
        self.timesteps = t[t >= tmin]

    def coefficients(self, t):
        """(a_t, a_prev) as 0-d fp32 tensors, reference :231-236."""
        t = int(t)
        prev = t - self.T // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        return a_t, a_prev

    def step(self, model_output, timestep, sample, use_clipped_model_output=False):
        a_t, a_prev = self.coefficients(timestep)
        b_t = 1 - a_t
        if self.prediction_type == "epsilon":
            x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
            eps = model_output
        elif self.prediction_type == "sample":
            x0 = model_output
            eps = (sample - a_t ** 0.5 * x0) / b_t ** 0.5
        elif self.prediction_type == "v_prediction":
            x0 = (a_t ** 0.5) * sample - (b_t ** 0.5) * model_output
            eps = (a_t ** 0.5) * model_output + (b_t ** 0.5) * sample
        else:
            raise NotImplementedError
        if self.clip_sample:
            x0 = x0.clamp(-self.clip_sample_range, self.clip_sample_range)
        if use_clipped_model_output:
            eps = (sample - a_t ** 0.5 * x0) / b_t ** 0.5
        prev = a_prev ** 0.5 * x0 + (1 - a_prev) ** 0.5 * eps
        return prev, x0

    def _bcast(self, v, ref):
        v = v.flatten()
        while v.dim() < ref.dim():
            v = v.unsqueeze(-1)
        return v

    def add_noise(self, x0, noise, timesteps, scale=1.0):
        ac = self.alphas_cumprod.to(x0.dtype)
        sa = self._bcast(ac[timesteps] ** 0.5, x0)
        sb = self._bcast((1 - ac[timesteps]) ** 0.5, x0)
        return sa * scale * x0 + sb * noise

    def remove_noise(self, xt, noise, timesteps, scale=1.0):
        ac = self.alphas_cumprod.to(xt.dtype)
        sa = self._bcast(ac[timesteps] ** 0.5, xt)
        sb = self._bcast((1 - ac[timesteps]) ** 0.5, xt)
        return (xt - sb * noise) / (sa * scale)
