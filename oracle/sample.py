"""Oracle: the DDIM sampling loop (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates TrainerDiffusion.sample
(/root/reference/ldmseg/trainers/trainers_ldm_cond.py:1045-1170) for the
default eval configuration (no descriptor model, no text encoder ->
multiplier 1, encoder_hidden_states None) and decode_latents (:397-442,
return_logits=True).  ``eps_fn(inputs, t)`` stands in for
``self.unet_model(inputs, t, encoder_hidden_states=None).sample``.

``sample_inpaint`` is the BUILD-DEFINED mask-inpainting sampler (SURVEY.md
section 8a row A9; the reference has none): RePaint-style paste of the re-noised
known latents after every step, True = known, final step pastes z0.
"""
import torch


def initial_noise(batch, L, seed):
    """:1088-1092 - CPU generator, same draw for every batch of equal size."""
    g = torch.Generator().manual_seed(seed) if seed is not None else None
    return torch.randn((batch, 4, L, L), generator=g)


def sample(eps_fn, sched, rgb_latents, seed, self_condition=True, return_all=False, noise=None):
    B, _, L, _ = rgb_latents.shape
    latents = initial_noise(B, L, seed) if noise is None else noise.clone()
    latents = latents * sched.init_noise_sigma
    cond = torch.zeros_like(rgb_latents)
    n = len(sched.timesteps)
    allv = []
    for i, t in enumerate(sched.timesteps):
        parts = [latents, rgb_latents] + ([cond] if self_condition else [])
        eps = eps_fn(torch.cat(parts, dim=1), t)
        prev, x0 = sched.step(eps, t, latents)
        if self_condition:
            cond = x0
        latents = x0 if i == n - 1 else prev
        if return_all:
            allv.append(latents)
    return torch.cat(allv, dim=0) if return_all else latents


def sample_inpaint(eps_fn, sched, rgb_latents, z0, known, seed, self_condition=True, noise=None):
    """known: bool [B,1,L,L], True = latent is given (trainers_ldm_cond.py:613-615
    convention).  After each step the known region is replaced by z0 re-noised
    to the *next* timestep with the same fixed noise draw; the last step pastes z0.
    ``noise`` (optional) = the rows of a larger batch's draw, for checking single images of that batch."""
    B, _, L, _ = rgb_latents.shape
    noise = initial_noise(B, L, seed) if noise is None else noise.clone()
    latents = noise * sched.init_noise_sigma
    cond = torch.zeros_like(rgb_latents)
    ts = sched.timesteps
    n = len(ts)
    m = known.to(latents.dtype)
    for i, t in enumerate(ts):
        parts = [latents, rgb_latents] + ([cond] if self_condition else [])
        eps = eps_fn(torch.cat(parts, dim=1), t)
        prev, x0 = sched.step(eps, t, latents)
        if self_condition:
            cond = x0
        if i == n - 1:
            latents = m * z0 + (1 - m) * x0
        else:
            t_next = ts[i + 1].reshape(1).expand(B)
            latents = m * sched.add_noise(z0, noise, t_next) + (1 - m) * prev
    return latents


def decode_latents(vae_decode_fn, latents, scaling_factor):
    """:421-425 with return_logits=True."""
    return vae_decode_fn(latents * (1.0 / scaling_factor)).float()
