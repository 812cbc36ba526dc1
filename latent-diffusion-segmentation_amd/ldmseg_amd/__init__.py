"""ldmseg_amd - MI355X (gfx950) implementation of the LDMSeg denoising path.

Package layout mirrors the reference's ``ldmseg`` package for the hot path only:
``schedulers.DDIMNoiseScheduler``, ``models.UNet``, ``models.GeneralVAESeg``,
``trainers.TrainerDiffusion`` (sample / decode_latents / encode_inputs).
"""
__version__ = "0.1.0"
