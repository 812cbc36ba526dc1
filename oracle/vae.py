"""Oracle: segmentation VAE (TEST INFRASTRUCTURE, see oracle/__init__.py).

Functional torch-CPU fp32 restatement of
/root/reference/ldmseg/models/vae.py::GeneralVAESeg for the default
``vae_model_kwargs`` (tools/configs/base/base.yaml:14-33: gaussian
parametrization, num_mid_blocks=0, num_upscalers=2):
  encoder  <- define_encoder   (:174-244)   keys encoder.{0,2,3,5,6,8,9,11,13,15}
  decoder  <- define_decoder   (:123-172)   keys decoder.{0,2,3,5,6,8,10}
  LayerNorm2d                  (:309-322)   biased variance over C per pixel
  posterior mode/sample        (:370-413)
  decode + bilinear x2         (:267-271)
Takes a state dict with the reference's key names.
"""
import torch
import torch.nn.functional as F


def layernorm2d(x, w, b, eps=1e-6):
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return w[:, None, None] * x + b[:, None, None]


def encode_moments(sd, x):
    def conv(i, h, stride=1):
        return F.conv2d(h, sd[f"encoder.{i}.weight"], sd[f"encoder.{i}.bias"], stride=stride, padding=1)
    h = F.silu(conv(0, x))
    h = conv(2, h)
    h = F.silu(conv(3, h, 2))
    h = conv(5, h)
    h = F.silu(conv(6, h, 2))
    h = conv(8, h)
    h = F.silu(conv(9, h, 2))
    h = conv(11, h)
    h = F.group_norm(h, 32, sd["encoder.13.weight"], sd["encoder.13.bias"], eps=1e-6)
    h = F.silu(h)
    return conv(15, h)


def encode_mode(sd, x):
    """posterior.mode() == mean (vae.py:388, 407-408)."""
    mean, _ = torch.chunk(encode_moments(sd, x), 2, dim=1)
    return mean


def encode_sample(sd, x, noise):
    mean, logvar = torch.chunk(encode_moments(sd, x), 2, dim=1)
    std = torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0))
    return mean + std * noise


def decode(sd, z, interpolate=True, interpolation_factor=2, checkpoints=None):
    h = F.conv2d(z, sd["decoder.0.weight"], sd["decoder.0.bias"], padding=1)
    for ct, ln in ((2, 3), (5, 6)):
        h = F.conv_transpose2d(h, sd[f"decoder.{ct}.weight"], sd[f"decoder.{ct}.bias"], stride=2)
        if checkpoints is not None:
            checkpoints[f"convt{ct}"] = h
        h = F.silu(layernorm2d(h, sd[f"decoder.{ln}.weight"], sd[f"decoder.{ln}.bias"]))
    h = F.silu(F.group_norm(h, 32, sd["decoder.8.weight"], sd["decoder.8.bias"], eps=1e-5))
    if checkpoints is not None:
        checkpoints["gn"] = h
    h = F.conv2d(h, sd["decoder.10.weight"], sd["decoder.10.bias"], padding=1)
    if interpolate:
        h = F.interpolate(h, scale_factor=interpolation_factor, mode="bilinear", align_corners=False)
    return h
