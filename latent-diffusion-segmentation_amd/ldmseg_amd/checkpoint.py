"""Readers for the reference's checkpoint layouts (torch pickles), so the library runs on released weights.

* LDM checkpoint ``ldmseg.pt`` = ``{'step','epoch','vae_image','vae_semseg','unet','ema','opt','p','scaler'}``
  (/root/reference/ldmseg/trainers/trainers_ldm_cond.py:1791-1814; loaded at :1863-1891).  ``unet`` carries no
  ``module.`` prefix but holds the duplicate ``new_conv.*`` alias of ``conv_in.*`` (unet.py:182,233).
* AE checkpoint ``ae.pt`` = ``{'step','epoch','vae','opt','p','scaler'}`` with ``module.``-prefixed keys
  (trainers_ae.py:497-505; stripped at vae.py:116-121).
"""
from collections import OrderedDict
from typing import Optional

import torch

from .weights import unet_schema, vae_schema


def _strip(sd):
    return OrderedDict((k[len("module."):] if k.startswith("module.") else k, v) for k, v in sd.items())


def unet_state_from(data: dict, use_ema: bool = False) -> "OrderedDict[str, torch.Tensor]":
    """Select and validate the UNet tensors of an LDM checkpoint dict."""
    sd = data["ema"] if use_ema and data.get("ema") is not None else data["unet"]
    if isinstance(sd, dict) and "shadow_params" in sd:
        raise NotImplementedError("diffusers EMAModel state (shadow_params list) is not a state dict")
    sd = _strip(sd)
    in_ch = int(sd["conv_in.weight"].shape[1])
    cross = any(".attn2." in k for k in sd)
    if cross:
        raise NotImplementedError("checkpoint has cross-attention (attn2) weights; only image_descriptors=remove is built")
    schema = unet_schema(in_ch, False)
    out = OrderedDict()
    for k, shp in schema.items():
        if k not in sd:
            raise KeyError(f"checkpoint lacks UNet tensor {k}")
        if tuple(sd[k].shape) != tuple(shp):
            raise ValueError(f"{k}: checkpoint shape {tuple(sd[k].shape)} != expected {tuple(shp)}")
        out[k] = sd[k]
    return out


def vae_state_from(data: dict) -> "OrderedDict[str, torch.Tensor]":
    """seg-VAE tensors from either checkpoint kind ('vae_semseg' of ldmseg.pt or 'vae' of ae.pt)."""
    sd = _strip(data["vae_semseg"] if "vae_semseg" in data else data["vae"])
    schema = vae_schema()
    return OrderedDict((k, sd[k]) for k in schema)


def _load(path: str, allow_pickle: Optional[bool] = None):
    """Tensors-only unpickling (``weights_only=True``).  The reference's checkpoints also pickle the config dict `p`
    (plain python containers: accepted by the safe loader) and, in some runs, optimizer / scaler objects that it refuses.
    The full unpickler executes code from the file, so it runs only on an explicit opt-in: ``allow_pickle=True`` or the
    environment variable ``LDMSEG_ALLOW_PICKLE=1``; I/O errors (missing / truncated file) are never retried."""
    import os
    import pickle
    if allow_pickle is None:
        allow_pickle = os.environ.get("LDMSEG_ALLOW_PICKLE", "0") == "1"
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError as e:
        if not allow_pickle:
            raise pickle.UnpicklingError(
                f"{path}: refused by the tensors-only loader ({e}); pass allow_pickle=True or set LDMSEG_ALLOW_PICKLE=1 "
                "to unpickle a checkpoint you trust") from e
        return torch.load(path, map_location="cpu", weights_only=False)


def load_ldm_checkpoint(path: str, use_ema: bool = False, allow_pickle: Optional[bool] = None):
    data = _load(path, allow_pickle)
    return {"unet": unet_state_from(data, use_ema), "vae_semseg": vae_state_from(data) if "vae_semseg" in data else None,
            "p": data.get("p"), "step": data.get("step"), "epoch": data.get("epoch")}


def load_ae_checkpoint(path: str, allow_pickle: Optional[bool] = None):
    data = _load(path, allow_pickle)
    return vae_state_from(data)


def build_models(ldm_path: str, ae_path: Optional[str] = None, device="cuda:0", compute_dtype="bf16",
                 scaling_factor: float = 0.18215):
    """UNet + seg-VAE on the MI355X from reference checkpoints (what main_ldm.py:137-168,215-216 assembles)."""
    from .models import UNet, GeneralVAESeg
    ck = load_ldm_checkpoint(ldm_path)
    vsd = load_ae_checkpoint(ae_path) if ae_path else ck["vae_semseg"]
    unet = UNet(ck["unet"], in_channels=int(ck["unet"]["conv_in.weight"].shape[1]), device=device,
                compute_dtype=compute_dtype)
    vae = GeneralVAESeg(vsd, scaling_factor=scaling_factor, device=device, compute_dtype=compute_dtype)
    return unet, vae
