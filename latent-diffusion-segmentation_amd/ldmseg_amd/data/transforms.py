"""Validation-time image transforms of the reference, host side (PIL), for the evaluation entry.

Stands behind /root/reference/ldmseg/data/util/pil_transforms.py: `CropResize` (:99-144; with the `crop_mode=None` the
reference hard-wires at :102 it is a plain PIL resize to size x size - bicubic for images, nearest for id maps, :131-135
and INT_MODES) and `ToTensor` (:150-170: images -> float [3,H,W] in [0,1], id maps -> int64).  These run on the host in
the reference's data loader too; PIL is the reference's own dependency and is what makes the pixels identical.
"""
from typing import Tuple

import numpy as np
import torch


def _resample(mode: str):
    from PIL import Image
    R = getattr(Image, "Resampling", Image)
    return {"bicubic": R.BICUBIC, "bilinear": R.BILINEAR, "nearest": R.NEAREST}[mode]


def crop_resize(img, size: Tuple[int, int], mode: str = "bicubic"):
    """PIL image -> PIL image of (height, width) = size (CropResize.crop_and_resize with crop_mode None)."""
    h, w = size
    return img.resize((w, h), resample=_resample(mode), reducing_gap=None)


def image_to_tensor(img) -> torch.Tensor:
    """torchvision ToTensor on a PIL RGB image: uint8 HWC -> float32 CHW in [0, 1]."""
    a = np.asarray(img.convert("RGB"), dtype=np.uint8)
    return torch.from_numpy(a).permute(2, 0, 1).float().div(255.0).contiguous()


def ids_to_tensor(img) -> torch.Tensor:
    return torch.from_numpy(np.array(img)).long()


def load_rgb(path: str, size: int):
    """One validation sample like COCO.__getitem__ + get_val_transforms: returns (image [3,size,size] in [0,1],
    original (h, w))."""
    from PIL import Image
    img = Image.open(path).convert("RGB")
    w, h = img.size
    return image_to_tensor(crop_resize(img, (size, size), "bicubic")), (h, w)
