"""Return-type contract of the reference call surface.

``OutputDict`` mirrors /root/reference/ldmseg/utils/utils.py:26-31: an ordered
dict whose items are also attributes (``out.sample`` == ``out['sample']``).
"""
from collections import OrderedDict

import numpy as np


class OutputDict(OrderedDict):
    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in OrderedDict(*args, **kwargs).items():
            self[k] = v

    def __setitem__(self, key, value):
        super().__setitem__(key, value)
        super().__setattr__(key, value)


class UNetOutput(OutputDict):           # unet.py:20-21
    pass


class DDIMNoiseSchedulerOutput(OutputDict):   # ddim_scheduler.py:21-23
    pass


class EncoderOutput(OutputDict):        # vae.py:31-33
    pass


class VAEOutput(OutputDict):            # vae.py:27-29
    pass


def color_map(N: int = 256, normalized: bool = False):
    """The PASCAL-VOC colour table the reference uses to paint id maps (utils.py:240-258): bit k of the id feeds
    bit 7 - k // 3 of channel k % 3.  Returns [N, 3] uint8 (or float32 in [0, 1])."""
    ids = np.arange(N, dtype=np.int64)
    cmap = np.zeros((N, 3), dtype=np.int64)
    for k in range(24):
        cmap[:, k % 3] |= ((ids >> k) & 1) << (7 - k // 3)
    return (cmap / 255).astype(np.float32) if normalized else cmap.astype(np.uint8)


def encode_seg(semseg, cmap=None):
    """TrainerDiffusion.encode_seg (trainers_ldm_cond.py:324-332): id map [B,H,W] -> colour image [B,H,W,3]
    (ids are taken modulo 256 like the reference's astype(uint8))."""
    if cmap is None:
        cmap = color_map()
    return cmap[np.asarray(semseg).astype(np.uint8)]
