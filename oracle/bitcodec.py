"""Oracle: bit encoding of segment ids (TEST INFRASTRUCTURE, see oracle/__init__.py).
Restates COCO.encode_bitmap / decode_bitmap (/root/reference/ldmseg/data/coco.py:377-390) in numpy."""
import numpy as np


def encode_bitmap(ids, n=7, fill_value=0.5, ignore_label=0):
    ids = np.asarray(ids, dtype=np.int64)
    ignore = ids == ignore_label
    bits = np.stack([np.mod(ids >> k, 2) for k in range(n)], axis=-3).astype(np.float32)
    bits[..., :, ignore] = fill_value if ids.ndim == 2 else bits[..., :, ignore]
    if ids.ndim == 3:
        for b in range(ids.shape[0]):
            bits[b][:, ignore[b]] = fill_value
    return bits, ignore


def decode_bitmap(x):
    x = np.asarray(x)
    n = x.shape[-3]
    w = (2.0 ** np.arange(n, dtype=np.float32)).reshape((n, 1, 1))
    return ((x > 0).astype(np.float32) * w).sum(axis=-3).astype(np.int64)
