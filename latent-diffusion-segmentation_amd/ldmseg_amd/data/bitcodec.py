"""Bit encoding of panoptic segment ids, on the GPU.

Stands behind ``COCO.encode_bitmap`` / ``COCO.decode_bitmap``
(/root/reference/ldmseg/data/coco.py:377-390): ids -> n bit planes (LSB first, void -> 0.5) and back
(bit k set iff plane k > 0).  Accepts [H,W] (the reference's per-sample call) or [B,H,W]; integer
results are bit-exact.  ``affine=(2, -1)`` fuses the ``2*x - 1`` of encode_inputs
(trainers_ldm_cond.py:369) into the encode.
"""
import torch

from .. import _lib


def encode_bitmap(x: torch.Tensor, n: int = 7, fill_value: float = 0.5, ignore_label: int = 0, affine=(1.0, 0.0)):
    if not x.is_cuda:
        raise RuntimeError("encode_bitmap needs a tensor on the MI355X (no CPU fallback)")
    single = x.dim() == 2
    ids = (x[None] if single else x).to(torch.int64).contiguous()
    B, H, W = ids.shape
    bits = torch.empty((B, n, H, W), device=ids.device, dtype=torch.float32)
    mask = torch.empty((B, H, W), device=ids.device, dtype=torch.uint8)
    with torch.cuda.device(ids.device):
        _lib.check(_lib.lib().ldmseg_bit_encode(_lib.ptr(ids), B, n, H * W, int(ignore_label), float(fill_value),
                                                float(affine[0]), float(affine[1]), _lib.ptr(bits), _lib.ptr(mask),
                                                _lib.stream_ptr(ids.device)), "ldmseg_bit_encode")
    mask = mask.bool()
    return (bits[0], mask[0]) if single else (bits, mask)


def decode_bitmap(x: torch.Tensor, n: int = 7) -> torch.Tensor:
    if not x.is_cuda:
        raise RuntimeError("decode_bitmap needs a tensor on the MI355X (no CPU fallback)")
    single = x.dim() == 3
    planes = (x[None] if single else x).to(torch.float32).contiguous()
    B, nb, H, W = planes.shape            # like the reference, the channel count decides the bit count
    out = torch.empty((B, H, W), device=planes.device, dtype=torch.int64)
    with torch.cuda.device(planes.device):
        _lib.check(_lib.lib().ldmseg_bit_decode(_lib.ptr(planes), B, nb, H * W, _lib.ptr(out),
                                                _lib.stream_ptr(planes.device)), "ldmseg_bit_decode")
    return out[0] if single else out
