"""GPU parity: ldmseg_panoptic_postprocess (through TrainerDiffusion.postprocess_panoptic) against the
CPU restatement of trainers_ldm_cond.py:1277-1313.  Integer outputs (labels, counts, keep, panoptic map)
are compared exactly."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import postprocess as o_pp

pytestmark = pytest.mark.gpu


def _segment_like_logits(B, C, H, W, seed, sharp=6.0):
    g = torch.Generator().manual_seed(seed)
    low = torch.randn(B, C, max(2, H // 16), max(2, W // 16), generator=g)
    x = F.interpolate(low, size=(H, W), mode="bilinear", align_corners=False) * sharp
    return x + 0.3 * torch.randn(B, C, H, W, generator=g)


def _trainer():
    from ldmseg_amd.trainers import TrainerDiffusion
    return TrainerDiffusion.__new__(TrainerDiffusion)      # post-processing needs no model objects


@pytest.mark.parametrize("B,C,H,W", [(2, 128, 96, 80), (1, 16, 33, 47), (3, 128, 64, 64)])
@pytest.mark.parametrize("kw", [
    dict(threshold_output=False, count_th=64, overlap_th=0.5, ignore_label=0),
    dict(threshold_output=True, threshold_mode="max", mask_th=0.5, count_th=32, overlap_th=0.3, ignore_label=0),
    dict(threshold_output=True, threshold_mode="topk_diff", mask_th=0.4, count_th=16, overlap_th=0.6, ignore_label=5),
    dict(threshold_output=True, threshold_mode="max", mask_th=0.9, count_th=1, overlap_th=0.0, ignore_label=-1),
])
def test_postprocess_matches_oracle(B, C, H, W, kw):
    x = _segment_like_logits(B, C, H, W, seed=H * 7 + C)
    tr = _trainer()
    res, st = tr.postprocess_panoptic(x.cuda(), return_stats=True, **kw)
    assert len(res) == B
    for b in range(B):
        pan, info, raw, ost = o_pp.panoptic_postprocess(x[b], **kw)
        assert np.array_equal(st["labels"][b].cpu().numpy(), raw)
        assert np.array_equal(st["counts"][b].cpu().numpy(), ost["counts"])
        assert np.array_equal(st["mask_counts"][b].cpu().numpy(), ost["mask_counts"])
        got_pan, got_info = res[b]["panoptic_seg"]
        assert got_info == info
        assert np.array_equal(got_pan.cpu().numpy(), pan)


def test_postprocess_full_size_properties():
    """BASELINE size (8 x 128 x 512 x 512 logits): size-independent invariants."""
    B, C, H, W = 8, 128, 512, 512
    g = torch.Generator(device="cuda").manual_seed(3)
    low = torch.randn(B, C, 32, 32, device="cuda", generator=g)
    x = F.interpolate(low, size=(H, W), mode="bilinear", align_corners=False) * 6
    tr = _trainer()
    res, st = tr.postprocess_panoptic(x, threshold_output=True, mask_th=0.5, count_th=512, overlap_th=0.5,
                                      ignore_label=0, return_stats=True)
    labels, counts, keep = st["labels"], st["counts"].long(), st["keep"].bool()
    assert int(counts.sum()) == int((labels >= 0).sum())                       # every non-void pixel counted once
    assert torch.equal(counts, torch.stack([torch.bincount(l[l >= 0].flatten().long(), minlength=C) for l in labels]))
    assert (counts[keep] >= 512).all() and not keep[:, 0].any()
    for b in range(B):
        pan, info = res[b]["panoptic_seg"]
        kept = {s["id"] for s in info}
        assert set(torch.unique(pan).tolist()) - {0} == kept                    # map and segments_info agree
        assert torch.equal(pan > 0, keep[b][labels[b].clamp(min=0).long()] & (labels[b] >= 0))
    # idempotent on its own output: one-hot logits of the kept map give the same map back
    again = tr.postprocess_panoptic(x, threshold_output=True, mask_th=0.5, count_th=512, overlap_th=0.5, ignore_label=0)
    assert all(torch.equal(a["panoptic_seg"][0], r["panoptic_seg"][0]) for a, r in zip(again, res))


def test_postprocess_rejects_host_tensors():
    with pytest.raises(RuntimeError):
        _trainer().postprocess_panoptic(torch.zeros(1, 4, 8, 8))
