"""CPU suite: known-answer checks of the panoptic post-processing oracle (restatement of
trainers_ldm_cond.py:1277-1313; the reference lines are inline in a trainer method and cannot be called
in isolation, so the oracle is checked against hand-derived answers)."""
import numpy as np
import torch

from oracle import postprocess as o_pp


def _logits_from_labels(lab, C, hi=6.0, lo=-6.0):
    x = torch.full((C,) + lab.shape, lo)
    x.scatter_(0, torch.as_tensor(lab)[None].long(), hi)
    return x


def test_small_segments_and_ignore_label_are_voided():
    lab = np.zeros((8, 8), np.int64)
    lab[:, 4:] = 2          # 32 px of class 2
    lab[0, 0:3] = 3         # 3 px of class 3 (small)
    lab[7, 0] = 1           # 1 px of class 1 (small)
    x = _logits_from_labels(lab, 5)
    pan, info, raw, st = o_pp.panoptic_postprocess(x, count_th=4, overlap_th=0.5, ignore_label=0)
    assert np.array_equal(raw, lab)
    assert st["counts"].tolist() == [28, 1, 32, 3, 0]
    # class 0 = ignore_label, classes 1 and 3 are below count_th, class 2 survives with id 3
    assert [s["id"] for s in info] == [3]
    assert np.array_equal(pan, np.where(lab == 2, 3, 0))
    # with another ignore label class 0 becomes a segment (id 1)
    pan, info, _, _ = o_pp.panoptic_postprocess(x, count_th=4, overlap_th=0.5, ignore_label=4)
    assert [s["id"] for s in info] == [1, 3]
    assert np.array_equal(pan, np.where(lab == 2, 3, np.where(lab == 0, 1, 0)))


def test_overlap_filter_uses_sigmoid_mask_area():
    # class 1 wins the argmax on 8 px but its sigmoid >= 0.5 on 32 px -> overlap 0.25
    x = torch.full((3, 8, 8), -4.0)
    x[2] = 1.0                       # background class 2 everywhere (sigmoid 0.73)
    x[1, :, :4] = 0.5                # class 1: sigmoid 0.62 on 32 px ...
    x[1, 0, :] = 3.0                 # ... and the argmax on the first row only (8 px)
    pan, info, raw, st = o_pp.panoptic_postprocess(x, count_th=1, overlap_th=0.3, ignore_label=0)
    assert st["counts"].tolist() == [0, 8, 56] and st["mask_counts"].tolist() == [0, 36, 64]
    assert [s["id"] for s in info] == [3]                     # 8/36 < 0.3 -> class 1 voided
    assert (pan[0] == 0).all() and (pan[1:] == 3).all()
    pan, info, _, _ = o_pp.panoptic_postprocess(x, count_th=1, overlap_th=0.2, ignore_label=0)
    assert [s["id"] for s in info] == [2, 3] and (pan[0] == 2).all()


def test_threshold_modes():
    x = torch.zeros(4, 2, 2)
    x[1, 0, 0] = 5.0                 # confident pixel: p(max) = 0.98
    x[2, 0, 1] = 0.2                 # flat pixel: p(max) = 0.29, top1 - top2 = 0.05
    x[3, 1, :] = 1.5                 # p(max) = 0.60, top1 - top2 = 0.46
    _, _, raw, _ = o_pp.panoptic_postprocess(x, threshold_output=True, threshold_mode="max", mask_th=0.5, count_th=0)
    assert raw.tolist() == [[1, -1], [3, 3]]
    _, _, raw, _ = o_pp.panoptic_postprocess(x, threshold_output=True, threshold_mode="topk_diff", mask_th=0.5, count_th=0)
    assert raw.tolist() == [[1, -1], [-1, -1]]
    pan, info, raw, _ = o_pp.panoptic_postprocess(x, threshold_output=False, count_th=0, overlap_th=0.0, ignore_label=-7)
    assert raw.tolist() == [[1, 2], [3, 3]] and [s["id"] for s in info] == [2, 3, 4]
