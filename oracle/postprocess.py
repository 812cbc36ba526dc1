"""TEST INFRASTRUCTURE ONLY - CPU restatement of the panoptic post-processing loop of the reference's
evaluation (ldmseg/trainers/trainers_ldm_cond.py:1277-1313).  Product code must not import this.

PARITY UNPINNED: the reference runs these lines inline in a trainer method that needs detectron2's
evaluator, a dataloader and model objects, so it cannot be called in isolation here; the restatement
follows the lines one by one (torch for the device-side part :1277-1290, numpy for the host-side segment
filtering :1292-1313) and is what the HIP kernels are compared against.
"""
import numpy as np
import torch
import torch.nn.functional as F


def panoptic_postprocess(mask_pred_result: torch.Tensor, threshold_output=False, threshold_mode="max", mask_th=0.5,
                         count_th=512, overlap_th=0.5, ignore_label=0):
    """One image.  mask_pred_result [C,H,W] fp32 logits at the output size.

    Returns (panoptic_pred + 1 [H,W] int64 numpy, segments_info, raw labels [H,W], stats dict)."""
    mask_pred_result = mask_pred_result.float()
    panoptic_pred = torch.argmax(mask_pred_result, dim=0)                         # :1278
    if threshold_output:                                                          # :1279-1286
        probs = F.softmax(mask_pred_result, dim=0)
        if threshold_mode == "topk_diff":
            topk = torch.topk(probs, k=2, dim=0)
            probs = topk.values[0] - topk.values[1]
        else:
            probs = probs.max(dim=0)[0]
        panoptic_pred[probs < mask_th] = -1
    panoptic_pred = panoptic_pred.cpu().numpy()                                   # :1289
    raw = panoptic_pred.copy()
    sig = torch.sigmoid(mask_pred_result).cpu().numpy()                           # :1290-1291
    segments_info = []
    counts = np.zeros(mask_pred_result.shape[0], np.int64)
    mask_counts = np.zeros(mask_pred_result.shape[0], np.int64)
    for c in range(mask_pred_result.shape[0]):
        mask_counts[c] = int((sig[c] >= np.float32(mask_th)).sum())
    for panoptic_label, count_i in zip(*np.unique(panoptic_pred, return_counts=True)):   # :1295
        if panoptic_label >= 0:
            counts[panoptic_label] = count_i
        if count_i < count_th or panoptic_label in {-1, ignore_label}:            # :1298-1300
            panoptic_pred[panoptic_pred == panoptic_label] = -1
            continue
        original_mask = sig[panoptic_label] >= np.float32(mask_th)                # :1303
        with np.errstate(divide="ignore"):
            ratio = np.float64((panoptic_pred == panoptic_label).sum()) / np.float64(original_mask.sum())
        if ratio < overlap_th:                                                    # :1304-1306
            panoptic_pred[panoptic_pred == panoptic_label] = -1
            continue
        segments_info.append({"id": int(panoptic_label) + 1, "category_id": 1, "isthing": True})   # :1308-1312
    return panoptic_pred + 1, segments_info, raw, {"counts": counts, "mask_counts": mask_counts}    # :1313
