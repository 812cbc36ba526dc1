"""Class-agnostic panoptic-quality evaluation with the reference's call surface.

Stands behind /root/reference/ldmseg/evaluations/panoptic_evaluation_agnostic.py::PanopticEvaluatorAgnostic
(`reset` :74, `process` :96-126, `evaluate` :128-185, `pq_compute` :188-230) as `compute_pq` drives it
(trainers_ldm_cond.py:1205-1207, 1315, 1335).  This is host-side bookkeeping in the reference too (numpy on the CPU,
run once per evaluation) and stays there; two things differ in mechanism, not in result:

* the reference delegates the metric to **panopticapi** (`pq_compute_multi_core`, pinned only as the git dependency
  `panopticapi @ git+https://github.com/cocodataset/panopticapi.git` in the reference's environment; absent from this
  image and from /root/reference).  `pq_compute_annotations` below restates its published algorithm
  (panopticapi/evaluation.py, `pq_compute_single_core` + `PQStat.pq_average`): segments match when IoU > 0.5 with the
  union reduced by the prediction's overlap with VOID, crowd ground truth is never a false negative, an unmatched
  prediction is no false positive when more than half of it lies on VOID / crowd, PQ = sum IoU / (TP + FP/2 + FN/2).
  PARITY UNPINNED by the reference (no tests, dependency absent); known-answer tests in tests/test_pq_cpu.py.
* the cross-rank gather of per-image predictions (`detectron2.utils.comm.gather` of pickled dicts over a gloo side
  group, :129-131) is `torch.distributed.gather_object` on the default group's CPU/gloo companion; the predictions are
  tiny (one id map per image), so this is not the latents all-gather of the sampling path and needs no RCCL.

Ground truth comes in the COCO panoptic format: a JSON with `annotations[*].segments_info` and one PNG per image whose
RGB encodes the segment id (id = R + 256 G + 256^2 B).  As in the reference every segment's category is rewritten to 1
(:59-72).  For data without a JSON (the reference's data/examples) `gt_from_png` derives the annotation from the PNG.
"""
import io
import json
import os
from collections import OrderedDict
from typing import Dict, List, Optional, Union

import numpy as np
import torch

VOID = 0
OFFSET = 256 * 256 * 256


def rgb2id(color: np.ndarray) -> np.ndarray:
    """COCO panoptic PNG colour -> segment id (panopticapi/utils.py rgb2id)."""
    color = np.asarray(color)
    if color.ndim == 3:
        c = color.astype(np.int64)
        return c[:, :, 0] + 256 * c[:, :, 1] + 256 * 256 * c[:, :, 2]
    return int(color[0] + 256 * color[1] + 256 * 256 * color[2])


def id2rgb(id_map: np.ndarray) -> np.ndarray:
    """segment id map [H,W] -> uint8 RGB [H,W,3] (panopticapi/utils.py id2rgb)."""
    ids = np.asarray(id_map).astype(np.int64)
    rgb = np.zeros(ids.shape + (3,), dtype=np.uint8)
    for i in range(3):
        rgb[..., i] = ids % 256
        ids = ids // 256
    return rgb


def gt_from_png(id_map: np.ndarray, image_id, file_name: str) -> dict:
    """Class-agnostic ground-truth annotation of one image from its id map (VOID = 0 carries no segment)."""
    ids, cnt = np.unique(id_map, return_counts=True)
    segs = [{"id": int(i), "category_id": 1, "iscrowd": 0, "area": int(c)} for i, c in zip(ids, cnt) if i != VOID]
    return {"image_id": image_id, "file_name": file_name, "segments_info": segs}


class PQStatCat(object):
    def __init__(self):
        self.iou, self.tp, self.fp, self.fn = 0.0, 0, 0, 0


def pq_compute_annotations(pairs, categories: Dict[int, dict]):
    """pairs: iterable of (gt_ann, gt_ids [H,W], pred_ann, pred_ids [H,W]).  Returns {category_id: PQStatCat}."""
    stat = {c: PQStatCat() for c in categories}
    for gt_ann, pan_gt, pred_ann, pan_pred in pairs:
        pan_gt = np.asarray(pan_gt).astype(np.uint64)
        pan_pred = np.asarray(pan_pred).astype(np.uint64)
        if pan_gt.shape != pan_pred.shape:
            raise ValueError(f"image {gt_ann['image_id']}: ground truth {pan_gt.shape} vs prediction {pan_pred.shape}")
        gt_segms = {el["id"]: dict(el) for el in gt_ann["segments_info"]}
        pred_segms = {el["id"]: dict(el) for el in pred_ann["segments_info"]}
        # areas of the predicted segments + sanity checks (every painted id is declared and vice versa)
        declared = set(pred_segms)
        labels, cnt = np.unique(pan_pred, return_counts=True)
        for label, c in zip(labels.tolist(), cnt.tolist()):
            if label not in pred_segms:
                if label == VOID:
                    continue
                raise KeyError(f"image {gt_ann['image_id']}: segment {label} is painted but not in segments_info")
            pred_segms[label]["area"] = c
            declared.discard(label)
            if pred_segms[label]["category_id"] not in categories:
                raise KeyError(f"image {gt_ann['image_id']}: segment {label} has unknown category")
        if declared:
            raise KeyError(f"image {gt_ann['image_id']}: segments {sorted(declared)} are in segments_info but not painted")
        for label, c in zip(*np.unique(pan_gt, return_counts=True)):
            if int(label) in gt_segms:
                gt_segms[int(label)].setdefault("area", int(c))
        # confusion counts
        pair_ids, inter = np.unique(pan_gt * np.uint64(OFFSET) + pan_pred, return_counts=True)
        gt_pred = {(int(p // OFFSET), int(p % OFFSET)): int(n) for p, n in zip(pair_ids.tolist(), inter.tolist())}
        gt_matched, pred_matched = set(), set()
        for (g, p), n in gt_pred.items():
            if g not in gt_segms or p not in pred_segms:
                continue
            if gt_segms[g].get("iscrowd", 0) == 1 or gt_segms[g]["category_id"] != pred_segms[p]["category_id"]:
                continue
            union = pred_segms[p]["area"] + gt_segms[g]["area"] - n - gt_pred.get((VOID, p), 0)
            iou = n / union
            if iou > 0.5:
                st = stat[gt_segms[g]["category_id"]]
                st.tp += 1
                st.iou += iou
                gt_matched.add(g)
                pred_matched.add(p)
        crowd_of_cat = {}
        for g, info in gt_segms.items():
            if g in gt_matched:
                continue
            if info.get("iscrowd", 0) == 1:
                crowd_of_cat[info["category_id"]] = g
                continue
            stat[info["category_id"]].fn += 1
        for p, info in pred_segms.items():
            if p in pred_matched:
                continue
            ign = gt_pred.get((VOID, p), 0)
            if info["category_id"] in crowd_of_cat:
                ign += gt_pred.get((crowd_of_cat[info["category_id"]], p), 0)
            if ign / info["area"] > 0.5:
                continue
            stat[info["category_id"]].fp += 1
    return stat


def pq_average(stat: Dict[int, PQStatCat], categories: Dict[int, dict], isthing: Optional[bool]):
    pq = sq = rq = 0.0
    n = 0
    per_class = {}
    for cid, info in categories.items():
        if isthing is not None and (info.get("isthing", 1) == 1) != isthing:
            continue
        st = stat[cid]
        if st.tp + st.fp + st.fn == 0:
            per_class[cid] = {"pq": 0.0, "sq": 0.0, "rq": 0.0}
            continue
        n += 1
        c_pq = st.iou / (st.tp + 0.5 * st.fp + 0.5 * st.fn)
        c_sq = st.iou / st.tp if st.tp else 0.0
        c_rq = st.tp / (st.tp + 0.5 * st.fp + 0.5 * st.fn)
        per_class[cid] = {"pq": c_pq, "sq": c_sq, "rq": c_rq}
        pq += c_pq
        sq += c_sq
        rq += c_rq
    if n == 0:
        return {"pq": 0.0, "sq": 0.0, "rq": 0.0, "n": 0}, per_class
    return {"pq": pq / n, "sq": sq / n, "rq": rq / n, "n": n}, per_class


def pq_compute(gt_json: dict, pred_json: dict, gt_maps: Dict, pred_maps: Dict):
    """`pq_compute` (:188-230) on in-memory data: gt_json / pred_json are COCO-panoptic dicts, gt_maps / pred_maps map
    image_id -> id map [H,W] (what the reference reads back from the PNG folders)."""
    categories = {el["id"]: el for el in gt_json["categories"]}
    preds = {el["image_id"]: el for el in pred_json["annotations"]}
    pairs = []
    for gt_ann in gt_json["annotations"]:
        iid = gt_ann["image_id"]
        if iid not in preds:
            continue                                                     # (:216-218: images without a prediction are skipped)
        pairs.append((gt_ann, gt_maps[iid], preds[iid], pred_maps[iid]))
    stat = pq_compute_annotations(pairs, categories)
    results = {}
    for name, isthing in (("All", None), ("Things", True)):
        results[name], per_class = pq_average(stat, categories, isthing)
        if name == "All":
            results["per_class"] = per_class
    return results, stat, len(preds)


def get_table(pq_res: dict) -> str:
    rows = ["|        |   PQ   |   SQ   |   RQ   | #categories |", "|:------:|:------:|:------:|:------:|:-----------:|"]
    for name in ("All", "Things", "Stuff"):
        if name in pq_res:
            r = pq_res[name]
            rows.append(f"| {name:^6} | {100 * r['pq']:6.3f} | {100 * r['sq']:6.3f} | {100 * r['rq']:6.3f} | {r['n']:^11} |")
    return "\n".join(rows)


class PanopticEvaluatorAgnostic(object):
    """reset() / process(file_names, image_ids, outputs) / evaluate() like the reference class.

    `meta` needs `panoptic_json` (COCO panoptic annotations) and `panoptic_root` (folder of ground-truth PNGs), as in
    the reference's dataset meta data; alternatively pass `gt_maps` / `gt_annotations` in memory (image_id -> id map /
    annotation dict), e.g. from `gt_from_png`.  Every ground-truth category becomes 1 ('object').
    """

    def __init__(self, output_dir: Optional[str] = None, meta: Optional[Dict] = None, gt_maps: Optional[Dict] = None,
                 gt_annotations: Optional[List[dict]] = None, group=None):
        self._metadata = meta or {}
        self.class_agnostic = True
        self._output_dir = output_dir
        self._group = group
        self._gt_maps = gt_maps
        self._gt_json = None
        if gt_annotations is not None:
            anns = [dict(a, segments_info=[dict(s, category_id=1) for s in a["segments_info"]]) for a in gt_annotations]
            self._gt_json = {"annotations": anns,
                             "categories": [{"id": 1, "name": "object", "supercategory": "object", "isthing": 1}]}
        if output_dir:
            os.makedirs(output_dir, exist_ok=True)
        self.reset()

    def reset(self):
        self._predictions = []

    def process(self, file_names: List[str], image_ids: List, outputs: List[Dict[str, Union[torch.Tensor, np.ndarray, tuple]]]):
        from PIL import Image
        for file_name, image_id, output in zip(file_names, image_ids, outputs):
            panoptic_img, segments_info = output["panoptic_seg"]
            if isinstance(panoptic_img, torch.Tensor):
                panoptic_img = panoptic_img.cpu().numpy()
            segs = [dict(s, category_id=1, isthing=True) for s in segments_info]
            png_name = os.path.splitext(os.path.basename(file_name))[0] + ".png"
            with io.BytesIO() as out:
                Image.fromarray(id2rgb(panoptic_img)).save(out, format="PNG")
                self._predictions.append({"image_id": image_id, "file_name": png_name, "png_string": out.getvalue(),
                                          "segments_info": segs})

    def _gather(self):
        """All ranks' predictions on rank 0 (detectron2 comm.gather, :129-131)."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(self._group) == 1:
            return [self._predictions], True
        rank, world = dist.get_rank(self._group), dist.get_world_size(self._group)
        dist.barrier(group=self._group)                                   # comm.synchronize()
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(self._predictions, gathered, dst=0, group=self._group)
        return gathered, rank == 0

    def _load_gt(self):
        from PIL import Image
        if self._gt_json is None:
            with open(self._metadata["panoptic_json"], "r") as f:
                gt = json.load(f)
            for anno in gt["annotations"]:                                # (:65-68) class agnostic ground truth
                for seg in anno["segments_info"]:
                    seg["category_id"] = 1
            gt["categories"] = [{"id": 1, "name": "object", "supercategory": "object", "isthing": 1}]
            self._gt_json = gt
        if self._gt_maps is None:
            root = self._metadata["panoptic_root"]
            self._gt_maps = LazyPngMaps(root, {a["image_id"]: a["file_name"] for a in self._gt_json["annotations"]})
        return self._gt_json, self._gt_maps

    def evaluate(self):
        from PIL import Image
        gathered, is_main = self._gather()
        if not is_main:
            return None
        predictions = [p for part in gathered for p in part]
        gt_json, gt_maps = self._load_gt()
        pred_maps = {}
        for p in predictions:
            pred_maps[p["image_id"]] = rgb2id(np.asarray(Image.open(io.BytesIO(p["png_string"])).convert("RGB")))
            if self._output_dir:
                with open(os.path.join(self._output_dir, p["file_name"]), "wb") as f:
                    f.write(p["png_string"])
        pred_json = {"annotations": [{k: v for k, v in p.items() if k != "png_string"} for p in predictions],
                     "categories": gt_json["categories"]}
        if self._output_dir:
            with open(os.path.join(self._output_dir, "predictions.json"), "w") as f:
                json.dump(pred_json, f)
        pq_res, stat, num_preds = pq_compute(gt_json, pred_json, gt_maps, pred_maps)
        res = {"PQ": 100 * pq_res["All"]["pq"], "SQ": 100 * pq_res["All"]["sq"], "RQ": 100 * pq_res["All"]["rq"],
               "PQ_th": 100 * pq_res["Things"]["pq"], "SQ_th": 100 * pq_res["Things"]["sq"], "RQ_th": 100 * pq_res["Things"]["rq"]}
        st = stat[1]
        res["precision"] = 100 * st.tp / (st.tp + st.fp + 1e-8)
        res["recall"] = 100 * st.tp / (st.tp + st.fn + 1e-8)
        res["num_predictions"] = num_preds
        self.table = get_table(pq_res)
        return OrderedDict({"panoptic_seg": res})


class LazyPngMaps(object):
    """image_id -> id map, read from a folder of COCO panoptic PNGs on access."""

    def __init__(self, root: str, names: Dict):
        self.root, self.names = root, names

    def __getitem__(self, image_id):
        from PIL import Image
        return rgb2id(np.asarray(Image.open(os.path.join(self.root, self.names[image_id])).convert("RGB")))
