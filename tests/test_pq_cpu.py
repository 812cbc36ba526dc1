"""CPU suite: class-agnostic panoptic-quality evaluation - known answers for the restated
panopticapi matching rules, the COCO-panoptic id <-> colour codec, the evaluator's process/evaluate round trip through
PNG bytes, and the cross-rank gather (world size 2, gloo)."""
import os
import sys

import numpy as np
import pytest
import torch

from ldmseg_amd.evaluations import PanopticEvaluatorAgnostic, id2rgb, rgb2id, pq_compute
from ldmseg_amd.evaluations.panoptic_evaluation_agnostic import gt_from_png, pq_compute_annotations


def ann(id_map, iid=0, crowd=()):
    a = gt_from_png(id_map, iid, f"{iid}.png")
    for s in a["segments_info"]:
        s["iscrowd"] = int(s["id"] in crowd)
    return a


def pred_out(id_map):
    ids = [int(i) for i in np.unique(id_map) if i != 0]
    return {"panoptic_seg": (id_map, [{"id": i, "category_id": 1, "isthing": True} for i in ids])}


CATS = {1: {"id": 1, "isthing": 1}}


def test_id_rgb_codec_roundtrip():
    ids = np.array([[0, 1, 255, 256], [65535, 65536, 16777215, 70000]])
    rgb = id2rgb(ids)
    assert rgb.dtype == np.uint8 and rgb.shape == (2, 4, 3)
    assert rgb[0, 2].tolist() == [255, 0, 0] and rgb[0, 3].tolist() == [0, 1, 0] and rgb[1, 1].tolist() == [0, 0, 1]
    assert np.array_equal(rgb2id(rgb), ids)


def test_perfect_prediction_is_pq_one():
    gt = np.zeros((8, 8), np.int64); gt[:4] = 5; gt[4:, :4] = 9          # two segments + void
    st = pq_compute_annotations([(ann(gt), gt, {"image_id": 0, "segments_info": [{"id": 5, "category_id": 1}, {"id": 9, "category_id": 1}]}, gt)], CATS)
    assert (st[1].tp, st[1].fp, st[1].fn) == (2, 0, 0) and abs(st[1].iou - 2.0) < 1e-12


def test_matching_rules_known_answers():
    # ground truth: A = rows 0-3 (32 px), B = rows 4-7 left half (16 px), void = rows 4-7 right half
    gt = np.zeros((8, 8), np.int64); gt[:4] = 1; gt[4:, :4] = 2
    # prediction: P1 = rows 0-2 (24 px, IoU with A = 24/32 = 0.75 -> TP); P2 = row 3 (8 px, inside A: IoU 8/32 -> FP);
    # P3 = rows 4-7 cols 2-7 (24 px: 8 on B, 16 on void) -> IoU with B = 8 / (24 + 16 - 8 - 16) = 0.5 -> NOT > 0.5, and
    # 16/24 > 0.5 of it lies on void -> ignored, not a false positive; B stays a false negative
    pr = np.zeros((8, 8), np.int64); pr[:3] = 11; pr[3] = 12; pr[4:, 2:] = 13
    pa = {"image_id": 0, "segments_info": [{"id": i, "category_id": 1} for i in (11, 12, 13)]}
    st = pq_compute_annotations([(ann(gt), gt, pa, pr)], CATS)[1]
    assert (st.tp, st.fp, st.fn) == (1, 1, 1) and abs(st.iou - 0.75) < 1e-12
    # crowd ground truth: never a false negative, and predictions mostly on it are ignored
    st = pq_compute_annotations([(ann(gt, crowd=(2,)), gt, pa, pr)], CATS)[1]
    assert (st.tp, st.fp, st.fn) == (1, 1, 0)
    # a painted id missing from segments_info is an error, as is a declared id that is not painted
    with pytest.raises(KeyError):
        pq_compute_annotations([(ann(gt), gt, {"image_id": 0, "segments_info": [{"id": 11, "category_id": 1}]}, pr)], CATS)
    with pytest.raises(KeyError):
        pq_compute_annotations([(ann(gt), gt, {"image_id": 0, "segments_info": pa["segments_info"] + [{"id": 99, "category_id": 1}]}, pr)], CATS)


def test_evaluator_roundtrip_and_metric():
    g = np.random.RandomState(0)
    gts, preds = {}, {}
    for iid in range(3):
        gt = np.zeros((20, 30), np.int64)
        gt[:10, :15] = 100 + iid; gt[:10, 15:] = 200 + iid; gt[10:, :] = 70000 + iid      # ids beyond 16 bits too
        gts[iid] = gt
        pr = np.zeros_like(gt)
        pr[:10, :15] = 1; pr[:8, 15:] = 2; pr[12:, :] = 3                                  # IoUs: 1.0, 0.8, 0.8
        preds[iid] = pr
    ev = PanopticEvaluatorAgnostic(gt_maps=gts, gt_annotations=[ann(gts[i], i) for i in range(3)])
    ev.process([f"/x/{i}.jpg" for i in range(3)], list(range(3)), [pred_out(preds[i]) for i in range(3)])
    res = ev.evaluate()["panoptic_seg"]
    assert abs(res["SQ"] - 100 * (1.0 + 0.8 + 0.8) / 3) < 1e-9 and abs(res["RQ"] - 100.0) < 1e-9
    assert abs(res["PQ"] - res["SQ"]) < 1e-9 and res["num_predictions"] == 3
    assert "PQ" in ev.table
    # images without a prediction are skipped (reference :216-218), predictions come back through PNG bytes
    ev.reset()
    ev.process(["0.jpg"], [0], [pred_out(torch.from_numpy(preds[0]))])
    assert abs(ev.evaluate()["panoptic_seg"]["RQ"] - 100.0) < 1e-9


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gts = {i: np.full((6, 6), 10 + i, np.int64) for i in range(4)}
    ev = PanopticEvaluatorAgnostic(gt_maps=gts, gt_annotations=[ann(gts[i], i) for i in range(4)])
    mine = [i for i in range(4) if i % world == rank]                     # images sharded over ranks
    preds = []
    for i in mine:
        p = np.full((6, 6), 1, np.int64)
        if i == 3:
            p[:, :4] = 2                                                  # image 3: split in two -> one TP (IoU 2/3) + one FP
        preds.append(pred_out(p))
    ev.process([f"{i}.jpg" for i in mine], mine, preds)
    out = ev.evaluate()
    q.put((rank, None if out is None else dict(out["panoptic_seg"])))
    dist.destroy_process_group()


def test_cross_rank_gather_gloo_world2():
    import torch.multiprocessing as mp
    import socket
    ctx = mp.get_context("spawn")
    got = None
    for attempt in range(3):        # (a rendezvous port can be taken between the probe and the store's bind: new port, once or twice)
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        q = ctx.Queue()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
        for p in procs:
            p.start()
        try:
            got = dict(q.get(timeout=120) for _ in range(2))
        except Exception:
            got = None
        for p in procs:
            p.join(60)
            if p.is_alive():
                p.kill()
                p.join(10)
        if got is not None:
            break
    assert got is not None, "two gloo ranks did not complete in three attempts"
    assert got[1] is None                                                  # only rank 0 evaluates
    r = got[0]
    assert r["num_predictions"] == 4                                       # rank 1's images arrived
    # TP = 4 (IoUs 1, 1, 1, 2/3), FP = 1, FN = 0
    assert abs(r["SQ"] - 100 * (3 + 2 / 3) / 4) < 1e-9 and abs(r["RQ"] - 100 * 4 / 4.5) < 1e-9


def test_eval_entry_batches_and_compute_pq_wiring(tmp_path, sched_kw):
    """Host logic of the evaluation entry without a GPU: tools/main_ldm_eval.py::batches (PIL resize, meta, padding masks)
    feeding TrainerDiffusion.compute_pq -> PanopticEvaluatorAgnostic with the model call replaced by a stand-in that
    returns the ground truth - file names / image ids / sizes must arrive intact, and the `max_iter` quirk of the
    reference loop (trainers_ldm_cond.py:1332: the check sits after the batch and uses `>`) must hold."""
    import importlib.util
    from PIL import Image
    from conftest import ROOT
    from ldmseg_amd.schedulers import DDIMNoiseScheduler
    from ldmseg_amd.trainers import TrainerDiffusion
    spec = importlib.util.spec_from_file_location("main_ldm_eval", os.path.join(ROOT, "tools", "main_ldm_eval.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sizes = [(40, 60), (64, 48), (50, 50), (33, 70), (20, 20)]
    g = np.random.RandomState(0)
    gts = {}
    for i, (h, w) in enumerate(sizes):
        Image.fromarray(g.randint(0, 255, (h, w, 3)).astype(np.uint8)).save(tmp_path / f"im{i}.png")
        gt = np.zeros((h, w), np.int64); gt[: h // 2] = 11 + i; gt[h // 2:] = 500 + i
        gts[f"im{i}"] = gt
    files = sorted(str(p) for p in tmp_path.glob("*.png"))
    got = list(mod.batches(files, 32, 2, None))
    assert [b["image"].shape for b in got] == [(2, 3, 32, 32), (2, 3, 32, 32), (1, 3, 32, 32)]
    assert all(b["mask"].dtype == torch.bool and bool(b["mask"].all()) and b["mask"].shape == (len(b["meta"]), 32, 32) for b in got)
    assert [m["im_size"] for b in got for m in b["meta"]] == sizes and got[1]["meta"][0]["image_id"] == "im2"
    assert float(got[0]["image"].min()) >= 0.0 and float(got[0]["image"].max()) <= 1.0
    assert TrainerDiffusion.padding_boxes(got[0]["mask"]).tolist() == [[0, 0, 32, 32], [0, 0, 32, 32]]
    m = torch.zeros(2, 8, 10, dtype=torch.bool); m[0, 2:5, 3:9] = True; m[1, 7, 0] = True
    assert TrainerDiffusion.padding_boxes(m).tolist() == [[2, 3, 3, 6], [7, 0, 1, 1]]
    assert tuple(TrainerDiffusion.crop_padding(torch.zeros(4, 8, 10), m[0]).shape) == (4, 3, 6)

    tr = TrainerDiffusion.__new__(TrainerDiffusion)
    tr.noise_scheduler = DDIMNoiseScheduler(**sched_kw)
    tr.device = torch.device("cpu")
    seen = []

    def fake_predict(rgb, im_sizes, masks, *a, **kw):
        seen.append((tuple(rgb.shape), [tuple(s) for s in im_sizes], None if masks is None else tuple(masks.shape)))
        k = len(seen) - 1
        return [pred_out(gts[f"im{2 * k + j}"]) for j in range(rgb.shape[0])]
    tr.predict_panoptic = fake_predict
    ev = PanopticEvaluatorAgnostic(gt_maps=gts, gt_annotations=[ann(gts[f"im{i}"], f"im{i}") for i in range(5)])
    res = tr.compute_pq(mod.batches(files, 32, 2, None), ev, num_inference_steps=5, seed=1, max_iter=0)["panoptic_seg"]
    assert len(seen) == 2 and seen[0] == ((2, 3, 32, 32), sizes[:2], (2, 32, 32))      # batches 0 and 1, then the break
    assert res["num_predictions"] == 4 and abs(res["PQ"] - 100.0) < 1e-9
    assert tr.noise_scheduler.timesteps.tolist() == [999, 799, 599, 399, 199]
    seen.clear()
    res = tr.compute_pq(mod.batches(files, 32, 2, None), ev, num_inference_steps=5, seed=1)["panoptic_seg"]
    assert res["num_predictions"] == 5
