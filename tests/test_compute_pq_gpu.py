"""The evaluation loop on the MI355X: `TrainerDiffusion.compute_pq` (the reference's trainers_ldm_cond.py:1181-1346) with
`PanopticEvaluatorAgnostic` over synthetic images written to disk - padding masks, original sizes different from the
network size, the `max_iter` quirk (:1332), ground truth from PNGs - checked against (a) the oracle chain (image-VAE
encode -> DDIM sampling -> seg-VAE decode -> interpolate -> crop -> interpolate -> post-processing on the CPU) and (b) a
brute-force restatement of the panoptic-quality metric on the product's own predictions; plus the eval entry
tools/main_ldm_eval.py as a subprocess."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import ROOT
from oracle import ddim as o_ddim, postprocess as o_post, sample as o_sample, unet as o_unet
from oracle import vae as o_vae, vae_image as o_img

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
S, LAT, STEPS = 128, 16, 3
# random weights give flat class distributions: thresholds low enough that segments survive, high enough that every branch runs
POST = dict(threshold_output=True, threshold_mode="max", mask_th=0.01, count_th=24, overlap_th=0.002, ignore_label=3)


def smooth_image(h, w, seed):
    g = torch.Generator().manual_seed(seed)
    low = torch.rand(1, 3, h // 16 + 2, w // 16 + 2, generator=g)
    return F.interpolate(low, size=(h, w), mode="bicubic", align_corners=False).clamp(0, 1)[0]


def brute_force_pq(gt_maps, pred_maps):
    """PQ / SQ / RQ of class-agnostic maps (0 = void) straight from the definition (Kirillov et al.; the matching rule
    panopticapi implements): a pair matches when IoU > 0.5, the union not counting the prediction's pixels on void;
    unmatched predictions with more than half their area on void are ignored."""
    tp = fp = fn = 0
    iou_sum = 0.0
    for gt, pr in zip(gt_maps, pred_maps):
        gids = [g for g in np.unique(gt) if g != 0]
        pids = [p for p in np.unique(pr) if p != 0]
        gm, pm = set(), set()
        for g in gids:
            G = gt == g
            for p in pids:
                P = pr == p
                inter = int((G & P).sum())
                if inter == 0:
                    continue
                union = int(G.sum()) + int(P.sum()) - inter - int((P & (gt == 0)).sum())
                if inter / union > 0.5:
                    tp += 1
                    iou_sum += inter / union
                    gm.add(g)
                    pm.add(p)
        fn += len(gids) - len(gm)
        for p in pids:
            if p in pm:
                continue
            P = pr == p
            if (P & (gt == 0)).sum() / P.sum() > 0.5:
                continue
            fp += 1
    den = tp + 0.5 * fp + 0.5 * fn
    return (100 * iou_sum / den if den else 0.0, 100 * iou_sum / tp if tp else 0.0, 100 * tp / den if den else 0.0, tp, fp, fn)


@pytest.fixture(scope="module")
def trainer(unet_sd, vae_sd, sched_kw):
    from ldmseg_amd import weights
    from ldmseg_amd.models import UNet, GeneralVAESeg, GeneralVAEImage
    from ldmseg_amd.schedulers import DDIMNoiseScheduler
    from ldmseg_amd.trainers import TrainerDiffusion
    isd = weights.generate(weights.vae_image_schema(), seed=11, norm_keys=weights.VAE_IMAGE_NORM_KEYS)
    unet = UNet(unet_sd, in_channels=12, device=DEV, compute_dtype="fp32")
    vae = GeneralVAESeg(vae_sd, scaling_factor=0.2, device=DEV, compute_dtype="fp32")
    enc = GeneralVAEImage(isd, scaling_factor=0.18215, device=DEV, compute_dtype="fp32")
    return TrainerDiffusion(vae, unet, DDIMNoiseScheduler(**sched_kw), vae_image=enc, latent_size=LAT), isd


def oracle_predictions(imgs, boxes, sizes, isd, unet_sd, vae_sd, sched_kw):
    """One BATCH through the oracle chain (the initial noise is drawn per batch, :1088-1091: image j of a batch gets slice j)."""
    with torch.no_grad():
        rgb_lat = o_img.encode_mode(isd, torch.stack(imgs), 0.18215)
        so = o_ddim.OracleDDIM(**sched_kw)
        so.set_timesteps_inference(STEPS)
        lat = o_sample.sample(lambda inp, t: o_unet.unet_forward(unet_sd, inp, t), so, rgb_lat, seed=42)
        logits = o_sample.decode_latents(lambda z: o_vae.decode(vae_sd, z), lat, 0.2)
        logits = F.interpolate(logits, size=(S, S), mode="bilinear", align_corners=False)          # :1252-1257
    out = []
    for j, ((y0, x0, ch, cw), hw) in enumerate(zip(boxes, sizes)):
        m = logits[j][:, y0:y0 + ch, x0:x0 + cw]                                                   # :1263
        final = F.interpolate(m[None], size=hw, mode="bilinear", align_corners=False)[0]           # :1266-1271
        pan, info, raw, _ = o_post.panoptic_postprocess(final, **POST)
        out.append((pan, info, final))
    return out


def test_compute_pq_loop_against_oracle_and_brute_force(trainer, unet_sd, vae_sd, sched_kw, tmp_path):
    from PIL import Image
    from ldmseg_amd.evaluations import PanopticEvaluatorAgnostic, id2rgb, rgb2id
    from ldmseg_amd.evaluations.panoptic_evaluation_agnostic import gt_from_png
    torch.set_num_threads(32)
    tr, isd = trainer
    # five images in three batches (2, 2, 1); padding boxes (y0, x0, height, width) in the S x S network grid, original sizes
    specs = [((0, 0, S, S), (96, 128)), ((0, 0, 100, S), (75, 96)), ((9, 4, 111, 86), (150, 117)), ((0, 0, S, 90), (128, 90)),
             ((0, 0, S, S), (64, 64))]
    imgs = [smooth_image(S, S, 20 + i) for i in range(len(specs))]
    masks = torch.zeros(len(specs), S, S, dtype=torch.bool)
    for i, (bx, _) in enumerate(specs):
        masks[i, bx[0]:bx[0] + bx[2], bx[1]:bx[1] + bx[3]] = True
    # ground truth: the oracle's own prediction for image 0 (so that true positives exist), coarse random blocks elsewhere
    gt_dir = tmp_path / "panoptic"
    gt_dir.mkdir()
    g = np.random.RandomState(3)
    gt_maps, gt_anns, ora = {}, [], {}
    for lo, hi in ((0, 2), (2, 4)):                       # only batches 0 and 1 are evaluated (max_iter quirk below)
        res_b = oracle_predictions(imgs[lo:hi], [sp[0] for sp in specs[lo:hi]], [sp[1] for sp in specs[lo:hi]], isd, unet_sd,
                                   vae_sd, sched_kw)
        for j, r in enumerate(res_b):
            ora[lo + j] = r
    for i, (bx, hw) in enumerate(specs):
        if i == 0:
            gt = ora[0][0].astype(np.int64) * 1000                                # ids beyond one byte
        else:
            gt = np.kron(g.randint(0, 4, (4, 4)), np.ones((hw[0] // 4 + 1, hw[1] // 4 + 1), np.int64))[:hw[0], :hw[1]] * 300
        Image.fromarray(id2rgb(gt)).save(gt_dir / f"img{i}.png")
        gt_maps[f"img{i}"] = rgb2id(np.asarray(Image.open(gt_dir / f"img{i}.png").convert("RGB")))
        assert np.array_equal(gt_maps[f"img{i}"], gt)
        gt_anns.append(gt_from_png(gt_maps[f"img{i}"], f"img{i}", f"img{i}.png"))

    def loader():
        for lo, hi in ((0, 2), (2, 4), (4, 5)):
            yield {"image": torch.stack(imgs[lo:hi]), "mask": masks[lo:hi],
                   "meta": [{"image_file": f"/data/img{i}.jpg", "image_id": f"img{i}", "im_size": specs[i][1]} for i in range(lo, hi)]}

    out_dir = tmp_path / "pred"
    ev = PanopticEvaluatorAgnostic(output_dir=str(out_dir), gt_maps=gt_maps, gt_annotations=gt_anns)
    kw = {k: v for k, v in POST.items() if k not in ("threshold_output", "threshold_mode")}
    # `if max_iter is not None and batch_idx > max_iter: break` (:1332) sits AFTER the batch: max_iter = 0 evaluates batches 0 and 1
    res = tr.compute_pq(loader(), ev, num_inference_steps=STEPS, seed=42, threshold_output=True, threshold_mode="max",
                        max_iter=0, **kw)["panoptic_seg"]
    assert res["num_predictions"] == 4
    assert sorted(os.listdir(out_dir)) == ["img0.png", "img1.png", "img2.png", "img3.png", "predictions.json"]
    preds = {}
    for i in range(4):
        preds[i] = rgb2id(np.asarray(Image.open(out_dir / f"img{i}.png").convert("RGB")))
        assert preds[i].shape == specs[i][1]                       # predictions live at the ORIGINAL size
    # (a) against the oracle chain: the same map except where the decision is within rounding of a tie
    for i in range(4):
        pan_ref, info_ref, final = ora[i]
        top2 = final.topk(2, dim=0)[0]
        clear = ((top2[0] - top2[1]) > 1e-2 * float(final.abs().max())).numpy()
        agree = preds[i] == pan_ref
        assert clear.mean() > 0.4 and agree[clear].mean() > 0.99 and agree.mean() > 0.95, (i, clear.mean(), agree.mean())
    # (b) the metric: brute force on the product's own predictions
    pq, sq, rq, tp, fp, fn = brute_force_pq([gt_maps[f"img{i}"] for i in range(4)], [preds[i] for i in range(4)])
    assert len(ora[0][1]) >= 1, "the oracle kept no segment on image 0: the thresholds of this test need retuning"
    assert tp >= 1                                                  # image 0's ground truth is (nearly) its prediction
    assert abs(res["PQ"] - pq) < 1e-9 and abs(res["SQ"] - sq) < 1e-9 and abs(res["RQ"] - rq) < 1e-9
    assert abs(res["precision"] - 100 * tp / (tp + fp + 1e-8)) < 1e-9 and abs(res["recall"] - 100 * tp / (tp + fn + 1e-8)) < 1e-9
    # the fused tail and the step-by-step path (logits materialised, torch interpolation) give the same maps
    r_f = tr.predict_panoptic(torch.stack(imgs[2:4]).to(DEV), [specs[2][1], specs[3][1]], masks[2:4].to(DEV), STEPS, seed=42, **POST)
    r_u = tr.predict_panoptic(torch.stack(imgs[2:4]).to(DEV), [specs[2][1], specs[3][1]], masks[2:4].to(DEV), STEPS, seed=42,
                              fused=False, **POST)
    for a, b in zip(r_f, r_u):
        assert float((a["panoptic_seg"][0] == b["panoptic_seg"][0]).float().mean()) > 0.999
    assert np.array_equal(r_f[0]["panoptic_seg"][0].cpu().numpy(), preds[2])        # and compute_pq ran the fused one


def test_main_ldm_eval_entry_subprocess(tmp_path):
    """tools/main_ldm_eval.py end to end (random weights, two tiny images, ground-truth PNGs): exit code, table, PNGs."""
    from PIL import Image
    from ldmseg_amd.evaluations import id2rgb
    img_dir, gt_dir, out_dir = tmp_path / "rgb", tmp_path / "pan", tmp_path / "out"
    img_dir.mkdir(); gt_dir.mkdir()
    for i, (h, w) in enumerate([(120, 160), (144, 100)]):
        a = (smooth_image(h, w, 40 + i).permute(1, 2, 0).numpy() * 255).astype(np.uint8)
        Image.fromarray(a).save(img_dir / f"{i:03d}.jpg")
        gt = np.zeros((h, w), np.int64); gt[:h // 2] = 7 + i; gt[h // 2:, :w // 2] = 3000
        Image.fromarray(id2rgb(gt)).save(gt_dir / f"{i:03d}.png")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "latent-diffusion-segmentation_amd")]))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "main_ldm_eval.py"), "--images", str(img_dir), "--panoptic", str(gt_dir),
                        "--size", "128", "--steps", "2", "--batch", "2", "--dtype", "bf16", "--count-th", "32", "--mask-th", "0.02",
                        "--out", str(out_dir)], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "PQ" in r.stdout and "num_predictions" in r.stdout
    assert sorted(os.listdir(out_dir)) == ["000.png", "001.png", "predictions.json"]
    assert np.asarray(Image.open(out_dir / "001.png")).shape[:2] == (144, 100)


def test_real_coco_pairs_through_bit_codec_and_eval_entry(golden, tmp_path):
    """The reference's own example pairs on the GPU path: (1) the bit codec kernels on the REAL remapped
    panoptic maps - bit-exact against what the reference's COCO.encode_bitmap / decode_bitmap produced (fixture), through
    the affine 2x-1 the sampler applies; (2) tools/main_ldm_eval.py on the two non-square images (480 x 640 and 640 x 427 ->
    PIL resize to 512 x 512 -> image VAE -> DDIM -> decode -> back to the original sizes) with the real panoptic PNGs as
    ground truth: the prediction PNGs must come back at each image's own size and the PQ table must be produced."""
    import io
    from PIL import Image
    from ldmseg_amd.data.bitcodec import decode_bitmap, encode_bitmap
    g = golden("real_coco.npz")
    for k in range(2):
        remapped = torch.from_numpy(g[f"remapped_{k}"].astype(np.int64)).cuda()
        bits, ign = encode_bitmap(remapped, n=7, fill_value=0.5, ignore_label=0)
        assert torch.equal(bits.cpu(), torch.from_numpy(g[f"bits_{k}"].astype(np.float32)))
        assert torch.equal(ign.cpu(), torch.from_numpy(g[f"ignore_{k}"]))
        net_in, _ = encode_bitmap(remapped, n=7, fill_value=0.5, ignore_label=0, affine=(2.0, -1.0))
        assert torch.equal(net_in, 2.0 * bits - 1.0)
        assert torch.equal(decode_bitmap(net_in).cpu(), torch.from_numpy(g[f"decoded_{k}"].astype(np.int64)))
    img_dir, gt_dir, out_dir = tmp_path / "rgb", tmp_path / "pan", tmp_path / "out"
    img_dir.mkdir(); gt_dir.mkdir()
    names = ["000000012280", "000000084752"]
    for k, n in enumerate(names):
        (img_dir / f"{n}.jpg").write_bytes(g[f"jpg_{k}"].tobytes())
        (gt_dir / f"{n}.png").write_bytes(g[f"png_{k}"].tobytes())
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "latent-diffusion-segmentation_amd")]))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "main_ldm_eval.py"), "--images", str(img_dir), "--panoptic", str(gt_dir),
                        "--size", "512", "--steps", "3", "--batch", "2", "--dtype", "bf16", "--out", str(out_dir)],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "PQ" in r.stdout and "num_predictions" in r.stdout
    assert sorted(os.listdir(out_dir)) == [names[0] + ".png", names[1] + ".png", "predictions.json"]
    for k, n in enumerate(names):
        w, h = Image.open(io.BytesIO(g[f"jpg_{k}"].tobytes())).size
        assert np.asarray(Image.open(out_dir / f"{n}.png")).shape[:2] == (h, w)
