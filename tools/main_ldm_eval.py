#!/usr/bin/env python
"""Evaluation entry of the MI355X build - the counterpart of `tools/main_ldm.py base.eval_only=True`
(/root/reference/tools/main_ldm.py:130-232; the reference script itself is hard-wired to CUDA, NCCL, hydra, diffusers and
detectron2 and cannot run here).

    python tools/main_ldm_eval.py --images DIR [--panoptic DIR] [--ldm ldmseg.pt --ae ae.pt --vae-image vae.pt]
                                  [--size 512] [--steps 50] [--batch 8] [--dtype bf16]

It assembles what main_worker assembles (image VAE encoder, seg-VAE, UNet with the 12-channel conv_in and cross-attention
removed, DDIM scheduler with base.yaml's noise_scheduler_kwargs), then runs `TrainerDiffusion.compute_pq`:
PIL resize (CropResize) -> 2x-1 -> image VAE -> 50-step DDIM sampling -> seg-VAE decode -> bilinear to the original
size -> argmax / thresholds / segment filtering -> class-agnostic PQ against the COCO panoptic PNGs.
One process per GPU under torch.distributed.run shards the images over ranks (DistributedSampler semantics,
trainers_ldm_cond.py:245); the evaluator gathers the predictions on rank 0.
Without checkpoints (none ship with this repo: the released ones are 3.3 GB downloads) deterministic random weights
of the right architecture are used - the pipeline then runs end to end but the PQ is meaningless.
"""
import argparse
import glob
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion-segmentation_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

NOISE_SCHEDULER_KWARGS = dict(prediction_type="epsilon", beta_schedule="scaled_linear", num_train_timesteps=1000,
                              beta_start=0.00085, beta_end=0.012, steps_offset=1, clip_sample=False, set_alpha_to_one=False,
                              thresholding=False, dynamic_thresholding_ratio=0.995, clip_sample_range=1.0,
                              sample_max_value=1.0, weight="none", max_snr=5.0)      # tools/configs/base/base.yaml:48-62


def build_trainer(args, device):
    from ldmseg_amd import checkpoint, weights
    from ldmseg_amd.models import UNet, GeneralVAESeg, GeneralVAEImage
    from ldmseg_amd.schedulers import DDIMNoiseScheduler
    from ldmseg_amd.trainers import TrainerDiffusion
    if args.ldm:
        ck = checkpoint.load_ldm_checkpoint(args.ldm)
        usd = ck["unet"]
        vsd = checkpoint.load_ae_checkpoint(args.ae) if args.ae else ck["vae_semseg"]
    else:
        usd = weights.generate(weights.unet_schema(12, False), seed=0)
        vsd = weights.generate(weights.vae_schema(), seed=7, norm_keys=weights.VAE_NORM_KEYS)
    if args.vae_image:
        isd = torch.load(args.vae_image, map_location="cpu", weights_only=True)
    else:
        isd = weights.generate(weights.vae_image_schema(), seed=11, norm_keys=weights.VAE_IMAGE_NORM_KEYS)
    unet = UNet(usd, in_channels=int(usd["conv_in.weight"].shape[1]), device=device, compute_dtype=args.dtype)
    vae = GeneralVAESeg(vsd, scaling_factor=args.scaling_factor, device=device, compute_dtype=args.dtype)
    enc = GeneralVAEImage(isd, scaling_factor=0.18215, device=device, compute_dtype=args.dtype)
    return TrainerDiffusion(vae, unet, DDIMNoiseScheduler(**NOISE_SCHEDULER_KWARGS), vae_image=enc,
                            latent_size=args.size // 8)


def batches(files, size, batch, panoptic_dir):
    from ldmseg_amd.data.transforms import load_rgb
    for i in range(0, len(files), batch):
        chunk = files[i:i + batch]
        imgs, meta = [], []
        for f in chunk:
            t, (h, w) = load_rgb(f, size)
            imgs.append(t)
            meta.append({"image_file": f, "image_id": os.path.splitext(os.path.basename(f))[0], "im_size": (h, w)})
        # CropResize with the crop_mode=None the reference hard-wires (pil_transforms.py:102) only resizes: nothing is padded,
        # the padding mask of every sample is all ones (compute_pq crops to its bounding box = the whole image)
        yield {"image": torch.stack(imgs), "mask": torch.ones(len(imgs), size, size, dtype=torch.bool), "meta": meta}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", required=True, help="folder of RGB images (e.g. data/examples/coco/rgb_images)")
    ap.add_argument("--panoptic", default=None, help="folder of COCO panoptic PNGs with the same stems (ground truth)")
    ap.add_argument("--ldm", default=None); ap.add_argument("--ae", default=None); ap.add_argument("--vae-image", default=None)
    ap.add_argument("--size", type=int, default=512); ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--batch", type=int, default=8); ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--scaling-factor", type=float, default=0.18215)      # tools/scripts/eval.sh:11
    ap.add_argument("--seed", type=int, default=42); ap.add_argument("--count-th", type=int, default=512)
    ap.add_argument("--mask-th", type=float, default=0.5); ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    trainer = build_trainer(args, device)
    files = sorted(glob.glob(os.path.join(args.images, "*.jpg")) + glob.glob(os.path.join(args.images, "*.png")))
    files = files[rank::world]                                              # images sharded over ranks
    from ldmseg_amd.evaluations import PanopticEvaluatorAgnostic, rgb2id
    from ldmseg_amd.evaluations.panoptic_evaluation_agnostic import gt_from_png
    gt_maps = gt_anns = None
    if args.panoptic:
        from PIL import Image
        gt_maps, gt_anns = {}, []
        for f in sorted(glob.glob(os.path.join(args.panoptic, "*.png"))):
            iid = os.path.splitext(os.path.basename(f))[0]
            gt_maps[iid] = rgb2id(np.asarray(Image.open(f).convert("RGB")))
            gt_anns.append(gt_from_png(gt_maps[iid], iid, os.path.basename(f)))
    if gt_anns is None:                                                     # no ground truth: predictions only
        from ldmseg_amd.data.transforms import load_rgb
        res = []
        for data in batches(files, args.size, args.batch, None):
            res += trainer.predict_panoptic(data["image"].to(device), [m["im_size"] for m in data["meta"]], None,
                                            args.steps, seed=args.seed, mask_th=args.mask_th, count_th=args.count_th)
        if rank == 0:
            print(f"{len(res)} images, segments per image: {[len(r['panoptic_seg'][1]) for r in res]}")
    else:
        gloo = dist.new_group(backend="gloo") if world > 1 else None       # object gather side group (detectron2 comm)
        ev = PanopticEvaluatorAgnostic(output_dir=args.out, gt_maps=gt_maps, gt_annotations=gt_anns, group=gloo)
        out = trainer.compute_pq(batches(files, args.size, args.batch, args.panoptic), ev, num_inference_steps=args.steps,
                                 seed=args.seed, threshold_output=True, mask_th=args.mask_th, count_th=args.count_th)
        if rank == 0:
            print(ev.table)
            print(dict(out["panoptic_seg"]))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
