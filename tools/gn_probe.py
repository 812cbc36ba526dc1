import sys, time; sys.path.insert(0,"latent-diffusion-segmentation_amd")
import torch
from ldmseg_amd import _lib, weights
from ldmseg_amd.models import UNet
from ldmseg_amd.schedulers import DDIMNoiseScheduler
from ldmseg_amd.trainers import TrainerDiffusion
sys.path.insert(0,"tests")
from conftest import SCHED_KW
u = UNet(weights.generate(weights.unet_schema(12, False), seed=0), 12, "cuda:0", "bf16")
tr = TrainerDiffusion(None, u, DDIMNoiseScheduler(**SCHED_KW))
for (B, L, inp) in ((8, 64, 0), (16, 64, 0), (16, 64, 1), (4, 128, 0), (8, 64, 0)):
    rgb = (0.18215 * torch.randn(B, 4, L, L)).cuda()
    for i in range(12):
        sx = DDIMNoiseScheduler(**SCHED_KW); sx.set_timesteps_inference(4)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if inp:
            gi = torch.Generator().manual_seed(7)
            z0 = (0.2 * torch.randn(rgb.shape, generator=gi)).cuda()
            known = (torch.rand(B, 1, L, L, generator=gi) < 0.5).cuda()
            o = tr.sample_inpaint([""] * B, known, z0, seed=42, rgb_latents=rgb, scheduler=sx)
        else:
            o = tr.sample([""] * B, num_inference_steps=4, seed=42, rgb_latents=rgb, scheduler=sx)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3 / 4
        print(f"B={B} L={L} inpaint={inp} call {i}: {dt:7.2f} ms/step  fallbacks {u.gn_fallbacks()}  backoff {u.gn_backoff()}", flush=True)
