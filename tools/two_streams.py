"""UNet forwards of B images on one stream against the same images split over several handles / streams (B = 8, 16; 2 x 4, 2 x 8, 4 x 2):
is there throughput in running half-batches concurrently?  (No: see profiles/r05_batch_and_streams.txt.)  python tools/two_streams.py"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R + "/latent-diffusion-segmentation_amd")
import torch
from ldmseg_amd import _lib, weights
from ldmseg_amd.models import UNet
usd = weights.generate(weights.unet_schema(12, False), seed=0)
L = 64
def bench(nstreams, B, iters=20):
    us = [UNet(usd, 12, "cuda:0", "bf16") for _ in range(nstreams)]
    ss = [torch.cuda.Stream() for _ in range(nstreams)]
    xs = [torch.randn(B, 12, L, L, device="cuda") for _ in range(nstreams)]
    t = torch.tensor(499, device="cuda")
    for _ in range(3):
        for u, s, x in zip(us, ss, xs):
            with torch.cuda.stream(s):
                u(x, t)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(iters):
            for u, s, x in zip(us, ss, xs):
                with torch.cuda.stream(s):
                    y = u(x, t).sample
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / iters * 1e3)
    fb = [_lib.lib().ldmseg_debug_get(10)]
    print(f"{nstreams} stream(s) x B={B}: {best:.3f} ms per round of {nstreams * B} images  ({nstreams * B / best * 1e3:.0f} image-forwards/s)  gn ring fallbacks {fb}", flush=True)
    del us
bench(1, 8)
bench(2, 4)
bench(2, 8)
bench(1, 16)
bench(4, 2)
bench(1, 8)
