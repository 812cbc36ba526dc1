#!/usr/bin/env python
"""Same-process A/B of the UNet forward under debug-knob settings:  python tools/ab_forward.py [--batch 8 --latent 64] "12=0" "12=3" "12=3,13=256"
Each setting: 3 warm-up forwards, then `--iters` timed forwards (HIP events), rounds interleaved `--rounds` times; prints ms per
forward (min / median over rounds) and the deviation of the output from the first setting's."""
import argparse
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion-segmentation_amd"))
import torch  # noqa: E402
from ldmseg_amd import _lib, weights  # noqa: E402
from ldmseg_amd.models import UNet  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("settings", nargs="+")
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--latent", type=int, default=64)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--rounds", type=int, default=3)
a = ap.parse_args()
L = _lib.lib()
usd = weights.generate(weights.unet_schema(12, False), seed=0)
u = UNet(usd, 12, "cuda:0", "bf16")
x = torch.randn(a.batch, 12, a.latent, a.latent, generator=torch.Generator().manual_seed(3)).cuda()
t = torch.tensor(499, device="cuda")
defaults = {}


def apply(setting):
    for kv in setting.split(","):
        if not kv or kv == "default":
            continue
        k, v = kv.split("=")
        k, v = int(k), int(v, 0)
        if k not in defaults:
            defaults[k] = L.ldmseg_debug_get(k)
        L.ldmseg_debug_set(k, v)


def restore():
    for k, v in defaults.items():
        L.ldmseg_debug_set(k, v)


times = {s: [] for s in a.settings}
outs = {}
for r in range(a.rounds):
    for s in a.settings:
        restore()
        apply(s)
        for _ in range(3):
            y = u(x, t).sample
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            y = u(x, t).sample
        e1.record()
        torch.cuda.synchronize()
        times[s].append(e0.elapsed_time(e1) / a.iters)
        outs.setdefault(s, y.float().cpu())
restore()
ref = outs[a.settings[0]]
for s in a.settings:
    v = sorted(times[s])
    d = float((outs[s] - ref).norm() / ref.norm())
    print(f"{s:24s} ms/forward min {v[0]:.3f} median {v[len(v) // 2]:.3f}   rel-L2 vs first {d:.2e}", flush=True)
