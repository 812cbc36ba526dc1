"""accuracy figures quoted in DESIGN.md (round 5): bf16x3 vs exact fp32; the three restructurings vs their old launch sets vs fp32"""
import sys, torch
sys.path.insert(0, "latent-diffusion-segmentation_amd")
from ldmseg_amd import weights, _lib
from ldmseg_amd.models import UNet
usd = weights.generate(weights.unet_schema(12, False), seed=0)
x = torch.randn(8, 12, 64, 64, generator=torch.Generator().manual_seed(5)).cuda()
l2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
u32 = UNet(usd, 12, "cuda:0", "fp32"); y32 = u32(x, 499).sample.clone(); del u32
u3 = UNet(usd, 12, "cuda:0", "bf16x3"); y3 = u3(x, 499).sample.clone(); del u3
print("bf16x3 vs fp32: rel-L2 %.3e  max-norm %.3e" % (l2(y3, y32), float((y3 - y32).abs().max() / y32.abs().max())))
u = UNet(usd, 12, "cuda:0", "bf16"); L = _lib.lib()
y = u(x, 499).sample.clone()
print("bf16 shipped vs fp32: rel-L2 %.3e" % l2(y, y32))
for k in (19, 20, 21):
    L.ldmseg_debug_set(k, 0); yo = u(x, 499).sample.clone(); L.ldmseg_debug_set(k, 1)
    print("key %d off: vs shipped %.3e, vs fp32 %.3e" % (k, l2(yo, y), l2(yo, y32)))
for k in (19, 20, 21): L.ldmseg_debug_set(k, 0)
yo = u(x, 499).sample.clone()
for k in (19, 20, 21): L.ldmseg_debug_set(k, 1)
print("all three off (round-4 launch set): vs shipped %.3e, vs fp32 %.3e" % (l2(yo, y), l2(yo, y32)))
