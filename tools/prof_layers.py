import sys, os, ctypes as C, torch, collections
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, os.path.join(R, "latent-diffusion-segmentation_amd")); sys.path.insert(0, R)
from ldmseg_amd import _lib, weights
from ldmseg_amd.models import UNet
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
usd = weights.generate(weights.unet_schema(12, False), seed=0)
import os
if os.environ.get("QF1"): _lib.lib().ldmseg_debug_set(2, int(os.environ["QF1"]))

if os.environ.get("DBG"): _lib.lib().ldmseg_debug_set(1, int(os.environ["DBG"], 0))
u = UNet(usd, 12, "cuda:0", dt)
x = torch.randn(B, 12, 64, 64, device="cuda")
for _ in range(2): u(x, 500)
torch.cuda.synchronize()
lib = _lib.lib(); lib.ldmseg_profile_reset(); lib.ldmseg_profile_enable(1)
R = 3
for _ in range(R): u(x, 500)
torch.cuda.synchronize(); lib.ldmseg_profile_enable(0)
os.makedirs("gpurun_out", exist_ok=True)
lib.ldmseg_profile_dump(b"gpurun_out/layers.csv")
agg = collections.OrderedDict()
for line in open("gpurun_out/layers.csv").read().splitlines()[1:]:
    fam, label, ms, fl = line.split(",")
    k = (fam, label); a = agg.setdefault(k, [0, 0.0, 0.0]); a[0] += 1; a[1] += float(ms); a[2] += float(fl)
tot = sum(a[1] for a in agg.values()) / R
print(f"total profiled ms/forward {tot:.3f}")
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
for (fam, label), (n, ms, fl) in rows[:45]:
    print(f"fam{fam} {label:70s} n/fwd={n/R:4.0f} ms/fwd={ms/R:7.3f} us/launch={1e3*ms/n:8.1f} TF={fl/ms/1e9 if ms>0 else 0:7.1f}")
