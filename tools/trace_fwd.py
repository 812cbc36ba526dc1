"""Plain UNet forwards (no per-launch events) for `rocprofv3 --kernel-trace`; tools/trace_join.py pairs the trace with the
launch labels tools/prof_layers.py wrote to gpurun_out/layers.csv."""
import sys, os, torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, os.path.join(R, "latent-diffusion-segmentation_amd")); sys.path.insert(0, R)
from ldmseg_amd import _lib, weights
from ldmseg_amd.models import UNet
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
L = int(sys.argv[3]) if len(sys.argv) > 3 else 64
if os.environ.get("DBG"): _lib.lib().ldmseg_debug_set(1, int(os.environ["DBG"], 0))
u = UNet(weights.generate(weights.unet_schema(12, False), seed=0), 12, "cuda:0", dt)
x = torch.randn(B, 12, L, L, device="cuda")
for _ in range(5): u(x, 500)
torch.cuda.synchronize()
