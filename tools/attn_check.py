import ctypes as C, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion-segmentation_amd"))
import torch
from ldmseg_amd import _lib
L = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr())
def ref(src, B, N, Cc):
    q, k, v = src.chunk(3, -1); d = Cc // 8
    q = q.view(B, N, 8, d).transpose(1, 2).double(); k = k.view(B, N, 8, d).transpose(1, 2).double(); v = v.view(B, N, 8, d).transpose(1, 2).double()
    return (torch.softmax(q @ k.transpose(-1, -2) * d ** -0.5, -1) @ v).transpose(1, 2).reshape(B, N, Cc).float()
variants = [int(a) for a in sys.argv[1:]] or [0, 5, 4, 1]
for (B, N, Cc) in [(1, 256, 320), (1, 384, 320), (1, 512, 320), (1, 1024, 320), (2, 1024, 320), (8, 1024, 320), (1, 1024, 640), (2, 4096, 320)]:
    g = torch.Generator().manual_seed(N + Cc)
    qkv = torch.randn(B, N, 3 * Cc, generator=g)
    r = ref(qkv.to(torch.bfloat16).float(), B, N, Cc)
    for var in variants:
        L.ldmseg_debug_set(2, var)
        outs = []
        for rep in range(2):
            out = torch.empty(B, N, Cc, device="cuda"); dq = qkv.cuda()
            rc = L.ldmseg_op_attention(P(dq), B, N, Cc, 8, 1, P(out), None); torch.cuda.synchronize()
            outs.append(out.cpu())
        o = outs[0]; nan = int(torch.isnan(o).sum())
        bad = (~torch.isfinite(o)) | ((o - r).abs() > 0.05 * r.abs().max())
        rows = bad.any(-1).nonzero()
        same = torch.equal(torch.nan_to_num(outs[0]), torch.nan_to_num(outs[1]))
        err = float((torch.nan_to_num(o) - r).abs().max() / r.abs().max())
        print(f"B={B} N={N} C={Cc} var {var}: nans {nan} rel_err {err:.3e} deterministic {same} bad_rows {rows.shape[0]} first {rows[:6].tolist()} heads_bad {sorted(set((bad.any(1).nonzero()[:,1]//(Cc//8)).tolist()))[:8]}")
