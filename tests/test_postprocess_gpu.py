"""GPU parity: ldmseg_panoptic_postprocess (through TrainerDiffusion.postprocess_panoptic) against the
CPU restatement of trainers_ldm_cond.py:1277-1313.  Integer outputs (labels, counts, keep, panoptic map)
are compared exactly."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import postprocess as o_pp

pytestmark = pytest.mark.gpu


def _segment_like_logits(B, C, H, W, seed, sharp=6.0):
    g = torch.Generator().manual_seed(seed)
    low = torch.randn(B, C, max(2, H // 16), max(2, W // 16), generator=g)
    x = F.interpolate(low, size=(H, W), mode="bilinear", align_corners=False) * sharp
    return x + 0.3 * torch.randn(B, C, H, W, generator=g)


def _trainer():
    from ldmseg_amd.trainers import TrainerDiffusion
    return TrainerDiffusion.__new__(TrainerDiffusion)      # post-processing needs no model objects


@pytest.mark.parametrize("B,C,H,W", [(2, 128, 96, 80), (1, 16, 33, 47), (3, 128, 64, 64)])
@pytest.mark.parametrize("kw", [
    dict(threshold_output=False, count_th=64, overlap_th=0.5, ignore_label=0),
    dict(threshold_output=True, threshold_mode="max", mask_th=0.5, count_th=32, overlap_th=0.3, ignore_label=0),
    dict(threshold_output=True, threshold_mode="topk_diff", mask_th=0.4, count_th=16, overlap_th=0.6, ignore_label=5),
    dict(threshold_output=True, threshold_mode="max", mask_th=0.9, count_th=1, overlap_th=0.0, ignore_label=-1),
])
def test_postprocess_matches_oracle(B, C, H, W, kw):
    x = _segment_like_logits(B, C, H, W, seed=H * 7 + C)
    tr = _trainer()
    res, st = tr.postprocess_panoptic(x.cuda(), return_stats=True, **kw)
    assert len(res) == B
    for b in range(B):
        pan, info, raw, ost = o_pp.panoptic_postprocess(x[b], **kw)
        assert np.array_equal(st["labels"][b].cpu().numpy(), raw)
        assert np.array_equal(st["counts"][b].cpu().numpy(), ost["counts"])
        assert np.array_equal(st["mask_counts"][b].cpu().numpy(), ost["mask_counts"])
        got_pan, got_info = res[b]["panoptic_seg"]
        assert got_info == info
        assert np.array_equal(got_pan.cpu().numpy(), pan)


def test_postprocess_full_size_properties():
    """BASELINE size (8 x 128 x 512 x 512 logits): size-independent invariants."""
    B, C, H, W = 8, 128, 512, 512
    g = torch.Generator(device="cuda").manual_seed(3)
    low = torch.randn(B, C, 32, 32, device="cuda", generator=g)
    x = F.interpolate(low, size=(H, W), mode="bilinear", align_corners=False) * 6
    tr = _trainer()
    res, st = tr.postprocess_panoptic(x, threshold_output=True, mask_th=0.5, count_th=512, overlap_th=0.5,
                                      ignore_label=0, return_stats=True)
    labels, counts, keep = st["labels"], st["counts"].long(), st["keep"].bool()
    assert int(counts.sum()) == int((labels >= 0).sum())                       # every non-void pixel counted once
    assert torch.equal(counts, torch.stack([torch.bincount(l[l >= 0].flatten().long(), minlength=C) for l in labels]))
    assert (counts[keep] >= 512).all() and not keep[:, 0].any()
    for b in range(B):
        pan, info = res[b]["panoptic_seg"]
        kept = {s["id"] for s in info}
        assert set(torch.unique(pan).tolist()) - {0} == kept                    # map and segments_info agree
        assert torch.equal(pan > 0, keep[b][labels[b].clamp(min=0).long()] & (labels[b] >= 0))
    # idempotent on its own output: one-hot logits of the kept map give the same map back
    again = tr.postprocess_panoptic(x, threshold_output=True, mask_th=0.5, count_th=512, overlap_th=0.5, ignore_label=0)
    assert all(torch.equal(a["panoptic_seg"][0], r["panoptic_seg"][0]) for a, r in zip(again, res))


def test_postprocess_rejects_host_tensors():
    with pytest.raises(RuntimeError):
        _trainer().postprocess_panoptic(torch.zeros(1, 4, 8, 8))


# ---------------------------------------------------------------------------------------------------------------------
# Fused evaluation tail (ldmseg_vae_decode_panoptic / postproc.hip resample_scan_kernel): the decoder's 4L output ->
# bilinear x2 -> bilinear to the input size -> crop -> bilinear to (h, w) -> post-processing, against the chain of torch
# ops the reference runs (trainers_ldm_cond.py:1252-1271 on top of vae.py:270) followed by the oracle post-processing.
def _chain(x4, in_size, box, hw):
    x8 = F.interpolate(x4[None], scale_factor=2, mode="bilinear", align_corners=False)           # vae.py:270
    xs = F.interpolate(x8, size=in_size, mode="bilinear", align_corners=False)[0]                 # :1252-1257
    y0, x0, ch, cw = box
    xs = xs[:, y0:y0 + ch, x0:x0 + cw]                                                            # :1263
    return F.interpolate(xs[None].float(), size=hw, mode="bilinear", align_corners=False)[0]      # :1266-1271


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("geo", [
    # (H4, W4), input size, [(crop box), ...], [(h, w), ...]
    ((32, 32), (64, 64), [(0, 0, 64, 64), (0, 0, 48, 64), (3, 5, 50, 41)], [(64, 64), (97, 133), (40, 30)]),     # input size = 8L: identity stage
    ((32, 32), (80, 72), [(0, 0, 80, 72), (7, 0, 60, 72)], [(120, 101), (33, 200)]),                            # a real middle resize
    ((16, 24), (32, 48), [(1, 2, 30, 44)], [(31, 47)]),                                                         # non-square map
])
def test_fused_tail_resampling_and_postprocess(geo, dt):
    import ctypes as C
    from ldmseg_amd import _lib
    (H4, W4), in_size, boxes, sizes = geo
    B, Cn = len(boxes), 128
    x4 = _segment_like_logits(B, Cn, H4, W4, seed=H4 + in_size[0] + dt, sharp=5.0)
    if dt == 1:
        x4 = x4.bfloat16().float()
    kw = dict(threshold_output=True, threshold_mode="max", mask_th=0.3, count_th=40, overlap_th=0.4, ignore_label=0)
    sz = np.asarray(sizes, np.int32)
    bx = np.asarray(boxes, np.int32)
    npix = sz[:, 0].astype(np.int64) * sz[:, 1]
    offs = np.concatenate([[0], np.cumsum(npix)[:-1]]).astype(np.int64)
    tot = int(npix.sum())
    dev = "cuda:0"
    labels = torch.empty(tot, dtype=torch.int32, device=dev)
    pan = torch.empty(tot, dtype=torch.int32, device=dev)
    keep = torch.empty(B, Cn, dtype=torch.uint8, device=dev)
    counts = torch.empty(B, Cn, dtype=torch.int32, device=dev)
    mcounts = torch.empty(B, Cn, dtype=torch.int32, device=dev)
    vol = torch.empty(tot * Cn, dtype=torch.float32, device=dev)
    dx = x4.to(dev)
    P = lambda t: C.c_void_p(t.data_ptr())
    A = lambda a: C.c_void_p(a.ctypes.data)
    r = _lib.lib().ldmseg_op_panoptic_from_decoder(P(dx), B, Cn, H4, W4, dt, in_size[0], in_size[1], A(bx), A(sz), A(offs), 1, 0,
                                                  kw["mask_th"], kw["count_th"], kw["overlap_th"], 0, P(labels), P(pan), P(keep),
                                                  P(counts), P(mcounts), P(vol), None)
    assert r == 0
    torch.cuda.synchronize()
    for b in range(B):
        h, w = sizes[b]
        ref = _chain(x4[b], in_size, boxes[b], (h, w))
        got = vol[Cn * int(offs[b]):Cn * int(offs[b]) + Cn * h * w].view(Cn, h, w).cpu()
        # the interpolation chain itself: one separable weighted sum against three chained fp32 interpolations
        assert float((got - ref).abs().max()) < 2e-5 * float(ref.abs().max()), (b, float((got - ref).abs().max()))
        # integer outputs: exact against the oracle post-processing evaluated on the kernel's own resampled logits
        pan_ref, info_ref, raw_ref, ost = o_pp.panoptic_postprocess(got, **kw)
        lab = labels[int(offs[b]):int(offs[b]) + h * w].view(h, w).cpu().numpy()
        top2 = got.topk(2, dim=0)[0]
        prob = torch.softmax(got, 0).max(0)[0]
        clear = ((top2[0] - top2[1]) > 1e-5) & ((prob - kw["mask_th"]).abs() > 1e-5)           # (fp32 exp differs in the last bits)
        assert np.array_equal(lab[clear.numpy()], raw_ref[clear.numpy()])
        nfuzzy = int((~clear).sum())
        assert nfuzzy <= max(3, (h * w) // 200)          # (bf16-quantised logits tie more often)
        assert np.abs(counts[b].cpu().numpy() - ost["counts"]).sum() <= 2 * nfuzzy
        sig_edge = int(((torch.sigmoid(got) - kw["mask_th"]).abs() < 1e-6).sum())
        assert np.abs(mcounts[b].cpu().numpy() - ost["mask_counts"]).sum() <= sig_edge
        if nfuzzy == 0 and sig_edge == 0:
            assert np.array_equal(pan[int(offs[b]):int(offs[b]) + h * w].view(h, w).cpu().numpy(), pan_ref)
            assert [int(c) + 1 for c in torch.nonzero(keep[b]).flatten().tolist()] == [s["id"] for s in info_ref]
        # and against the chain of torch ops: same labels wherever the decision is not within rounding of a tie
        _, _, raw_chain, _ = o_pp.panoptic_postprocess(ref, **kw)
        t2 = ref.topk(2, dim=0)[0]
        pr = torch.softmax(ref, 0).max(0)[0]
        clear2 = ((t2[0] - t2[1]) > 1e-3) & ((pr - kw["mask_th"]).abs() > 1e-3)
        assert clear2.float().mean() > 0.9
        assert np.array_equal(lab[clear2.numpy()], raw_chain[clear2.numpy()])


def test_decode_panoptic_fused_vs_unfused(vae_sd):
    """ldmseg_vae_decode_panoptic against the unfused product path (decode -> torch interpolate -> crop -> interpolate ->
    ldmseg_panoptic_postprocess) on the same latents, padding masks and ragged output sizes; bf16 and fp32."""
    from ldmseg_amd.models import GeneralVAESeg
    from ldmseg_amd.trainers import TrainerDiffusion
    for cd in ("fp32", "bf16"):
        vae = GeneralVAESeg(vae_sd, scaling_factor=0.2, device="cuda:0", compute_dtype=cd)
        g = torch.Generator().manual_seed(3)
        B, L = 3, 16
        z = (0.2 * 3.0 * torch.randn(B, 4, L, L, generator=g)).cuda()
        S = 8 * L
        masks = torch.zeros(B, S, S, dtype=torch.bool)
        masks[0] = True
        masks[1, :100, :] = True
        masks[2, 9:120, 4:90] = True
        sizes = [(128, 128), (75, 96), (200, 155)]
        kw = dict(threshold_output=True, threshold_mode="topk_diff", mask_th=0.005, count_th=30, overlap_th=0.2, ignore_label=0)
        boxes = TrainerDiffusion.padding_boxes(masks.cuda())
        assert boxes.tolist() == [[0, 0, 128, 128], [0, 0, 100, 128], [9, 4, 111, 86]]
        outs, st = vae.decode_panoptic(z, (S, S), sizes, boxes, z_scale=1 / 0.2, return_stats=True, **kw)
        tr = _trainer()
        logits = vae.decode(z, z_scale=1 / 0.2).float()
        logits = F.interpolate(logits, size=(S, S), mode="bilinear", align_corners=False)
        for b in range(B):
            m = TrainerDiffusion.crop_padding(logits[b], masks[b].cuda())
            m = F.interpolate(m[None], size=sizes[b], mode="bilinear", align_corners=False)
            res, ust = tr.postprocess_panoptic(m.contiguous(), return_stats=True, **kw)
            lab_u = ust["labels"][0]
            lab_f = st["labels"][b]
            assert lab_f.shape == lab_u.shape == sizes[b]
            # bf16: the unfused path interpolates the fp32 copy of bf16-rounded 4L logits through an fp32 8L tensor - the same values
            agree = float((lab_f == lab_u).float().mean())
            assert agree > 0.999, (cd, b, agree)
            pan_f, kept = outs[b]
            pan_u, info_u = res[0]["panoptic_seg"]
            assert float((pan_f == pan_u).float().mean()) > 0.995, (cd, b)
