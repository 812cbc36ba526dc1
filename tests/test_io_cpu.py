"""CPU suite, part 3: bit-codec oracle against the reference-generated golden, and the readers for
the reference's checkpoint layouts (host logic only)."""
import numpy as np
import pytest
import torch

from oracle import bitcodec as o_bits


def test_bitcodec_oracle_vs_reference_golden(golden):
    g = golden("bitcodec.npz")
    bits, ign = o_bits.encode_bitmap(g["ids"])
    assert np.array_equal(bits, g["bits"]) and np.array_equal(ign, g["ignore"])
    assert np.array_equal(o_bits.decode_bitmap(2 * g["bits"] - 1), g["decoded"])
    assert np.array_equal(o_bits.decode_bitmap(g["dec_in"]), g["dec_out"])
    # SURVEY App. C known answers: id 1 -> [1,0,0,0,0,0,0], id 5 -> [1,0,1,...], 127 -> ones, void -> 0.5
    b, _ = o_bits.encode_bitmap(np.array([[0, 1, 2], [5, 127, 64]]))
    assert b[:, 0, 1].tolist() == [1, 0, 0, 0, 0, 0, 0]
    assert b[:, 1, 0].tolist() == [1, 0, 1, 0, 0, 0, 0]
    assert b[:, 1, 1].tolist() == [1] * 7 and b[:, 0, 0].tolist() == [0.5] * 7
    assert o_bits.decode_bitmap(2 * b - 1).tolist() == [[0, 1, 2], [5, 127, 64]]


def _tiny_like(schema):
    # tensors with the right shapes but shared zero storage (cheap): the loader only checks keys/shapes
    return {k: torch.zeros(1).expand(shp) for k, shp in schema.items()}


def test_checkpoint_readers(tmp_path):
    from ldmseg_amd import checkpoint, weights
    usd = _tiny_like(weights.unet_schema(12, False))
    usd["new_conv.weight"] = usd["conv_in.weight"]            # duplicate alias the reference saves (unet.py:182,233)
    usd["new_conv.bias"] = usd["conv_in.bias"]
    vsd = weights.generate(weights.vae_schema(), seed=1, norm_keys=weights.VAE_NORM_KEYS)
    data = {"step": 7, "epoch": 1, "vae_image": {}, "vae_semseg": vsd, "unet": usd, "ema": None, "opt": None,
            "p": {"x": 1}, "scaler": None}
    out = checkpoint.unet_state_from(data)
    assert list(out) == list(weights.unet_schema(12, False)) and "new_conv.weight" not in out
    assert list(checkpoint.vae_state_from(data)) == list(weights.vae_schema())
    ae = {"step": 1, "epoch": 0, "vae": {"module." + k: v for k, v in vsd.items()}, "opt": None, "p": {}, "scaler": None}
    path = tmp_path / "ae.pt"
    torch.save(ae, str(path))
    back = checkpoint.load_ae_checkpoint(str(path))
    assert all(torch.equal(back[k], vsd[k]) for k in vsd)
    bad = dict(data, unet={k: v for k, v in usd.items() if k != "conv_out.bias"})
    with pytest.raises(KeyError):
        checkpoint.unet_state_from(bad)
    bad = dict(data, unet=dict(usd, **{"conv_out.bias": torch.zeros(5)}))
    with pytest.raises(ValueError):
        checkpoint.unet_state_from(bad)
    cross = dict(usd, **{"mid_block.attentions.0.transformer_blocks.0.attn2.to_q.weight": torch.zeros(1)})
    with pytest.raises(NotImplementedError):
        checkpoint.unet_state_from(dict(data, unet=cross))


def test_color_map_and_encode_seg_vs_reference_golden(golden):
    """utils.color_map / encode_seg against arrays produced by importing the reference (utils.py:240-258,
    trainers_ldm_cond.py:324-332)."""
    import numpy as np
    from ldmseg_amd.utils import color_map, encode_seg
    g = golden("colormap.npz")
    assert np.array_equal(color_map(), g["cmap"]) and color_map().dtype == np.uint8
    assert np.allclose(color_map(normalized=True), g["cmap_norm"], atol=1e-7)
    assert np.array_equal(encode_seg(g["ids"]), g["painted"])
    assert np.array_equal(color_map(8), g["cmap"][:8])


def test_scheduler_timesteps_cache_follows_reassignment(sched_kw):
    """The host copy of `timesteps` (used so that step() needs no D2H sync) must never outlive the tensor it mirrors."""
    import torch
    from ldmseg_amd.schedulers import DDIMNoiseScheduler
    s = DDIMNoiseScheduler(**sched_kw)
    assert s.timesteps_host()[:3] == [999, 998, 997] and len(s.timesteps) == 1000
    s.set_timesteps_inference(50)
    assert s.timesteps_host()[:2] == [999, 979]
    full = s.timesteps
    s.timesteps = full[10:]
    assert s.timesteps_host() == [int(v) for v in full[10:]]
    assert s._timestep_int(s.timesteps[0]) == int(full[10])
    s.timesteps = torch.tensor([7, 5, 3])
    assert s.timesteps_host() == [7, 5, 3]
    s.move_timesteps_to("cpu")
    assert s.timesteps_host() == [7, 5, 3]
    s.set_timesteps_inference(10)
    assert s.timesteps_host() == [int(v) for v in s.timesteps] == list(range(999, 0, -100))


def test_val_transforms_match_pil(tmp_path):
    """data.transforms: CropResize(crop_mode=None) is a plain PIL resize (bicubic for images) and ToTensor scales to [0, 1]."""
    from PIL import Image
    from ldmseg_amd.data import transforms as T
    g = np.random.RandomState(1)
    arr = g.randint(0, 256, size=(37, 53, 3)).astype(np.uint8)
    p = tmp_path / "im.png"
    Image.fromarray(arr).save(p)
    t, (h, w) = T.load_rgb(str(p), 64)
    assert (h, w) == (37, 53) and t.shape == (3, 64, 64) and t.dtype == torch.float32
    R = getattr(Image, "Resampling", Image)
    ref = np.asarray(Image.fromarray(arr).resize((64, 64), resample=R.BICUBIC, reducing_gap=None), dtype=np.float32) / 255
    assert np.array_equal(t.permute(1, 2, 0).numpy(), ref)
    ids = Image.fromarray(g.randint(0, 200, size=(10, 12)).astype(np.uint8))
    small = T.ids_to_tensor(T.crop_resize(ids, (5, 6), "nearest"))
    assert small.dtype == torch.int64 and small.shape == (5, 6)


def test_real_coco_pairs_host_side(golden):
    """Two of the reference's own example pairs (tests/golden/real_coco.npz: the jpg / png bytes + what the reference's
    functions make of them, tests/golden/make_golden.py::real_coco_golden): the host side of the I/O rows - PNG colours ->
    segment ids (evaluator's rgb2id = coco.py:500-501), the oracle bit codec on the REAL remapped id maps, and the PIL
    resize to 512 x 512 (CropResize: bicubic for the image, nearest for ids) on a portrait and a landscape image."""
    import hashlib
    import io
    from PIL import Image
    from ldmseg_amd.data.transforms import crop_resize
    from ldmseg_amd.evaluations import rgb2id
    g = golden("real_coco.npz")
    sizes = []
    for k in range(2):
        img = Image.open(io.BytesIO(g[f"jpg_{k}"].tobytes())).convert("RGB")
        sem = np.array(Image.open(io.BytesIO(g[f"png_{k}"].tobytes())).convert("RGB"))
        sizes.append(img.size)
        ids = rgb2id(sem)
        assert np.array_equal(ids, g[f"ids_{k}"])
        assert ids.shape == (img.size[1], img.size[0]) and len(np.unique(ids)) >= 10       # real scenes: 12 / 19 segments + void
        # the reference's random relabelling is a bijection of the segments onto 1..127 with void fixed at 0
        remapped, mapping = g[f"remapped_{k}"], dict(g[f"mapping_{k}"].tolist())
        assert set(np.unique(ids)) - {0} == set(mapping) and len(set(mapping.values())) == len(mapping)
        assert np.array_equal(np.vectorize(lambda v: mapping.get(int(v), 0))(ids), remapped)
        bits, ign = o_bits.encode_bitmap(remapped.astype(np.int64))
        assert np.array_equal(bits, g[f"bits_{k}"].astype(np.float32)) and np.array_equal(ign, g[f"ignore_{k}"])
        assert np.array_equal(o_bits.decode_bitmap(2 * bits - 1), g[f"decoded_{k}"]) and np.array_equal(g[f"decoded_{k}"], remapped)
        r_img = np.asarray(crop_resize(img, (512, 512), "bicubic"))
        # JPEG decoding (libjpeg-turbo) and PIL's bicubic filter are allowed to differ by a rounding step between library versions
        # (ADVICE r04): the subsampled pixels are compared with a tolerance everywhere, bit for bit (and through the sha of the whole
        # image) only under the Pillow version the fixture was generated with
        import PIL
        diff = np.abs(r_img[::8, ::8].astype(np.int16) - g[f"resized_sample_{k}"].astype(np.int16))
        assert diff.max() <= 2 and (diff > 0).mean() < 0.02, (PIL.__version__, int(diff.max()))
        if PIL.__version__ == str(g["pillow_version"]):                                    # (recorded by make_golden.py)
            assert np.array_equal(r_img[::8, ::8], g[f"resized_sample_{k}"])
            assert hashlib.sha256(r_img.tobytes()).digest() == g[f"resized_sha_{k}"].tobytes()
        r_ids = np.asarray(crop_resize(Image.fromarray(remapped), (512, 512), "nearest"))
        assert np.array_equal(r_ids, g[f"resized_ids_{k}"])
    assert sizes == [(480, 640), (640, 427)]                      # (width, height): one portrait, one landscape
