"""CPU suite: structural pins of the image-VAE (AutoencoderKL encoder) restatement.  The arithmetic is diffusers'
(absent here) - parity unpinned; what can be pinned is the published SD-1.x VAE encoder size and the key layout."""
import torch

from ldmseg_amd import weights
from oracle import vae_image as o_vi


def test_schema_matches_sd_vae_encoder_size():
    sc = weights.vae_image_schema()
    assert weights.count_params(sc) == 34_163_592 + 72          # SD-1.x VAE encoder + quant_conv
    assert len(sc) == 108
    assert sc["encoder.down_blocks.1.resnets.0.conv_shortcut.weight"] == (256, 128, 1, 1)
    assert "encoder.down_blocks.3.downsamplers.0.conv.weight" not in sc
    assert sc["encoder.conv_out.weight"] == (8, 512, 3, 3) and sc["quant_conv.weight"] == (8, 8, 1, 1)


def test_oracle_shapes_and_attention_key_aliases():
    sd = weights.generate(weights.vae_image_schema(), seed=11, norm_keys=weights.VAE_IMAGE_NORM_KEYS)
    x = torch.rand(2, 3, 64, 32, generator=torch.Generator().manual_seed(0))
    mom = o_vi.encode_moments(sd, 2 * x - 1)
    assert mom.shape == (2, 8, 8, 4) and torch.isfinite(mom).all()
    # later diffusers names (to_q ...) give the same result
    ren = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}
    sd2 = {}
    for k, v in sd.items():
        parts = k.split(".")
        if "attentions" in parts and parts[-2] in ren:
            k = ".".join(parts[:-2] + [ren[parts[-2]], parts[-1]])
        sd2[k] = v
    assert torch.equal(o_vi.encode_moments(sd2, 2 * x - 1), mom)
    # batch independence and the asymmetric stride-2 padding: shifting content at the right/bottom edge matters
    assert torch.allclose(o_vi.encode_moments(sd, 2 * x[:1] - 1), mom[:1], atol=1e-5)
    lat = o_vi.encode_mode(sd, x)
    assert torch.equal(lat, mom[:, :4] * 0.18215)
