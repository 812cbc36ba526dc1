"""State-dict schemas and a deterministic weight generator.

The schemas enumerate the exact key set / shapes a reference checkpoint holds:
  * UNet: ``data['unet']`` of ldmseg.pt
    (/root/reference/ldmseg/trainers/trainers_ldm_cond.py:1791-1814) - the
    diffusers SD-1.x UNet2DConditionModel key set after
    ``remove_cross_attention`` (unet.py:83-105) and ``modify_encoder``
    (unet.py:178-233).  Validated by the known totals: 859,520,964 params /
    686 tensors (vanilla) and 815,556,484 / 574 (12-ch, no cross-attn).
  * seg-VAE: ``GeneralVAESeg.state_dict()`` (vae.py:123-244), 34 tensors,
    2,023,208 params for base.yaml:14-33.

No pretrained weights exist offline, so benchmarks and tests run on weights
drawn by ``generate`` (per-tensor seed = crc32(key) ^ seed).  Matrices/filters
are U(+-sqrt(3/fan_in)) (variance preserving, so activations stay O(1) through
38 blocks and softmax is non-trivial); biases U(+-0.05); norm gains 1 +- 0.1.
"""
import zlib
from collections import OrderedDict

import torch

BLOCK_OUT = (320, 640, 1280, 1280)
TIME_DIM = 1280
CROSS_DIM = 768


def _resnet(sd, p, cin, cout):
    sd[p + "norm1.weight"] = (cin,)
    sd[p + "norm1.bias"] = (cin,)
    sd[p + "conv1.weight"] = (cout, cin, 3, 3)
    sd[p + "conv1.bias"] = (cout,)
    sd[p + "time_emb_proj.weight"] = (cout, TIME_DIM)
    sd[p + "time_emb_proj.bias"] = (cout,)
    sd[p + "norm2.weight"] = (cout,)
    sd[p + "norm2.bias"] = (cout,)
    sd[p + "conv2.weight"] = (cout, cout, 3, 3)
    sd[p + "conv2.bias"] = (cout,)
    if cin != cout:
        sd[p + "conv_shortcut.weight"] = (cout, cin, 1, 1)
        sd[p + "conv_shortcut.bias"] = (cout,)


def _transformer(sd, p, c, cross):
    sd[p + "norm.weight"] = (c,)
    sd[p + "norm.bias"] = (c,)
    sd[p + "proj_in.weight"] = (c, c, 1, 1)
    sd[p + "proj_in.bias"] = (c,)
    b = p + "transformer_blocks.0."
    sd[b + "norm1.weight"] = (c,)
    sd[b + "norm1.bias"] = (c,)
    for n in ("to_q", "to_k", "to_v"):
        sd[b + f"attn1.{n}.weight"] = (c, c)
    sd[b + "attn1.to_out.0.weight"] = (c, c)
    sd[b + "attn1.to_out.0.bias"] = (c,)
    if cross:
        sd[b + "norm2.weight"] = (c,)
        sd[b + "norm2.bias"] = (c,)
        sd[b + "attn2.to_q.weight"] = (c, c)
        sd[b + "attn2.to_k.weight"] = (c, CROSS_DIM)
        sd[b + "attn2.to_v.weight"] = (c, CROSS_DIM)
        sd[b + "attn2.to_out.0.weight"] = (c, c)
        sd[b + "attn2.to_out.0.bias"] = (c,)
    sd[b + "norm3.weight"] = (c,)
    sd[b + "norm3.bias"] = (c,)
    sd[b + "ff.net.0.proj.weight"] = (8 * c, c)
    sd[b + "ff.net.0.proj.bias"] = (8 * c,)
    sd[b + "ff.net.2.weight"] = (c, 4 * c)
    sd[b + "ff.net.2.bias"] = (c,)
    sd[p + "proj_out.weight"] = (c, c, 1, 1)
    sd[p + "proj_out.bias"] = (c,)


def unet_schema(in_channels=12, cross_attention=False):
    """Ordered key -> shape for the SD-1.x UNet as LDMSeg instantiates it."""
    sd = OrderedDict()
    sd["conv_in.weight"] = (BLOCK_OUT[0], in_channels, 3, 3)
    sd["conv_in.bias"] = (BLOCK_OUT[0],)
    sd["time_embedding.linear_1.weight"] = (TIME_DIM, BLOCK_OUT[0])
    sd["time_embedding.linear_1.bias"] = (TIME_DIM,)
    sd["time_embedding.linear_2.weight"] = (TIME_DIM, TIME_DIM)
    sd["time_embedding.linear_2.bias"] = (TIME_DIM,)
    skip_ch = [BLOCK_OUT[0]]
    c = BLOCK_OUT[0]
    for i, co in enumerate(BLOCK_OUT):
        for j in range(2):
            _resnet(sd, f"down_blocks.{i}.resnets.{j}.", c, co)
            c = co
            if i < 3:
                _transformer(sd, f"down_blocks.{i}.attentions.{j}.", c, cross_attention)
            skip_ch.append(c)
        if i < 3:
            sd[f"down_blocks.{i}.downsamplers.0.conv.weight"] = (c, c, 3, 3)
            sd[f"down_blocks.{i}.downsamplers.0.conv.bias"] = (c,)
            skip_ch.append(c)
    _resnet(sd, "mid_block.resnets.0.", c, c)
    _transformer(sd, "mid_block.attentions.0.", c, cross_attention)
    _resnet(sd, "mid_block.resnets.1.", c, c)
    for i, co in enumerate(reversed(BLOCK_OUT)):
        for j in range(3):
            _resnet(sd, f"up_blocks.{i}.resnets.{j}.", c + skip_ch.pop(), co)
            c = co
            if i > 0:
                _transformer(sd, f"up_blocks.{i}.attentions.{j}.", c, cross_attention)
        if i < 3:
            sd[f"up_blocks.{i}.upsamplers.0.conv.weight"] = (c, c, 3, 3)
            sd[f"up_blocks.{i}.upsamplers.0.conv.bias"] = (c,)
    sd["conv_norm_out.weight"] = (c,)
    sd["conv_norm_out.bias"] = (c,)
    sd["conv_out.weight"] = (4, c, 3, 3)
    sd["conv_out.bias"] = (4,)
    return sd


def vae_schema(in_channels=7, int_channels=256, out_channels=128,
               block_out_channels=(32, 64, 128, 256), latent_channels=4,
               num_latents=2, num_upscalers=2, upscale_channels=256):
    """Ordered key -> shape of GeneralVAESeg (gaussian, num_mid_blocks=0)."""
    sd = OrderedDict()

    def conv(name, co, ci, k=3):
        sd[name + ".weight"] = (co, ci, k, k)
        sd[name + ".bias"] = (co,)

    boc = tuple(block_out_channels)
    conv("encoder.0", boc[0], in_channels)
    idx = 2
    for i in range(len(boc) - 1):
        conv(f"encoder.{idx}", boc[i], boc[i])
        conv(f"encoder.{idx + 1}", boc[i + 1], boc[i])
        idx += 3
    conv(f"encoder.{idx}", int_channels, boc[-1])          # 11
    sd[f"encoder.{idx + 2}.weight"] = (int_channels,)      # 13 GroupNorm
    sd[f"encoder.{idx + 2}.bias"] = (int_channels,)
    conv(f"encoder.{idx + 4}", latent_channels * num_latents, int_channels)  # 15
    conv("decoder.0", int_channels, latent_channels)
    idx = 2
    cin = int_channels
    for _ in range(num_upscalers):
        sd[f"decoder.{idx}.weight"] = (cin, upscale_channels, 2, 2)   # ConvTranspose2d [in,out,kh,kw]
        sd[f"decoder.{idx}.bias"] = (upscale_channels,)
        sd[f"decoder.{idx + 1}.weight"] = (upscale_channels,)         # LayerNorm2d
        sd[f"decoder.{idx + 1}.bias"] = (upscale_channels,)
        cin = upscale_channels
        idx += 3
    sd[f"decoder.{idx}.weight"] = (upscale_channels,)                  # GroupNorm
    sd[f"decoder.{idx}.bias"] = (upscale_channels,)
    conv(f"decoder.{idx + 2}", out_channels, upscale_channels)
    return sd


def vae_image_schema(old_attention_names=True):
    """Ordered key -> shape of the encoder half of the SD-1.x AutoencoderKL (GeneralVAEImage with the decoder
    removed, tools/main_ldm.py:137-139) + quant_conv.  diffusers 0.16.1 names the attention projections
    query/key/value/proj_attn; later releases to_q/to_k/to_v/to_out.0."""
    sd = OrderedDict()

    def conv(name, co, ci, k=3):
        sd[name + ".weight"] = (co, ci, k, k)
        sd[name + ".bias"] = (co,)

    def norm(name, c):
        sd[name + ".weight"] = (c,)
        sd[name + ".bias"] = (c,)

    def resnet(p, ci, co):
        norm(p + "norm1", ci)
        conv(p + "conv1", co, ci)
        norm(p + "norm2", co)
        conv(p + "conv2", co, co)
        if ci != co:
            conv(p + "conv_shortcut", co, ci, 1)

    ch = (128, 256, 512, 512)
    conv("encoder.conv_in", ch[0], 3)
    cin = ch[0]
    for i, co in enumerate(ch):
        for j in range(2):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}.", cin, co)
            cin = co
        if i < 3:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", co, co)
    resnet("encoder.mid_block.resnets.0.", 512, 512)
    ap = "encoder.mid_block.attentions.0."
    norm(ap + "group_norm", 512)
    for nm in (("query", "key", "value", "proj_attn") if old_attention_names else ("to_q", "to_k", "to_v", "to_out.0")):
        sd[ap + nm + ".weight"] = (512, 512)
        sd[ap + nm + ".bias"] = (512,)
    resnet("encoder.mid_block.resnets.1.", 512, 512)
    norm("encoder.conv_norm_out", 512)
    conv("encoder.conv_out", 8, 512)
    conv("quant_conv", 8, 8, 1)
    return sd


VAE_IMAGE_NORM_KEYS = ("encoder.mid_block.attentions.0.group_norm.weight", "encoder.mid_block.attentions.0.group_norm.bias")


def count_params(schema):
    n = 0
    for shp in schema.values():
        k = 1
        for s in shp:
            k *= s
        n += k
    return n


def _is_norm(key):
    parts = key.split(".")
    leaf = parts[-2]
    return leaf.startswith("norm") or leaf == "conv_norm_out"


def generate(schema, seed=0, device="cpu", dtype=torch.float32, norm_keys=()):
    """Draw a deterministic state dict for ``schema`` (see module docstring)."""
    out = OrderedDict()
    for key, shp in schema.items():
        g = torch.Generator(device="cpu").manual_seed((zlib.crc32(key.encode()) ^ seed) & 0x7FFFFFFF)
        is_norm = _is_norm(key) or key in norm_keys
        u = torch.rand(shp, generator=g, dtype=torch.float32) * 2 - 1
        if len(shp) == 1:
            if is_norm and key.endswith("weight"):
                t = 1.0 + 0.1 * u
            elif is_norm:
                t = 0.1 * u
            else:
                t = 0.05 * u
        else:
            if len(shp) == 4 and key.startswith("decoder.") and shp[2] == 2:
                fan_in = shp[0]               # ConvTranspose2d k2s2: one tap per output pixel
            else:
                fan_in = 1
                for s in shp[1:]:
                    fan_in *= s
            t = u * (3.0 / fan_in) ** 0.5
        out[key] = t.to(device=device, dtype=dtype)
    return out


VAE_NORM_KEYS = ("encoder.13.weight", "encoder.13.bias", "decoder.3.weight", "decoder.3.bias",
                 "decoder.6.weight", "decoder.6.bias", "decoder.8.weight", "decoder.8.bias")
