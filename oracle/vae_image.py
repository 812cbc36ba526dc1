"""TEST INFRASTRUCTURE ONLY - torch-CPU restatement of the RGB image encoder the reference uses as
`encode_func` (ldmseg/models/vae.py:36-39 GeneralVAEImage(AutoencoderKL), decoder removed at
tools/main_ldm.py:137-139; called from trainers_ldm_cond.py:360-375).  Product code must not import this.

PARITY UNPINNED: the arithmetic lives in diffusers==0.16.1 (data/environment.yml:50), which is neither vendored
by the reference nor installed here.  Restated from its published architecture (models/vae.py Encoder,
models/unet_2d_blocks.py DownEncoderBlock2D / UNetMidBlock2D, models/resnet.py ResnetBlock2D / Downsample2D,
models/attention.py AttentionBlock) for the SD-1.x VAE config: block_out_channels (128,256,512,512),
layers_per_block 2, norm_num_groups 32, resnet eps 1e-6, act silu, one attention head over 512 channels,
double_z.  Structural pin: 34,163,592 encoder parameters + 72 for quant_conv (tests/test_oracle_cpu.py).
"""
import math

import torch
import torch.nn.functional as F


def _gn(sd, p, x, silu):
    h = F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps=1e-6)
    return F.silu(h) if silu else h


def _resnet(sd, p, x):
    h = F.conv2d(_gn(sd, p + "norm1", x, True), sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1)
    h = F.conv2d(_gn(sd, p + "norm2", h, True), sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1)
    if p + "conv_shortcut.weight" in sd:
        x = F.conv2d(x, sd[p + "conv_shortcut.weight"], sd[p + "conv_shortcut.bias"])
    return x + h                                           # output_scale_factor = 1


def _attention(sd, p, x):
    old = p + "query.weight" in sd
    qn, kn, vn, pn = ("query", "key", "value", "proj_attn") if old else ("to_q", "to_k", "to_v", "to_out.0")
    B, C, H, W = x.shape
    h = _gn(sd, p + "group_norm", x, False).view(B, C, H * W).transpose(1, 2)      # [B, HW, C]
    lin = lambda n, t: F.linear(t, sd[p + n + ".weight"].reshape(C, C), sd[p + n + ".bias"])
    q, k, v = lin(qn, h), lin(kn, h), lin(vn, h)
    scale = 1.0 / math.sqrt(C / 1)                         # one head of width C
    attn = torch.softmax(torch.bmm(q, k.transpose(1, 2)) * scale, dim=-1)
    o = lin(pn, torch.bmm(attn, v)).transpose(1, 2).reshape(B, C, H, W)
    return o + x                                           # rescale_output_factor = 1


def encode_moments(sd, x):
    """x [B,3,H,W] (already 2*img-1) -> moments [B,8,H/8,W/8] = quant_conv(encoder(x))."""
    h = F.conv2d(x, sd["encoder.conv_in.weight"], sd["encoder.conv_in.bias"], padding=1)
    for i in range(4):
        for j in range(2):
            h = _resnet(sd, f"encoder.down_blocks.{i}.resnets.{j}.", h)
        if i < 3:
            d = f"encoder.down_blocks.{i}.downsamplers.0.conv."
            h = F.conv2d(F.pad(h, (0, 1, 0, 1)), sd[d + "weight"], sd[d + "bias"], stride=2)
    h = _resnet(sd, "encoder.mid_block.resnets.0.", h)
    h = _attention(sd, "encoder.mid_block.attentions.0.", h)
    h = _resnet(sd, "encoder.mid_block.resnets.1.", h)
    h = F.conv2d(_gn(sd, "encoder.conv_norm_out", h, True), sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"],
                 padding=1)
    return F.conv2d(h, sd["quant_conv.weight"], sd["quant_conv.bias"])


def encode_mode(sd, images, scaling_factor=0.18215):
    """encode_inputs(sample_posterior=False) for RGB images in [0,1] (trainers_ldm_cond.py:369-375, 391-393)."""
    mom = encode_moments(sd, 2.0 * images - 1.0)
    return mom[:, :4] * scaling_factor
