"""Oracle: SD-1.x UNet2DConditionModel forward as LDMSeg instantiates it.
TEST INFRASTRUCTURE (see oracle/__init__.py).  **PARITY UNPINNED**: the block
arithmetic belongs to diffusers==0.16.1 (pinned in
/root/reference/data/environment.yml:50), which is absent here.  This file
restates its published architecture with
torch.nn.functional primitives and follows the reference's own control flow:

  forward            <- /root/reference/ldmseg/models/unet.py:281-436
  conv_in surgery    <- unet.py:178-233 (conv_in takes 8|12 channels)
  cross-attn removal <- unet.py:83-105  (attn2 / norm2 absent -> skipped)

Weights come in as a state dict with the reference's (= diffusers') key names.
"""
import math

import torch
import torch.nn.functional as F

BLOCK_OUT = (320, 640, 1280, 1280)
HEADS = 8
GROUPS = 32


def timestep_embedding(timesteps, dim=320):
    """diffusers Timesteps(320, flip_sin_to_cos=True, freq_shift=0): [cos | sin]."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=timesteps.device) / half
    arg = timesteps.float()[:, None] * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(arg), torch.sin(arg)], dim=-1)


def resnet(sd, p, x, emb):
    h = F.silu(F.group_norm(x, GROUPS, sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps=1e-5))
    h = F.conv2d(h, sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1)
    t = F.linear(F.silu(emb), sd[p + "time_emb_proj.weight"], sd[p + "time_emb_proj.bias"])
    h = h + t[:, :, None, None]
    h = F.silu(F.group_norm(h, GROUPS, sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps=1e-5))
    h = F.conv2d(h, sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1)
    if p + "conv_shortcut.weight" in sd:
        x = F.conv2d(x, sd[p + "conv_shortcut.weight"], sd[p + "conv_shortcut.bias"])
    return x + h


def attention(sd, p, x, ctx=None):
    """diffusers Attention: heads=8, channel c = head*d + i, scale d^-0.5."""
    B, N, C = x.shape
    d = C // HEADS
    src = x if ctx is None else ctx
    q = F.linear(x, sd[p + "to_q.weight"])
    k = F.linear(src, sd[p + "to_k.weight"])
    v = F.linear(src, sd[p + "to_v.weight"])
    q = q.view(B, N, HEADS, d).transpose(1, 2)
    k = k.view(B, -1, HEADS, d).transpose(1, 2)
    v = v.view(B, -1, HEADS, d).transpose(1, 2)
    # (query rows are independent: long sequences - 16384 tokens at 1024x1024 - go through in chunks so that the score
    # matrix stays near 1 GB; the arithmetic per row is unchanged)
    chunk = N if N <= 4096 else 2048
    outs = []
    for q0 in range(0, N, chunk):
        s = torch.softmax((q[:, :, q0:q0 + chunk] @ k.transpose(-1, -2)) * (d ** -0.5), dim=-1)
        outs.append(s @ v)
    o = (outs[0] if len(outs) == 1 else torch.cat(outs, dim=2)).transpose(1, 2).reshape(B, N, C)
    return F.linear(o, sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])


def transformer(sd, p, x, ctx=None):
    B, C, H, W = x.shape
    res = x
    h = F.group_norm(x, GROUPS, sd[p + "norm.weight"], sd[p + "norm.bias"], eps=1e-6)
    h = F.conv2d(h, sd[p + "proj_in.weight"], sd[p + "proj_in.bias"])
    h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    b = p + "transformer_blocks.0."
    n = F.layer_norm(h, (C,), sd[b + "norm1.weight"], sd[b + "norm1.bias"], eps=1e-5)
    h = attention(sd, b + "attn1.", n) + h
    if b + "attn2.to_q.weight" in sd:
        n = F.layer_norm(h, (C,), sd[b + "norm2.weight"], sd[b + "norm2.bias"], eps=1e-5)
        h = attention(sd, b + "attn2.", n, ctx) + h
    n = F.layer_norm(h, (C,), sd[b + "norm3.weight"], sd[b + "norm3.bias"], eps=1e-5)
    g = F.linear(n, sd[b + "ff.net.0.proj.weight"], sd[b + "ff.net.0.proj.bias"])
    a, gate = g.chunk(2, dim=-1)
    g = a * F.gelu(gate)                       # exact (erf) GELU
    h = F.linear(g, sd[b + "ff.net.2.weight"], sd[b + "ff.net.2.bias"]) + h
    h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
    h = F.conv2d(h, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"])
    return h + res


def unet_forward(sd, sample, timestep, encoder_hidden_states=None, taps=None):
    """sample [B, 8|12, L, L] fp32, timestep 0-d or [B] int tensor -> [B,4,L,L]."""
    B = sample.shape[0]
    t = torch.as_tensor(timestep, device=sample.device).reshape(-1).expand(B)
    emb = timestep_embedding(t).to(sd["time_embedding.linear_1.weight"].dtype)   # (weights cast to bf16 / moved to a GPU: tools/yardstick.py)
    emb = F.linear(emb, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])
    emb = F.linear(F.silu(emb), sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])

    h = F.conv2d(sample, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
    skips = [h]
    for i in range(4):
        for j in range(2):
            h = resnet(sd, f"down_blocks.{i}.resnets.{j}.", h, emb)
            if i < 3:
                h = transformer(sd, f"down_blocks.{i}.attentions.{j}.", h, encoder_hidden_states)
            skips.append(h)
        if i < 3:
            h = F.conv2d(h, sd[f"down_blocks.{i}.downsamplers.0.conv.weight"],
                         sd[f"down_blocks.{i}.downsamplers.0.conv.bias"], stride=2, padding=1)
            skips.append(h)
        if taps is not None:
            taps[f"down{i}"] = h

    h = resnet(sd, "mid_block.resnets.0.", h, emb)
    h = transformer(sd, "mid_block.attentions.0.", h, encoder_hidden_states)
    h = resnet(sd, "mid_block.resnets.1.", h, emb)
    if taps is not None:
        taps["mid"] = h

    for i in range(4):
        for j in range(3):
            h = torch.cat([h, skips.pop()], dim=1)
            h = resnet(sd, f"up_blocks.{i}.resnets.{j}.", h, emb)
            if i > 0:
                h = transformer(sd, f"up_blocks.{i}.attentions.{j}.", h, encoder_hidden_states)
        if i < 3:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = F.conv2d(h, sd[f"up_blocks.{i}.upsamplers.0.conv.weight"],
                         sd[f"up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
        if taps is not None:
            taps[f"up{i}"] = h

    h = F.silu(F.group_norm(h, GROUPS, sd["conv_norm_out.weight"], sd["conv_norm_out.bias"], eps=1e-5))
    return F.conv2d(h, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)
