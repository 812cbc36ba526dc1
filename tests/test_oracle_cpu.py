"""CPU suite, part 1: the oracle against the golden vectors produced by importing the reference,
and the product's host-side scheduler logic against the same vectors."""
import numpy as np
import pytest
import torch

from oracle import ddim as o_ddim, vae as o_vae, sample as o_sample, unet as o_unet
from conftest import rel_err


def test_scheduler_tables_bit_exact(golden, sched_kw):
    g = golden("scheduler.npz")
    s = o_ddim.OracleDDIM(**sched_kw)
    assert np.array_equal(s.betas.numpy(), g["betas"])
    assert np.array_equal(s.alphas_cumprod.numpy(), g["alphas_cumprod"])
    assert np.float32(s.final_alpha_cumprod) == g["final_alpha_cumprod"]
    assert float(s.alphas_cumprod[0]) == 0.9991499781608582          # SURVEY App. C known answers
    assert float(s.alphas_cumprod[999]) == 0.00466009508818388
    for sched in ("linear", "squaredcos_cap_v2", "sigmoid"):
        s2 = o_ddim.OracleDDIM(**{**sched_kw, "beta_schedule": sched})
        assert np.array_equal(s2.alphas_cumprod.numpy(), g[f"alphas_cumprod_{sched}"])


@pytest.mark.parametrize("n", [10, 50, 30, 7, 1000])
def test_scheduler_timesteps_bit_exact(golden, sched_kw, n):
    g = golden("scheduler.npz")
    s = o_ddim.OracleDDIM(**sched_kw)
    assert np.array_equal(s.timesteps.numpy(), g["timesteps_default"])
    s.set_timesteps_inference(n)
    assert s.timesteps.dtype == torch.int64
    assert np.array_equal(s.timesteps.numpy(), g[f"timesteps_{n}"])


def test_scheduler_known_grids(sched_kw):
    s = o_ddim.OracleDDIM(**sched_kw)
    s.set_timesteps_inference(50)
    assert s.timesteps.tolist() == list(range(999, 0, -20))
    s.set_timesteps_inference(10)
    assert s.timesteps.tolist() == list(range(999, 0, -100))
    s.set_timesteps_inference(50, tmin=500)
    assert len(s.timesteps) == 25 and s.timesteps[-2:].tolist() == [539, 519]


@pytest.mark.parametrize("pt", ["epsilon", "sample", "v_prediction"])
@pytest.mark.parametrize("clip", [False, True])
@pytest.mark.parametrize("ucmo", [False, True])
def test_scheduler_step_bit_exact(golden, sched_kw, pt, clip, ucmo):
    g = golden("scheduler.npz")
    s = o_ddim.OracleDDIM(**{**sched_kw, "prediction_type": pt, "clip_sample": clip})
    s.set_timesteps_inference(50)
    eps, x = torch.from_numpy(g["step_eps"]), torch.from_numpy(g["step_x"])
    key = f"{pt}_clip{int(clip)}_ucmo{int(ucmo)}"
    for i, t in enumerate(s.timesteps):
        prev, x0 = s.step(eps, t, x, use_clipped_model_output=ucmo)
        assert np.array_equal(prev.numpy(), g[f"step_prev_{key}"][i]), (key, i)
        assert np.array_equal(x0.numpy(), g[f"step_x0_{key}"][i]), (key, i)


def test_scheduler_known_step_values(sched_kw):
    s = o_ddim.OracleDDIM(**sched_kw)
    s.set_timesteps_inference(50)
    prev, x0 = s.step(torch.full((1,), -0.25), 999, torch.full((1,), 0.5))
    assert float(prev) == 0.5948936939239502 and float(x0) == 10.978072166442871
    prev, x0 = s.step(torch.full((1,), -0.25), 19, torch.full((1,), 0.5))
    assert float(prev) == 0.5305942296981812 and float(x0) == 0.5381117463111877


def test_scheduler_noise_ops(golden, sched_kw):
    g = golden("scheduler.npz")
    s = o_ddim.OracleDDIM(**sched_kw)
    x0, noise, t = (torch.from_numpy(g[k]) for k in ("an_x0", "an_noise", "an_t"))
    assert np.array_equal(s.add_noise(x0, noise, t).numpy(), g["an_out"])
    assert np.array_equal(s.add_noise(x0, noise, t, scale=0.5).numpy(), g["an_out_scale"])
    assert np.array_equal(s.remove_noise(torch.from_numpy(g["an_out"]), noise, t).numpy(), g["rn_out"])


@pytest.mark.parametrize("mode", ["none", "max_clamp_snr", "linear", "fixed"])
def test_scheduler_weights(golden, sched_kw, mode):
    g = golden("scheduler.npz")
    s = o_ddim.OracleDDIM(**{**sched_kw, "weight": mode})
    assert np.array_equal(s.weights.numpy().astype(np.float32), g[f"weights_{mode}"])


# ------------------------------------------------------------------ product host logic
def test_product_scheduler_host_logic(golden, sched_kw):
    from ldmseg_amd.schedulers import DDIMNoiseScheduler
    g = golden("scheduler.npz")
    s = DDIMNoiseScheduler(**sched_kw)
    assert np.array_equal(s.alphas_cumprod.numpy(), g["alphas_cumprod"])
    assert np.array_equal(s.betas.numpy(), g["betas"])
    assert np.float32(s.final_alpha_cumprod) == g["final_alpha_cumprod"]
    assert np.array_equal(s.timesteps.numpy(), g["timesteps_default"])
    assert len(s) == 1000 and s.init_noise_sigma == 1.0 and s.num_inference_steps is None
    for n in (10, 50, 30, 7, 1000):
        s.set_timesteps_inference(n)
        assert s.timesteps.dtype == torch.int64
        assert np.array_equal(s.timesteps.numpy(), g[f"timesteps_{n}"])
        assert s.timesteps_host() == g[f"timesteps_{n}"].tolist()
    s.set_timesteps_inference(50, tmin=500)
    assert np.array_equal(s.timesteps.numpy(), g["timesteps_50_tmin500"])
    s.set_timesteps_inference(50)
    assert np.array_equal(s.coefficient_table(), g["coef_50"])         # fp32 scalars, bit-exact
    assert s.steps_offset == 19
    for mode in ("none", "max_clamp_snr", "linear", "fixed"):
        s2 = DDIMNoiseScheduler(**{**sched_kw, "weight": mode})
        assert np.array_equal(s2.weights.numpy().astype(np.float32), g[f"weights_{mode}"])
    for sched in ("linear", "squaredcos_cap_v2", "sigmoid"):
        s2 = DDIMNoiseScheduler(**{**sched_kw, "beta_schedule": sched})
        assert np.array_equal(s2.alphas_cumprod.numpy(), g[f"alphas_cumprod_{sched}"])
    with pytest.raises(NotImplementedError):
        DDIMNoiseScheduler(**{**sched_kw, "beta_schedule": "nope"})
    assert "DDIMScheduler(" in str(s)


def test_product_scheduler_refuses_cpu_tensors(sched_kw):
    """No CPU fallback: the step must fail loudly on host tensors."""
    from ldmseg_amd.schedulers import DDIMNoiseScheduler
    s = DDIMNoiseScheduler(**sched_kw)
    s.set_timesteps_inference(10)
    with pytest.raises(RuntimeError):
        s.step(torch.zeros(1, 4, 8, 8), 999, torch.zeros(1, 4, 8, 8))


# ------------------------------------------------------------------ seg-VAE oracle
def test_vae_oracle_vs_reference_golden(golden, vae_sd):
    g = golden("vae.npz")
    assert int(g["n_params"]) == 2023208 and g["attrs"].tolist() == [8, 2, 2]
    x = torch.from_numpy(g["enc_x"])
    with torch.no_grad():
        mom = o_vae.encode_moments(vae_sd, x)
        assert rel_err(mom, g["enc_moments"]) < 1e-5
        assert rel_err(o_vae.encode_mode(vae_sd, x), g["enc_mode"]) < 1e-5
        assert rel_err(o_vae.encode_sample(vae_sd, x, torch.from_numpy(g["enc_noise"])), g["enc_sample"]) < 1e-5
        z = torch.from_numpy(g["dec_z"])
        ck = {}
        assert rel_err(o_vae.decode(vae_sd, z, interpolate=False, checkpoints=ck), g["dec_logits_4L"]) < 1e-5
        assert rel_err(ck["convt2"], g["dec_after_convt2"]) < 1e-5
        assert rel_err(o_vae.decode(vae_sd, z, interpolate=True), g["dec_logits_8L"]) < 1e-5
        fw = o_vae.decode(vae_sd, o_vae.encode_mode(vae_sd, x), interpolate=False)
        assert rel_err(fw, g["fwd_sample"]) < 1e-5


# ------------------------------------------------------------------ sampling loop oracle
@pytest.mark.parametrize("n", [10, 50])
def test_sample_loop_vs_reference_scheduler(golden, sched_kw, n):
    g = golden("sample_loop.npz")
    wmix = torch.from_numpy(g["wmix"])

    def eps_net(inp, t):
        return torch.tanh(torch.nn.functional.conv2d(inp, wmix, padding=1)) * (1.0 + float(t) / 1000.0)

    s = o_ddim.OracleDDIM(**sched_kw)
    s.set_timesteps_inference(n)
    out = o_sample.sample(eps_net, s, torch.from_numpy(g[f"rgb_{n}"]), seed=42, self_condition=True)
    assert np.array_equal(out.numpy(), g[f"final_{n}"])


def test_sample_loop_noise_semantics():
    a = o_sample.initial_noise(2, 8, 42)
    b = o_sample.initial_noise(2, 8, 42)
    assert torch.equal(a, b)                       # same draw for every batch of a size (:1088-1091)
    from ldmseg_amd.trainers import TrainerDiffusion
    assert torch.equal(TrainerDiffusion.draw_noise(2, 8, 42), a)


# ------------------------------------------------------------------ UNet oracle (structural pins)
def test_unet_schema_counts():
    from ldmseg_amd import weights
    s = weights.unet_schema(4, True)
    assert len(s) == 686 and weights.count_params(s) == 859_520_964      # vanilla SD-1.x
    s = weights.unet_schema(8, False)
    assert len(s) == 574 and weights.count_params(s) == 815_544_964
    s = weights.unet_schema(12, False)
    assert len(s) == 574 and weights.count_params(s) == 815_556_484
    assert s["conv_in.weight"] == (320, 12, 3, 3)
    assert s["up_blocks.1.resnets.2.conv1.weight"] == (1280, 1920, 3, 3)
    assert s["up_blocks.3.resnets.0.conv_shortcut.weight"] == (320, 960, 1, 1)
    assert "down_blocks.3.attentions.0.norm.weight" not in s
    assert not any("attn2" in k or "norm2" in k.split("transformer_blocks.0.")[-1] and "transformer_blocks" in k
                   for k in s)


def test_timestep_embedding_layout():
    e = o_unet.timestep_embedding(torch.tensor([0, 999]))
    assert e.shape == (2, 320)
    assert torch.all(e[0, :160] == 1.0) and torch.all(e[0, 160:] == 0.0)   # cos first (flip_sin_to_cos)
    assert abs(float(e[1, 160]) - float(torch.sin(torch.tensor(999.0)))) < 1e-6


@pytest.mark.timeout(600)
def test_unet_oracle_structure(unet_sd):
    torch.manual_seed(0)
    x = torch.randn(1, 12, 8, 8)
    with torch.no_grad():
        y = o_unet.unet_forward(unet_sd, x, torch.tensor(500))
        assert y.shape == (1, 4, 8, 8) and torch.isfinite(y).all()
        # per-sample timesteps == broadcast scalar
        y2 = o_unet.unet_forward(unet_sd, x.repeat(2, 1, 1, 1), torch.tensor([500, 500]))
        assert rel_err(y2[1:], y) < 1e-5
        # zero conv_out weight -> output equals its bias everywhere
        sd = dict(unet_sd)
        sd["conv_out.weight"] = torch.zeros_like(sd["conv_out.weight"])
        y0 = o_unet.unet_forward(sd, x, torch.tensor(500))
        assert torch.allclose(y0, sd["conv_out.bias"][None, :, None, None].expand_as(y0))


def test_unet_oracle_blocks_against_independent_torch_modules():
    """The UNet oracle is unpinned by the reference (diffusers absent); its building blocks are cross-checked against
    torch's own, independently written modules: multi-head attention (nn.MultiheadAttention uses the same
    channel = head*d + i split), GroupNorm/SiLU/conv resnet arithmetic (nn modules), exact-erf GEGLU."""
    import torch.nn as nn
    import torch.nn.functional as F
    from oracle import unet as o_unet
    g = torch.Generator().manual_seed(0)
    C, N, B = 320, 24, 2
    p = "a."
    sd = {p + k: torch.randn(C, C, generator=g) / C ** 0.5 for k in ("to_q.weight", "to_k.weight", "to_v.weight", "to_out.0.weight")}
    sd[p + "to_out.0.bias"] = torch.randn(C, generator=g) * 0.1
    x = torch.randn(B, N, C, generator=g)
    mha = nn.MultiheadAttention(C, 8, bias=True, batch_first=True)
    with torch.no_grad():
        mha.in_proj_weight.copy_(torch.cat([sd[p + "to_q.weight"], sd[p + "to_k.weight"], sd[p + "to_v.weight"]]))
        mha.in_proj_bias.zero_()
        mha.out_proj.weight.copy_(sd[p + "to_out.0.weight"])
        mha.out_proj.bias.copy_(sd[p + "to_out.0.bias"])
        ref = mha(x, x, x, need_weights=False)[0]
    assert torch.allclose(o_unet.attention(sd, p, x), ref, atol=2e-5, rtol=1e-4)

    # ResnetBlock2D arithmetic with nn modules (time embedding projected and broadcast over pixels)
    cin, cout = 64, 96
    mods = dict(norm1=nn.GroupNorm(32, cin, eps=1e-5), conv1=nn.Conv2d(cin, cout, 3, padding=1), temb=nn.Linear(1280, cout),
                norm2=nn.GroupNorm(32, cout, eps=1e-5), conv2=nn.Conv2d(cout, cout, 3, padding=1), sc=nn.Conv2d(cin, cout, 1))
    rs = {}
    for name, key in (("norm1", "norm1"), ("conv1", "conv1"), ("temb", "time_emb_proj"), ("norm2", "norm2"), ("conv2", "conv2"),
                      ("sc", "conv_shortcut")):
        for pn, t in mods[name].named_parameters():
            with torch.no_grad():
                t.copy_(torch.randn(t.shape, generator=g) * (0.05 if t.dim() > 1 else 0.2) + (1.0 if "norm" in name and pn == "weight" else 0.0))
            rs["r." + key + "." + pn] = t.detach()
    xx = torch.randn(2, cin, 8, 8, generator=g)
    emb = torch.randn(2, 1280, generator=g)
    with torch.no_grad():
        h = mods["conv1"](F.silu(mods["norm1"](xx))) + mods["temb"](F.silu(emb))[:, :, None, None]
        h = mods["conv2"](F.silu(mods["norm2"](h)))
        ref = mods["sc"](xx) + h
    assert torch.allclose(o_unet.resnet(rs, "r.", xx, emb), ref, atol=2e-5, rtol=1e-4)
