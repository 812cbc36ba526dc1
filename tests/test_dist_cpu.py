"""world_size-2 gloo test of the data-parallel decomposition: images shard over ranks,
every rank runs its shard locally, one all-gather of the final latents.  The per-rank sampler is
stubbed with a deterministic CPU function so that the SHARDING / NOISE / GATHER logic of
TrainerDiffusion.sample_sharded is what is tested (the GPU path is covered by -m gpu tests)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_sample(self, prompts, num_inference_steps=50, seed=None, rgb_latents=None, scheduler=None, latents=None, **kw):
    # stand-in for the HIP loop: any deterministic per-image function of (noise, rgb)
    return torch.tanh(latents + 3.0 * rgb_latents.cpu()) * num_inference_steps


def _worker(rank, world, port, mode, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "latent-diffusion-segmentation_amd"))
    from ldmseg_amd.trainers.sampler import TrainerDiffusion
    tr = TrainerDiffusion.__new__(TrainerDiffusion)
    tr.device = torch.device("cpu")
    TrainerDiffusion.sample = _fake_sample
    B, L = 4, 8
    rgb = torch.randn(B, 4, L, L, generator=torch.Generator().manual_seed(1234))
    out = tr.sample_sharded([""] * B, num_inference_steps=7, seed=42, rgb_latents=rgb, noise_mode=mode)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["reference", "global"])
@pytest.mark.timeout(300)
def test_sharded_sampling_gloo(mode):
    world = 2
    ctx = mp.get_context("spawn")
    res = None
    for attempt in range(3):        # (a rendezvous port can be taken between _free_port() and the store's bind: new port, once or twice)
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, mode, q)) for r in range(world)]
        for p in procs:
            p.start()
        try:
            got = dict(q.get(timeout=120) for _ in range(world))
        except Exception:
            got = None
        for p in procs:
            p.join(60)
            if p.is_alive():
                p.kill()
                p.join(10)
        if got is not None and all(p.exitcode == 0 for p in procs):
            res = got
            break
    assert res is not None, "two gloo ranks did not complete in three attempts"
    B, L = 4, 8
    rgb = torch.randn(B, 4, L, L, generator=torch.Generator().manual_seed(1234))
    g = torch.Generator().manual_seed(42)
    if mode == "global":
        noise = torch.randn((B, 4, L, L), generator=g)
    else:  # reference: every rank draws the same randn(B/W) (trainers_ldm_cond.py:1088-1091)
        noise = torch.randn((B // world, 4, L, L), generator=g).repeat(world, 1, 1, 1)
    expect = torch.tanh(noise + 3.0 * rgb) * 7
    assert torch.equal(res[0], res[1])            # every rank holds the gathered batch
    assert torch.equal(res[0], expect)


def test_bench_direct_form_spawns_ranks_that_refuse_without_a_gpu():
    """`python bench.py --gpus 2` with no launcher re-executes itself under torch.distributed.run.  Here
    (no GPU) the spawned ranks must refuse loudly - there is no CPU fallback - and the RCCL form must refuse before spawning."""
    import os, subprocess, sys
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("the GPU suite runs the direct form for real (tests/test_bench_gpu.py)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=dict(env, LDMSEG_BENCH_BACKEND="nccl"))
    assert r.returncode != 0 and "visible devices" in r.stderr
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=dict(env, LDMSEG_BENCH_BACKEND="gloo"))
    assert r.returncode != 0 and "needs an MI355X" in (r.stderr + r.stdout)       # both ranks started and refused
