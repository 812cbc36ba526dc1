"""bench.py's multi-rank control flow (launch contract, barrier / max-over-ranks timing, sharded sampling + all-gather,
rank-0 JSON line) on whatever box runs the GPU suite.  The driver's box has ONE GPU, so two ranks share it over gloo
(LDMSEG_BENCH_BACKEND=gloo): the line must say so - `n_gpus` counts distinct devices, never ranks - and the RCCL
configuration must refuse to start when there are fewer devices than ranks instead of remapping silently."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(nproc, extra_env, port):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", str(nproc), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--profile-steps", "0"]
    return subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)


def test_two_ranks_over_gloo_share_the_device_and_say_so():
    r = run_bench(2, {"LDMSEG_BENCH_BACKEND": "gloo"}, 29541)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]            # rank 0 prints ONE line
    out = json.loads(lines[0])
    ndev = torch.cuda.device_count()
    assert out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["ranks"] == 2 and out["config"]["backend"] == "gloo"
    assert out["config"]["global_batch"] == 16 and out["config"]["batch_per_gpu"] == 8
    assert out["n_gpus"] == min(2, ndev)
    assert out["config"]["ranks_share_device"] == (ndev < 2)
    if ndev < 2:
        assert "NOT a multi-GPU measurement" in out["note"]
    assert out["value"] > 0 and out["images_per_s_50step_ddim_incl_decode"] > 0
    # the decomposition a multi-GPU run is read by: every rank's own step time and the one collective on its own
    assert len(out["per_rank_ms_per_step"]) == 2 and max(out["per_rank_ms_per_step"]) == pytest.approx(out["ms_per_step"], rel=1e-6)
    assert out["allgather_latents"]["bytes_per_rank"] == 8 * 4 * 64 * 64 * 4 and out["allgather_latents"]["us"] > 0
    # two PROCESSES share the device here: the cooperative GroupNorm grids of one cannot count on co-residency.  Every rank's
    # latents of every timed region must be finite (all-reduced MIN over ranks in bench.py) - VERDICT r03 weak 9
    assert out["finite"] is True
    assert out["timed_regions"]["n"] == 3 and len(out["timed_regions"]["ms_per_step"]) == 3
    assert out["timed_regions"]["ms_per_step"][0] == pytest.approx(out["ms_per_step"], rel=1e-3)
    assert out["timed_regions"]["min_ms_per_step"] <= out["timed_regions"]["median_ms_per_step"]


def test_two_ranks_over_rccl_when_two_devices_are_visible():
    """The measured configuration - one rank per GPU, RCCL all-gather of the latents over xGMI - whenever the box has at
    least two devices (the driver's GPU-test box has one: skipped there; the 8-GPU scaling run is the driver's)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible devices")
    r = run_bench(2, {"LDMSEG_BENCH_BACKEND": "nccl"}, 29543)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 2 and out["config"]["backend"] == "nccl" and not out["config"]["ranks_share_device"]
    assert out["config"]["global_batch"] == 16 and len(out["per_rank_ms_per_step"]) == 2
    assert out["allgather_latents"]["backend"] == "nccl" and out["allgather_latents"]["us"] > 0
    assert out["value"] > 0 and "note" not in out


def test_rccl_refuses_more_ranks_than_devices():
    ndev = torch.cuda.device_count()
    if ndev >= 2:
        pytest.skip("needs a box with fewer GPUs than ranks")
    r = run_bench(2, {"LDMSEG_BENCH_BACKEND": "nccl"}, 29542)
    assert r.returncode != 0
    assert "visible devices" in (r.stderr + r.stdout)


def test_direct_form_spawns_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher (the form the driver's 1-GPU command suggests) re-executes itself under
    torch.distributed.run - VERDICT r04 item 4; the reference's entry spawns its ranks too (tools/main_ldm.py:59-69,108-111)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", LDMSEG_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
           "--no-extras", "--no-images", "--profile-steps", "0", "--repeats", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["config"]["ranks"] == 2 and out["config"]["global_batch"] == 16 and out["finite"] is True
    assert out["n_gpus"] == min(2, torch.cuda.device_count()) and out["value"] > 0


def test_direct_form_refuses_rccl_without_enough_devices():
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer GPUs than ranks")
    env = dict(os.environ, LDMSEG_BENCH_BACKEND="nccl")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and "visible devices" in r.stderr
