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


def test_rccl_refuses_more_ranks_than_devices():
    ndev = torch.cuda.device_count()
    if ndev >= 2:
        pytest.skip("needs a box with fewer GPUs than ranks")
    r = run_bench(2, {"LDMSEG_BENCH_BACKEND": "nccl"}, 29542)
    assert r.returncode != 0
    assert "visible devices" in (r.stderr + r.stdout)
