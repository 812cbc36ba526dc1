"""Build libldmseg_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

The shared library is written in-tree (next to this file) so that it travels with
the repository snapshot to the GPU box; it is git-ignored.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.normpath(os.path.join(HERE, "..", "csrc"))
INCLUDE = os.path.normpath(os.path.join(HERE, "..", "..", "include"))
LIB_PATH = os.path.join(HERE, "libldmseg_hip.so")
SOURCES = ["igemm.hip", "norm.hip", "attention.hip", "attention3.hip", "attention_fp8.hip", "misc.hip", "postproc.hip", "sched.hip", "engine.hip", "ops_api.hip"]
EXTRA_FLAGS = {
    "sched.hip": ["-ffp-contract=off"],                       # bit-exact scheduler arithmetic
    "attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],    # scores feed VALU softmax: keep MFMA results in VGPRs
    "attention3.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans"],
    "attention_fp8.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans"],
}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    """Compile every .hip source and link the shared library. Returns its path."""
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in ("common.h", "kernels.h", "igemm_tuned.inc")]
    headers += [os.path.join(INCLUDE, h) for h in ("ldmseg_hip.h", "ldmseg_hip_ops.h")]
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stderr[-4000:])

    with ThreadPoolExecutor(max_workers=max(1, min(6, os.cpu_count() or 1))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _stale(LIB_PATH, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
