"""Build libldmseg_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

The shared library is written in-tree (next to this file) so that it travels with
the repository snapshot to the GPU box; it is git-ignored.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.normpath(os.path.join(HERE, "..", "csrc"))
INCLUDE = os.path.normpath(os.path.join(HERE, "..", "..", "include"))
LIB_PATH = os.path.join(HERE, "libldmseg_hip.so")
SOURCES = ["igemm.hip", "tfuse.hip", "tproj.hip", "tail.hip", "norm.hip", "attention.hip", "attention3.hip", "attention4.hip", "attention_fp8.hip", "attention_mx.hip", "misc.hip", "postproc.hip", "sched.hip", "engine.hip", "ops_api.hip"]
EXTRA_FLAGS = {
    "sched.hip": ["-ffp-contract=off"],                       # bit-exact scheduler arithmetic
    "tail.hip": ["-ffp-contract=off"],                        # the fused step tail carries the same arithmetic
    "attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],    # scores feed VALU softmax: keep MFMA results in VGPRs
    "attention3.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans"],
    "attention4.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans"],
    "attention_fp8.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans"],
    "attention_mx.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans"],
}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _digest(cmd, deps):
    """Identity of one compile: the command line (compiler, flags, output) and the bytes of the source and every header."""
    h = hashlib.sha256("\0".join(cmd).encode())
    for d in deps:
        with open(d, "rb") as f:
            h.update(b"\0" + os.path.basename(d).encode() + b"\0" + f.read())
    return h.hexdigest()


def _current(target, stamp, digest):
    if not (os.path.exists(target) and os.path.exists(stamp)):
        return False
    with open(stamp) as f:
        return f.read().strip() == digest


def build_library(force=False, verbose=False):
    """Compile every .hip source and link the shared library. Returns its path.

    An object is rebuilt when the hash of (command line, source, headers) differs from the one recorded next to it, so a
    flag change or a touched-but-identical file behaves correctly (mtimes are not consulted)."""
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in ("common.h", "kernels.h", "igemm_tuned.inc", "sched_math.h")]
    headers += [os.path.join(INCLUDE, h) for h in ("ldmseg_hip.h", "ldmseg_hip_ops.h")]
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", s, "-o", o]
        dig = _digest(cmd, [s] + headers)
        if force or not _current(o, o + ".sha", dig):
            jobs.append((cmd, o + ".sha", dig))

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stderr[-4000:])

    def compile_one(job):
        cmd, stamp, dig = job
        if os.path.exists(stamp):
            os.remove(stamp)
        run(cmd)
        with open(stamp, "w") as f:
            f.write(dig)

    with ThreadPoolExecutor(max_workers=max(1, min(6, os.cpu_count() or 1))) as ex:
        list(ex.map(compile_one, jobs))
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    h = hashlib.sha256("\0".join(link).encode())
    for o in objs:
        with open(o + ".sha") as f:
            h.update(f.read().encode())
    ldig = h.hexdigest()
    lstamp = os.path.join(objdir, "lib.sha")
    if jobs or force or not _current(LIB_PATH, lstamp, ldig):
        run(link)
        with open(lstamp, "w") as f:
            f.write(ldig)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
