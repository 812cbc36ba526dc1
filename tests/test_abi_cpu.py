"""CPU suite, part 2: the C-ABI library builds, loads and exports what include/*.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libpath():
    from ldmseg_amd import build
    return build.build_library()


def declared_symbols():
    names = set()
    for h in ("ldmseg_hip.h", "ldmseg_hip_ops.h"):
        txt = open(os.path.join(ROOT, "include", h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names |= set(re.findall(r"\b(ldmseg_[a-z0-9_]+)\s*\(", txt))
    return sorted(names)


def test_library_exports_every_declared_symbol(libpath):
    lib = ctypes.CDLL(libpath)
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/ but not exported"


def test_binding_covers_header():
    from ldmseg_amd import _lib
    assert set(_lib.SIGNATURES) == set(declared_symbols())
    h = _lib.lib()
    assert b"gfx950" in h.ldmseg_version()


def test_code_objects_are_gfx950_only(libpath):
    data = open(libpath, "rb").read()
    assert b"gfx950" in data
    for other in (b"gfx942", b"gfx90a", b"sm_90"):
        assert other not in data


def test_oracle_is_not_imported_by_product():
    pkg = os.path.join(ROOT, "latent-diffusion-segmentation_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "import oracle" not in txt and "from oracle" not in txt, os.path.join(dp, f)


def test_fastdiv_formula_is_exact():
    """The multiply-shift divisor the kernels use instead of runtime integer division (csrc/kernels.h fastdiv_make,
    common.h fd_div): s = ceil(log2 d), M = ceil(2^(31+s) / d), q = (n * M) >> (31 + s), claimed exact for
    0 <= n < 2^31.  Restated here in exact integer arithmetic and checked at the places such schemes break: around
    every multiple of d near both ends of the range, for every small divisor, the UNet's own divisors and random
    large ones.  (tests/test_ops_gpu.py::test_fastdiv_on_device runs the device code.)"""
    import random

    def make(d):
        if d == 1:
            return 0, 0
        s = (d - 1).bit_length()
        m = -(-(1 << (31 + s)) // d)
        assert m < (1 << 32), d
        return m, s - 1

    def div(n, d, m, sh):
        return n if m == 0 else ((n * m) >> 32) >> sh

    rng = random.Random(0)
    top = (1 << 31) - 1
    divisors = list(range(1, 1025)) + [4096, 16384, 65536, 160, 320, 2880, 11520, 23040, 40960, 3, 7, 9, 45, 180]
    divisors += [rng.randrange(1, 1 << 31) for _ in range(300)] + [top, top - 1, (1 << 30) + 1, (1 << 30) - 1]
    for d in divisors:
        m, sh = make(d)
        ns = {0, 1, d - 1, d, d + 1, top, top - 1}
        kmax = top // d
        for k in {1, 2, kmax // 2, kmax - 1, kmax, rng.randrange(0, kmax + 1)}:
            for off in (-1, 0, 1):
                ns.add(k * d + off)
        for _ in range(20):
            ns.add(rng.randrange(0, top + 1))
        for n in ns:
            if 0 <= n <= top:
                assert div(n, d, m, sh) == n // d, (n, d)
