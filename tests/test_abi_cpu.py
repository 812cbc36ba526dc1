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
