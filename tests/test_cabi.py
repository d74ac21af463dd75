"""CPU-only checks of the C-ABI boundary: the shared library builds for gfx950, loads, and
exports every symbol include/vch.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def V():
    import vch_amd
    vch_amd.build()
    return vch_amd


def _declared():
    src = open(os.path.join(ROOT, "include", "vch.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(vch(?:1d|2d)?_[a-z0-9_]+)\s*\(", src)
    return sorted(set(n for n in names if not n.endswith("_params") and n not in ("vch_opt_params", "vch_stats")))


def test_every_declared_symbol_is_exported_and_bound(V):
    lib = ctypes.CDLL(V.LIB_PATH)
    from importlib import import_module
    sigs = import_module(V.PKG_NAME + "._lib").SIGNATURES if hasattr(V, "PKG_NAME") else None
    declared = _declared()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/vch.h but not exported"
        if sigs is not None:
            assert name in sigs, f"{name} has no ctypes signature in _lib.py"


def test_abi_version_and_error_string(V):
    lib = V.load()
    assert lib.vch_abi_version() == 3
    assert isinstance(lib.vch_last_error(), bytes)


def test_no_cpu_fallback_without_device(V):
    """Without a visible HIP device the product path must fail loudly, not fall back."""
    lib = V.load()
    if lib.vch_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(V.VchError):
        V.Engine2D(Nx=16, Ny=16)


def test_struct_layouts_match_header():
    import vch_amd
    L = __import__("importlib").import_module(vch_amd.PKG_NAME + "._lib")
    assert ctypes.sizeof(L.Params2D) == 2 * 4 + 7 * 8
    assert ctypes.sizeof(L.OptParams) == 5 * 8 + 8 + 2 * 8      # int32 + padding before u_min
    assert ctypes.sizeof(L.Stats) == 10 * 8           # ABI version 3: + unconverged_solves


def test_time_grid_rule(V):
    import numpy as np
    t, dts = V.time_grid(0.045, 1e-2)
    assert len(dts) == 5 and abs(dts[-1] - 0.005) < 1e-15 and t[-1] == 0.045
    t, dts = V.time_grid(1.0, 1e-3)
    assert len(dts) == 1000 and t[-1] == 1.0
