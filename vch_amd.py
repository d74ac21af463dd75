"""Import shim: the package directory required by the build contract
(`sparse-optimal-control-of-viscous-chan-hilliard-via-gradient-descent--1d-2d_amd/`) is not a
valid identifier, so it is loaded with importlib and re-exported here.

    import vch_amd
    eng = vch_amd.Engine2D(Nx=128, Ny=128)
    F2 = vch_amd.module("Vch_control_2D.Forward2_solver")
"""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

PKG_NAME = "sparse-optimal-control-of-viscous-chan-hilliard-via-gradient-descent--1d-2d_amd"
pkg = importlib.import_module(PKG_NAME)
globals().update({k: getattr(pkg, k) for k in pkg.__all__})


def module(rel):
    """Import a sub-module of the package, e.g. module("Vch_control_2D.Forward2_solver")."""
    return importlib.import_module(PKG_NAME + "." + rel)
