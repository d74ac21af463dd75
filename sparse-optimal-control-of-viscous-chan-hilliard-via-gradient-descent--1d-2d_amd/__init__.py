"""MI355X-native proximal-gradient optimal-control engine for the viscous Cahn-Hilliard
system: hand-written HIP kernels (csrc/) behind a C ABI (include/vch.h), and a host-side
mirror of the reference's Python interface (Vch_control_2D/, Vch_control_1D/).

The directory name is not a valid Python identifier; import it through the `vch_amd`
shim at the repository root (`import vch_amd`) or with importlib.
"""
from ._lib import VchError, build, load, LIB_PATH          # noqa: F401
from . import parallel                                         # noqa: F401
from .engine import Engine1D, Engine2D, make_opt, time_grid  # noqa: F401

__all__ = ["parallel", "VchError", "build", "load", "LIB_PATH", "Engine1D", "Engine2D", "make_opt", "time_grid"]
