"""ctypes binding of libvch_hip.so (the C ABI declared in include/vch.h).

The shared library is built in-tree by `build()` (hipcc, gfx950) and is the ONLY compute
path of this package: there is no CPU fallback.  Loading fails loudly when the library is
missing, and every engine call fails loudly when no HIP device is present.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.environ.get("VCH_LIB") or os.path.join(HERE, "libvch_hip.so")     # VCH_LIB: A/B builds
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
SOURCES = ["vch_hip.hip"]
DEPS = ["vch_hip.hip", "vch_comm.hip", "vch_engine2d.hip", "vch_kernels2d.h", "vch_gemm.h", "vch_fft.h", "vch_common.h", "vch_engine1d.hip", "vch_kernels1d.h",
        os.path.join(ROOT, "include", "vch.h")]


class VchError(RuntimeError):
    pass


HASH_PATH = LIB_PATH + ".srchash"


BUILD_FLAGS = ["-O3", "--offload-arch=gfx950", "-shared", "-fPIC", "-std=c++17", "-Wno-unused-value", "-Wno-unused-result"]


def source_hash(defines=()) -> str:
    """sha256 over the contents of every file the shared library is built from (names included), the compiler flags and
    the -D defines of the build: a tuning build (A/B defines) never passes for the default one."""
    h = hashlib.sha256()
    h.update(" ".join(BUILD_FLAGS + sorted(f"-D{d}" for d in defines)).encode())
    for d in DEPS:
        p = d if os.path.isabs(d) else os.path.join(CSRC, d)
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _stale():
    """The library is current iff the hash recorded beside it at build time equals the hash of the sources as they
    are now (mtimes say nothing after a checkout or a copy to the GPU box)."""
    if not os.path.exists(LIB_PATH) or not os.path.exists(HASH_PATH):
        return True
    with open(HASH_PATH) as fh:
        return fh.read().strip() != source_hash()


def build(force: bool = False, verbose: bool = False, defines=(), out=None) -> str:
    """Compile the HIP engine for gfx950 into the package directory (hipcc cross-compiles
    without a GPU).  Returns the path of the shared library.  `defines`/`out`: tuning builds."""
    if out is None and os.environ.get("VCH_LIB"):
        return LIB_PATH                      # an explicitly selected prebuilt variant
    if out is None and not force and not _stale():
        return LIB_PATH
    cmd = [HIPCC] + BUILD_FLAGS + ["-I", CSRC, "-o", out or LIB_PATH] + [f"-D{d}" for d in defines] + \
          [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise VchError("hipcc failed:\n" + r.stdout + r.stderr)
    if out is None:
        with open(HASH_PATH, "w") as fh:
            fh.write(source_hash(defines) + "\n")      # with defines: differs from source_hash(), so the next default build() rebuilds
    return out or LIB_PATH


class Params2D(C.Structure):
    _fields_ = [("Nx", C.c_int32), ("Ny", C.c_int32), ("Lx", C.c_double), ("Ly", C.c_double),
                ("tau", C.c_double), ("gamma", C.c_double), ("c1", C.c_double), ("c2", C.c_double),
                ("kappa", C.c_double)]


class Params1D(C.Structure):
    _fields_ = [("N", C.c_int32), ("Lx", C.c_double), ("tau", C.c_double), ("gamma", C.c_double),
                ("c1", C.c_double), ("c2", C.c_double), ("kappa", C.c_double)]


class OptParams(C.Structure):
    _fields_ = [("b1", C.c_double), ("b2", C.c_double), ("b3", C.c_double),
                ("kappa_sparsity", C.c_double), ("alpha_max", C.c_double), ("max_iter", C.c_int32),
                ("u_min", C.c_double), ("u_max", C.c_double)]


class Stats(C.Structure):
    _fields_ = [("newton_iters", C.c_int64), ("linear_solves", C.c_int64), ("linear_iters", C.c_int64),
                ("armijo_trials", C.c_int64), ("max_lin_relres", C.c_double), ("seconds", C.c_double),
                ("max_lin_absres", C.c_double), ("host_syncs", C.c_int64), ("launches", C.c_int64),
                ("unconverged_solves", C.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_P = C.c_void_p
_D = C.POINTER(C.c_double)
_I32 = C.POINTER(C.c_int32)

# name -> (restype, argtypes); must list every symbol declared in include/vch.h
SIGNATURES = {
    "vch_last_error": (C.c_char_p, []),
    "vch_abi_version": (C.c_int, []),
    "vch_device_count": (C.c_int, []),
    "vch2d_create": (_P, [C.POINTER(Params2D), C.c_int, C.c_int, C.c_int]),
    "vch2d_destroy": (None, [_P]),
    "vch2d_batch": (C.c_int, [_P]),
    "vch2d_uses_fft": (C.c_int, [_P]),
    "vch2d_apply_laplacian": (C.c_int, [_P, _D, _D]),
    "vch2d_initialize_mu": (C.c_int, [_P, _D, _D, _D]),
    "vch2d_solve_w": (C.c_int, [_P, _D, C.c_double, _D, _D, _D]),
    "vch2d_residuals": (C.c_int, [_P, _D, _D, _D, _D, _D, _D, C.c_double, _D, _D, _D]),
    "vch2d_jacobian_apply": (C.c_int, [_P, _D, C.c_double, _D, _D, _D, _D]),
    "vch2d_jacobian_solve": (C.c_int, [_P, _D, C.c_double, _D, _D, _D, _D, C.POINTER(Stats)]),
    "vch2d_schur_apply": (C.c_int, [_P, _D, C.c_double, _D, _D]),
    "vch2d_adjoint_apply": (C.c_int, [_P, C.c_int, _D, C.c_double, _D, _D]),
    "vch2d_adjoint_solve": (C.c_int, [_P, _D, C.c_double, _D, _D, C.POINTER(Stats)]),
    "vch2d_spectral_solve": (C.c_int, [_P, C.c_double, C.c_double, C.c_double, _D, _D]),
    "vch2d_newton_raphson": (C.c_int, [_P, _D, _D, _D, _D, C.c_double, _D, _D, _D, C.c_int, _I32,
                                       C.POINTER(Stats)]),
    "vch2d_forward": (C.c_int, [_P, _D, _D, C.c_int, _D, C.c_int, _D, C.POINTER(Stats)]),
    "vch2d_backward": (C.c_int, [_P, _D, C.c_int, _D, C.c_double, C.c_double, C.c_double, C.c_double,
                                 _D, _D, _D, _D, _D, C.POINTER(Stats)]),
    "vch2d_cost": (C.c_int, [_P, _D, _D, _D, _D, C.c_int, _D, _D, _D, C.POINTER(OptParams), _D]),
    "vch2d_grad_prox": (C.c_int, [_P, _D, _D, C.c_int, _D, C.POINTER(OptParams), _D]),
    "vch2d_pgd_init": (C.c_int, [_P, _D, _D, _D, C.c_int, C.c_double, _D, C.c_int, _D, _D,
                                 C.POINTER(OptParams), _D]),
    "vch2d_pgd_iterate": (C.c_int, [_P, C.c_int, _D, _D, _I32, _D, _D]),
    "vch2d_pgd_get": (C.c_int, [_P, C.c_int, _D]),
    "vch2d_pgd_errors": (C.c_int, [_P, C.c_int, _D, _D]),
    "vch1d_pgd_errors": (C.c_int, [_P, C.c_int, _D, _D]),
    "vch2d_pgd_cost_dev": (C.c_int, [_P, C.POINTER(C.c_void_p)]),
    "vch1d_create": (_P, [C.POINTER(Params1D), C.c_int, C.c_int, C.c_int]),
    "vch1d_destroy": (None, [_P]),
    "vch1d_apply_laplacian": (C.c_int, [_P, _D, _D]),
    "vch1d_residuals": (C.c_int, [_P, _D, _D, _D, _D, _D, _D, C.c_double, _D, _D]),
    "vch1d_jacobian_solve": (C.c_int, [_P, _D, C.c_double, _D, _D, _D, _D]),
    "vch1d_adjoint_solve": (C.c_int, [_P, _D, C.c_double, _D, _D]),
    "vch1d_newton_raphson": (C.c_int, [_P, _D, _D, _D, _D, C.c_double, _D, _D, _D, C.c_int, _I32]),
    "vch1d_forward": (C.c_int, [_P, _D, _D, C.c_int, _D, C.c_int, _D, C.POINTER(Stats)]),
    "vch1d_backward": (C.c_int, [_P, _D, C.c_int, _D, C.c_double, C.c_double, C.c_double, _D, _D, _D, _D, _D]),
    "vch1d_cost": (C.c_int, [_P, _D, _D, _D, _D, C.c_int, _D, _D, C.POINTER(OptParams), _D]),
    "vch1d_grad_prox": (C.c_int, [_P, _D, _D, C.c_int, _D, C.POINTER(OptParams), _D]),
    "vch2d_free_energy": (C.c_int, [_P, _D, C.c_int, _D, C.c_double, C.c_double, C.c_double, _D]),
    "vch1d_free_energy": (C.c_int, [_P, _D, C.c_int, _D, C.c_double, C.c_double, _D]),
    "vch1d_pgd_init": (C.c_int, [_P, _D, _D, _D, _D, _D, C.c_int, _D, C.POINTER(OptParams), _D]),
    "vch1d_pgd_iterate": (C.c_int, [_P, C.c_int, _D, _D, _I32, _D, _D]),
    "vch1d_pgd_get": (C.c_int, [_P, C.c_int, _D]),
    "vch2d_counters": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "vch_comm_unique_id": (C.c_int, [C.POINTER(C.c_ubyte)]),
    "vch_comm_create": (_P, [C.POINTER(C.c_ubyte), C.c_int, C.c_int, C.c_int]),
    "vch_comm_destroy": (None, [_P]),
    "vch_comm_allreduce_cost": (C.c_int, [_P, C.POINTER(C.c_void_p), C.c_int, C.c_long, _D]),
    "vch2d_prof_begin": (C.c_int, [_P, C.c_int]),
    "vch2d_prof_end": (C.c_int, [_P, _D, C.POINTER(C.c_int64), C.c_int]),
    "vch2d_prof_spans": (C.c_int, [_P, _I32, C.POINTER(C.c_float), C.c_int]),
}

_lib = None


def load():
    """Load libvch_hip.so and bind every declared symbol; raises VchError when the library
    has not been built (run `python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VchError(f"{LIB_PATH} not found: the HIP engine has not been built "
                       "(there is no CPU fallback); call build() first")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError = ABI mismatch, by design
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return load().vch_last_error().decode("utf-8", "replace")


def check(rc: int):
    if rc < 0:
        msg = last_error()
        if rc == -1:
            raise ValueError(msg)
        raise VchError(f"engine error {rc}: {msg}")
    return rc
