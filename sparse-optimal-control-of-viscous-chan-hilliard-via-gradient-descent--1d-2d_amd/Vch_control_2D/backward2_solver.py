"""Host-side mirror of the reference's adjoint module (src/2D/Vch_control_2D/
backward2_solver.py): `run_backward` with the same signature, checks and return tuple; the
sweep itself runs on the GPU (vch2d_backward)."""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

from ._ctx import engine_for_config
from .config import ForwardSolverConfig


def fpp_log(phi, c1, c2, eps: float = 1e-8):
    """f''(phi) = 2 c1/(1 - p^2) - 2 c2 with p clipped to +-(1-eps) (B2:41-72); host helper for
    API parity, the engine evaluates it inside its adjoint kernels."""
    p = np.clip(phi, -1.0 + eps, 1.0 - eps)
    return 2.0 * c1 / (1.0 - p * p) - 2.0 * c2


def run_backward(phi_hist: np.ndarray, x: np.ndarray, y: np.ndarray, t_hist: np.ndarray,
                 config: ForwardSolverConfig, b1: float, b2: float, phi_Q: Optional[np.ndarray] = None,
                 phi_T_target: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Adjoint sweep -> (p, q, r), each (M+1, Nx+1, Ny+1) (B2:75-246).  The shape checks of
    B2:141-145 raise AssertionError like the reference."""
    assert phi_hist.ndim == 3, "phi_hist must be (M+1, Nx+1, Ny+1)"
    M1, nx1, ny1 = phi_hist.shape
    assert x.ndim == 1 and y.ndim == 1, "x and y must be 1D arrays"
    assert x.size >= 2 and y.size >= 2, "x and y must have at least 2 points"
    assert t_hist.ndim == 1 and t_hist.shape[0] == M1, "t_hist must align with phi_hist"
    hx, hy = float(x[1] - x[0]), float(y[1] - y[0])
    # grid taken from the history and the coordinate arrays, as in B2:152-155
    cfg = config.model_copy(update=dict(Nx=nx1 - 1, Ny=ny1 - 1, Lx=hx * (nx1 - 1), Ly=hy * (ny1 - 1))) \
        if hasattr(config, "model_copy") else config
    eng = engine_for_config(cfg, max_steps=max(M1 - 1, 1))
    p, q, r, _ = eng.backward(phi_hist, t_hist, b1, b2, phi_Q, phi_T_target, hx=hx, hy=hy)
    return p, q, r
