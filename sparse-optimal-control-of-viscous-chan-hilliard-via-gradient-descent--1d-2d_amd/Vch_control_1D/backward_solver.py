"""Host-side mirror of src/1D/Vch_control_1D/backward_solver.py."""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

from ._ctx import engine_for
from .config import ForwardSolverConfig

# B1:29-33: physical parameters are frozen from a DEFAULT config at import time
_cfg = ForwardSolverConfig()
c1, c2, tau, gamma = _cfg.c1, _cfg.c2, _cfg.tau, _cfg.gamma
kappa = _cfg.kappa


def fpp_log(phi, eps: float = 1e-8):
    """B1:36-46."""
    p = np.clip(phi, -1 + eps, 1 - eps)
    return 2.0 * c1 / (1.0 - p ** 2) - 2.0 * c2


def run_backward(phi_hist: np.ndarray, x: np.ndarray, t_hist: np.ndarray, b1: float, b2: float,
                 phi_Q: Optional[np.ndarray] = None,
                 phi_T_target: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """B1:48-126 -> (p, q, r), each (M+2, N+1); row 0 stays zero (B1:110)."""
    rows, n = phi_hist.shape
    h = float(x[1] - x[0])
    eng = engine_for(n - 1, h * (n - 1), max_steps=max(rows - 2, 1))
    return eng.backward(phi_hist, t_hist, b1, b2, phi_Q, phi_T_target, h=h)
