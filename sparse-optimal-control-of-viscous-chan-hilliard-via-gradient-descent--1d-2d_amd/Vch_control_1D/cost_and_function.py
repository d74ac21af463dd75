"""Host-side mirror of src/1D/Vch_control_1D/cost_and_function.py."""
from __future__ import annotations

import numpy as np

from ..engine import make_opt
from ._ctx import engine_for


def calculate_cost(phi_hist, u, phi_Q_target, phi_T_target, x, t_hist, b1, b2, b3, kappa, verbose: bool = True) -> float:
    """C1:26-84 (prints its summary regardless of `verbose`, as the reference does)."""
    rows, n = phi_hist.shape
    eng = engine_for(n - 1, float(x[-1] - x[0]), max_steps=max(rows - 2, 1))
    J = eng.cost(phi_hist, u, phi_Q_target, phi_T_target, x, t_hist, make_opt(b1=b1, b2=b2, b3=b3, kappa_sparsity=kappa))
    print(f"  Tracking Cost (J1): {J[0]:.6g}")
    print(f"  Terminal Cost (J2): {J[1]:.6g}")
    print(f"  Control Energy (J3): {J[2]:.6g}")
    print(f"  Sparsity Cost (J4): {J[3]:.6g}")
    print("-----------------------------")
    print(f"  Total Cost: {J[4]:.6g}")
    return float(J[4])


def calculate_gradient(r, u, b3: float):
    """C1:86-100."""
    return r + b3 * u


def perform_gradient_step(u_current, grad_smooth, alpha: float):
    """C1:103-112."""
    return u_current - alpha * grad_smooth
