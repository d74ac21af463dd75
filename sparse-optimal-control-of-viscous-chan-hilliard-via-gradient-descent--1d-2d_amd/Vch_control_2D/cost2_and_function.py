"""Host-side mirror of src/2D/Vch_control_2D/cost2_and_function.py: cost, gradient and
proximal step with the reference's signatures, evaluated by the HIP engine."""
from __future__ import annotations

import numpy as np

from ..engine import Engine2D, make_opt
from ._ctx import engine_for
from .config import OptimizationConfig

PRINT_COST_SUMMARY = True      # the reference prints six lines per call (C2:113-118)


def _engine_like(arr, x=None, y=None) -> Engine2D:
    nx1, ny1 = arr.shape[-2:]
    Lx = float(x[-1] - x[0]) if x is not None else 1.0
    Ly = float(y[-1] - y[0]) if y is not None else 1.0
    return engine_for(nx1 - 1, ny1 - 1, Lx, Ly, 0.05, 10.0, 0.75, 1.0, 1e-4, max_steps=max(arr.shape[0] - 1, 1))


def calculate_cost(phi_hist, u, phi_Q_target, phi_T_target, x, y, t_hist, opt_config: OptimizationConfig) -> float:
    """J = J1 + J2 + J3 + J4 by nested trapezoid rules (C2:19-120)."""
    eng = _engine_like(phi_hist, x, y)
    J = eng.cost(phi_hist, u, phi_Q_target, phi_T_target, t_hist, opt_config, x, y)
    if PRINT_COST_SUMMARY:
        print(f"  Tracking Cost (J1): {J[0]:.6g}")
        print(f"  Terminal Cost (J2): {J[1]:.6g}")
        print(f"  Control Energy (J3): {J[2]:.6g}")
        print(f"  Sparsity Cost (J4): {J[3]:.6g}")
        print("  -----------------------------")
        print(f"  Total Cost:         {J[4]:.6g}")
    return float(J[4])


def calculate_gradient(r, u, opt_config: OptimizationConfig):
    """r + b3 u (C2:123-150).  One fused axpy; the PGD loop never materialises it (the engine's
    grad+prox kernel consumes r and u directly), so this stand-alone form stays on the host."""
    return r + opt_config.b3 * u


def proximal_step(u_current, grad_smooth, alpha: float, opt_config: OptimizationConfig):
    """Gradient step + soft threshold + box projection (C2:153-200) on the GPU.  The engine's
    kernel takes (u, r) and forms r + b3 u itself, so r is recovered from the gradient."""
    u_current = np.asarray(u_current, dtype=np.float64)
    grad_smooth = np.asarray(grad_smooth, dtype=np.float64)
    eng = _engine_like(u_current)
    o = make_opt(opt_config, b3=0.0)           # grad_smooth already contains b3*u
    return eng.grad_prox(u_current, grad_smooth, alpha, o)
