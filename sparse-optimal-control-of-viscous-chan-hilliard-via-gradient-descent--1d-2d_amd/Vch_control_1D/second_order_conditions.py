"""Host-side mirror of src/1D/Vch_control_1D/second_order_conditions.py: finite-difference
coercivity test with the kink-aware critical cone; forward marches and costs run on the GPU."""
from __future__ import annotations

import contextlib
import io
from typing import List

import numpy as np

from .Forward_solver import run_main_simulation
from .cost_and_function import calculate_cost
from .config import ForwardSolverConfig


def _generate_direction(u_star, r_star, u_min, u_max, kappa, b3, rng, tol=1e-8, tol_s=1e-9):
    """Unit direction in the critical cone incl. the L1 kink at u = 0 (S1:33-55)."""
    v = rng.standard_normal(size=u_star.shape)
    s_star = r_star + b3 * u_star
    lower = u_star <= (u_min + tol)
    upper = u_star >= (u_max - tol)
    at_zero = np.abs(u_star) <= tol
    kink_interior = at_zero & (np.abs(s_star) < (kappa - tol_s))
    kink_plus = at_zero & (s_star >= (kappa - tol_s))
    kink_minus = at_zero & (s_star <= (-kappa + tol_s))
    if np.any(lower):
        v[lower] = np.abs(v[lower])
    if np.any(upper):
        v[upper] = -np.abs(v[upper])
    if np.any(kink_interior):
        v[kink_interior] = 0.0
    if np.any(kink_plus):
        v[kink_plus] = -np.abs(v[kink_plus])
    if np.any(kink_minus):
        v[kink_minus] = np.abs(v[kink_minus])
    nrm = np.linalg.norm(v)
    if nrm == 0:
        idx = np.unravel_index(np.argmax(np.abs(s_star)), s_star.shape)
        v[idx] = 1.0
        nrm = 1.0
    return v / nrm


def _coerce_rng(seed_or_rng=None):
    if isinstance(seed_or_rng, np.random.Generator):
        return seed_or_rng
    if seed_or_rng is None:
        return np.random.default_rng()
    try:
        return np.random.default_rng(int(seed_or_rng))
    except Exception:
        return np.random.default_rng()


def approximate_second_order_condition(fwd_config: ForwardSolverConfig, u_star, r_star, phi_star, x, t_hist, b1, b2, b3,
                                       kappa, phi_Q_target, phi_T_target, u_min, u_max, num_directions: int = 10,
                                       epsilon: float = 1e-4, seed=None, rng=None) -> List[float]:
    """S1:71-177."""
    rng = _coerce_rng(rng if rng is not None else seed)
    quiet = lambda: contextlib.redirect_stdout(io.StringIO())
    with quiet():
        cost_star = calculate_cost(phi_star, u_star, phi_Q_target, phi_T_target, x, t_hist, b1, b2, b3, kappa, verbose=False)
    grad_star = r_star + b3 * u_star
    out: List[float] = []
    for _ in range(num_directions):
        h = _generate_direction(u_star, r_star, u_min, u_max, kappa, b3, rng)
        u_p = u_star + epsilon * h
        phi_p, _, _ = run_main_simulation(fwd_config=fwd_config, store_history=True, control_input=u_p, verbose=False)
        with quiet():
            cost_p = calculate_cost(phi_p, u_p, phi_Q_target, phi_T_target, x, t_hist, b1, b2, b3, kappa, verbose=False)
        out.append((cost_p - cost_star - epsilon * np.sum(grad_star * h)) / (0.5 * epsilon ** 2))
    return out
