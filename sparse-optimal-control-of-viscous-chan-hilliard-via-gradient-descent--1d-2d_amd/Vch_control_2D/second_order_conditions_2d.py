"""Host-side mirror of src/2D/Vch_control_2D/second_order_conditions_2d.py: finite-difference
coercivity test (N extra forward marches + costs, all on the GPU through the mirrored
`run_main_simulation` / `calculate_cost`) and the sparsity (KKT) match statistic."""
from __future__ import annotations

import contextlib
import io
from typing import List, Optional

import numpy as np

from .Forward2_solver import run_main_simulation
from .cost2_and_function import calculate_cost
from .config import ForwardSolverConfig, OptimizationConfig


def _generate_direction(u_star, r_star, u_min, u_max, rng, tol: float = 1e-8):
    """Random unit direction in the bound-only critical cone (S2:35-88): components pointing
    out of the box at saturated nodes are flipped inwards."""
    v = rng.standard_normal(size=u_star.shape)
    lower = u_star <= (u_min + tol)
    upper = u_star >= (u_max - tol)
    if np.any(lower):
        v[lower] = np.abs(v[lower])
    if np.any(upper):
        v[upper] = -np.abs(v[upper])
    nv = np.linalg.norm(v)
    if nv < 1e-12:
        v = np.zeros_like(v)
        v.ravel()[0] = 1.0
        nv = 1.0
    return v / nv


def _ensure_opt_config(b1, b2, b3, kappa_sparsity, opt_config) -> OptimizationConfig:
    """Either a full opt_config or all four legacy scalars (S2:91-117)."""
    if opt_config is not None:
        return opt_config
    if any(v is None for v in (b1, b2, b3, kappa_sparsity)):
        raise ValueError("Either provide opt_config or all of (b1, b2, b3, kappa_sparsity).")
    return OptimizationConfig(b1=float(b1), b2=float(b2), b3=float(b3), kappa_sparsity=float(kappa_sparsity))


def approximate_second_order_condition_2d(u_star, r_star, phi_star, x, y, t_hist,
                                          opt_config: Optional[OptimizationConfig] = None, b1=None, b2=None, b3=None,
                                          kappa=None, phi_Q_target=None, phi_T_target=None, u_min: float = -np.inf,
                                          u_max: float = np.inf, num_directions: int = 10, epsilon: float = 1e-4,
                                          seed: Optional[int] = None,
                                          fwd_config: Optional[ForwardSolverConfig] = None) -> List[float]:
    """d2 ~ (J(u* + eps h) - J(u*) - eps <grad J(u*), h>) / (eps^2 / 2) for random h (S2:120-235)."""
    rng = np.random.default_rng(seed)
    opt = _ensure_opt_config(b1, b2, b3, kappa, opt_config)
    if phi_Q_target is None:
        phi_Q_target = np.zeros_like(phi_star)
    if phi_T_target is None:
        phi_T_target = np.zeros_like(phi_star[-1])
    quiet = lambda: contextlib.redirect_stdout(io.StringIO())
    with quiet():
        cost_star = calculate_cost(phi_star, u_star, phi_Q_target, phi_T_target, x, y, t_hist, opt)
    grad_star = r_star + opt.b3 * u_star
    out: List[float] = []
    print(f"Testing {num_directions} random directions in the critical cone...")
    for i in range(num_directions):
        h = _generate_direction(u_star, r_star, u_min, u_max, rng)
        u_p = u_star + epsilon * h
        phi_p, _, _ = run_main_simulation(config=fwd_config, store_history=True, control_input=u_p, verbose=False)
        with quiet():
            cost_p = calculate_cost(phi_p, u_p, phi_Q_target, phi_T_target, x, y, t_hist, opt)
        d2 = (cost_p - cost_star - epsilon * np.sum(grad_star * h)) / (0.5 * epsilon ** 2)
        out.append(float(d2))
        print(f"  Direction {i+1}/{num_directions}: estimated d²J/dh² ≈ {d2:.6e}")
    return out


def verify_sparsity_condition(u_optimal, r_optimal, kappa: float, tol: float = 1e-6):
    """u* = 0 <=> |r*| <= kappa match statistics (S2:238-297); prints like the reference and
    additionally returns (sparsity %, |r|<=kappa %, match %)."""
    u_flat, r_flat = np.ravel(u_optimal), np.ravel(r_optimal)
    is_u_zero = np.abs(u_flat) < tol
    is_r_small = np.abs(r_flat) <= kappa
    total = u_flat.size
    nz, nr, nm = int(np.sum(is_u_zero)), int(np.sum(is_r_small)), int(np.sum(is_u_zero == is_r_small))
    print("\n" + "=" * 60 + "\nVERIFYING SPARSITY CONDITION\nCondition: u*(x,t) = 0  <=>  |r*(x,t)| <= kappa\n" + "=" * 60)
    print(f"Sparsity of final control (u* ≈ 0): {100.0 * nz / total:.2f}% ({nz}/{total} points)")
    print(f"Region where |r*| <= kappa:          {100.0 * nr / total:.2f}% ({nr}/{total} points)")
    print(f"Percentage of points where the conditions match: {100.0 * nm / total:.2f}%")
    print("\n✓ The sparsity condition is satisfied." if 100.0 * nm / total > 99.0
          else "\n⚠ The sparsity condition is not fully satisfied.")
    print("=" * 60)
    return 100.0 * nz / total, 100.0 * nr / total, 100.0 * nm / total
