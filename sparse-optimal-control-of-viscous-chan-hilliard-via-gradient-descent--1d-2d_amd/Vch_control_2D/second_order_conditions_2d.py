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
    """Random unit direction in the bound-only critical cone (S2:35-88): one `standard_normal` draw, then the
    components at saturated nodes are reflected into the box (where both bounds are hit, the upper one wins, as in
    the reference's assignment order); r_star is part of the signature only."""
    draw = rng.standard_normal(size=u_star.shape)
    inward = np.zeros(u_star.shape, dtype=np.int8)          # +1 at the lower bound, -1 at the upper bound
    inward[u_star <= u_min + tol] = 1
    inward[u_star >= u_max - tol] = -1
    v = np.where(inward == 0, draw, inward * np.abs(draw))
    length = np.linalg.norm(v)
    if length < 1e-12:
        v = np.zeros_like(v)
        v.flat[0] = 1.0
        length = 1.0
    return v / length


def _ensure_opt_config(b1, b2, b3, kappa_sparsity, opt_config) -> OptimizationConfig:
    """The weights as an OptimizationConfig: the object if one is given, else built from the four legacy
    scalars, all of which must then be present (S2:91-117)."""
    if opt_config is not None:
        return opt_config
    weights = dict(b1=b1, b2=b2, b3=b3, kappa_sparsity=kappa_sparsity)
    if None in weights.values():
        raise ValueError("Either provide opt_config or all of (b1, b2, b3, kappa_sparsity).")
    return OptimizationConfig(**{k: float(v) for k, v in weights.items()})


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
    print(f"Testing {num_directions} random directions in the critical cone...")
    # all directions are drawn first -- the same draws in the same order as the reference's loop (S2:178-186), so seeded
    # results are unchanged -- and the perturbed controls are then marched as ONE batch of trajectories (MAX_BATCH at a
    # time) with one batched cost evaluation, instead of num_directions marches of batch 1
    dirs = [_generate_direction(u_star, r_star, u_min, u_max, rng) for _ in range(num_directions)]
    costs = _perturbed_costs([u_star + epsilon * h for h in dirs], fwd_config, phi_Q_target, phi_T_target, x, y, t_hist, opt)
    out: List[float] = []
    for i, (h, cost_p) in enumerate(zip(dirs, costs)):
        d2 = (cost_p - cost_star - epsilon * np.sum(grad_star * h)) / (0.5 * epsilon ** 2)
        out.append(float(d2))
        print(f"  Direction {i+1}/{num_directions}: estimated d²J/dh² ≈ {d2:.6e}")
    return out


MAX_BATCH = 8      # perturbed controls marched side by side (each needs its own state and control history on the device)


def _perturbed_costs(controls, fwd_config, phi_Q_target, phi_T_target, x, y, t_hist, opt):
    """J(u) for every control of the list: the marches of `run_main_simulation(config, control_input=u)` (default start:
    init_phi_random(amp=0.1, seed=42), F2:517) and `calculate_cost`, batched over the controls."""
    from ..engine import time_grid
    from ._ctx import engine_for_config
    from .Forward2_solver import init_phi_random, DELTA_SEP
    cfg = fwd_config
    Nx, Ny = int(cfg.Nx), int(cfg.Ny)
    _, dts = time_grid(float(cfg.T), float(cfg.dt_initial))
    phi0 = init_phi_random(Nx, Ny, DELTA_SEP, amp=0.1, seed=42)
    costs = []
    for k0 in range(0, len(controls), MAX_BATCH):
        chunk = controls[k0:k0 + MAX_BATCH]
        nb = len(chunk)
        U = np.stack(chunk)
        rows = U.shape[1]
        eng = engine_for_config(cfg, batch=nb, max_steps=max(len(dts), rows - 1, 1))
        Um = U[:, :len(dts) + 1] if rows > len(dts) + 1 else U            # rows beyond the march are never read (F2:545-548)
        phi_p, _ = eng.forward(np.broadcast_to(phi0, (nb,) + phi0.shape), dts, u=np.ascontiguousarray(Um), store=True)
        phi_p = phi_p.reshape((nb,) + phi_p.shape[-3:])
        tile = lambda a: np.broadcast_to(a, (nb,) + np.shape(a))
        J = eng.cost(phi_p, U, tile(phi_Q_target), tile(phi_T_target), t_hist, opt, x, y)
        costs.extend(float(v) for v in np.atleast_2d(J)[:, 4])
    return costs


def sparsity_statistics(u_optimal, r_optimal, kappa: float, tol: float = 1e-6):
    """Counts behind the KKT sparsity check `u* = 0 <=> |r*| <= kappa`: (nodes with |u*| < tol, nodes with
    |r*| <= kappa, nodes where the two predicates agree, nodes in total)."""
    u_is_zero = np.abs(np.ravel(u_optimal)) < tol
    r_is_small = np.abs(np.ravel(r_optimal)) <= kappa
    return int(u_is_zero.sum()), int(r_is_small.sum()), int((u_is_zero == r_is_small).sum()), int(u_is_zero.size)


def verify_sparsity_condition(u_optimal, r_optimal, kappa: float, tol: float = 1e-6):
    """The reference's report (S2:238-297) from `sparsity_statistics`; additionally returns the three percentages
    (sparsity of u*, share of |r*| <= kappa, share of matching nodes) that the reference only prints."""
    n_zero, n_small, n_match, total = sparsity_statistics(u_optimal, r_optimal, kappa, tol)
    pct = tuple(100.0 * k / total for k in (n_zero, n_small, n_match))
    bar = "=" * 60
    print(f"\n{bar}\nVERIFYING SPARSITY CONDITION\nCondition: u*(x,t) = 0  <=>  |r*(x,t)| <= kappa\n{bar}")
    print(f"Sparsity of final control (u* ≈ 0): {pct[0]:.2f}% ({n_zero}/{total} points)")
    print(f"Region where |r*| <= kappa:          {pct[1]:.2f}% ({n_small}/{total} points)")
    print(f"Percentage of points where the conditions match: {pct[2]:.2f}%")
    print("\n✓ The sparsity condition is satisfied." if pct[2] > 99.0 else "\n⚠ The sparsity condition is not fully satisfied.")
    print(bar)
    return pct
