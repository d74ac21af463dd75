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


# sign code of a node of the critical cone: FREE keeps the drawn value, +1 / -1 force the sign, 0 pins the component
_FREE = 2


def _cone_sign_codes(u_star, s_star, u_min, u_max, kappa, tol, tol_s):
    """Admissible sign per node of the critical cone of `J + kappa |u|_1` over the box (S1:33-55).

    Rules, later ones overriding earlier ones where they overlap (the reference assigns them in this order):
    lower bound -> +, upper bound -> -, and at the kink u = 0 of the L1 term, with s = r + b3 u the smooth
    gradient: |s| < kappa -> pinned, s >= kappa -> -, s <= -kappa -> + (each with the slack tol_s)."""
    code = np.full(u_star.shape, _FREE, dtype=np.int8)
    kink = np.abs(u_star) <= tol
    for mask, c in ((u_star <= u_min + tol, 1),
                    (u_star >= u_max - tol, -1),
                    (kink & (np.abs(s_star) < kappa - tol_s), 0),
                    (kink & (s_star >= kappa - tol_s), -1),
                    (kink & (s_star <= -kappa + tol_s), 1)):
        code[mask] = c
    return code


def _generate_direction(u_star, r_star, u_min, u_max, kappa, b3, rng, tol=1e-8, tol_s=1e-9):
    """Unit direction in the critical cone incl. the L1 kink at u = 0: ONE `standard_normal` draw of the shape of
    u_star (the only thing the seeded goldens pin), signs imposed by `_cone_sign_codes`; an all-pinned cone falls
    back to the coordinate direction of the largest |r + b3 u|."""
    draw = rng.standard_normal(size=u_star.shape)
    s_star = r_star + b3 * u_star
    code = _cone_sign_codes(u_star, s_star, u_min, u_max, kappa, tol, tol_s)
    v = np.where(code == _FREE, draw, code * np.abs(draw))
    nrm = np.linalg.norm(v)
    if nrm == 0:
        v[np.unravel_index(np.argmax(np.abs(s_star)), s_star.shape)] = 1.0
        nrm = 1.0
    return v / nrm


def _as_generator(source=None):
    """A numpy Generator from a Generator, a seed-like value (anything int() accepts) or nothing; values that
    cannot be read as a seed give an unseeded generator, as the reference's helper does (S1:57-68)."""
    if isinstance(source, np.random.Generator):
        return source
    seed = None
    if source is not None:
        try:
            seed = int(source)
        except (TypeError, ValueError):
            seed = None
    try:
        return np.random.default_rng(seed)
    except (TypeError, ValueError):          # e.g. a negative seed
        return np.random.default_rng()


def approximate_second_order_condition(fwd_config: ForwardSolverConfig, u_star, r_star, phi_star, x, t_hist, b1, b2, b3,
                                       kappa, phi_Q_target, phi_T_target, u_min, u_max, num_directions: int = 10,
                                       epsilon: float = 1e-4, seed=None, rng=None) -> List[float]:
    """S1:71-177."""
    rng = _as_generator(rng if rng is not None else seed)
    quiet = lambda: contextlib.redirect_stdout(io.StringIO())
    with quiet():
        cost_star = calculate_cost(phi_star, u_star, phi_Q_target, phi_T_target, x, t_hist, b1, b2, b3, kappa, verbose=False)
    grad_star = r_star + b3 * u_star
    # all directions first (the same draws in the same order as the reference's loop, S1:137-150), then their perturbed
    # controls marched as one batch of trajectories with one batched cost evaluation
    dirs = [_generate_direction(u_star, r_star, u_min, u_max, kappa, b3, rng) for _ in range(num_directions)]
    costs = _perturbed_costs([u_star + epsilon * h for h in dirs], fwd_config, phi_Q_target, phi_T_target, x, t_hist,
                             b1, b2, b3, kappa)
    return [(cost_p - cost_star - epsilon * np.sum(grad_star * h)) / (0.5 * epsilon ** 2) for h, cost_p in zip(dirs, costs)]


MAX_BATCH = 64


def _perturbed_costs(controls, fwd_config, phi_Q_target, phi_T_target, x, t_hist, b1, b2, b3, kappa):
    """J(u) for every control of the list: `run_main_simulation(fwd_config, control_input=u)` (default start:
    init_phi_random(amp=0.01, seed=42), F1:316) and `calculate_cost`, batched over the controls."""
    from ..engine import time_grid, make_opt
    from ._ctx import engine_for_config
    from .Forward_solver import init_phi_random, delta_sep
    cfg = fwd_config if fwd_config is not None else ForwardSolverConfig()
    N = int(cfg.N)
    _, dts = time_grid(float(cfg.T), float(cfg.dt_initial))
    phi0 = init_phi_random(N, delta_sep, amp=0.01, seed=42, enforce_zero_mean=True)
    opt = make_opt(b1=b1, b2=b2, b3=b3, kappa_sparsity=kappa)
    costs = []
    for k0 in range(0, len(controls), MAX_BATCH):
        chunk = controls[k0:k0 + MAX_BATCH]
        nb = len(chunk)
        U = np.stack(chunk)
        eng = engine_for_config(cfg, batch=nb, max_steps=max(len(dts), 1))
        phi_p, _ = eng.forward(np.broadcast_to(phi0, (nb, N + 1)), dts, u=U, store=True)
        phi_p = phi_p.reshape((nb,) + phi_p.shape[-2:])
        tile = lambda a: np.broadcast_to(a, (nb,) + np.shape(a))
        J = eng.cost(phi_p, U, tile(phi_Q_target), tile(phi_T_target), x, t_hist, opt)
        costs.extend(float(v) for v in np.atleast_2d(J)[:, 4])
    return costs
