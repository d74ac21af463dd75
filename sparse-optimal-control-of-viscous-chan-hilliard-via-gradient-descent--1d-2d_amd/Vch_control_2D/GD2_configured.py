"""Host-side mirror of the reusable parts of src/2D/Vch_control_2D/GD2_configured.py: target
construction (G2:149-228), the backtracking line search (G2:71-146) and the optimisation loop
of the `__main__` block (G2:291-382) as a callable that runs device-resident
(vch2d_pgd_init / vch2d_pgd_iterate).  Prompts, previews and plots are out of scope."""
from __future__ import annotations

import contextlib
import io
import time
from typing import Tuple

import numpy as np

from ..engine import make_opt, time_grid
from ._ctx import engine_for_config
from .Forward2_solver import run_main_simulation, init_phi_random, DELTA_SEP
from .backward2_solver import run_backward                                   # noqa: F401
from .cost2_and_function import calculate_cost, calculate_gradient, proximal_step   # noqa: F401
from .config import ForwardSolverConfig, OptimizationConfig, load_params, save_params

INTERACTIVE = False
DEFAULT_TARGET_CHOICE = 1
DEFAULT_TRACKING_CHOICE = 1


def build_targets(x, y, t_hist, phi_initial, Lx, Ly, T, interactive=False, choice_t=1, choice_q=1):
    """phi_T: 0.7 sin(2 pi x/Lx) cos(pi y/Ly) or a centred disc; phi_Q: ramp with the CONFIG T or
    zeros (G2:184-226)."""
    xx, yy = np.meshgrid(x, y, indexing="ij")
    if choice_t == 1:
        phi_T = 0.7 * np.sin(2 * np.pi * xx / Lx) * np.cos(np.pi * yy / Ly)
    else:
        phi_T = -np.ones_like(xx)
        phi_T[(xx - Lx / 2) ** 2 + (yy - Ly / 2) ** 2 < (Lx / 3.5) ** 2] = 1.0
    if choice_q == 1:
        tp = (t_hist / T)[:, np.newaxis, np.newaxis]
        phi_Q = (1 - tp) * phi_initial + tp * phi_T
    else:
        phi_Q = np.zeros((len(t_hist), len(x), len(y)))
    return phi_T, phi_Q


def perform_backtracking_line_search_2D(u_k, cost_k, grad_smooth, phi_Q_target, phi_T_target, x, y, fwd_config,
                                        opt_config, alpha_init: float = 1.0, beta: float = 0.8,
                                        max_ls_iter: int = 10) -> Tuple:
    """G2:71-146 through the function-level mirrors (each call one engine invocation)."""
    alpha, attempts = alpha_init, 0
    u_next, phi_next, t_next, cost_next = u_k, None, None, cost_k
    t0 = time.perf_counter()
    for _ in range(max_ls_iter):
        attempts += 1
        u_next = proximal_step(u_k, grad_smooth, alpha, opt_config)
        phi_next, _, t_next = run_main_simulation(config=fwd_config, store_history=True, control_input=u_next, verbose=False)
        with contextlib.redirect_stdout(io.StringIO()):
            cost_next = calculate_cost(phi_next, u_next, phi_Q_target, phi_T_target, x, y, t_next, opt_config)
        if cost_next < cost_k:
            return alpha, u_next, cost_next, phi_next, t_next, time.perf_counter() - t0, attempts
        alpha *= beta
    return alpha, u_next, cost_next, phi_next, t_next, time.perf_counter() - t0, attempts


def run_optimization(fwd_config: ForwardSolverConfig, opt_config: OptimizationConfig, n_iter=None, seeds=(42,),
                     choice_t=DEFAULT_TARGET_CHOICE, choice_q=DEFAULT_TRACKING_CHOICE, amp=0.1, device=0):
    """The PGD loop of G2:291-382 for a batch of initial conditions (`seeds`), device-resident.
    Returns dict(costs [B][it+1], alphas, attempts, changes, tracking_error, terminal_error (the two relative
    error histories the driver appends per iteration, G2:336-363, from the cost kernel's sums), u, phi, r, seconds)."""
    Nx, Ny = int(fwd_config.Nx), int(fwd_config.Ny)
    t_hist, dts = time_grid(float(fwd_config.T), float(fwd_config.dt_initial))
    B = len(seeds)
    eng = engine_for_config(fwd_config, batch=B, max_steps=len(dts), device=device)
    phi0 = np.stack([init_phi_random(Nx, Ny, DELTA_SEP, amp=amp, seed=int(s)) for s in seeds])
    phi_T, _ = build_targets(eng.x, eng.y, t_hist, phi0[0], fwd_config.Lx, fwd_config.Ly, fwd_config.T,
                             choice_t=choice_t, choice_q=2)
    phi_Tb = np.broadcast_to(phi_T, phi0.shape).copy()
    J0 = eng.pgd_init(phi0, phi_Tb, t_hist, make_opt(opt_config), ramp=(choice_q == 1), T=float(fwd_config.T))
    n = int(opt_config.max_iter if n_iter is None else n_iter)
    res = eng.pgd_iterate(n)
    costs = np.concatenate([J0[:, 4:5], res["cost"]], axis=1)
    return dict(costs=costs, alphas=res["alpha"], attempts=res["attempts"], changes=res["change"],
                tracking_error=res["tracking_error"], terminal_error=res["terminal_error"], iters=res["iters"], seconds=res["seconds"], u=eng.pgd_get("u"), phi=eng.pgd_get("phi"),
                r=eng.pgd_get("r"), phi_T=phi_T, t_hist=t_hist, x=eng.x.copy(), y=eng.y.copy())


def main(n_iter=None, params_file="last_run_config_2d.json", num_directions=5, verbose=True):
    """Non-interactive equivalent of the reference's `__main__` block (G2:230-441) without previews and
    plots: parameters from the last-run JSON (defaults if absent, K2:181-190), uncontrolled march, targets
    (choice 1/1), the PGD loop (device-resident), the time-study summary (G2:403-416), the final adjoint,
    the coercivity finite-difference test and the sparsity statistic (G2:424-437), `save_params`
    (G2:440).  Returns the run_optimization dict extended by `hessian_values` and `sparsity`."""
    from .second_order_conditions_2d import approximate_second_order_condition_2d, verify_sparsity_condition
    say = print if verbose else (lambda *a, **k: None)
    allp = load_params(params_file)
    fwd, opt = allp.forward_solver, allp.optimization
    say("=" * 60 + "\n    2D GRADIENT DESCENT OPTIMIZATION \n" + "=" * 60)
    say("Forward Solver Config:\n" + fwd.model_dump_json(indent=2) + "\nOptimization Config:\n" + opt.model_dump_json(indent=2))
    t0 = time.time()
    res = run_optimization(fwd, opt, n_iter=n_iter)
    B = res["costs"].shape[0]
    assert B == 1
    costs = res["costs"][0]
    it = int(res["iters"])
    say(f"Completed Iterations: {it}\nFinal Cost: {costs[it]:.5f}\nCost Reduction: {100 * (1 - costs[it] / costs[0]):.2f}%")
    if it > 0:
        say(f"Relative tracking error: {res['tracking_error'][0, it - 1]:.6e}   "
            f"relative terminal error: {res['terminal_error'][0, it - 1]:.6e}")
    sec = res["seconds"]
    say("\n" + "=" * 50 + "\n  TIME STUDY SUMMARY\n" + "=" * 50)
    say(f"Total backward-solve time:        {sec['backward']:.3f}s")
    say(f"Total optimistic forward time:    {sec['optimistic_forward']:.3f}s")
    say(f"Total optimistic cost time:       {sec['cost']:.3f}s")
    say(f"Total backtracking time:          {sec['backtracking']:.3f}s  (attempts={int(res['attempts'].sum())})")
    ok = res["attempts"][0, :it] == 0
    if ok.any():
        say(f"ALPHA ADVISOR: a good initial alpha_max for the next run is {float(np.mean(res['alphas'][0, :it][ok])):.4f}")
    x, y, t_hist = res["x"], res["y"], res["t_hist"]
    phi_T, phi_Q = build_targets(x, y, t_hist, res["phi"][0], fwd.Lx, fwd.Ly, fwd.T, choice_t=DEFAULT_TARGET_CHOICE,
                                 choice_q=DEFAULT_TRACKING_CHOICE)
    out = dict(res, hessian_values=None, sparsity=None, runtime_s=None)
    try:
        _, _, r_opt = run_backward(res["phi"], x, y, t_hist, fwd, opt.b1, opt.b2, phi_Q, phi_T)
        sink = contextlib.nullcontext() if verbose else contextlib.redirect_stdout(io.StringIO())
        with sink:
            hv = approximate_second_order_condition_2d(
                u_star=res["u"], r_star=r_opt, phi_star=res["phi"], x=x, y=y, t_hist=t_hist, b1=opt.b1, b2=opt.b2, b3=opt.b3,
                kappa=opt.kappa_sparsity, phi_Q_target=phi_Q, phi_T_target=phi_T, u_min=opt.u_min, u_max=opt.u_max,
                num_directions=num_directions, epsilon=1e-4, seed=42, fwd_config=fwd)
            say("Coercivity condition appears to hold in tested directions." if all(v > 0 for v in hv)
                else "Coercivity condition may fail; non-positive second derivatives found.")
            out["sparsity"] = verify_sparsity_condition(res["u"], r_opt, opt.kappa_sparsity)
        out["hessian_values"] = hv
    except Exception as e:          # G2:438-439
        say(f"[Warning] Could not perform final analysis (second-order/sparsity): {e}")
    save_params(fwd, opt, it, params_file)
    out["runtime_s"] = time.time() - t0
    return out

