"""Host-side mirror of the reusable parts of src/1D/Vch_control_1D/GD_1D.py: the prox, the
line search, the sparsity (KKT) check, target construction, and the optimisation loop of the
`__main__` block (G1:333-477) as a callable.  Prompts and plots are out of scope."""
from __future__ import annotations

import contextlib
import io

import numpy as np

from ..engine import make_opt
from .Forward_solver import run_main_simulation
from .backward_solver import run_backward
from .cost_and_function import calculate_cost, calculate_gradient, perform_gradient_step
from .config import ForwardSolverConfig, OptimizationConfig, load_params, save_params
from ._ctx import engine_for

INTERACTIVE = False
DEFAULT_TARGET_CHOICE = 1
DEFAULT_TRACKING_CHOICE = 1


def perform_proximal_and_projection(u_temp, alpha, kappa, u_min, u_max):
    """Soft threshold alpha*kappa then box projection (G1:56-71), on the GPU: the engine's fused
    kernel is called with r = 0, b3 = 0 and step 0 on the already stepped control."""
    u_temp = np.asarray(u_temp, dtype=np.float64)
    shp = u_temp.shape
    u2 = u_temp.reshape(-1, shp[-1]) if u_temp.ndim > 1 else u_temp.reshape(1, -1)
    eng = engine_for(u2.shape[1] - 1, max_steps=max(u2.shape[0], 1))
    # prox(v) with v = u - alpha*(r + b3 u): choose r = 0, b3 = 0 so that v = u
    o = make_opt(b3=0.0, kappa_sparsity=float(kappa), u_min=float(u_min), u_max=float(u_max))
    out = eng.grad_prox(u2, np.zeros_like(u2), float(alpha), o)
    return np.asarray(out).reshape(shp)


def perform_backtracking_line_search(u_k, cost_k, grad_smooth, phi_Q_target, phi_T_target, x, t_hist, b1, b2, b3,
                                     kappa, u_min, u_max, fwd_config, alpha_init=10.0, beta=0.8, max_ls_iter=5):
    """G1:73-113."""
    alpha, n_trials = alpha_init, 0
    u_next = phi_next = None
    cost_next = cost_k
    for _ in range(max_ls_iter):
        n_trials += 1
        u_next = perform_proximal_and_projection(perform_gradient_step(u_k, grad_smooth, alpha), alpha, kappa, u_min, u_max)
        phi_next, _, _ = run_main_simulation(fwd_config, store_history=True, control_input=u_next, verbose=False)
        with contextlib.redirect_stdout(io.StringIO()):
            cost_next = calculate_cost(phi_next, u_next, phi_Q_target, phi_T_target, x, t_hist, b1, b2, b3, kappa, verbose=False)
        if cost_next < cost_k:
            return alpha, u_next, cost_next, phi_next, 0.0, 0.0, n_trials
        alpha *= beta
    return alpha, u_next, cost_next, phi_next, 0.0, 0.0, n_trials


def verify_sparsity_condition(u_optimal, r_optimal, kappa, tol=1e-6, verbose=True):
    """u* = 0  <=>  |r*| <= kappa match statistics (G1:115-147).  Returns (sparsity %, |r|<=kappa %,
    match %)."""
    is_u_zero = np.abs(u_optimal) < tol
    is_r_small = np.abs(r_optimal) <= kappa
    total = u_optimal.size
    res = (100.0 * np.sum(is_u_zero) / total, 100.0 * np.sum(is_r_small) / total,
           100.0 * np.sum(is_u_zero == is_r_small) / total)
    if verbose:
        print(f"Sparsity of final control (u* ~ 0): {res[0]:.2f}%")
        print(f"Region where |r*| <= kappa:          {res[1]:.2f}%")
        print(f"Percentage of points where the conditions match: {res[2]:.2f}%")
    return res


def build_targets_1d(x, t_hist, phi_initial, Lx, T, interactive=False, choice_t=DEFAULT_TARGET_CHOICE,
                     choice_q=DEFAULT_TRACKING_CHOICE, A_T=0.7, k_tan=0.45):
    """G1:151-254 (non-interactive)."""
    if choice_t == 1:
        phi_T = A_T * np.sin(2.0 * np.pi * x / Lx)
    elif choice_t == 2:
        phi_T = A_T * np.cos(2.0 * np.pi * x / Lx)
    else:
        tr = np.tan(2.0 * np.pi * k_tan * (x / Lx - 0.5))
        sc = np.max(np.abs(tr))
        phi_T = A_T * (tr / (sc if sc > 1e-12 else 1.0))
    if choice_q == 1:
        tp = (t_hist / (t_hist[-1] if t_hist[-1] > 0 else 1.0))[:, np.newaxis]
        phi_Q = (1.0 - tp) * phi_initial + tp * phi_T
    else:
        phi_Q = np.zeros((len(t_hist), len(x)))
    return phi_T, phi_Q


def relative_errors(phi, phi_Q, phi_T, x, t_hist):
    """(tracking, terminal) relative L2 errors of a state history as the driver reports them per iteration
    (G1:425-450): trapezoid rule in x then t, sqrt(L T) as denominator when the tracking target is ~0."""
    trapz = getattr(np, "trapezoid", None) or np.trapz
    sq_x = lambda a: trapz(np.square(a), x=x, axis=-1)
    scale = np.sqrt(max(float(x[-1] - x[0]), 1e-30) * max(float(t_hist[-1] - t_hist[0]), 1e-30))
    num_q = np.sqrt(trapz(sq_x(phi - phi_Q), x=t_hist))
    den_q = np.sqrt(trapz(sq_x(phi_Q), x=t_hist))
    if den_q < 1e-9 * scale:
        den_q = scale
    return float(num_q / (den_q + 1e-12)), float(np.sqrt(sq_x(phi[-1] - phi_T)) / (np.sqrt(sq_x(phi_T)) + 1e-12))


def run_optimization(fwd_config: ForwardSolverConfig, opt_config: OptimizationConfig, n_iter=None, choice_t=1,
                     choice_q=1):
    """The loop of G1:333-477 -> dict(costs, alphas, trials, tracking_error, terminal_error, u, phi, r, converged)."""
    O = opt_config
    quiet = contextlib.redirect_stdout(io.StringIO())
    phi_k, x, t_hist = run_main_simulation(fwd_config, store_history=True, verbose=False)
    u_k = np.zeros_like(phi_k)
    phi_T, phi_Q = build_targets_1d(x, t_hist, phi_k[0].copy(), float(fwd_config.Lx), float(fwd_config.T),
                                    choice_t=choice_t, choice_q=choice_q)
    with quiet:
        cost_k = calculate_cost(phi_k, u_k, phi_Q, phi_T, x, t_hist, O.b1, O.b2, O.b3, O.kappa_sparsity)
    costs, alphas, trials, track, term = [cost_k], [], [], [], []
    alpha_prev, plateau, converged = O.alpha_max, 0, False
    r_k = None
    for k in range(O.max_iter if n_iter is None else n_iter):
        _, _, r_k = run_backward(phi_k, x, t_hist, O.b1, O.b2, phi_Q, phi_T)
        g = calculate_gradient(r_k, u_k, O.b3)
        u_o = perform_proximal_and_projection(perform_gradient_step(u_k, g, alpha_prev), alpha_prev,
                                              O.kappa_sparsity, O.u_min, O.u_max)
        phi_o, _, _ = run_main_simulation(fwd_config, store_history=True, control_input=u_o, verbose=False)
        with contextlib.redirect_stdout(io.StringIO()):
            c_o = calculate_cost(phi_o, u_o, phi_Q, phi_T, x, t_hist, O.b1, O.b2, O.b3, O.kappa_sparsity, verbose=False)
        if c_o < cost_k:
            a_k, u_n, c_n, phi_n, nt = alpha_prev, u_o, c_o, phi_o, 1
        else:
            a_k, u_n, c_n, phi_n, _, _, nt = perform_backtracking_line_search(
                u_k, cost_k, g, phi_Q, phi_T, x, t_hist, O.b1, O.b2, O.b3, O.kappa_sparsity, O.u_min, O.u_max,
                fwd_config, alpha_init=alpha_prev)
        costs.append(c_n); alphas.append(a_k); trials.append(nt)
        e_q, e_t = relative_errors(phi_n, phi_Q, phi_T, x, t_hist)
        track.append(e_q); term.append(e_t)
        plateau = plateau + 1 if (k > 0 and abs(costs[-1] - costs[-2]) < 1e-7) else 0
        if plateau >= 10:
            alpha_prev, plateau = min(O.alpha_max, a_k * 2.0), 0
        else:
            alpha_prev = min(O.alpha_max, a_k * 1.2)
        change = np.linalg.norm(u_n - u_k) / (np.linalg.norm(u_k) + 1e-9)
        if change < 1e-5 and k > 10:
            u_k, converged = u_n.copy(), True
            break
        u_k, cost_k, phi_k = u_n.copy(), c_n, phi_n
    return dict(costs=costs, alphas=alphas, trials=trials, tracking_error=track, terminal_error=term, u=u_k, phi=phi_k,
                r=r_k, converged=converged, phi_T=phi_T, phi_Q=phi_Q, x=x, t_hist=t_hist)


def run_optimization_resident(fwd_config: ForwardSolverConfig, opt_config: OptimizationConfig, n_iter=None, choice_t=1,
                              choice_q=1, initial_phi=None, seeds=None):
    """The same loop (G1:333-477) run device-resident through `vch1d_pgd_*`: control, state history,
    adjoint and targets never leave HBM between iterations.  `seeds` (a list) or `initial_phi`
    ((B, N+1)) gives a batch of trajectories; default = the reference's seed 42.  Returns the same
    dict as run_optimization, with a leading trajectory axis when batched."""
    from ..engine import Engine1D, time_grid
    from .Forward_solver import init_phi_random, delta_sep
    O = opt_config
    N = int(fwd_config.N)
    tg, dts = time_grid(float(fwd_config.T), float(fwd_config.dt_initial))
    t_hist = np.concatenate([[0.0], tg])
    if initial_phi is not None:
        phi0 = np.atleast_2d(np.asarray(initial_phi, dtype=np.float64))
    else:
        phi0 = np.stack([init_phi_random(N, delta_sep, amp=0.01, seed=s, enforce_zero_mean=True)
                         for s in (seeds if seeds is not None else [42])])
    B = phi0.shape[0]
    eng = Engine1D(N=N, Lx=float(fwd_config.Lx), tau=float(fwd_config.tau), gamma=float(fwd_config.gamma),
                   c1=float(fwd_config.c1), c2=float(fwd_config.c2), kappa=float(fwd_config.kappa), batch=B,
                   max_steps=max(len(dts), 1))
    try:
        x = eng.x.copy()
        phi_T = np.stack([build_targets_1d(x, t_hist, phi0[b], float(fwd_config.Lx), float(fwd_config.T),
                                           choice_t=choice_t, choice_q=2)[0] for b in range(B)])
        phi_Q = None if choice_q == 1 else np.zeros((B, len(t_hist), N + 1))
        J0 = eng.pgd_init(phi0, phi_T, t_hist, dts, make_opt(O), phi_Q=phi_Q, x=x)
        n = O.max_iter if n_iter is None else n_iter
        res = eng.pgd_iterate(n)
        sq = (lambda a: a[0]) if B == 1 else (lambda a: a)
        costs = np.concatenate([J0.reshape(B, 5)[:, 4:5], res["cost"]], axis=1)
        out = dict(costs=sq(costs), alphas=sq(res["alpha"]), trials=sq(res["trials"]), change=sq(res["change"]),
                   tracking_error=sq(res["tracking_error"]), terminal_error=sq(res["terminal_error"]), iters=res["iters"], u=eng.pgd_get("u"), phi=eng.pgd_get("phi"), r=eng.pgd_get("r"),
                   phi_T=sq(phi_T), phi_Q=eng.pgd_get("phi_Q"), x=x, t_hist=t_hist, seconds=res["seconds"])
    finally:
        eng.close()
    return out


def main(n_iter=None, params_file="last_run_config.json", control_file="optimal_control.npy", num_directions=3,
         verbose=True):
    """Non-interactive equivalent of the reference's `__main__` block (G1:256-609) without prompts and
    plots: parameters from the last-run JSON (defaults if absent), the PGD loop (device-resident), the
    final adjoint, `optimal_control.npy` (G1:487), the coercivity finite-difference test (G1:490-509), the
    sparsity statistic (G1:518) and `save_params` (G1:608)."""
    from .second_order_conditions import approximate_second_order_condition
    say = print if verbose else (lambda *a, **k: None)
    allp = load_params(params_file)
    fwd, opt = allp.forward_solver, allp.optimization
    res = run_optimization_resident(fwd, opt, n_iter=n_iter)
    it = int(res["iters"])
    costs = np.asarray(res["costs"])
    say(f"Completed Iterations: {it}\nFinal Cost: {costs[it]:.5f}\nCost Reduction: {100 * (1 - costs[it] / costs[0]):.2f}%")
    x, t_hist = res["x"], res["t_hist"]
    _, _, r_opt = run_backward(res["phi"], x, t_hist, opt.b1, opt.b2, res["phi_Q"], res["phi_T"])
    np.save(control_file, res["u"])
    say(f"Optimal control saved as '{control_file}'")
    sink = contextlib.nullcontext() if verbose else contextlib.redirect_stdout(io.StringIO())
    with sink:
        hv = approximate_second_order_condition(fwd_config=fwd, u_star=res["u"], r_star=r_opt, phi_star=res["phi"], x=x,
                                                t_hist=t_hist, b1=opt.b1, b2=opt.b2, b3=opt.b3, kappa=opt.kappa_sparsity,
                                                phi_Q_target=res["phi_Q"], phi_T_target=res["phi_T"], u_min=opt.u_min,
                                                u_max=opt.u_max, num_directions=num_directions, epsilon=1e-4, seed=42)
        for i, d2 in enumerate(hv, start=1):
            say(f"  Direction {i}: estimated second derivative = {d2:.6e}")
        sp = verify_sparsity_condition(res["u"], r_opt, opt.kappa_sparsity, verbose=verbose)
    save_params(fwd, opt, it, params_file)
    return dict(res, hessian_values=hv, sparsity=sp, r_optimal=r_opt)

