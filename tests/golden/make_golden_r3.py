#!/usr/bin/env python3
"""Round-3 golden vectors from the reference (same rules as make_golden.py: TEST INFRASTRUCTURE, runs only in
the build container, imports the reference from /root/reference, copies nothing, writes .npz data files only).

    python tests/golden/make_golden_r3.py [--only config1|stress256]

  g1d_config1_256.npz   BASELINE config 1 AT FULL SIZE: 1D N = 256, T = 1, dt_initial = 5e-3 (200 steps, 202 history
                        rows), default K1 weights: natural march, adjoint sweep on it, cost, and three iterations of
                        the PGD loop G1:353-480 (costs, step lengths, trial counts, final control and state, error
                        metrics through make_golden_r2.errs_1d)
  g2d_stress_512.npz    (only with --only stress512) the same at the benched grid 512^2, first step only.  NOT committed: the first
                        Newton call had not returned after 2 h 50 min of SuperLU time in the build container
  g2d_stress_1024.npz   (only with --only stress1024) the same at BASELINE config 5's own grid, first step only.  NOT committed: the
                        reference's first Newton call there had not returned after 5 h 45 min of SuperLU time in the build container
  g2d_stress_256.npz    amp = 1.0 start at 256^2 (the FFT path, twice the size of g2d_stress_128), dt = 1e-3, 3 steps:
                        Newton residual histories and residual-evaluation counts per step (F2:377-423: step ceiling,
                        Armijo, best-trial fallback), sub-sampled fields, per-level norms
"""
import argparse
import os
import sys
import tempfile

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import _fresh_import, quiet, save, REF, REF2D, REF1D     # noqa: E402
from make_golden_r2 import errs_1d                                          # noqa: E402


def gen_config1():
    F1, B1, C1, K1, G1 = _fresh_import(REF1D, ["Forward_solver", "backward_solver", "cost_and_function", "config", "GD_1D"])
    N, T, dt, n_iter = 256, 1.0, 5e-3, 3
    cfg = K1.ForwardSolverConfig(N=N, T=T, dt_initial=dt)
    opt = K1.OptimizationConfig()
    with quiet():
        phi_k, x, t_hist = F1.run_main_simulation(cfg, store_history=True, verbose=False)
        u_k = np.zeros_like(phi_k)
        phi_T, phi_Q = G1.build_targets_1d(x, t_hist, phi_k[0].copy(), cfg.Lx, cfg.T, interactive=False, choice_t=1, choice_q=1)
        cost_k = C1.calculate_cost(phi_k, u_k, phi_Q, phi_T, x, t_hist, opt.b1, opt.b2, opt.b3, opt.kappa_sparsity)
        p0, q0, r0 = B1.run_backward(phi_k, x, t_hist, opt.b1, opt.b2, phi_Q, phi_T)
    phi_nat = phi_k.copy()
    print(f"    config 1: natural march {phi_nat.shape}, J(u0) = {cost_k:.12f}", flush=True)
    costs, alphas, trials, trk, trm = [cost_k], [], [], [], []
    alpha_prev, plateau = opt.alpha_max, 0
    for k in range(n_iter):
        with quiet():
            _, _, r_k = B1.run_backward(phi_k, x, t_hist, opt.b1, opt.b2, phi_Q, phi_T)
            g = C1.calculate_gradient(r_k, u_k, opt.b3)
            u_o = G1.perform_proximal_and_projection(C1.perform_gradient_step(u_k, g, alpha_prev), alpha_prev,
                                                     opt.kappa_sparsity, opt.u_min, opt.u_max)
            phi_o, _, _ = F1.run_main_simulation(cfg, store_history=True, control_input=u_o, verbose=False)
            c_o = C1.calculate_cost(phi_o, u_o, phi_Q, phi_T, x, t_hist, opt.b1, opt.b2, opt.b3, opt.kappa_sparsity, verbose=False)
            if c_o < cost_k:
                a_k, u_n, c_n, phi_n, nt = alpha_prev, u_o, c_o, phi_o, 1
            else:
                a_k, u_n, c_n, phi_n, _, _, nt = G1.perform_backtracking_line_search(
                    u_k, cost_k, g, phi_Q, phi_T, x, t_hist, opt.b1, opt.b2, opt.b3, opt.kappa_sparsity, opt.u_min, opt.u_max,
                    cfg, alpha_init=alpha_prev)
        costs.append(c_n); alphas.append(a_k); trials.append(nt)
        e1, e2 = errs_1d(phi_n, phi_Q, phi_T, x, t_hist)
        trk.append(e1); trm.append(e2)
        if k > 0 and abs(costs[-1] - costs[-2]) < 1e-7:
            plateau += 1
        else:
            plateau = 0
        alpha_prev = min(opt.alpha_max, a_k * (2.0 if plateau >= 10 else 1.2))
        print(f"    config 1 k={k} J={c_n:.12f} alpha={a_k:.6g} trials={nt}", flush=True)
        u_k, cost_k, phi_k = u_n.copy(), c_n, phi_n
    # every second history row of the fields (the file stays below 1 MB) + the row norms of the full arrays
    l2 = lambda A: np.sqrt((A ** 2).sum(axis=1))
    save("g1d_config1_256.npz", N=N, T=T, dt=dt, n_iter=n_iter, x=x, t_hist=t_hist, phi_nat_sub=phi_nat[::2], r_nat_sub=r0[::2],
         p_nat_sub=p0[::4], nrm_phi_nat=l2(phi_nat), nrm_r_nat=l2(r0), nrm_p_nat=l2(p0), nrm_q_nat=l2(q0),
         phi_T=phi_T, costs=np.array(costs), alphas=np.array(alphas), trials=np.array(trials), tracking=np.array(trk),
         terminal=np.array(trm), u_final_sub=u_k[::2], phi_final_sub=phi_k[::2], nrm_u_final=l2(u_k), nrm_phi_final=l2(phi_k))


def gen_stress(N, M, sub, name):
    F2, K2 = _fresh_import(REF2D, ["Forward2_solver", "config"])
    dt = 1e-3
    cfg = K2.ForwardSolverConfig(Nx=N, Ny=N, T=M * dt, dt_initial=dt)
    hists, evals = [], []
    orig_newton, orig_res, orig_init = F2.newton_raphson, F2.solve_phi_residual, F2.init_phi_random
    count = [0]

    def counting_res(*a, **k):
        count[0] += 1
        return orig_res(*a, **k)

    def recording_newton(*a, **k):
        count[0] = 0
        k["return_residual_history"] = True
        pn, mn, hist = orig_newton(*a, **k)
        hists.append(np.array(hist)); evals.append(count[0])
        print(f"    {N}^2 stress step {len(hists)}: {len(hist)} norms {np.array2string(np.array(hist), precision=3)}, "
              f"{count[0]} residual evaluations", file=sys.stderr, flush=True)
        return pn, mn
    F2.newton_raphson, F2.solve_phi_residual = recording_newton, counting_res
    F2.init_phi_random = lambda a, b, d, amp=0.1, seed=42, **k: orig_init(a, b, d, amp=1.0, seed=seed)
    try:
        with quiet():
            phi, (x, y), t_hist = F2.run_main_simulation(cfg, store_history=True, control_input=None, verbose=False)
    finally:
        F2.newton_raphson, F2.solve_phi_residual, F2.init_phi_random = orig_newton, orig_res, orig_init
    H = np.full((M, max(len(h) for h in hists)), np.nan)
    for i, h in enumerate(hists):
        H[i, :len(h)] = h
    l2 = lambda A: np.sqrt((A.reshape(A.shape[0], -1) ** 2).sum(axis=1))
    save(name, N=N, M=M, dt=dt, t_hist=t_hist, hists=H, n_hist=np.array([len(h) for h in hists]),
         res_evals=np.array(evals), phi_sub=phi[:, ::sub, ::sub], nrm_phi=l2(phi), clipped_frac0=np.mean(np.abs(phi[0]) >= 0.99))


def gen_stress256():
    gen_stress(256, 3, 4, "g2d_stress_256.npz")


def gen_stress512():
    """The near-singular start at the benched grid, 512^2: the first step of the reference (four SuperLU solves of the 1 M-row
    block system)."""
    gen_stress(512, 1, 8, "g2d_stress_512.npz")


def gen_stress1024():
    """BASELINE config 5's own grid: the FIRST step of the reference at 1024^2 (SuperLU on a 4.2 M-row block system per Newton
    iteration: more than an hour each here; the second step would run hundreds of damped Newton iterations)."""
    gen_stress(1024, 1, 16, "g2d_stress_1024.npz")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", choices=["config1", "stress256", "stress512", "stress1024"])
    a = ap.parse_args()
    if not os.path.isdir(REF):
        sys.exit("reference checkout not present: golden vectors can only be regenerated in the build container")
    os.chdir(tempfile.mkdtemp(prefix="vch_golden_"))
    import warnings
    warnings.filterwarnings("ignore")
    for name, fn in (("config1", gen_config1), ("stress256", gen_stress256), ("stress512", gen_stress512), ("stress1024", gen_stress1024)):
        if a.only == name or (a.only is None and name not in ("stress512", "stress1024")):
            print(name); fn()
