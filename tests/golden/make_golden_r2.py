#!/usr/bin/env python3
"""Round-2 golden vectors from the reference (same rules as make_golden.py: TEST INFRASTRUCTURE,
runs only in the build container, imports the reference from /root/reference, copies nothing,
writes .npz data files only).

    python tests/golden/make_golden_r2.py [--only pgd2d|pgd1d|fwd256|stress128|soc]

What it adds over make_golden.py:
  g2d_pgd_16_plateau.npz   G2:295-382 with alpha_max = 4e4, 9 iterations: deep backtracking ("return last
                           try", G2:144-146) and the plateau boost x1.5 (G2:365-371) visible in the alphas
  g2d_pgd_16_stop.npz      kappa_sparsity = 10 (the zero control is the fixed point): costs bit-constant, every
                           line search exhausts its 10 attempts, boosts every 5 iterations, and the loop leaves
                           through `change < 1e-5 and k > 20` (G2:375-381) at k = 21
  g2d_pgd_16_err.npz       tracking / terminal error histories (G2:336-363) of the runs above and of g2d_pgd_16
  g1d_pgd_32_stop.npz      the 1D twin: plateau x2.0 after 10 iterations (G1:453-463), stop at k = 11 (G1:466-473)
  g1d_pgd_32_err.npz       G1:425-450 error histories for g1d_pgd_32 and the stop run
  g2d_forward_256.npz      BASELINE config 3 grid (256^2, dt = 1/400): 5 steps natural + controlled, adjoint
                           sweep, cost; fields sub-sampled [::4, ::4], per-level L2 norms of the full fields
  g2d_stress_128.npz       amp = 1.0 start at 128^2 (the FFT path), dt = 1e-3, 5 steps: Newton residual
                           histories and residual-evaluation counts per step, sub-sampled fields
  g2d_soc_16_pct.npz, g1d_soc_32_pct.npz   the printed statistics of verify_sparsity_condition (S2:238, G1:115)
"""
import argparse
import contextlib
import io
import os
import re
import sys
import tempfile

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import _fresh_import, quiet, save, REF, REF2D, REF1D     # noqa: E402


# ---- the error metrics of the drivers (inline closures there, so the arithmetic is restated) ----
def errs_2d(phi, phi_Q, phi_T, x, y, t):          # G2:336-363
    def l2_xy(A):
        return float(np.sqrt(max(np.trapz(np.trapz(A ** 2, x=y, axis=1), x=x), 0.0)))

    def l2_xyt(A):
        return float(np.sqrt(max(np.trapz(np.array([l2_xy(A[k]) ** 2 for k in range(A.shape[0])]), x=t), 0.0)))
    rms = float(np.sqrt(max(float((x[-1] - x[0]) * (y[-1] - y[0])), 1e-30) * max(float(t[-1] - t[0]), 1e-30)))
    numQ, denQ = l2_xyt(phi - phi_Q), l2_xyt(phi_Q)
    if denQ < 1e-9 * rms:
        denQ = rms
    return numQ / (denQ + 1e-12), l2_xy(phi[-1] - phi_T) / (l2_xy(phi_T) + 1e-12)


def errs_1d(phi, phi_Q, phi_T, x, t):             # G1:425-450
    l2_xt = lambda A: np.sqrt(np.trapz(np.trapz(A ** 2, x=x, axis=1), x=t))
    l2_x = lambda a: np.sqrt(np.trapz(a ** 2, x=x))
    rms = np.sqrt(max(float(x[-1] - x[0]), 1e-30) * max(float(t[-1] - t[0]), 1e-30))
    numQ, denQ = l2_xt(phi - phi_Q), l2_xt(phi_Q)
    if denQ < 1e-9 * rms:
        denQ = rms
    return float(numQ / (denQ + 1e-12)), float(l2_x(phi[-1] - phi_T) / (l2_x(phi_T) + 1e-12))


def gen_pgd2d():
    F2, B2, C2, K2, G2 = _fresh_import(REF2D, ["Forward2_solver", "backward2_solver", "cost2_and_function", "config",
                                               "GD2_configured"])

    def run(tag, N, T, dt, n_iter, zero_target_q=False, **optkw):
        cfg = K2.ForwardSolverConfig(Nx=N, Ny=N, T=T, dt_initial=dt)
        opt = K2.OptimizationConfig(**optkw)
        with quiet():
            phi_k, (x, y), t_k = F2.run_main_simulation(cfg, store_history=True, control_input=None, verbose=False)
            u_k = np.zeros_like(phi_k)
            phi_T, phi_Q = G2.build_targets(x, y, t_k, phi_k[0].copy(), cfg.Lx, cfg.Ly, cfg.T, interactive=False,
                                            choice_t=1, choice_q=1)
            if zero_target_q:
                phi_Q = np.zeros_like(phi_Q)      # exercises the RMS fallback of the tracking error (G2:353-354)
            cost_k = C2.calculate_cost(phi_k, u_k, phi_Q, phi_T, x, y, t_k, opt)
        costs, alphas, attempts, changes, trk, trm = [cost_k], [], [], [], [], []
        alpha_prev, plateau, stopped_at = opt.alpha_max, 0, -1
        for k in range(n_iter):
            with quiet():
                _, _, r_k = B2.run_backward(phi_k, x, y, t_k, cfg, opt.b1, opt.b2, phi_Q, phi_T)
                g = C2.calculate_gradient(r_k, u_k, opt)
                u_o = C2.proximal_step(u_k, g, alpha_prev, opt)
                phi_o, _, t_o = F2.run_main_simulation(cfg, store_history=True, control_input=u_o, verbose=False)
                c_o = C2.calculate_cost(phi_o, u_o, phi_Q, phi_T, x, y, t_o, opt)
                if c_o < cost_k:
                    a_k, u_n, c_n, phi_n, att = alpha_prev, u_o, c_o, phi_o, 0
                else:
                    a_k, u_n, c_n, phi_n, _, _, att = G2.perform_backtracking_line_search_2D(
                        u_k, cost_k, g, phi_Q, phi_T, x, y, cfg, opt, alpha_init=alpha_prev * 0.8)
            costs.append(c_n); alphas.append(a_k); attempts.append(att)
            e1, e2 = errs_2d(phi_n, phi_Q, phi_T, x, y, t_k)
            trk.append(e1); trm.append(e2)
            if k > 0 and abs(costs[-1] - costs[-2]) < 1e-5:
                plateau += 1
            else:
                plateau = 0
            if plateau >= 5:
                alpha_prev, plateau = min(opt.alpha_max, a_k * 1.5), 0
            else:
                alpha_prev = min(opt.alpha_max, a_k * 1.2)
            change = np.linalg.norm(u_n - u_k) / (np.linalg.norm(u_k) + 1e-9)
            changes.append(change)
            print(f"    {tag} k={k} J={c_n:.12f} alpha={a_k:.6g} att={att} change={change:.3e}", flush=True)
            if change < 1e-5 and k > 20:
                u_k, phi_k = u_n, phi_n
                stopped_at = k
                break
            u_k, cost_k, phi_k = u_n, c_n, phi_n
        return dict(N=N, T=T, dt=dt, n_iter=n_iter, alpha_max=opt.alpha_max, b3=opt.b3,
                    kappa_sparsity=opt.kappa_sparsity, costs=np.array(costs), alphas=np.array(alphas),
                    attempts=np.array(attempts), changes=np.array(changes), tracking=np.array(trk),
                    terminal=np.array(trm), stopped_at=stopped_at, u_final=u_k, phi_final=phi_k, phi_T=phi_T,
                    t_hist=t_k, zero_target_q=int(zero_target_q))

    save("g2d_pgd_16_plateau.npz", **run("plateau", 16, 0.1, 1e-2, 9, alpha_max=4.0e4))
    save("g2d_pgd_16_stop.npz", **run("stop", 16, 0.05, 1e-2, 40, kappa_sparsity=10.0))
    base = run("base", 16, 0.1, 1e-2, 4)
    old = np.load(os.path.join(HERE, "g2d_pgd_16.npz"))
    assert np.array_equal(old["costs"], base["costs"]), "g2d_pgd_16 is not reproduced"
    zq = run("zeroQ", 16, 0.05, 1e-2, 2, zero_target_q=True)
    save("g2d_pgd_16_err.npz", tracking=base["tracking"], terminal=base["terminal"], costs=base["costs"],
         zq_tracking=zq["tracking"], zq_terminal=zq["terminal"], zq_costs=zq["costs"], zq_alphas=zq["alphas"],
         zq_T=zq["T"], zq_dt=zq["dt"])


def gen_pgd1d():
    F1, B1, C1, K1, G1 = _fresh_import(REF1D, ["Forward_solver", "backward_solver", "cost_and_function", "config", "GD_1D"])

    def run(tag, N, T, dt, n_iter, **optkw):
        cfg = K1.ForwardSolverConfig(N=N, T=T, dt_initial=dt)
        opt = K1.OptimizationConfig(**optkw)
        with quiet():
            phi_k, x, t_hist = F1.run_main_simulation(cfg, store_history=True, verbose=False)
            u_k = np.zeros_like(phi_k)
            phi_T, phi_Q = G1.build_targets_1d(x, t_hist, phi_k[0].copy(), cfg.Lx, cfg.T, interactive=False,
                                               choice_t=1, choice_q=1)
            cost_k = C1.calculate_cost(phi_k, u_k, phi_Q, phi_T, x, t_hist, opt.b1, opt.b2, opt.b3, opt.kappa_sparsity)
        costs, alphas, trials, changes, trk, trm = [cost_k], [], [], [], [], []
        alpha_prev, plateau, stopped_at = opt.alpha_max, 0, -1
        for k in range(n_iter):
            with quiet():
                _, _, r_k = B1.run_backward(phi_k, x, t_hist, opt.b1, opt.b2, phi_Q, phi_T)
                g = C1.calculate_gradient(r_k, u_k, opt.b3)
                u_o = G1.perform_proximal_and_projection(C1.perform_gradient_step(u_k, g, alpha_prev), alpha_prev,
                                                         opt.kappa_sparsity, opt.u_min, opt.u_max)
                phi_o, _, _ = F1.run_main_simulation(cfg, store_history=True, control_input=u_o, verbose=False)
                c_o = C1.calculate_cost(phi_o, u_o, phi_Q, phi_T, x, t_hist, opt.b1, opt.b2, opt.b3,
                                        opt.kappa_sparsity, verbose=False)
                if c_o < cost_k:
                    a_k, u_n, c_n, phi_n, nt = alpha_prev, u_o, c_o, phi_o, 1
                else:
                    a_k, u_n, c_n, phi_n, _, _, nt = G1.perform_backtracking_line_search(
                        u_k, cost_k, g, phi_Q, phi_T, x, t_hist, opt.b1, opt.b2, opt.b3, opt.kappa_sparsity,
                        opt.u_min, opt.u_max, cfg, alpha_init=alpha_prev)
            costs.append(c_n); alphas.append(a_k); trials.append(nt)
            e1, e2 = errs_1d(phi_n, phi_Q, phi_T, x, t_hist)
            trk.append(e1); trm.append(e2)
            if k > 0 and abs(costs[-1] - costs[-2]) < 1e-7:
                plateau += 1
            else:
                plateau = 0
            if plateau >= 10:
                alpha_prev, plateau = min(opt.alpha_max, a_k * 2.0), 0
            else:
                alpha_prev = min(opt.alpha_max, a_k * 1.2)
            change = np.linalg.norm(u_n - u_k) / (np.linalg.norm(u_k) + 1e-9)
            changes.append(change)
            print(f"    {tag} k={k} J={c_n:.12f} alpha={a_k:.6g} trials={nt} change={change:.3e}", flush=True)
            if change < 1e-5 and k > 10:
                u_k = u_n.copy()            # G1:468: phi_k is NOT advanced on this path
                stopped_at = k
                break
            u_k, cost_k, phi_k = u_n.copy(), c_n, phi_n
        return dict(N=N, T=T, dt=dt, n_iter=n_iter, alpha_max=opt.alpha_max, kappa_sparsity=opt.kappa_sparsity,
                    costs=np.array(costs), alphas=np.array(alphas), trials=np.array(trials),
                    changes=np.array(changes), tracking=np.array(trk), terminal=np.array(trm),
                    stopped_at=stopped_at, u_final=u_k, phi_final=phi_k, phi_T=phi_T, phi_Q=phi_Q, t_hist=t_hist)

    save("g1d_pgd_32_stop.npz", **run("stop", 32, 0.1, 1e-2, 30, kappa_sparsity=10.0))
    base = run("base", 32, 0.1, 1e-2, 4, alpha_max=100.0)
    old = np.load(os.path.join(HERE, "g1d_pgd_32.npz"))
    assert np.array_equal(old["costs"], base["costs"]), "g1d_pgd_32 is not reproduced"
    save("g1d_pgd_32_err.npz", tracking=base["tracking"], terminal=base["terminal"], costs=base["costs"])


def gen_fwd256():
    F2, B2, C2, K2, G2 = _fresh_import(REF2D, ["Forward2_solver", "backward2_solver", "cost2_and_function", "config",
                                               "GD2_configured"])
    N, M = 256, 5
    dt = 1.0 / 400.0                       # BASELINE config 3: 400 steps over T = 1
    cfg = K2.ForwardSolverConfig(Nx=N, Ny=N, T=M * dt, dt_initial=dt)
    opt = K2.OptimizationConfig()
    with quiet():
        phi_nat, (x, y), t_hist = F2.run_main_simulation(cfg, store_history=True, control_input=None, verbose=False)
    print("    256^2 natural march done", flush=True)
    u = np.random.default_rng(256).uniform(-1.0, 1.0, phi_nat.shape)
    with quiet():
        phi_u, _, t_u = F2.run_main_simulation(cfg, store_history=True, control_input=u, verbose=False)
        phi_T, phi_Q = G2.build_targets(x, y, t_hist, phi_nat[0].copy(), cfg.Lx, cfg.Ly, cfg.T, interactive=False,
                                        choice_t=1, choice_q=1)
    print("    256^2 controlled march done", flush=True)
    with quiet():
        p, q, r = B2.run_backward(phi_u, x, y, t_hist, cfg, opt.b1, opt.b2, phi_Q, phi_T)
        J = C2.calculate_cost(phi_u, u, phi_Q, phi_T, x, y, t_hist, opt)
    l2 = lambda A: np.sqrt((A.reshape(A.shape[0], -1) ** 2).sum(axis=1))
    s = (slice(None), slice(None, None, 4), slice(None, None, 4))
    save("g2d_forward_256.npz", N=N, M=M, dt=dt, t_hist=t_hist, u_seed=256, phi_nat_sub=phi_nat[s], phi_u_sub=phi_u[s],
         p_sub=p[s], r_sub=r[s], J=J, nrm_phi_nat=l2(phi_nat), nrm_phi_u=l2(phi_u), nrm_p=l2(p), nrm_q=l2(q), nrm_r=l2(r),
         mass_nat=np.array([np.sum(F2.trapz_weights(N + 1)[:, None] * F2.trapz_weights(N + 1)[None, :] * lv) for lv in phi_nat]))


def gen_stress128():
    F2, K2 = _fresh_import(REF2D, ["Forward2_solver", "config"])
    N, M, dt = 128, 5, 1e-3
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
        print(f"    128^2 stress step {len(hists)}: {len(hist)} norms, {count[0]} residual evaluations, last {hist[-1]:.3e}", flush=True)
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
    save("g2d_stress_128.npz", N=N, M=M, dt=dt, t_hist=t_hist, hists=H, n_hist=np.array([len(h) for h in hists]),
         res_evals=np.array(evals), phi_sub=phi[:, ::2, ::2], nrm_phi=l2(phi), clipped_frac0=np.mean(np.abs(phi[0]) >= 0.99))


def _pct(text):
    m = re.findall(r"([0-9.]+)% \((\d+)/(\d+) points\)", text)
    mm = re.search(r"conditions match: ([0-9.]+)%", text)
    return np.array([float(m[0][0]), float(m[1][0]), float(mm.group(1))]), np.array([int(m[0][1]), int(m[1][1]), int(m[0][2])])


def gen_soc():
    (S2,) = _fresh_import(REF2D, ["second_order_conditions_2d"])
    gp, gs = np.load(os.path.join(HERE, "g2d_pgd_16.npz")), np.load(os.path.join(HERE, "g2d_soc_16.npz"))
    out = {}
    for tag, kap, tol in (("default", 1e-4, 1e-6), ("loose", 5e-2, 1e-2)):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            S2.verify_sparsity_condition(gp["u_final"], gs["r_opt"], kap, tol=tol)
        out[f"pct_{tag}"], out[f"cnt_{tag}"] = _pct(buf.getvalue())
        out[f"kappa_{tag}"], out[f"tol_{tag}"] = kap, tol
    save("g2d_soc_16_pct.npz", **out)
    (G1,) = _fresh_import(REF1D, ["GD_1D"])
    gp, gs = np.load(os.path.join(HERE, "g1d_pgd_32.npz")), np.load(os.path.join(HERE, "g1d_soc_32.npz"))
    out = {}
    for tag, kap, tol in (("default", 1e-4, 1e-6), ("loose", 5e-2, 1e-2)):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            G1.verify_sparsity_condition(gp["u_final"], gs["r_opt"], kap, tol=tol)
        out[f"pct_{tag}"], out[f"cnt_{tag}"] = _pct(buf.getvalue())
        out[f"kappa_{tag}"], out[f"tol_{tag}"] = kap, tol
    save("g1d_soc_32_pct.npz", **out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", choices=["pgd2d", "pgd1d", "fwd256", "stress128", "soc"])
    a = ap.parse_args()
    if not os.path.isdir(REF):
        sys.exit("reference checkout not present: golden vectors can only be regenerated in the build container")
    os.chdir(tempfile.mkdtemp(prefix="vch_golden_"))
    import warnings
    warnings.filterwarnings("ignore")
    for name, fn in (("pgd2d", gen_pgd2d), ("pgd1d", gen_pgd1d), ("soc", gen_soc), ("stress128", gen_stress128),
                     ("fwd256", gen_fwd256)):
        if a.only in (None, name):
            print(name); fn()
