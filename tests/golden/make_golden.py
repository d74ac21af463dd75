#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/ from the reference.

TEST INFRASTRUCTURE.  Runs ONLY in the build container, where the reference
checkout is mounted read-only at /root/reference.  It imports the reference's
Python modules (never copies them), calls them on small seeded inputs and stores
inputs + outputs as .npz data files.  The GPU box has no /root/reference; tests
there read the .npz files only.

Usage:  python tests/golden/make_golden.py [--only 2d|1d] [--big]

`--big` additionally records norms / sub-sampled fields of two full-size (512^2)
reference forward steps (minutes of SuperLU time; output stays < 1 MB).

Nothing is written under /root/reference: bytecode writing is disabled and the
working directory is a scratch directory.
"""
import argparse
import contextlib
import importlib
import io
import os
import sys
import tempfile

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src"
REF2D = os.path.join(REF, "2D", "Vch_control_2D")
REF1D = os.path.join(REF, "1D", "Vch_control_1D")


def _fresh_import(path, names):
    """Import reference modules by bare name from `path` (they import each other
    that way), dropping any same-named module imported before (1D vs 2D config)."""
    for n in ("config", "Forward2_solver", "backward2_solver", "cost2_and_function",
              "GD2_configured", "second_order_conditions_2d", "visualization_3d",
              "Forward_solver", "backward_solver", "cost_and_function", "GD_1D",
              "second_order_conditions"):
        sys.modules.pop(n, None)
    sys.path[:] = [p for p in sys.path if p not in (REF2D, REF1D)]
    sys.path.insert(0, path)
    with contextlib.redirect_stdout(io.StringIO()):
        mods = [importlib.import_module(n) for n in names]
    return mods


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()):
        yield


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f"  wrote {name}: {os.path.getsize(path)/1024:.1f} KiB")


# --------------------------------------------------------------------------
# 2D
# --------------------------------------------------------------------------
def gen_2d(big=False):
    F2, B2, C2, K2, G2 = _fresh_import(
        REF2D, ["Forward2_solver", "backward2_solver", "cost2_and_function", "config",
                "GD2_configured"])
    import scipy.sparse as sps

    delta_sep = 1e-2

    # ---- 1. operators on a square and a non-square grid ------------------
    for tag, (Nx, Ny, Lx, Ly) in {"16": (16, 16, 1.0, 1.0), "12x9": (12, 9, 1.0, 0.7)}.items():
        rng = np.random.default_rng(1234 + Nx)
        hx, hy = Lx / Nx, Ly / Ny
        shp = (Nx + 1, Ny + 1)
        L = F2.laplacian_matrix_neumann(Nx, Ny, hx, hy)
        v = rng.standard_normal(shp)
        phi_new = rng.uniform(-0.9, 0.9, shp)
        phi_old = rng.uniform(-0.9, 0.9, shp)
        # a few nodes beyond the clips (regularized_log clip 0.995, jacobian clip, fpp clip)
        phi_new.flat[3] = 0.9991
        phi_new.flat[7] = -0.99995
        phi_old.flat[5] = 0.9999999995
        mu_new = rng.standard_normal(shp)
        mu_old = rng.standard_normal(shp)
        w_new = rng.standard_normal(shp)
        w_old = rng.standard_normal(shp)
        u_n = rng.uniform(-1, 1, shp)
        u_np1 = rng.uniform(-1, 1, shp)
        dt, tau, gamma, c1, c2, kappa = 1e-2, 0.05, 10.0, 0.75, 1.0, 1e-4
        Lv = F2.apply_laplacian(L, v, Nx, Ny)
        LLv = F2.apply_laplacian(L, Lv, Nx, Ny)
        L2v = ((L @ L) @ v.ravel()).reshape(shp)
        mu0 = F2.initialize_mu(phi_old, w_new, c1, c2, kappa, L, Nx, Ny, delta_sep)
        wn = F2.solve_w(w_old, dt, gamma, u_n, u_np1)
        Rphi = F2.solve_phi_residual(phi_new, phi_old, mu_new, mu_old, w_new, w_old, dt, tau,
                                     c1, c2, kappa, L, Nx, Ny, delta_sep)
        Rmu = F2.solve_mu_residual(phi_new, phi_old, mu_new, mu_old, dt, L, Nx, Ny)
        J = F2.assemble_jacobian(phi_new, dt, tau, c1, kappa, L, delta_sep)
        dvec = rng.standard_normal(2 * v.size)
        Jd = J @ dvec
        rhs = rng.standard_normal(2 * v.size)
        from scipy.sparse.linalg import spsolve
        Jsol = spsolve(J.tocsc(), rhs)
        fpp = B2.fpp_log(phi_old, c1, c2)
        Nloc = v.size
        I = sps.eye(Nloc, format="csr")
        LL = (L @ L).tocsr()
        Dn = sps.diags(B2.fpp_log(phi_new.ravel(), c1, c2), 0, format="csr")
        A = (I - tau * L + 0.5 * dt * LL - 0.5 * dt * (Dn @ L)).tocsr()       # B2:198
        Bm = (I - tau * L - 0.5 * dt * LL + 0.5 * dt * (Dn @ L)).tocsr()      # B2:203
        Av = (A @ v.ravel()).reshape(shp)
        Bv = (Bm @ v.ravel()).reshape(shp)
        Asol = spsolve(A.tocsc(), v.ravel()).reshape(shp)
        AT = (I - tau * L).tocsc()
        ATsol = spsolve(AT, v.ravel()).reshape(shp)
        reglog = F2.regularized_log(phi_new, delta_sep)
        E = F2.free_energy(phi_old, kappa, c1, c2, hx, hy, w=w_old, eps=0.5 * delta_sep)
        save(f"g2d_ops_{tag}.npz", Nx=Nx, Ny=Ny, Lx=Lx, Ly=Ly, dt=dt, tau=tau, gamma=gamma,
             c1=c1, c2=c2, kappa=kappa, delta_sep=delta_sep,
             v=v, phi_new=phi_new, phi_old=phi_old, mu_new=mu_new, mu_old=mu_old,
             w_new=w_new, w_old=w_old, u_n=u_n, u_np1=u_np1, dvec=dvec, rhs=rhs,
             Lv=Lv, LLv=LLv, L2v=L2v, mu0=mu0, w_filt=wn, Rphi=Rphi, Rmu=Rmu, Jd=Jd,
             Jsol=Jsol, fpp=fpp, Av=Av, Bv=Bv, Asol=Asol, ATsol=ATsol, reglog=reglog,
             free_energy=E, trapz_w=F2.trapz_weights(Nx + 1))

    # ---- 2. init_phi_random ------------------------------------------------
    ic = {}
    for N in (16, 32):
        for seed in (42, 43, 44, 45):
            for amp in (0.1, 1.0):
                ic[f"N{N}_s{seed}_a{amp}"] = F2.init_phi_random(N, N, delta_sep, amp=amp, seed=seed)
    ic["N14x11_s42_a0.1"] = F2.init_phi_random(14, 11, delta_sep, amp=0.1, seed=42)
    save("g2d_init_phi.npz", **ic)

    # ---- 3. one Newton call (inputs in the style of T2f:425-440) ----------
    Nx = Ny = 32
    cfg = K2.ForwardSolverConfig(Nx=Nx, Ny=Ny)
    hx, hy = cfg.Lx / Nx, cfg.Ly / Ny
    L = F2.laplacian_matrix_neumann(Nx, Ny, hx, hy)
    phi0 = F2.init_phi_random(Nx, Ny, delta_sep, amp=0.1, seed=42)
    w0 = np.zeros_like(phi0)
    mu_init = F2.initialize_mu(phi0, w0, cfg.c1, cfg.c2, cfg.kappa, L, Nx, Ny, delta_sep)
    rng = np.random.default_rng(7)
    w1 = 0.01 * rng.standard_normal(phi0.shape)
    newton = {}
    for tag, dt in (("dt1e-2", 1e-2), ("dt1e-3", 1e-3)):
        pn, mn, hist = F2.newton_raphson(phi0, mu_init, w0, w1, dt, cfg.tau, cfg.c1, cfg.c2,
                                         cfg.kappa, delta_sep, L, Nx, Ny, hx, hy,
                                         return_residual_history=True)
        newton[f"phi_new_{tag}"] = pn
        newton[f"mu_new_{tag}"] = mn
        newton[f"hist_{tag}"] = np.array(hist)
    # near-singular start: amp=1.0 (about a third of the nodes clipped at +-0.99)
    phis = F2.init_phi_random(Nx, Ny, delta_sep, amp=1.0, seed=42)
    mus = F2.initialize_mu(phis, w0, cfg.c1, cfg.c2, cfg.kappa, L, Nx, Ny, delta_sep)
    pn, mn, hist = F2.newton_raphson(phis, mus, w0, w0, 1e-3, cfg.tau, cfg.c1, cfg.c2,
                                     cfg.kappa, delta_sep, L, Nx, Ny, hx, hy,
                                     return_residual_history=True)
    save("g2d_newton_32.npz", phi0=phi0, mu_init=mu_init, w0=w0, w1=w1,
         phi_stress=phis, mu_stress=mus, phi_new_stress=pn, mu_new_stress=mn,
         hist_stress=np.array(hist), **newton)

    # ---- 4./5./6. forward, backward, cost/grad/prox ------------------------
    def forward_case(tag, Nx, Ny, Lx, Ly, T, dt, with_u, amp=None, extra=None, drop=()):
        cfg = K2.ForwardSolverConfig(Nx=Nx, Ny=Ny, Lx=Lx, Ly=Ly, T=T, dt_initial=dt)
        opt = K2.OptimizationConfig()
        orig = F2.init_phi_random
        if amp is not None:       # the tests' own monkey-patch idiom (T2f:294-297)
            F2.init_phi_random = lambda a, b, d, amp=0.1, seed=42, **k: orig(a, b, d, amp=amp_, seed=seed)
            amp_ = amp
        try:
            with quiet():
                phi_nat, (x, y), t_hist = F2.run_main_simulation(cfg, store_history=True,
                                                                 control_input=None, verbose=False)
            out = dict(Nx=Nx, Ny=Ny, Lx=Lx, Ly=Ly, T=T, dt=dt, x=x, y=y, t_hist=t_hist,
                       phi_nat=phi_nat)
            if with_u:
                rng = np.random.default_rng(99)
                u = rng.uniform(-1.0, 1.0, phi_nat.shape)
                with quiet():
                    phi_u, _, t_u = F2.run_main_simulation(cfg, store_history=True,
                                                           control_input=u, verbose=False)
                    # control array shorter than the march: zeros beyond it (F2:545-548)
                    phi_us, _, _ = F2.run_main_simulation(cfg, store_history=True,
                                                          control_input=u[:4], verbose=False)
                assert np.array_equal(t_u, t_hist)
                out.update(u=u, phi_u=phi_u, phi_ushort=phi_us)
                for (ct, cq) in ((1, 1), (2, 2)):
                    with quiet():
                        phi_T, phi_Q = G2.build_targets(x, y, t_hist, phi_nat[0].copy(), Lx, Ly, T,
                                                        interactive=False, choice_t=ct, choice_q=cq)
                        p, q, r = B2.run_backward(phi_u, x, y, t_hist, cfg, opt.b1, opt.b2,
                                                  phi_Q, phi_T)
                        J = C2.calculate_cost(phi_u, u, phi_Q, phi_T, x, y, t_hist, opt)
                    out.update({f"phi_T_{ct}{cq}": phi_T, f"phi_Q_{ct}{cq}": phi_Q,
                                f"p_{ct}{cq}": p, f"q_{ct}{cq}": q, f"r_{ct}{cq}": r,
                                f"J_{ct}{cq}": J})
                    if (ct, cq) == (1, 1):
                        g = C2.calculate_gradient(r, u, opt)
                        for a in (0.5, 50.0):
                            out[f"prox_a{a}"] = C2.proximal_step(u, g, a, opt)
                        out["grad"] = g
                        # cost components one by one (C2:80-106)
                        J1 = (opt.b1 / 2) * np.trapz(np.trapz(np.trapz((phi_u - phi_Q) ** 2, y, axis=-1), x, axis=-1), x=t_hist)
                        J2 = (opt.b2 / 2) * np.trapz(np.trapz((phi_u[-1] - phi_T) ** 2, y, axis=-1), x, axis=-1)
                        J3 = (opt.b3 / 2) * np.trapz(np.trapz(np.trapz(u ** 2, y, axis=-1), x, axis=-1), x=t_hist)
                        J4 = opt.kappa_sparsity * np.trapz(np.trapz(np.trapz(np.abs(u), y, axis=-1), x, axis=-1), x=t_hist)
                        out["Jparts"] = np.array([J1, J2, J3, J4])
                with quiet():
                    # backward with no targets (None -> zeros, B2:167-168)
                    p0, q0, r0 = B2.run_backward(phi_u, x, y, t_hist, cfg, 1.3, 0.7, None, None)
                out.update(p_none=p0, q_none=q0, r_none=r0)
            if extra:
                out.update(extra(cfg, F2))
        finally:
            F2.init_phi_random = orig
        for k in drop:              # keep each fixture around 1 MB
            out.pop(k, None)
        save(f"g2d_forward_{tag}.npz", **out)

    forward_case("16", 16, 16, 1.0, 1.0, 0.1, 1e-2, True)
    forward_case("16_ragged", 16, 16, 1.0, 1.0, 0.045, 1e-2, True)       # dt does not divide T
    forward_case("32", 32, 32, 1.0, 1.0, 0.1, 1e-2, True,
                 drop=("p_22", "q_22", "phi_Q_22", "p_none", "q_none", "phi_ushort", "prox_a0.5"))
    forward_case("14x11", 14, 11, 1.0, 0.8, 0.05, 1e-2, True)
    forward_case("32_stress", 32, 32, 1.0, 1.0, 5e-3, 1e-3, False, amp=1.0)
    forward_case("64_fine", 64, 64, 1.0, 1.0, 5e-3, 1e-3, True,
                 drop=("p_22", "q_22", "r_22", "phi_Q_22", "phi_T_22", "J_22", "p_none", "q_none",
                       "r_none", "phi_ushort", "prox_a0.5", "prox_a50.0", "grad", "q_11", "phi_Q_11"))

    # ---- 7. PGD iterations through the reference functions (G2:291-382) -----
    def pgd_case(tag, N, T, dt, alpha_max, n_iter, b3=None):
        cfg = K2.ForwardSolverConfig(Nx=N, Ny=N, T=T, dt_initial=dt)
        kw = dict(alpha_max=alpha_max)
        if b3 is not None:
            kw["b3"] = b3
        opt = K2.OptimizationConfig(**kw)
        with quiet():
            phi_k, (x, y), t_k = F2.run_main_simulation(cfg, store_history=True, control_input=None, verbose=False)
            u_k = np.zeros_like(phi_k)
            phi_T, phi_Q = G2.build_targets(x, y, t_k, phi_k[0].copy(), cfg.Lx, cfg.Ly, cfg.T,
                                            interactive=False, choice_t=1, choice_q=1)
            cost_k = C2.calculate_cost(phi_k, u_k, phi_Q, phi_T, x, y, t_k, opt)
        costs, alphas, attempts_l, changes = [cost_k], [], [], []
        alpha_prev = opt.alpha_max
        plateau = 0
        for k in range(n_iter):
            with quiet():
                _, _, r_k = B2.run_backward(phi_k, x, y, t_k, cfg, opt.b1, opt.b2, phi_Q, phi_T)
                g = C2.calculate_gradient(r_k, u_k, opt)
                u_o = C2.proximal_step(u_k, g, alpha_prev, opt)
                phi_o, _, t_o = F2.run_main_simulation(cfg, store_history=True, control_input=u_o, verbose=False)
                c_o = C2.calculate_cost(phi_o, u_o, phi_Q, phi_T, x, y, t_o, opt)
                if c_o < cost_k:
                    a_k, u_n, c_n, phi_n, t_n, att = alpha_prev, u_o, c_o, phi_o, t_o, 0
                else:
                    a_k, u_n, c_n, phi_n, t_n, _, att = G2.perform_backtracking_line_search_2D(
                        u_k, cost_k, g, phi_Q, phi_T, x, y, cfg, opt, alpha_init=alpha_prev * 0.8)
            costs.append(c_n)
            alphas.append(a_k)
            attempts_l.append(att)
            if k > 0 and abs(costs[-1] - costs[-2]) < 1e-5:
                plateau += 1
            else:
                plateau = 0
            if plateau >= 5:
                alpha_prev, plateau = min(opt.alpha_max, a_k * 1.5), 0
            else:
                alpha_prev = min(opt.alpha_max, a_k * 1.2)
            changes.append(np.linalg.norm(u_n - u_k) / (np.linalg.norm(u_k) + 1e-9))
            u_k, cost_k, phi_k, t_k = u_n, c_n, phi_n, t_n
        save(f"g2d_pgd_{tag}.npz", N=N, T=T, dt=dt, alpha_max=alpha_max, n_iter=n_iter,
             b3=opt.b3, costs=np.array(costs), alphas=np.array(alphas),
             attempts=np.array(attempts_l), changes=np.array(changes), u_final=u_k,
             phi_final=phi_k, phi_T=phi_T, phi_Q=phi_Q, r_last=r_k, t_hist=t_k)

    pgd_case("16", 16, 0.1, 1e-2, 50.0, 4)
    # huge step: optimistic step overshoots -> backtracking engages (G2:320-328)
    pgd_case("16_bt", 16, 0.1, 1e-2, 4.0e4, 3)

    # ---- 8. second-order / sparsity diagnostics on the PGD result (G2:424-437, S2) -------
    S2 = importlib.import_module("second_order_conditions_2d")
    gp = np.load(os.path.join(HERE, "g2d_pgd_16.npz"))
    cfg = K2.ForwardSolverConfig(Nx=16, Ny=16, T=float(gp["T"]), dt_initial=float(gp["dt"]))
    opt = K2.OptimizationConfig()
    x = np.linspace(0, 1, 17); y = np.linspace(0, 1, 17)
    with quiet():
        _, _, r_opt = B2.run_backward(gp["phi_final"], x, y, gp["t_hist"], cfg, opt.b1, opt.b2, gp["phi_Q"], gp["phi_T"])
        d2 = S2.approximate_second_order_condition_2d(
            u_star=gp["u_final"], r_star=r_opt, phi_star=gp["phi_final"], x=x, y=y, t_hist=gp["t_hist"],
            b1=opt.b1, b2=opt.b2, b3=opt.b3, kappa=opt.kappa_sparsity, phi_Q_target=gp["phi_Q"],
            phi_T_target=gp["phi_T"], u_min=opt.u_min, u_max=opt.u_max, num_directions=3, epsilon=1e-4, seed=42,
            fwd_config=cfg)
    rng = np.random.default_rng(42)
    h0 = S2._generate_direction(gp["u_final"], r_opt, opt.u_min, opt.u_max, rng)
    usat = gp["u_final"].copy(); usat.flat[::7] = 1.0; usat.flat[3::11] = -1.0
    hs = S2._generate_direction(usat, r_opt, opt.u_min, opt.u_max, np.random.default_rng(7))
    save("g2d_soc_16.npz", r_opt=r_opt, d2=np.array(d2), h0=h0, u_sat=usat, h_sat=hs)

    if big:
        # Full-size spot check: residual-norm histories of the first two 512^2 steps
        # (the round-off-limited regime of SURVEY 7 'hard parts') + sub-sampled fields.
        N = 512
        cfg = K2.ForwardSolverConfig(Nx=N, Ny=N, T=2e-3, dt_initial=1e-3)
        hx = hy = 1.0 / N
        L = F2.laplacian_matrix_neumann(N, N, hx, hy)
        phi = F2.init_phi_random(N, N, delta_sep, amp=0.1, seed=42)
        w = np.zeros_like(phi)
        mu = F2.initialize_mu(phi, w, cfg.c1, cfg.c2, cfg.kappa, L, N, N, delta_sep)
        hists = []
        fields = [phi[::8, ::8].copy()]
        _norm = np.linalg.norm

        def _loud(a, *k, **kw):          # progress: every residual norm as it is computed
            val = _norm(a, *k, **kw)
            print(f"      |R| = {val:.6e}", flush=True)
            return val
        np.linalg.norm = _loud
        for step in range(2):
            phi, mu, hist = F2.newton_raphson(phi, mu, w, w, 1e-3, cfg.tau, cfg.c1, cfg.c2,
                                              cfg.kappa, delta_sep, L, N, N, hx, hy,
                                              return_residual_history=True)
            hists.append(np.array(hist))
            fields.append(phi[::8, ::8].copy())
            print("   512^2 newton step", step, hist, flush=True)
        np.linalg.norm = _norm
        save("g2d_newton_512.npz", hist0=hists[0], hist1=hists[1], sub=np.array(fields))


# --------------------------------------------------------------------------
# 1D
# --------------------------------------------------------------------------
def gen_1d():
    F1, B1, C1, K1, G1 = _fresh_import(
        REF1D, ["Forward_solver", "backward_solver", "cost_and_function", "config", "GD_1D"])
    delta_sep = 1e-2

    # operators
    N, Lx = 24, 1.0
    h = Lx / N
    rng = np.random.default_rng(321)
    L = F1.laplacian_matrix_neumann(N, h)
    v = rng.standard_normal(N + 1)
    phi_new = rng.uniform(-0.9, 0.9, N + 1)
    phi_old = rng.uniform(-0.9, 0.9, N + 1)
    phi_new[3] = 0.9991
    phi_old[5] = -0.9999999995
    mu_new, mu_old, w_new, w_old = (rng.standard_normal(N + 1) for _ in range(4))
    dt, tau, gamma, c1, c2, kappa = 1e-2, 0.05, 10.0, 0.75, 1.0, 9e-4
    Rphi = F1.solve_phi_residual(phi_new, phi_old, mu_new, mu_old, w_new, w_old, dt, tau, c1, c2, L, kappa)
    Rmu = F1.solve_mu_residual(phi_new, phi_old, mu_new, mu_old, dt, L)
    J = F1.assemble_jacobian(phi_new, dt, tau, c1, L, kappa)
    dvec = rng.standard_normal(2 * (N + 1))
    I = np.eye(N + 1)
    Dn = np.diag(B1.fpp_log(phi_new))
    A = I - B1.tau * L + 0.5 * dt * (L @ L) - 0.5 * dt * (Dn @ L)       # B1:101
    Bm = I - B1.tau * L - 0.5 * dt * (L @ L) + 0.5 * dt * (Dn @ L)      # B1:105
    save("g1d_ops_24.npz", N=N, Lx=Lx, dt=dt, tau=tau, gamma=gamma, c1=c1, c2=c2, kappa=kappa,
         v=v, phi_new=phi_new, phi_old=phi_old, mu_new=mu_new, mu_old=mu_old, w_new=w_new,
         w_old=w_old, dvec=dvec, Lv=L @ v, LLv=L @ (L @ v),
         mu0=F1.initialize_mu(phi_old, w_new, c1, c2, L, kappa), Rphi=Rphi, Rmu=Rmu,
         Jd=J @ dvec, Jsol=np.linalg.solve(J, dvec), fpp=B1.fpp_log(phi_old),
         Av=A @ v, Bv=Bm @ v, Asol=np.linalg.solve(A, v),
         ATsol=np.linalg.solve(I - B1.tau * L, v),
         free_energy=F1.free_energy(phi_old, kappa, c1, c2, h, w=w_old))

    ic = {}
    for n in (32, 64, 256):
        for seed in (42, 43):
            ic[f"N{n}_s{seed}"] = F1.init_phi_random(n, delta_sep, amp=0.01, seed=seed)
    save("g1d_init_phi.npz", **ic)

    def case(tag, N, T, dt, initial_phi=None):
        cfg = K1.ForwardSolverConfig(N=N, T=T, dt_initial=dt)
        opt = K1.OptimizationConfig()
        with quiet():
            phi_nat, x, t_hist = F1.run_main_simulation(cfg, store_history=True, verbose=False,
                                                        initial_phi=initial_phi)
        rng = np.random.default_rng(5)
        u = rng.uniform(-1, 1, phi_nat.shape)
        out = dict(N=N, T=T, dt=dt, x=x, t_hist=t_hist, phi_nat=phi_nat, u=u)
        if initial_phi is not None:
            out["initial_phi"] = initial_phi
        with quiet():
            phi_u, _, t_u = F1.run_main_simulation(cfg, store_history=True, control_input=u,
                                                   verbose=False, initial_phi=initial_phi)
            # control with exactly step+1 rows at the last step: the hold-last branch F1:351-353
            # (shorter arrays raise IndexError in the reference, so that is the only legal short form)
            nrow = phi_nat.shape[0] - 2
            phi_us, _, _ = F1.run_main_simulation(cfg, store_history=True, control_input=u[:nrow],
                                                  verbose=False, initial_phi=initial_phi)
        out.update(phi_u=phi_u, phi_ushort=phi_us)
        for ct in (1, 2, 3):
            with quiet():
                phi_T, phi_Q = G1.build_targets_1d(x, t_hist, phi_nat[0].copy(), cfg.Lx, cfg.T,
                                                   interactive=False, choice_t=ct, choice_q=1)
            out[f"phi_T_{ct}"] = phi_T
            out[f"phi_Q_{ct}"] = phi_Q
        phi_T, phi_Q = out["phi_T_1"], out["phi_Q_1"]
        with quiet():
            p, q, r = B1.run_backward(phi_u, x, t_hist, opt.b1, opt.b2, phi_Q, phi_T)
            Jv = C1.calculate_cost(phi_u, u, phi_Q, phi_T, x, t_hist, opt.b1, opt.b2, opt.b3,
                                   opt.kappa_sparsity)
            p0, q0, r0 = B1.run_backward(phi_u, x, t_hist, 1.3, 0.7, None, None)
        g = C1.calculate_gradient(r, u, opt.b3)
        ut = C1.perform_gradient_step(u, g, 7.0)
        out.update(p=p, q=q, r=r, J=Jv, grad=g, gstep=ut, p_none=p0, q_none=q0, r_none=r0,
                   prox=G1.perform_proximal_and_projection(ut, 7.0, opt.kappa_sparsity,
                                                           opt.u_min, opt.u_max))
        save(f"g1d_forward_{tag}.npz", **out)

    case("32", 32, 0.1, 1e-2)
    case("64", 64, 0.1, 5e-3)
    case("64_ragged", 64, 0.033, 1e-2)
    xs = np.linspace(0, 1, 65)
    case("64_ic", 64, 0.05, 5e-3, initial_phi=0.3 * np.cos(2 * np.pi * xs) + 0.1 * np.cos(5 * np.pi * xs))

    # one Newton call, N=64 with history + the stalled N=4096 call (norms only)
    N = 64
    cfg = K1.ForwardSolverConfig(N=N)
    h = cfg.Lx / N
    L = F1.laplacian_matrix_neumann(N, h)
    phi0 = F1.init_phi_random(N, delta_sep, amp=0.01, seed=42)
    w0 = np.zeros(N + 1)
    mu0 = F1.initialize_mu(phi0, w0, cfg.c1, cfg.c2, L, cfg.kappa)
    w1 = 0.01 * np.random.default_rng(3).standard_normal(N + 1)
    pn, mn, hist = F1.newton_raphson(phi0, mu0, w0, w1, 1e-2, cfg.tau, cfg.c1, cfg.c2, h,
                                     delta_sep, L, cfg.kappa, return_residual_history=True)
    out = dict(phi0=phi0, mu0=mu0, w0=w0, w1=w1, phi_new=pn, mu_new=mn, hist=np.array(hist))
    save("g1d_newton_64.npz", **out)

    # PGD iterations through the reference functions (G1:333-477)
    def pgd_case(tag, N, T, dt, alpha_max, n_iter):
        cfg = K1.ForwardSolverConfig(N=N, T=T, dt_initial=dt)
        opt = K1.OptimizationConfig(alpha_max=alpha_max)
        with quiet():
            phi_k, x, t_hist = F1.run_main_simulation(cfg, store_history=True, verbose=False)
            u_k = np.zeros_like(phi_k)
            phi_T, phi_Q = G1.build_targets_1d(x, t_hist, phi_k[0].copy(), cfg.Lx, cfg.T,
                                               interactive=False, choice_t=1, choice_q=1)
            cost_k = C1.calculate_cost(phi_k, u_k, phi_Q, phi_T, x, t_hist, opt.b1, opt.b2,
                                       opt.b3, opt.kappa_sparsity)
        costs, alphas, trials = [cost_k], [], []
        alpha_prev = opt.alpha_max
        plateau = 0
        for k in range(n_iter):
            with quiet():
                _, _, r_k = B1.run_backward(phi_k, x, t_hist, opt.b1, opt.b2, phi_Q, phi_T)
                g = C1.calculate_gradient(r_k, u_k, opt.b3)
                u_o = G1.perform_proximal_and_projection(
                    C1.perform_gradient_step(u_k, g, alpha_prev), alpha_prev,
                    opt.kappa_sparsity, opt.u_min, opt.u_max)
                phi_o, _, _ = F1.run_main_simulation(cfg, store_history=True, control_input=u_o, verbose=False)
                c_o = C1.calculate_cost(phi_o, u_o, phi_Q, phi_T, x, t_hist, opt.b1, opt.b2,
                                        opt.b3, opt.kappa_sparsity, verbose=False)
                if c_o < cost_k:
                    a_k, u_n, c_n, phi_n, nt = alpha_prev, u_o, c_o, phi_o, 1
                else:
                    a_k, u_n, c_n, phi_n, _, _, nt = G1.perform_backtracking_line_search(
                        u_k, cost_k, g, phi_Q, phi_T, x, t_hist, opt.b1, opt.b2, opt.b3,
                        opt.kappa_sparsity, opt.u_min, opt.u_max, cfg, alpha_init=alpha_prev)
            costs.append(c_n)
            alphas.append(a_k)
            trials.append(nt)
            if k > 0 and abs(costs[-1] - costs[-2]) < 1e-7:
                plateau += 1
            else:
                plateau = 0
            if plateau >= 10:
                alpha_prev, plateau = min(opt.alpha_max, a_k * 2.0), 0
            else:
                alpha_prev = min(opt.alpha_max, a_k * 1.2)
            u_k, cost_k, phi_k = u_n.copy(), c_n, phi_n
        save(f"g1d_pgd_{tag}.npz", N=N, T=T, dt=dt, alpha_max=alpha_max, n_iter=n_iter,
             costs=np.array(costs), alphas=np.array(alphas), trials=np.array(trials),
             u_final=u_k, phi_final=phi_k, phi_T=phi_T, phi_Q=phi_Q, r_last=r_k, t_hist=t_hist)

    pgd_case("32", 32, 0.1, 1e-2, 100.0, 4)
    pgd_case("32_bt", 32, 0.1, 1e-2, 2.0e5, 3)

    # second-order / sparsity diagnostics on the PGD result (G1:493-518, S1)
    S1 = importlib.import_module("second_order_conditions")
    gp = np.load(os.path.join(HERE, "g1d_pgd_32.npz"))
    cfg = K1.ForwardSolverConfig(N=32, T=float(gp["T"]), dt_initial=float(gp["dt"]))
    opt = K1.OptimizationConfig()
    x = np.linspace(0, 1, 33)
    with quiet():
        _, _, r_opt = B1.run_backward(gp["phi_final"], x, gp["t_hist"], opt.b1, opt.b2, gp["phi_Q"], gp["phi_T"])
        d2 = S1.approximate_second_order_condition(cfg, gp["u_final"], r_opt, gp["phi_final"], x, gp["t_hist"],
                                                   opt.b1, opt.b2, opt.b3, opt.kappa_sparsity, gp["phi_Q"], gp["phi_T"],
                                                   opt.u_min, opt.u_max, num_directions=3, epsilon=1e-4, seed=42)
    h0 = S1._generate_direction(gp["u_final"], r_opt, opt.u_min, opt.u_max, opt.kappa_sparsity, opt.b3, np.random.default_rng(42))
    save("g1d_soc_32.npz", r_opt=r_opt, d2=np.array(d2), h0=h0)

    N = 4096
    cfg = K1.ForwardSolverConfig(N=N)
    h = cfg.Lx / N
    L = F1.laplacian_matrix_neumann(N, h)
    phi0 = F1.init_phi_random(N, delta_sep, amp=0.01, seed=42)
    w0 = np.zeros(N + 1)
    mu0 = F1.initialize_mu(phi0, w0, cfg.c1, cfg.c2, L, cfg.kappa)
    norms = []
    _norm = np.linalg.norm
    np.linalg.norm = lambda a, *k, **kw: (norms.append(float(_norm(a, *k, **kw))) or norms[-1])
    try:
        pn, mn = F1.newton_raphson(phi0, mu0, w0, w0, 1e-3, cfg.tau, cfg.c1, cfg.c2, h,
                                   delta_sep, L, cfg.kappa)
    finally:
        np.linalg.norm = _norm
    save("g1d_newton_4096_norms.npz", norms=np.array(norms), phi_new_sub=pn[::16],
         dphi_inf=np.max(np.abs(pn - phi0)))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", choices=["1d", "2d"])
    ap.add_argument("--big", action="store_true")
    a = ap.parse_args()
    if not os.path.isdir(REF):
        sys.exit("reference checkout not present: golden vectors can only be regenerated "
                 "in the build container")
    scratch = tempfile.mkdtemp(prefix="vch_golden_")
    os.chdir(scratch)
    import warnings
    warnings.filterwarnings("ignore")
    if a.only in (None, "2d"):
        print("2D goldens"); gen_2d(big=a.big)
    if a.only in (None, "1d"):
        print("1D goldens"); gen_1d()
