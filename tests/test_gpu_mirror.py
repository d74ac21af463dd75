"""The reference-named Python modules (the drop-in function seam) on the GPU: the same calls the
reference's drivers and tests make, checked against the goldens."""
import contextlib
import io

import numpy as np
import pytest

from conftest import golden, relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def V():
    import vch_amd
    vch_amd.build()
    return vch_amd


def test_2d_function_seam(V):
    F2 = V.module("Vch_control_2D.Forward2_solver")
    B2 = V.module("Vch_control_2D.backward2_solver")
    C2 = V.module("Vch_control_2D.cost2_and_function")
    K2 = V.module("Vch_control_2D.config")
    g = golden("g2d_forward_16.npz")
    cfg = K2.ForwardSolverConfig(Nx=16, Ny=16, T=float(g["T"]), dt_initial=float(g["dt"]))
    opt = K2.OptimizationConfig()
    phi, (x, y), t = F2.run_main_simulation(cfg, store_history=True, control_input=None, verbose=False)
    assert np.array_equal(t, g["t_hist"]) and np.array_equal(x, g["x"])
    assert relerr(phi, g["phi_nat"]) < 1e-9
    phi_u, _, _ = F2.run_main_simulation(cfg, store_history=True, control_input=g["u"], verbose=False)
    assert relerr(phi_u, g["phi_u"]) < 1e-9
    with pytest.raises(ValueError):
        F2.run_main_simulation(cfg, store_history=True, control_input=np.zeros((3, 5, 5)), verbose=False)
    p, q, r = B2.run_backward(g["phi_u"], x, y, t, cfg, opt.b1, opt.b2, g["phi_Q_11"], g["phi_T_11"])
    assert relerr(r, g["r_11"]) < 1e-9 and relerr(p, g["p_11"]) < 1e-9
    with pytest.raises(AssertionError):
        B2.run_backward(g["phi_u"][0], x, y, t, cfg, 1.0, 1.0)
    with contextlib.redirect_stdout(io.StringIO()) as out:
        J = C2.calculate_cost(g["phi_u"], g["u"], g["phi_Q_11"], g["phi_T_11"], x, y, t, opt)
    assert abs(J / float(g["J_11"]) - 1) < 1e-12 and "Tracking Cost (J1)" in out.getvalue()
    gr = C2.calculate_gradient(g["r_11"], g["u"], opt)
    assert np.array_equal(gr, g["grad"])
    assert relerr(C2.proximal_step(g["u"], gr, 0.5, opt), g["prox_a0.5"]) < 1e-14
    # operator handles
    go = golden("g2d_ops_16.npz")
    L = F2.laplacian_matrix_neumann(16, 16, 1 / 16, 1 / 16)
    assert relerr(F2.apply_laplacian(L, go["v"], 16, 16), go["Lv"]) < 1e-12
    assert relerr((L @ go["v"].ravel()).reshape(17, 17), go["Lv"]) < 1e-12
    with pytest.raises(ValueError):
        F2.apply_laplacian(L, go["v"][:5], 16, 16)
    J = F2.assemble_jacobian(go["phi_new"], float(go["dt"]), 0.05, 0.75, 1e-4, L, 1e-2)
    assert relerr(J @ go["dvec"], go["Jd"]) < 1e-12
    assert relerr(J.solve(go["rhs"]), go["Jsol"]) < 1e-9
    gn = golden("g2d_newton_32.npz")
    L32 = F2.laplacian_matrix_neumann(32, 32, 1 / 32, 1 / 32)
    pn, mn, hist = F2.newton_raphson(gn["phi0"], gn["mu_init"], gn["w0"], gn["w1"], 1e-2, 0.05, 0.75, 1.0, 1e-4, 1e-2,
                                     L32, 32, 32, 1 / 32, 1 / 32, return_residual_history=True)
    assert len(hist) == len(gn["hist_dt1e-2"]) and relerr(pn, gn["phi_new_dt1e-2"]) < 1e-9
    assert np.array_equal(F2.init_phi_random(16, 16, 1e-2, amp=0.1, seed=43), golden("g2d_init_phi.npz")["N16_s43_a0.1"])


def test_2d_driver_loop(V):
    G2 = V.module("Vch_control_2D.GD2_configured")
    K2 = V.module("Vch_control_2D.config")
    g = golden("g2d_pgd_16_bt.npz")
    cfg = K2.ForwardSolverConfig(Nx=16, Ny=16, T=float(g["T"]), dt_initial=float(g["dt"]))
    opt = K2.OptimizationConfig(alpha_max=float(g["alpha_max"]))
    res = G2.run_optimization(cfg, opt, n_iter=int(g["n_iter"]))
    assert np.allclose(res["costs"][0], g["costs"], rtol=1e-8)
    assert list(res["attempts"][0]) == list(g["attempts"])
    assert relerr(res["u"], g["u_final"]) < 1e-8
    phi_T, phi_Q = G2.build_targets(res["x"], res["y"], res["t_hist"], res["phi"][0], 1.0, 1.0, cfg.T, choice_t=1, choice_q=1)
    assert relerr(phi_T, g["phi_T"]) < 1e-15


def test_1d_function_seam_and_driver(V):
    F1 = V.module("Vch_control_1D.Forward_solver")
    B1 = V.module("Vch_control_1D.backward_solver")
    C1 = V.module("Vch_control_1D.cost_and_function")
    G1 = V.module("Vch_control_1D.GD_1D")
    K1 = V.module("Vch_control_1D.config")
    g = golden("g1d_forward_64.npz")
    cfg = K1.ForwardSolverConfig(N=64, T=float(g["T"]), dt_initial=float(g["dt"]))
    opt = K1.OptimizationConfig()
    phi, x, t = F1.run_main_simulation(cfg, store_history=True, verbose=False)
    assert np.array_equal(t, g["t_hist"]) and relerr(phi, g["phi_nat"]) < 1e-9
    p, q, r = B1.run_backward(g["phi_u"], x, t, opt.b1, opt.b2, g["phi_Q_1"], g["phi_T_1"])
    assert relerr(r, g["r"]) < 1e-9
    with contextlib.redirect_stdout(io.StringIO()):
        J = C1.calculate_cost(g["phi_u"], g["u"], g["phi_Q_1"], g["phi_T_1"], x, t, opt.b1, opt.b2, opt.b3, opt.kappa_sparsity)
    assert abs(J / float(g["J"]) - 1) < 1e-12
    ut = C1.perform_gradient_step(g["u"], C1.calculate_gradient(g["r"], g["u"], opt.b3), 7.0)
    assert relerr(G1.perform_proximal_and_projection(ut, 7.0, opt.kappa_sparsity, opt.u_min, opt.u_max), g["prox"]) < 1e-14
    for ct in (1, 2, 3):
        phi_T, phi_Q = G1.build_targets_1d(x, t, g["phi_nat"][0], 1.0, cfg.T, choice_t=ct, choice_q=1)
        assert relerr(phi_T, g[f"phi_T_{ct}"]) < 1e-15 and relerr(phi_Q, g[f"phi_Q_{ct}"]) < 1e-15
    for tag in ("32", "32_bt"):
        gp = golden(f"g1d_pgd_{tag}.npz")
        cfgp = K1.ForwardSolverConfig(N=int(gp["N"]), T=float(gp["T"]), dt_initial=float(gp["dt"]))
        optp = K1.OptimizationConfig(alpha_max=float(gp["alpha_max"]))
        res = G1.run_optimization(cfgp, optp, n_iter=int(gp["n_iter"]))
        assert np.allclose(res["costs"], gp["costs"], rtol=1e-9)
        assert np.allclose(res["alphas"], gp["alphas"], rtol=1e-14) and list(res["trials"]) == list(gp["trials"])
        assert relerr(res["u"], gp["u_final"]) < 1e-8
    st = G1.verify_sparsity_condition(res["u"], res["r"], optp.kappa_sparsity, verbose=False)
    assert 0.0 <= st[2] <= 100.0


def test_second_order_and_sparsity_diagnostics(V):
    """the coercivity finite-difference test and the KKT sparsity statistic,
    run through the mirrored modules on the GPU, against the reference's own numbers."""
    S2 = V.module("Vch_control_2D.second_order_conditions_2d")
    K2 = V.module("Vch_control_2D.config")
    B2 = V.module("Vch_control_2D.backward2_solver")
    gp, gs = golden("g2d_pgd_16.npz"), golden("g2d_soc_16.npz")
    cfg = K2.ForwardSolverConfig(Nx=16, Ny=16, T=float(gp["T"]), dt_initial=float(gp["dt"]))
    opt = K2.OptimizationConfig()
    x = y = np.linspace(0, 1, 17)
    _, _, r_opt = B2.run_backward(gp["phi_final"], x, y, gp["t_hist"], cfg, opt.b1, opt.b2, gp["phi_Q"], gp["phi_T"])
    assert relerr(r_opt, gs["r_opt"]) < 1e-9
    assert np.array_equal(S2._generate_direction(gp["u_final"], gs["r_opt"], opt.u_min, opt.u_max, np.random.default_rng(42)), gs["h0"])
    assert np.array_equal(S2._generate_direction(gs["u_sat"], gs["r_opt"], opt.u_min, opt.u_max, np.random.default_rng(7)), gs["h_sat"])
    with contextlib.redirect_stdout(io.StringIO()):
        d2 = S2.approximate_second_order_condition_2d(
            u_star=gp["u_final"], r_star=gs["r_opt"], phi_star=gp["phi_final"], x=x, y=y, t_hist=gp["t_hist"], b1=opt.b1,
            b2=opt.b2, b3=opt.b3, kappa=opt.kappa_sparsity, phi_Q_target=gp["phi_Q"], phi_T_target=gp["phi_T"],
            u_min=opt.u_min, u_max=opt.u_max, num_directions=3, epsilon=1e-4, seed=42, fwd_config=cfg)
        st = S2.verify_sparsity_condition(gp["u_final"], gs["r_opt"], opt.kappa_sparsity)
    assert np.allclose(d2, gs["d2"], rtol=1e-5), (d2, gs["d2"])
    assert 0 <= st[2] <= 100
    # 1D
    S1 = V.module("Vch_control_1D.second_order_conditions")
    K1 = V.module("Vch_control_1D.config")
    gp, gs = golden("g1d_pgd_32.npz"), golden("g1d_soc_32.npz")
    cfg = K1.ForwardSolverConfig(N=32, T=float(gp["T"]), dt_initial=float(gp["dt"]))
    o = K1.OptimizationConfig()
    x = np.linspace(0, 1, 33)
    assert np.array_equal(S1._generate_direction(gp["u_final"], gs["r_opt"], o.u_min, o.u_max, o.kappa_sparsity, o.b3,
                                                 np.random.default_rng(42)), gs["h0"])
    d2 = S1.approximate_second_order_condition(cfg, gp["u_final"], gs["r_opt"], gp["phi_final"], x, gp["t_hist"], o.b1, o.b2,
                                               o.b3, o.kappa_sparsity, gp["phi_Q"], gp["phi_T"], o.u_min, o.u_max,
                                               num_directions=3, epsilon=1e-4, seed=42)
    assert np.allclose(d2, gs["d2"], rtol=1e-5), (d2, gs["d2"])


def test_free_energy_device_reduction(V):
    """free_energy (F2:256-319, F1:243-262) as a device reduction, against the oracle's
    restatement (itself pinned by the reference-made golden) on square, non-square and history inputs."""
    from oracle import vch2d_oracle as O2, vch1d_oracle as O1
    F2 = V.module("Vch_control_2D.Forward2_solver")
    F1 = V.module("Vch_control_1D.Forward_solver")
    rng = np.random.default_rng(0)
    for shp, hx, hy in (((9, 7), 0.1, 0.2), ((17, 17), 1 / 16, 1 / 16), ((33, 65), 0.03, 0.015)):
        phi = rng.uniform(-0.999, 0.999, shp)
        w = rng.standard_normal(shp)
        for ww, eps in ((None, None), (w, 0.5e-2)):
            ref = O2.free_energy(phi, 1e-4, 0.75, 1.0, hx, hy, w=ww, eps=eps)
            assert abs(F2.free_energy(phi, 1e-4, 0.75, 1.0, hx, hy, w=ww, eps=eps) - ref) <= 1e-13 * max(1.0, abs(ref))
    hist = rng.uniform(-0.9, 0.9, (5, 17, 17))
    E = F2.free_energy_history(hist, 2e-4, 0.7, 1.1, 1 / 16, 1 / 16)
    assert np.allclose(E, [O2.free_energy(h, 2e-4, 0.7, 1.1, 1 / 16, 1 / 16) for h in hist], rtol=1e-13)
    g = golden("g2d_ops_16.npz")
    assert abs(F2.free_energy(g["phi_old"], float(g["kappa"]), float(g["c1"]), float(g["c2"]), 1 / 16, 1 / 16, w=g["w_old"],
                              eps=0.5e-2) - float(g["free_energy"])) < 1e-12 * max(1.0, abs(float(g["free_energy"])))
    p1 = rng.uniform(-0.999, 0.999, (6, 41))
    w1 = rng.standard_normal((6, 41))
    E1 = F1.free_energy_history(p1, 9e-4, 0.75, 1.0, 1 / 40, w_hist=w1)
    assert np.allclose(E1, [O1.free_energy(a, 9e-4, 0.75, 1.0, 1 / 40, w=b) for a, b in zip(p1, w1)], rtol=1e-13)
    assert abs(F1.free_energy(p1[0], 9e-4, 0.75, 1.0, 1 / 40) - O1.free_energy(p1[0], 9e-4, 0.75, 1.0, 1 / 40)) < 1e-13
    g1 = golden("g1d_ops_24.npz")
    E24 = F1.free_energy(g1["phi_old"], float(g1["kappa"]), float(g1["c1"]), float(g1["c2"]), float(g1["Lx"]) / int(g1["N"]), w=g1["w_old"])
    assert abs(E24 - float(g1["free_energy"])) < 1e-13 * max(1.0, abs(float(g1["free_energy"])))



def test_driver_mains_roundtrip(V, tmp_path):
    """the non-interactive `main()` of both drivers -- parameters from the last-run
    JSON, PGD loop, final adjoint, coercivity test, sparsity statistic, `save_params`, and (1D)
    `optimal_control.npy` -- reproduce the reference-made PGD goldens and write the reference's files."""
    import json
    K2 = V.module("Vch_control_2D.config")
    G2 = V.module("Vch_control_2D.GD2_configured")
    K1 = V.module("Vch_control_1D.config")
    G1 = V.module("Vch_control_1D.GD_1D")
    gp = golden("g2d_pgd_16.npz")
    f2 = str(tmp_path / "last_run_config_2d.json")
    K2.save_params(K2.ForwardSolverConfig(Nx=16, Ny=16, T=float(gp["T"]), dt_initial=float(gp["dt"])),
                   K2.OptimizationConfig(alpha_max=float(gp["alpha_max"])), 0, f2)
    out = G2.main(n_iter=int(gp["n_iter"]), params_file=f2, num_directions=2, verbose=False)
    assert np.allclose(out["costs"][0], gp["costs"], rtol=1e-9) and relerr(out["u"], gp["u_final"]) < 1e-8
    assert len(out["hessian_values"]) == 2 and all(np.isfinite(out["hessian_values"])) and 0 <= out["sparsity"][2] <= 100
    saved = json.load(open(f2))
    assert saved["last_run_iterations"] == int(gp["n_iter"]) and saved["forward_solver"]["Nx"] == 16
    assert K2.load_params(f2).optimization.alpha_max == float(gp["alpha_max"])
    g1 = golden("g1d_pgd_32.npz")
    f1, fu = str(tmp_path / "last_run_config.json"), str(tmp_path / "optimal_control.npy")
    K1.save_params(K1.ForwardSolverConfig(N=int(g1["N"]), T=float(g1["T"]), dt_initial=float(g1["dt"])),
                   K1.OptimizationConfig(alpha_max=float(g1["alpha_max"])), 0, f1)
    o1 = G1.main(n_iter=int(g1["n_iter"]), params_file=f1, control_file=fu, num_directions=2, verbose=False)
    assert np.allclose(o1["costs"], g1["costs"], rtol=1e-9)
    assert relerr(np.load(fu), g1["u_final"]) < 1e-8 and relerr(o1["r_optimal"], golden("g1d_soc_32.npz")["r_opt"]) < 1e-8
    assert len(o1["hessian_values"]) == 2 and json.load(open(f1))["last_run_iterations"] == int(g1["n_iter"])
