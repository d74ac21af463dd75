"""GPU parity tests of the 1D HIP engine (persistent per-trajectory workgroups, block cyclic
reduction) through the C ABI, against the reference's golden vectors and the CPU oracle.

Tolerances: SOLVE 1e-9 for N <= 64 (well-conditioned: cyclic reduction and LAPACK agree to
round-off); the N = 4096 adjoint system has condition number ~1e13, where LAPACK's own residual
is ~1e-4, so only the regime is pinned there (SURVEY 7 'round-off-limited regimes')."""
import numpy as np
import pytest

from conftest import golden, relerr

pytestmark = pytest.mark.gpu
OPS, SOLVE = 1e-12, 1e-9


@pytest.fixture(scope="module")
def V():
    import vch_amd
    vch_amd.build()
    return vch_amd


@pytest.fixture(scope="module")
def O1():
    from oracle import vch1d_oracle
    return vch1d_oracle


def test_operators_and_solves_vs_golden(V):
    g = golden("g1d_ops_24.npz")
    N, dt = int(g["N"]), float(g["dt"])
    e = V.Engine1D(N=N, Lx=float(g["Lx"]), tau=float(g["tau"]), gamma=float(g["gamma"]), c1=float(g["c1"]),
                   c2=float(g["c2"]), kappa=float(g["kappa"]))
    assert relerr(e.apply_laplacian(g["v"]), g["Lv"]) < OPS
    assert relerr(e.apply_laplacian(e.apply_laplacian(g["v"])), g["LLv"]) < OPS
    Rp, Rm = e.residuals(g["phi_new"], g["phi_old"], g["mu_new"], g["mu_old"], g["w_new"], g["w_old"], dt)
    assert relerr(Rp, g["Rphi"]) < OPS and relerr(Rm, g["Rmu"]) < OPS
    n = N + 1
    d = g["dvec"]
    dphi, dmu = e.jacobian_solve(g["phi_new"], dt, d[:n], d[n:])           # np.linalg.solve(J, d), F1:185
    assert relerr(dphi, g["Jsol"][:n]) < SOLVE and relerr(dmu, g["Jsol"][n:]) < SOLVE
    assert relerr(e.adjoint_solve(g["phi_new"], dt, g["v"]), g["Asol"]) < SOLVE      # B1:116
    assert relerr(e.adjoint_solve(None, 0.0, g["v"]), g["ATsol"]) < SOLVE            # B1:94


@pytest.mark.parametrize("N", [24, 100, 256, 1024, 1500, 2048, 4096])
def test_cyclic_reduction_all_level_counts(V, O1, N):
    """0, 1 and 2 implicit CR levels, power-of-two and ragged sizes, batch 3: the solution
    satisfies the Newton system (checked with the oracle's matrix-free operator)."""
    rng = np.random.default_rng(N)
    n = N + 1
    e = V.Engine1D(N=N, batch=3)
    P = O1.Params1D(N=N)
    phi = rng.uniform(-0.9, 0.9, (3, n))
    a, b_ = rng.standard_normal((3, n)), rng.standard_normal((3, n))
    dt = 1e-3
    dphi, dmu = e.jacobian_solve(phi, dt, a, b_)
    for k in range(3):
        ref = O1._solve_newton_banded(phi[k], dt, P, 1.0 / N, -np.concatenate([a[k], b_[k]]))
        tol = 1e-9 if N <= 256 else 1e-6
        assert relerr(dphi[k], ref[:n]) < tol and relerr(dmu[k], ref[n:]) < tol, (N, k)


def test_newton_vs_golden(V):
    g = golden("g1d_newton_64.npz")
    e = V.Engine1D(N=64)
    pn, mn, hist = e.newton_raphson(g["phi0"], g["mu0"], g["w0"], g["w1"], 1e-2)
    assert len(hist) == len(g["hist"]), (hist, g["hist"])
    assert np.allclose(hist[:-1], g["hist"][:-1], rtol=1e-6)
    assert relerr(pn, g["phi_new"]) < SOLVE and relerr(mn, g["mu_new"]) < SOLVE


@pytest.mark.parametrize("tag", ["32", "64", "64_ragged", "64_ic"])
def test_forward_backward_cost_vs_golden(V, O1, tag):
    g = golden(f"g1d_forward_{tag}.npz")
    N = int(g["N"])
    t = g["t_hist"]
    rows = len(t)
    M = rows - 2
    e = V.Engine1D(N=N, max_steps=M)
    tg, dts = V.time_grid(float(g["T"]), float(g["dt"]))
    assert np.array_equal(np.concatenate([[0.0], tg]), t)          # duplicated t = 0 row (F1:329-336)
    phi0 = g["phi_nat"][0]
    ph, st = e.forward(phi0, dts)
    assert ph.shape == (rows, N + 1) and np.array_equal(ph[0], ph[1])
    assert relerr(ph, g["phi_nat"]) < SOLVE, st
    ph_u, st = e.forward(phi0, dts, u=g["u"])
    assert relerr(ph_u, g["phi_u"]) < SOLVE, st
    ph_s, _ = e.forward(phi0, dts, u=g["u"][:rows - 2])            # hold-last branch F1:351-353
    assert relerr(ph_s, g["phi_ushort"]) < SOLVE
    with pytest.raises(IndexError):
        e.forward(phi0, dts, u=g["u"][:3])
    opt = O1.OptParams1D()
    p, q, r = e.backward(g["phi_u"], t, opt.b1, opt.b2, g["phi_Q_1"], g["phi_T_1"])
    assert relerr(p, g["p"]) < SOLVE and relerr(q, g["q"]) < SOLVE and relerr(r, g["r"]) < SOLVE
    assert not p[0].any() and not r[0].any()                        # B1:110: the dt = 0 row stays zero
    _, _, r0 = e.backward(g["phi_u"], t, 1.3, 0.7, None, None)
    assert relerr(r0, g["r_none"]) < SOLVE
    J = e.cost(g["phi_u"], g["u"], g["phi_Q_1"], g["phi_T_1"], g["x"], t, V.make_opt(opt))
    assert abs(J[4] / float(g["J"]) - 1) < 1e-12
    un = e.grad_prox(g["u"], g["r"], 7.0, V.make_opt(opt))
    assert relerr(un, g["prox"]) < 1e-14


def test_frozen_adjoint_parameters(V, O1):
    """B1:29-33: the adjoint uses the default tau, gamma, c1, c2 whatever the run-time config is."""
    g = golden("g1d_forward_32.npz")
    t = g["t_hist"]
    e = V.Engine1D(N=32, tau=0.2, gamma=3.0, c1=0.5, c2=0.9, max_steps=len(t))
    p, q, r = e.backward(g["phi_u"], t, 0.3, 13.0, g["phi_Q_1"], g["phi_T_1"])
    assert relerr(r, g["r"]) < SOLVE


def test_forward_batch_and_invariants(V, O1):
    """Batch of 4 trajectories = 4 single runs (bitwise); mass drift <= 1e-12 (T1f:185-223)."""
    N, T, dt = 128, 0.2, 1e-2
    _, dts = V.time_grid(T, dt)
    phi0 = np.stack([O1.init_phi_random(N, 1e-2, amp=0.01, seed=42 + i) for i in range(4)])
    e4 = V.Engine1D(N=N, batch=4, max_steps=len(dts))
    e1 = V.Engine1D(N=N, batch=1, max_steps=len(dts))
    ph4, _ = e4.forward(phi0, dts)
    for b in range(4):
        ph1, _ = e1.forward(phi0[b], dts)
        assert np.array_equal(ph4[b], ph1)
    w = O1.trapz_weights(N + 1) / N
    mass = ph4 @ w
    assert np.max(np.abs(mass - mass[:, :1])) <= 1e-12
    P = O1.Params1D(N=N, T=T, dt_initial=dt)
    ref, _, _ = O1.forward(P, seed=42)
    assert relerr(ph4[0], ref) < SOLVE


def test_n4096_roundoff_regime(V, O1):
    """Config 2 (N = 4096, white-noise IC): Newton stalls above tol and leaves through the
    line-search-failure return like the reference (norm history recorded from the reference:
    1.3e12 -> 9.0e-4 -> 5.9e-5 -> 5.0e-5)."""
    g = golden("g1d_newton_4096_norms.npz")
    N = 4096
    e = V.Engine1D(N=N)
    P = O1.Params1D(N=N)
    h = 1.0 / N
    phi0 = O1.init_phi_random(N, 1e-2, amp=0.01, seed=42)
    w0 = np.zeros(N + 1)
    mu0 = O1.mu_init(phi0, w0, P, h)
    pn, mn, hist = e.newton_raphson(phi0, mu0, w0, w0, 1e-3)
    assert abs(hist[0] / g["norms"][0] - 1) < 1e-9
    assert 1e-6 < hist[-1] < 1e-2 and len(hist) <= 8
    assert relerr(pn[::16], g["phi_new_sub"]) < 1e-6
    # a short full-size march runs and conserves mass
    _, dts = V.time_grid(5e-3, 1e-3)
    ef = V.Engine1D(N=N, max_steps=len(dts))
    ph, st = ef.forward(phi0, dts)
    w = O1.trapz_weights(N + 1) / N
    assert np.max(np.abs(ph @ w - (ph @ w)[0])) <= 1e-12 and np.all(np.isfinite(ph))


# ---------------------------------------------------------------------------------------
# device-resident PGD (vch1d_pgd_*)
# ---------------------------------------------------------------------------------------
def test_pgd_resident_vs_reference_golden(V):
    """The device-resident loop against PGD iterations made with the reference's own functions
    (incl. a forced backtracking case): costs, steps, trial counts, final control/state, targets."""
    G1 = V.module("Vch_control_1D.GD_1D")
    K1 = V.module("Vch_control_1D.config")
    for tag in ("32", "32_bt"):
        gp = golden(f"g1d_pgd_{tag}.npz")
        cfg = K1.ForwardSolverConfig(N=int(gp["N"]), T=float(gp["T"]), dt_initial=float(gp["dt"]))
        opt = K1.OptimizationConfig(alpha_max=float(gp["alpha_max"]))
        res = G1.run_optimization_resident(cfg, opt, n_iter=int(gp["n_iter"]))
        assert np.allclose(res["costs"], gp["costs"], rtol=1e-9), (res["costs"], gp["costs"])
        assert np.allclose(res["alphas"], gp["alphas"], rtol=1e-14) and list(res["trials"]) == list(gp["trials"])
        assert relerr(res["u"], gp["u_final"]) < 1e-8 and relerr(res["phi"], gp["phi_final"]) < 1e-8
        assert relerr(res["phi_Q"], gp["phi_Q"]) < 1e-15 and relerr(res["r"], gp["r_last"]) < 1e-8
        assert np.allclose(res["t_hist"], gp["t_hist"], rtol=0, atol=0)


def test_pgd_resident_batch_matches_function_seam(V):
    """A batch of three seeds (line searches of different length, the accepted trajectories skipping
    the remaining trial marches) against the function-seam loop run seed by seed."""
    G1 = V.module("Vch_control_1D.GD_1D")
    K1 = V.module("Vch_control_1D.config")
    F1 = V.module("Vch_control_1D.Forward_solver")
    cfg = K1.ForwardSolverConfig(N=48, T=0.08, dt_initial=1e-2)
    opt = K1.OptimizationConfig(alpha_max=2.0e5)
    seeds = [42, 43, 44]
    phi0 = np.stack([F1.init_phi_random(48, 1e-2, amp=a, seed=s) for s, a in zip(seeds, (0.01, 0.2, 0.05))])
    rb = G1.run_optimization_resident(cfg, opt, n_iter=3, initial_phi=phi0)
    assert rb["u"].shape == (3, 10, 49)
    assert len({tuple(t) for t in rb["trials"]}) > 1, rb["trials"]        # the searches really differ
    for b in range(3):
        r1 = G1.run_optimization_resident(cfg, opt, n_iter=3, initial_phi=phi0[b])
        assert np.array_equal(rb["trials"][b], r1["trials"])
        assert np.allclose(rb["costs"][b], r1["costs"], rtol=1e-13)
        assert relerr(rb["u"][b], r1["u"]) < 1e-12 and relerr(rb["phi"][b], r1["phi"]) < 1e-12
    # seed 42 is the reference's own start: the function-seam loop must agree too
    rs = G1.run_optimization(cfg, opt, n_iter=3)
    assert list(rs["trials"]) == list(rb["trials"][0])
    assert np.allclose(rs["costs"], rb["costs"][0], rtol=1e-12)
    assert relerr(rs["u"], rb["u"][0]) < 1e-10


def test_bitwise_reproducibility_1d(V):
    F1 = V.module("Vch_control_1D.Forward_solver")
    N, M = 200, 12
    _, dts = V.time_grid(M * 1e-2, 1e-2)
    phi0 = np.stack([F1.init_phi_random(N, 1e-2, amp=0.05, seed=s) for s in (1, 2)])
    runs = []
    for rep in range(2):
        e = V.Engine1D(N=N, batch=2, max_steps=len(dts))
        runs.append(e.forward(phi0, dts)[0])
        e.close()
    assert np.array_equal(runs[0], runs[1])


def test_config1_full_size_vs_reference(V, O1):
    """BASELINE config 1 at its full size (1D N = 256, T = 1, dt_initial = 5e-3: 200 steps, 202 history rows, default
    K1 weights) against the reference's own run (tests/golden/make_golden_r3.py): forward march, adjoint sweep, and three
    iterations of the PGD loop G1:353-480 -- the second of them a line search that returns its last try (5 trials) --
    through vch1d_forward / _backward / _pgd_* at full size; error metrics against G1:425-450."""
    g = golden("g1d_config1_256.npz")
    N, T, dt = int(g["N"]), float(g["T"]), float(g["dt"])
    t = g["t_hist"]
    tg, dts = V.time_grid(T, dt)
    assert len(dts) == 200 and np.array_equal(np.concatenate([[0.0], tg]), t)
    F1 = V.module("Vch_control_1D.Forward_solver")
    phi0 = F1.init_phi_random(N, 1e-2, amp=0.01, seed=42, enforce_zero_mean=True)
    e = V.Engine1D(N=N, max_steps=len(dts))
    ph, st = e.forward(phi0, dts)
    assert ph.shape == (202, 257)
    assert relerr(ph[::2], g["phi_nat_sub"]) < SOLVE, st
    assert np.allclose(np.sqrt((ph ** 2).sum(axis=1)), g["nrm_phi_nat"], rtol=1e-9)
    opt = O1.OptParams1D()
    x = np.linspace(0.0, 1.0, N + 1)
    phi_T, phi_Q = O1.build_targets(x, t, ph[0].copy(), 1.0, T, 1, 1)
    assert relerr(phi_T, g["phi_T"]) < 1e-15
    p, q, r = e.backward(ph, t, opt.b1, opt.b2, phi_Q, phi_T)
    assert relerr(r[::2], g["r_nat_sub"]) < 1e-8 and relerr(p[::4], g["p_nat_sub"]) < 1e-8
    assert np.allclose(np.sqrt((q ** 2).sum(axis=1)), g["nrm_q_nat"], rtol=1e-7)
    e.close()
    G1 = V.module("Vch_control_1D.GD_1D")
    K1 = V.module("Vch_control_1D.config")
    res = G1.run_optimization_resident(K1.ForwardSolverConfig(N=N, T=T, dt_initial=dt), K1.OptimizationConfig(),
                                       n_iter=int(g["n_iter"]))
    assert np.allclose(res["costs"], g["costs"], rtol=1e-9), (res["costs"], g["costs"])
    assert np.allclose(res["alphas"], g["alphas"], rtol=1e-13) and list(res["trials"]) == list(g["trials"])
    assert max(res["trials"]) == 5
    assert relerr(res["u"][::2], g["u_final_sub"]) < 1e-7 and relerr(res["phi"][::2], g["phi_final_sub"]) < 1e-7
    assert np.allclose(res["tracking_error"], g["tracking"], rtol=1e-8)
    assert np.allclose(res["terminal_error"], g["terminal"], rtol=1e-8)
