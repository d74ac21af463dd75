"""The reference's own test invariants (tests_2D/Test_2d_*/, tests_1D/Test_1d_*/)
re-expressed against the HIP engine through the reference-named mirror modules: same set-ups,
same thresholds (file:line of the original assertion in each docstring)."""
import contextlib
import io

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
DELTA_SEP = 1e-2


@pytest.fixture(scope="module")
def V():
    import vch_amd
    vch_amd.build()
    return vch_amd


@pytest.fixture(scope="module")
def M2(V):
    class NS:
        F = V.module("Vch_control_2D.Forward2_solver")
        B = V.module("Vch_control_2D.backward2_solver")
        C = V.module("Vch_control_2D.cost2_and_function")
        K = V.module("Vch_control_2D.config")
    return NS


@pytest.fixture(scope="module")
def default_run(M2):
    """Default 128^2 configuration, T = 1, dt = 1e-2, no control (the reference's `cfg` fixture)."""
    cfg = M2.K.ForwardSolverConfig()
    phi_hist, (x, y), t = M2.F.run_main_simulation(cfg, store_history=True, verbose=False)
    return cfg, phi_hist, x, y, t


def test_laplacian_on_known_function(M2):
    """T2f:155: cos(pi x/Lx) cos(pi y/Ly) is an eigenfunction; interior rtol 1e-3."""
    N = 128
    x = np.linspace(0, 1, N + 1)
    X, Y = np.meshgrid(x, x, indexing="ij")
    v = np.cos(np.pi * X) * np.cos(np.pi * Y)
    L = M2.F.laplacian_matrix_neumann(N, N, 1 / N, 1 / N)
    num = M2.F.apply_laplacian(L, v, N, N)
    np.testing.assert_allclose(num[1:-1, 1:-1], (-2 * np.pi ** 2 * v)[1:-1, 1:-1], rtol=1e-3, atol=1e-8)


def test_initial_phi_and_solve_w(M2):
    """T2f:175 (weighted zero mean <= 5e-14, bounds) and T2f:193 (solve_w closed form rtol 1e-15)."""
    F = M2.F
    phi0 = F.init_phi_random(128, 128, DELTA_SEP, amp=1, enforce_zero_mean=True)
    w = np.outer(F.trapz_weights(129), F.trapz_weights(129))
    assert abs(np.sum(w * phi0) / np.sum(w)) <= 5e-14 and np.all(np.abs(phi0) <= 1 - DELTA_SEP)
    w_old, u = np.ones((129, 129)) * 0.1, np.ones((129, 129)) * 0.5
    g = 10.0 / 0.1
    np.testing.assert_allclose(F.solve_w(w_old, 0.1, 10.0, u, u), ((g - 0.5) * w_old + 0.5 * (u + u)) / (g + 0.5), rtol=1e-15)


def test_mass_conservation_and_energy_decrease(M2, default_run):
    """T2f:213-249 (max |M(t) - M(0)| <= 1e-11 over 100 steps) and T2f:252-279 (free energy
    non-increasing, diffs <= 1e-9)."""
    cfg, phi_hist, x, y, t = default_run
    F = M2.F
    assert phi_hist.shape == (101, 129, 129)
    w = np.outer(F.trapz_weights(129), F.trapz_weights(129))
    mass = np.array([np.sum(w * p) for p in phi_hist])
    assert np.abs(mass - mass[0]).max() <= 1e-11
    E = F.free_energy_history(phi_hist, cfg.kappa, cfg.c1, cfg.c2, 1 / 128, 1 / 128)       # one device reduction
    assert np.all(np.diff(E) <= 1e-9), np.diff(E).max()
    assert E[-1] < E[0]


def test_symmetry_preservation(M2):
    """T2f:282-299: a left-right symmetric start stays symmetric (atol 1e-8)."""
    cfg = M2.K.ForwardSolverConfig()
    x = np.linspace(0, 1, 129)
    phi0 = np.tile((0.5 * np.cos(2 * np.pi * x))[:, None], (1, 129))
    ph, _, _ = M2.F.run_main_simulation(cfg, store_history=True, verbose=False, initial_phi=phi0)
    np.testing.assert_allclose(ph[-1], np.fliplr(ph[-1]), atol=1e-8)


def test_time_integrator_convergence_order(M2):
    """T2f:304-356: error vs dt/8 reference over T = 5 * 0.005, slope in (1, 2.2)."""
    base = 0.005
    K = M2.K
    run = lambda dt: M2.F.run_main_simulation(K.ForwardSolverConfig(dt_initial=dt, T=5 * base), store_history=True, verbose=False)[0][-1]
    ref = run(base / 8)
    dts = np.array([base, base / 2, base / 4])
    errs = [np.linalg.norm(run(d) - ref) for d in dts]
    slope = np.polyfit(np.log(dts), np.log(np.array(errs) + 1e-30), 1)[0]
    assert 1 < slope < 2.2, (slope, errs)


def test_unconditional_stability_large_dt(M2):
    """T2f:358-369: dt = 1, T = 3 stays finite."""
    ph, _, _ = M2.F.run_main_simulation(M2.K.ForwardSolverConfig(dt_initial=1.0, T=3.0), store_history=True, verbose=False)
    assert np.all(np.isfinite(ph[-1]))


def test_linear_growth_rate(M2):
    """T2f:371-401: growth of a small cosine mode over T = 1e-5 vs the dispersion relation, rtol 1e-2."""
    cfg = M2.K.ForwardSolverConfig()
    x = np.linspace(0, 1, 129)
    X, Y = np.meshgrid(x, x, indexing="ij")
    kx, ky = 4 * np.pi, 2 * np.pi
    k2 = kx ** 2 + ky ** 2
    theo = (k2 * (2 * cfg.c2 - 2 * cfg.c1 - cfg.kappa * k2)) / (1 + cfg.tau * k2)
    phi0 = 1e-4 * np.cos(kx * X) * np.cos(ky * Y)
    T = 1e-5
    ph, _, _ = M2.F.run_main_simulation(M2.K.ForwardSolverConfig(dt_initial=cfg.dt_initial / 10, T=T), store_history=True,
                                        verbose=False, initial_phi=phi0)
    num = np.log(np.linalg.norm(ph[-1]) / np.linalg.norm(phi0)) / T
    np.testing.assert_allclose(num, theo, rtol=1e-2)


def test_newton_quadratic_convergence(M2):
    """T2f:404-491: residual tail monotone, final < 1e-6, log-log slope in (1.5, 2.5)."""
    F = M2.F
    cfg = M2.K.ForwardSolverConfig()
    N = 128
    L = F.laplacian_matrix_neumann(N, N, 1 / N, 1 / N)
    phi_old = F.init_phi_random(N, N, DELTA_SEP, seed=99)
    w_old = np.zeros_like(phi_old)
    mu_old = F.initialize_mu(phi_old, w_old, cfg.c1, cfg.c2, cfg.kappa, L, N, N, DELTA_SEP)
    _, _, res = F.newton_raphson(phi_old, mu_old, w_old, w_old + 0.01, cfg.dt_initial, cfg.tau, cfg.c1, cfg.c2, cfg.kappa,
                                 DELTA_SEP, L, N, N, 1 / N, 1 / N, return_residual_history=True)
    res = np.asarray(res)
    assert res.size >= 2
    tail = res[-4:] if res.size >= 4 else res
    assert np.all(tail[1:] <= tail[:-1] + 1e-12) and tail[-1] < 1e-6
    if tail.size >= 3:
        slope = np.polyfit(np.log(tail[:-1] + 1e-300), np.log(tail[1:] + 1e-300), 1)[0]
        assert 1.5 < slope < 2.5, (slope, tail)


def test_cost_terms_isolated(M2):
    """T2c:208-300: J1..J4 one at a time against analytic constants (rtol 1e-10); T2c:170
    gradient r + b3 u; T2c:189 prox with kappa_s = 0 = clipped gradient step."""
    Nx, Ny, M = 12, 11, 5
    x, y, t = np.linspace(0, 1, Nx + 1), np.linspace(0, 1, Ny + 1), np.linspace(0, 1, M + 1)
    shp = (M + 1, Nx + 1, Ny + 1)
    K, C = M2.K, M2.C
    q = contextlib.redirect_stdout(io.StringIO())
    z, zt = np.zeros(shp), np.zeros(shp[1:])
    with q:
        J1 = C.calculate_cost(np.full(shp, 0.3), z, np.full(shp, 0.1), np.full(shp[1:], 0.3), x, y, t, K.OptimizationConfig(b1=2.0, b2=0, b3=0, kappa_sparsity=0))
        J2 = C.calculate_cost(np.full(shp, 0.3), z, np.full(shp, 0.3), np.full(shp[1:], -0.2), x, y, t, K.OptimizationConfig(b1=0, b2=3.0, b3=0, kappa_sparsity=0))
        J3 = C.calculate_cost(z, np.full(shp, -0.5), z, zt, x, y, t, K.OptimizationConfig(b1=0, b2=0, b3=4.0, kappa_sparsity=0))
        J4 = C.calculate_cost(z, np.full(shp, -0.5), z, zt, x, y, t, K.OptimizationConfig(b1=0, b2=0, b3=0, kappa_sparsity=0.7))
    np.testing.assert_allclose([J1, J2, J3, J4], [0.5 * 2.0 * 0.04, 0.5 * 3.0 * 0.25, 0.5 * 4.0 * 0.25, 0.7 * 0.5], rtol=1e-10)
    rng = np.random.default_rng(0)
    r, u = rng.standard_normal(shp), rng.uniform(-1, 1, shp)
    opt = K.OptimizationConfig(b3=0.3, kappa_sparsity=0.0)
    g = C.calculate_gradient(r, u, opt)
    np.testing.assert_allclose(g, r + 0.3 * u, rtol=0, atol=0)
    np.testing.assert_allclose(C.proximal_step(u, g, 0.2, opt), np.clip(u - 0.2 * g, -1, 1), rtol=1e-15, atol=1e-16)


def test_proximal_identities(M2):
    """T2p:134 one ISTA step = soft threshold; T2p:168 with box = soft-then-clip; T2p:229 fixed point."""
    K, C = M2.K, M2.C
    rng = np.random.default_rng(1)
    shp = (4, 12, 12)
    u, g = rng.uniform(-2, 2, shp), rng.standard_normal(shp)
    opt = K.OptimizationConfig(kappa_sparsity=0.3, u_min=-1.0, u_max=1.0)
    v = u - 0.5 * g
    soft = np.sign(v) * np.maximum(np.abs(v) - 0.15, 0)
    np.testing.assert_allclose(C.proximal_step(u, g, 0.5, opt), np.clip(soft, -1, 1), rtol=1e-15, atol=1e-16)
    wide = K.OptimizationConfig(kappa_sparsity=0.3, u_min=-1e6, u_max=1e6)
    np.testing.assert_allclose(C.proximal_step(u, g, 0.5, wide), soft, rtol=1e-15, atol=1e-16)
    # fixed point: u* = 0 where |g| <= kappa
    g_small = 0.2 * np.sign(g)
    assert not C.proximal_step(np.zeros(shp), g_small, 0.5, opt).any()


def test_backward_correct_vs_swapped_ordering(M2, V):
    """T2b:299-414: on a real forward history (32^2) the per-step identity A(phi_n) p_n =
    B(phi_{n+1}) p_{n+1} + src holds (rel < 5e-7) and is violated by > 100x when the roles of phi_n and
    phi_{n+1} are swapped."""
    from oracle import vch2d_oracle as O2           # checker only
    cfg = M2.K.ForwardSolverConfig(Nx=32, Ny=32, T=0.1, dt_initial=1e-2)
    ph, (x, y), t = M2.F.run_main_simulation(cfg, store_history=True, verbose=False)
    b1, b2 = 1.3, 0.7
    p, q, r = M2.B.run_backward(ph, x, y, t, cfg, b1, b2, None, None)
    P = O2.Params2D(Nx=32, Ny=32)
    hx = hy = 1 / 32
    good, bad = [], []
    for n in range(len(t) - 1):
        dt = t[n + 1] - t[n]
        src = 0.5 * dt * b1 * (ph[n] + ph[n + 1])
        rhs = O2.adjoint_B_apply(ph[n + 1], p[n + 1], dt, P, hx, hy) + src
        good.append(np.linalg.norm(O2.adjoint_A_apply(ph[n], p[n], dt, P, hx, hy) - rhs) / np.linalg.norm(rhs))
        rhs_s = O2.adjoint_B_apply(ph[n], p[n + 1], dt, P, hx, hy) + src
        bad.append(np.linalg.norm(O2.adjoint_A_apply(ph[n + 1], p[n], dt, P, hx, hy) - rhs_s) / np.linalg.norm(rhs_s))
    assert max(good) < 5e-7 and np.median(bad) / max(max(good), 1e-16) > 100


def test_1d_reference_invariants(V):
    """T1f:185-223 mass <= 1e-12; T1f:253-296 CN order slope in (1.2, 2.2) at N = 512; T1f:342-395
    Newton < 10 iterations, final < 1e-6; T1f:300-319 symmetry via initial_phi."""
    F = V.module("Vch_control_1D.Forward_solver")
    K = V.module("Vch_control_1D.config")
    cfg = K.ForwardSolverConfig()
    ph, x, t = F.run_main_simulation(cfg, store_history=True, verbose=False)
    mass = ph @ (F.trapz_weights(129) / 128)
    assert np.abs(mass - mass[0]).max() <= 1e-12
    E = F.free_energy_history(ph, cfg.kappa, cfg.c1, cfg.c2, 1 / 128)
    assert np.all(np.diff(E) <= 1e-9)
    base = 0.005
    run = lambda dt: F.run_main_simulation(K.ForwardSolverConfig(N=512, dt_initial=dt), store_history=True, verbose=False)[0][-1]   # default T, as T1f:267
    ref = run(base / 8)
    dts = np.array([base, base / 2, base / 4])
    slope = np.polyfit(np.log(dts), np.log([np.linalg.norm(run(d) - ref) + 1e-30 for d in dts]), 1)[0]
    assert 1.2 < slope < 2.2, slope
    L = F.laplacian_matrix_neumann(128, 1 / 128)
    phi_old = F.init_phi_random(128, 1e-2, seed=99)
    w = np.zeros(129)
    mu = F.initialize_mu(phi_old, w, cfg.c1, cfg.c2, L, cfg.kappa)
    _, _, res = F.newton_raphson(phi_old, mu, w, w + 0.01, cfg.dt_initial, cfg.tau, cfg.c1, cfg.c2, 1 / 128, 1e-2, L, cfg.kappa,
                                 return_residual_history=True)
    assert len(res) < 10 and res[-1] < 1e-6
    xs = np.linspace(0, 1, 129)
    ps, _, _ = F.run_main_simulation(cfg, store_history=True, verbose=False, initial_phi=0.5 * np.cos(2 * np.pi * xs))
    np.testing.assert_allclose(ps[-1], ps[-1][::-1], atol=1e-8)
