"""GPU parity tests of the 2D HIP engine, through the C ABI (ctypes), against
 (a) the golden vectors produced by the reference itself (tests/golden/*.npz) and
 (b) the CPU oracle (oracle/vch2d_oracle.py) on seeded inputs at sizes it finishes in seconds.

Tolerances (float64; the reference solves its linear systems with SuperLU, the engine with a
DCT-preconditioned conjugate-gradient iteration converged to round-off, so results agree to solver
round-off, not bit for bit):
  OPS   1e-12  pure stencil / element-wise arithmetic (different summation order only)
  SOLVE 1e-9   anything that passed through a linear solve or a short time march
  MARCH 1e-8   multi-step marches / PGD iterates (round-off amplified by the CH instability)
"""
import numpy as np
import pytest

from conftest import golden, relerr

pytestmark = pytest.mark.gpu

OPS, SOLVE, MARCH = 1e-12, 1e-9, 1e-8


@pytest.fixture(scope="module")
def V():
    import vch_amd
    vch_amd.build()
    return vch_amd


@pytest.fixture(scope="module")
def O2():
    from oracle import vch2d_oracle
    return vch2d_oracle


def _eng(V, g=None, B=1, max_steps=16, **kw):
    if g is not None:
        kw = dict(Nx=int(g["Nx"]), Ny=int(g["Ny"]), Lx=float(g["Lx"]), Ly=float(g["Ly"]), **kw)
        for k in ("tau", "gamma", "c1", "c2", "kappa"):
            if k in g.files:
                kw[k] = float(g[k])
    return V.Engine2D(batch=B, max_steps=max_steps, **kw)


# ---------------------------------------------------------------------------------------
# kernel level
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(16, 16, 1.0, 1.0), (12, 9, 1.0, 0.7), (70, 45, 1.3, 0.9), (129, 64, 1.0, 1.0),
                                   (128, 128, 1.0, 1.0), (64, 32, 1.0, 0.5), (32, 256, 2.0, 1.0), (1024, 1024, 1.0, 1.0),
                                   (512, 512, 1.0, 1.0), (512, 32, 1.0, 0.3), (64, 512, 1.0, 2.0),
                                   (2048, 16, 1.0, 1.0), (16, 2048, 1.0, 1.0)])
def test_spectral_solve_matches_operator(V, O2, shape):
    """Both DCT back ends (MFMA f64 GEMM for general grids, in-LDS FFT of every image size for
    power-of-two grids): (c0 + M(c1 + c2 M)) z == v for z = spectral_solve(v)."""
    Nx, Ny, Lx, Ly = shape
    e = V.Engine2D(Nx=Nx, Ny=Ny, Lx=Lx, Ly=Ly, batch=2)
    pow2 = lambda n: n >= 16 and n & (n - 1) == 0
    assert e.uses_fft == (pow2(Nx) and pow2(Ny))
    rng = np.random.default_rng(3)
    v = rng.standard_normal((2, Nx + 1, Ny + 1))
    c0, c1, c2 = 7.0, 0.9, 3e-3
    z = e.spectral_solve(c0, c1, c2, v)
    hx, hy = Lx / Nx, Ly / Ny
    mmax = 4.0 / hx ** 2 + 4.0 / hy ** 2
    cond = (c0 + mmax * (c1 + c2 * mmax)) / c0            # the forward check amplifies round-off by cond
    tol = max(1e-10, 200 * np.finfo(float).eps * cond)
    for b in range(2):
        M = lambda a: -O2.lap(a, hx, hy)
        chk = c0 * z[b] + M(c1 * z[b] + c2 * M(z[b]))
        assert relerr(chk, v[b]) < tol, (shape, b, relerr(chk, v[b]), tol)
    # and against the dense eigen-decomposition route on a smooth field (well-conditioned check)
    xs, ys = np.linspace(0, Lx, Nx + 1), np.linspace(0, Ly, Ny + 1)
    if Nx == Ny and hx == hy:
        k, l = 3, 2
        f = np.cos(k * np.pi * xs / Lx)[:, None] * np.cos(l * np.pi * ys / Ly)[None, :]      # eigenfunction of M
        m = (2 - 2 * np.cos(k * np.pi / Nx)) / hx ** 2 + (2 - 2 * np.cos(l * np.pi / Ny)) / hy ** 2
        zf = e.spectral_solve(c0, c1, c2, np.stack([f, 2 * f]))
        assert relerr(zf[0], f / (c0 + m * (c1 + c2 * m))) < 1e-11


@pytest.mark.parametrize("tag", ["16", "12x9"])
def test_operators_vs_golden(V, tag):
    g = golden(f"g2d_ops_{tag}.npz")
    e = _eng(V, g)
    dt = float(g["dt"])
    assert relerr(e.apply_laplacian(g["v"]), g["Lv"]) < OPS
    assert relerr(e.apply_laplacian(e.apply_laplacian(g["v"])), g["LLv"]) < OPS
    assert relerr(e.initialize_mu(g["phi_old"], g["w_new"]), g["mu0"]) < OPS
    assert relerr(e.solve_w(g["w_old"], dt, g["u_n"], g["u_np1"]), g["w_filt"]) < 1e-15
    Rp, Rm, nrm = e.residuals(g["phi_new"], g["phi_old"], g["mu_new"], g["mu_old"], g["w_new"], g["w_old"], dt)
    assert relerr(Rp, g["Rphi"]) < OPS and relerr(Rm, g["Rmu"]) < OPS
    assert abs(nrm / np.sqrt(np.sum(g["Rphi"] ** 2) + np.sum(g["Rmu"] ** 2)) - 1) < 1e-13
    n = g["v"].size
    d = g["dvec"]
    shp = g["v"].shape
    top, bot = e.jacobian_apply(g["phi_new"], dt, d[:n].reshape(shp), d[n:].reshape(shp))
    assert relerr(np.concatenate([top.ravel(), bot.ravel()]), g["Jd"]) < OPS
    assert relerr(e.adjoint_apply("A", g["phi_new"], dt, g["v"]), g["Av"]) < 1e-11
    assert relerr(e.adjoint_apply("B", g["phi_new"], dt, g["v"]), g["Bv"]) < 1e-11
    with pytest.raises(ValueError):
        e.apply_laplacian(np.zeros((3, 3)))


@pytest.mark.parametrize("tag", ["16", "12x9"])
def test_linear_solves_vs_golden(V, tag):
    """The spsolve seams: Newton system (F2:370), adjoint step (B2:229), terminal (B2:185)."""
    g = golden(f"g2d_ops_{tag}.npz")
    e = _eng(V, g)
    dt = float(g["dt"])
    n = g["v"].size
    shp = g["v"].shape
    rhs = g["rhs"]
    dphi, dmu, st = e.jacobian_solve(g["phi_new"], dt, rhs[:n].reshape(shp), rhs[n:].reshape(shp))
    sol = g["Jsol"]
    assert relerr(dphi, sol[:n].reshape(shp)) < SOLVE, st
    assert relerr(dmu, sol[n:].reshape(shp)) < SOLVE, st
    assert st["max_lin_relres"] < 1e-11, st
    p, st = e.adjoint_solve(g["phi_new"], dt, g["v"])
    assert relerr(p, g["Asol"]) < SOLVE, st
    p, st = e.adjoint_solve(None, 0.0, g["v"])
    assert relerr(p, g["ATsol"]) < SOLVE, st


def test_batch_independence(V, O2):
    """Trajectories of one batch do not interact: a 3-batch equals three 1-batches."""
    rng = np.random.default_rng(11)
    Nx = Ny = 40
    phi = rng.uniform(-0.8, 0.8, (3, Nx + 1, Ny + 1))
    x = rng.standard_normal((3, Nx + 1, Ny + 1))
    e3 = V.Engine2D(Nx=Nx, Ny=Ny, batch=3)
    e1 = V.Engine2D(Nx=Nx, Ny=Ny, batch=1)
    y3 = e3.schur_apply(phi, 1e-3, x)
    P = O2.Params2D(Nx=Nx, Ny=Ny)
    for b in range(3):
        assert np.array_equal(y3[b], e1.schur_apply(phi[b], 1e-3, x[b]))
        assert relerr(y3[b], O2.schur_apply(phi[b], x[b], 1e-3, P, 1 / Nx, 1 / Ny)) < 1e-11


# ---------------------------------------------------------------------------------------
# Newton step
# ---------------------------------------------------------------------------------------
def test_newton_vs_golden(V):
    g = golden("g2d_newton_32.npz")
    e = V.Engine2D(Nx=32, Ny=32)
    for tag, dt in (("dt1e-2", 1e-2), ("dt1e-3", 1e-3)):
        pn, mn, hist, st = e.newton_raphson(g["phi0"], g["mu_init"], g["w0"], g["w1"], dt)
        ref = g[f"hist_{tag}"]
        assert len(hist) == len(ref), (hist, ref, st)
        assert np.allclose(hist[:-1], ref[:-1], rtol=1e-6), (hist, ref)
        assert hist[-1] < 1e-6
        assert relerr(pn, g[f"phi_new_{tag}"]) < SOLVE and relerr(mn, g[f"mu_new_{tag}"]) < SOLVE, st
    # near-singular start (a third of the nodes at +-0.99): step ceiling + Armijo path
    pn, mn, hist, st = e.newton_raphson(g["phi_stress"], g["mu_stress"], g["w0"], g["w0"], 1e-3)
    ref = g["hist_stress"]
    assert len(hist) == len(ref), (hist, ref, st)
    assert np.allclose(hist[:-1], ref[:-1], rtol=1e-5), (hist, ref)
    assert relerr(pn, g["phi_new_stress"]) < SOLVE, st


def test_newton_batch_divergent_control_flow(V):
    """Two trajectories with different Newton iteration counts in one batch."""
    g = golden("g2d_newton_32.npz")
    e = V.Engine2D(Nx=32, Ny=32, batch=2)
    z = g["w0"]
    po = np.stack([g["phi0"], g["phi_stress"]])
    mo = np.stack([g["mu_init"], g["mu_stress"]])
    w0 = np.stack([z, z])
    w1 = np.stack([g["w1"], z])
    pn, mn, hists, st = e.newton_raphson(po, mo, w0, w1, 1e-3)
    assert len(hists[0]) == len(g["hist_dt1e-3"]) and len(hists[1]) == len(g["hist_stress"]), hists
    assert relerr(pn[0], g["phi_new_dt1e-3"]) < SOLVE and relerr(pn[1], g["phi_new_stress"]) < SOLVE


def test_linear_solve_degenerate_right_hand_sides(V, O2):
    """Corner cases of the single-reduction CG in one batch: a zero right-hand side (x = 0, no iteration),
    a constant field (constant Jacobian diagonal: P differs from A by a multiple of M only, a handful of
    iterations, the predicted <z,z> collapsing to round-off) and a generic field, each against the
    oracle's direct solve."""
    N, dt = 16, 1e-2
    rng = np.random.default_rng(11)
    P = O2.Params2D(Nx=N, Ny=N)
    h = 1.0 / N
    phi = np.stack([rng.uniform(-0.9, 0.9, (N + 1, N + 1)), np.full((N + 1, N + 1), 0.3), rng.uniform(-0.5, 0.5, (N + 1, N + 1))])
    a = rng.standard_normal((3, N + 1, N + 1))
    b = rng.standard_normal((3, N + 1, N + 1))
    a[0] = 0.0
    b[0] = 0.0
    e = V.Engine2D(Nx=N, Ny=N, batch=3)
    dphi, dmu, st = e.jacobian_solve(phi, dt, a, b)
    assert not dphi[0].any() and not dmu[0].any()
    L = O2.lap_matrix(N, N, h, h)
    for k in (1, 2):
        J = O2.jac_matrix(phi[k], dt, P, L)
        from scipy.sparse.linalg import spsolve
        ref = spsolve(J.tocsc(), np.concatenate([a[k].ravel(), b[k].ravel()]))
        n = (N + 1) ** 2
        assert relerr(dphi[k].ravel(), ref[:n]) < SOLVE and relerr(dmu[k].ravel(), ref[n:]) < SOLVE, (k, st)
    assert st["max_lin_relres"] <= 1e-14


def test_half_size_dct_variant(V, O2, monkeypatch):
    """The opt-in half-size DCT-I (VCH_DCT_HALF=1: real-even algorithm on an FFT of length N, 512-interval axes
    only): the spectral solve inverts the operator, and a Newton step agrees with the long transform."""
    monkeypatch.setenv("VCH_DCT_HALF", "1")
    rng = np.random.default_rng(9)
    for Nx, Ny in ((512, 512), (512, 16), (32, 512)):
        e = V.Engine2D(Nx=Nx, Ny=Ny, batch=2)
        v = rng.standard_normal((2, Nx + 1, Ny + 1))
        z = e.spectral_solve(7.0, 0.9, 3e-3, v)
        hx, hy = 1.0 / Nx, 1.0 / Ny
        Mz = np.stack([-O2.lap(a, hx, hy) for a in z])
        back = 7.0 * z + 0.9 * Mz + 3e-3 * np.stack([-O2.lap(a, hx, hy) for a in Mz])
        assert relerr(back, v) < 1e-6       # the check multiplies by the operator (cond ~ 1e9): round-off amplified
        e.close()
    N = 512
    phi = O2.init_phi_random(N, N, 1e-2, amp=0.1, seed=42)
    w = np.zeros_like(phi)
    eh = V.Engine2D(Nx=N, Ny=N, max_steps=2)
    ph, mh, hh, _ = eh.newton_raphson(phi, eh.initialize_mu(phi, w), w, w, 1e-3)
    eh.close()
    monkeypatch.delenv("VCH_DCT_HALF")
    ef = V.Engine2D(Nx=N, Ny=N, max_steps=2)
    pf, mf, hf, _ = ef.newton_raphson(phi, ef.initialize_mu(phi, w), w, w, 1e-3)
    assert len(hh) == len(hf) and np.allclose(hh[:-1], hf[:-1], rtol=1e-6) and relerr(ph, pf) < 1e-11


# ---------------------------------------------------------------------------------------
# forward march, adjoint sweep, cost, prox
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["16", "16_ragged", "32", "14x11", "64_fine"])
def test_forward_backward_cost_vs_golden(V, O2, tag):
    g = golden(f"g2d_forward_{tag}.npz")
    t = g["t_hist"]
    M = len(t) - 1
    e = _eng(V, g, max_steps=M)
    tg, dts = V.time_grid(float(g["T"]), float(g["dt"]))
    assert np.array_equal(tg, t)
    phi0 = g["phi_nat"][0]
    ph, st = e.forward(phi0, dts)
    assert relerr(ph, g["phi_nat"]) < SOLVE, st
    ph_u, st = e.forward(phi0, dts, u=g["u"])
    assert relerr(ph_u, g["phi_u"]) < SOLVE, st
    if "phi_ushort" in g.files:         # control shorter than the march: zeros beyond (F2:545-548)
        ph_s, _ = e.forward(phi0, dts, u=g["u"][:4])
        assert relerr(ph_s, g["phi_ushort"]) < SOLVE
    opt = O2.OptParams()
    for ct, cq in ((1, 1), (2, 2)):
        if f"r_{ct}{cq}" not in g.files:
            continue
        phi_T, phi_Q = O2.build_targets(g["x"], g["y"], t, phi0, float(g["Lx"]), float(g["Ly"]), float(g["T"]), ct, cq)
        p, q, r, st = e.backward(g["phi_u"], t, opt.b1, opt.b2, phi_Q, phi_T)
        assert relerr(r, g[f"r_{ct}{cq}"]) < SOLVE, st
        if f"p_{ct}{cq}" in g.files:
            assert relerr(p, g[f"p_{ct}{cq}"]) < SOLVE, st
        if f"q_{ct}{cq}" in g.files:
            assert relerr(q, g[f"q_{ct}{cq}"]) < SOLVE, st
        if f"J_{ct}{cq}" in g.files:
            J = e.cost(g["phi_u"], g["u"], phi_Q, phi_T, t, opt, g["x"], g["y"])
            assert abs(J[4] / float(g[f"J_{ct}{cq}"]) - 1) < 1e-12
            if (ct, cq) == (1, 1) and "Jparts" in g.files:
                assert np.allclose(J[:4], g["Jparts"], rtol=1e-12)
    if "r_none" in g.files:             # targets None -> zeros (B2:167-168)
        _, _, r0, _ = e.backward(g["phi_u"], t, 1.3, 0.7, None, None, want=("r",))
        assert relerr(r0, g["r_none"]) < SOLVE
    if "grad" in g.files:
        for a in (0.5, 50.0):
            if f"prox_a{a}" in g.files:
                un = e.grad_prox(g["u"], g["r_11"], a, opt)
                assert relerr(un, g[f"prox_a{a}"]) < 1e-14
    # resident-state path: backward/cost straight after a forward, nothing re-uploaded
    ph_u, _ = e.forward(phi0, dts, u=g["u"], store=False)
    phi_T, phi_Q = O2.build_targets(g["x"], g["y"], t, phi0, float(g["Lx"]), float(g["Ly"]), float(g["T"]), 1, 1)
    _, _, r, _ = e.backward(None, t, opt.b1, opt.b2, phi_Q, phi_T, want=("r",))
    assert relerr(r, g["r_11"]) < MARCH


def test_forward_stress_vs_golden(V):
    """amp=1.0 initial data: ~1/3 of the nodes clipped at +-0.99 (config 5 regime)."""
    g = golden("g2d_forward_32_stress.npz")
    e = V.Engine2D(Nx=32, Ny=32, max_steps=8)
    _, dts = V.time_grid(float(g["T"]), float(g["dt"]))
    ph, st = e.forward(g["phi_nat"][0], dts)
    assert relerr(ph, g["phi_nat"]) < MARCH, st


def test_forward_invariants(V, O2):
    """The reference's own forward tests: mass drift <= 1e-11 (T2f:213-249), left-right
    symmetry atol 1e-8 (T2f:282-299), finite at dt = 1 (T2f:358)."""
    N = 48
    e = V.Engine2D(Nx=N, Ny=N, max_steps=20)
    phi0 = O2.init_phi_random(N, N, 1e-2, amp=0.1, seed=7)
    _, dts = V.time_grid(0.2, 1e-2)
    ph, st = e.forward(phi0, dts)
    w = np.outer(O2.trapz_weights(N + 1), O2.trapz_weights(N + 1)) / N ** 2
    mass = np.sum(w * ph, axis=(1, 2))
    assert np.max(np.abs(mass - mass[0])) <= 1e-11
    assert np.max(np.abs(ph)) <= 0.99 + 1e-15
    xs = np.linspace(0, 1, N + 1)
    sym0 = 0.3 * np.cos(2 * np.pi * xs)[:, None] * np.cos(np.pi * xs)[None, :]
    ps, _ = e.forward(sym0, dts[:10])
    assert np.max(np.abs(ps[-1] - ps[-1][::-1, :])) < 1e-8
    p1, _ = e.forward(phi0, np.array([1.0]))
    assert np.all(np.isfinite(p1))


def test_adjoint_identities(V, O2):
    """The reference's backward tests on its own 7x6 synthetic history (T2b:176-297):
    terminal conditions, per-step A p_n = B p_{n+1} + src, q = -L p, r filter."""
    Nx, Ny, M = 6, 5, 5
    x, y, t = np.linspace(0, 1, Nx + 1), np.linspace(0, 1, Ny + 1), np.linspace(0, 0.2, M + 1)
    X, Y = np.meshgrid(x, y, indexing="ij")
    phi = np.array([0.2 * np.sin(np.pi * X) * np.sin(np.pi * Y) * (1 + 0.2 * np.cos(2 * np.pi * tt / 0.2)) for tt in t])
    e = V.Engine2D(Nx=Nx, Ny=Ny, max_steps=M)
    P = O2.Params2D(Nx=Nx, Ny=Ny)
    b1, b2 = 1.3, 0.7
    p, q, r, _ = e.backward(phi, t, b1, b2, None, None)
    hx, hy = x[1] - x[0], y[1] - y[0]
    assert relerr(p[-1] - P.tau * O2.lap(p[-1], hx, hy), b2 * phi[-1]) < 1e-10
    assert not r[-1].any()
    for n in range(M):
        dt = t[n + 1] - t[n]
        lhs = O2.adjoint_A_apply(phi[n], p[n], dt, P, hx, hy)
        rhs = O2.adjoint_B_apply(phi[n + 1], p[n + 1], dt, P, hx, hy) + 0.5 * dt * b1 * (phi[n] + phi[n + 1])
        assert relerr(lhs, rhs) < 1e-9
        assert np.max(np.abs(q[n] + O2.lap(p[n], hx, hy))) < 1e-9
        den = P.gamma + 0.5 * dt
        assert np.max(np.abs(r[n] - ((P.gamma - 0.5 * dt) / den * r[n + 1] + 0.5 * dt / den * (q[n] + q[n + 1])))) < 1e-9


@pytest.mark.parametrize("tag", ["16", "16_bt"])
def test_pgd_vs_golden(V, O2, tag):
    """Device-resident PGD loop vs 3-4 iterations of the reference (incl. forced backtracking)."""
    g = golden(f"g2d_pgd_{tag}.npz")
    N, M = int(g["N"]), len(g["t_hist"]) - 1
    e = V.Engine2D(Nx=N, Ny=N, max_steps=M)
    opt = V.make_opt(alpha_max=float(g["alpha_max"]), b3=float(g["b3"]))
    phi0 = O2.init_phi_random(N, N, 1e-2, amp=0.1, seed=42)
    J0 = e.pgd_init(phi0, g["phi_T"], g["t_hist"], opt, ramp=True, T=float(g["T"]))
    assert abs(J0[0, 4] / g["costs"][0] - 1) < 1e-10
    res = e.pgd_iterate(int(g["n_iter"]))
    assert res["iters"] == int(g["n_iter"])
    assert np.allclose(res["cost"][0], g["costs"][1:], rtol=1e-8), (res["cost"], g["costs"])
    assert np.allclose(res["alpha"][0], g["alphas"], rtol=1e-13), (res["alpha"], g["alphas"])
    assert list(res["attempts"][0]) == list(g["attempts"])
    assert np.allclose(res["change"][0], g["changes"], rtol=1e-6)
    assert relerr(e.pgd_get("u"), g["u_final"]) < MARCH
    assert relerr(e.pgd_get("phi"), g["phi_final"]) < MARCH
    assert relerr(e.pgd_get("phi_Q"), g["phi_Q"]) < 1e-15


def test_pgd_batch_matches_single(V, O2):
    """A batch of 3 seeds gives the same iterates as three single-trajectory runs."""
    N, T, dt = 24, 0.05, 1e-2
    t, dts = V.time_grid(T, dt)
    M = len(dts)
    xs = np.linspace(0, 1, N + 1)
    phi_T = 0.7 * np.sin(2 * np.pi * xs)[:, None] * np.cos(np.pi * xs)[None, :]
    phi0 = np.stack([O2.init_phi_random(N, N, 1e-2, amp=0.1, seed=42 + i) for i in range(3)])
    opt = V.make_opt()
    e3 = V.Engine2D(Nx=N, Ny=N, batch=3, max_steps=M)
    e3.pgd_init(phi0, np.stack([phi_T] * 3), t, opt, ramp=True, T=T)
    r3 = e3.pgd_iterate(2)
    u3 = e3.pgd_get("u")
    e1 = V.Engine2D(Nx=N, Ny=N, batch=1, max_steps=M)
    for b in range(3):
        e1.pgd_init(phi0[b], phi_T, t, opt, ramp=True, T=T)
        r1 = e1.pgd_iterate(2)
        # bit for bit: every decision that shapes a trajectory's arithmetic (form and length of its solves, orders of its
        # starting guesses) is taken per trajectory from its own history
        assert np.array_equal(r3["cost"][b], r1["cost"][0])
        assert np.array_equal(u3[b], e1.pgd_get("u"))


def test_pgd_batch_uneven_line_search(V, O2):
    """Trajectories whose line searches end after different numbers of trials (the accepted ones sit
    the remaining trial marches out): every trajectory still reproduces its single-trajectory run."""
    N, T, dt = 16, 0.1, 1e-2
    t, dts = V.time_grid(T, dt)
    M = len(dts)
    xs = np.linspace(0, 1, N + 1)
    base = np.sin(2 * np.pi * xs)[:, None] * np.cos(np.pi * xs)[None, :]
    amps = [0.7, 0.3, 0.02, 0.95]
    phi_T = np.stack([a * base for a in amps])
    phi0 = np.stack([O2.init_phi_random(N, N, 1e-2, amp=0.1, seed=7 + i) for i in range(4)])
    opt = V.make_opt(alpha_max=4.0e4)     # overshooting optimistic step, as in the reference-made g2d_pgd_16_bt
    e4 = V.Engine2D(Nx=N, Ny=N, batch=4, max_steps=M)
    e4.pgd_init(phi0, phi_T, t, opt, ramp=True, T=T)
    r4 = e4.pgd_iterate(4)
    u4, p4 = e4.pgd_get("u"), e4.pgd_get("phi")
    assert len({tuple(a) for a in r4["attempts"]}) > 1, r4["attempts"]      # the searches really differ
    e1 = V.Engine2D(Nx=N, Ny=N, batch=1, max_steps=M)
    for b in range(4):
        e1.pgd_init(phi0[b], phi_T[b], t, opt, ramp=True, T=T)
        r1 = e1.pgd_iterate(4)
        assert np.array_equal(r4["attempts"][b], r1["attempts"][0])
        # ... bit for bit, although its batch mates sit out different trial marches and their solves take other forms and
        # lengths: nothing in a trajectory's arithmetic depends on them (round 2: equal to the solves' tolerance only)
        assert np.array_equal(r4["cost"][b], r1["cost"][0])
        assert np.array_equal(u4[b], e1.pgd_get("u")) and np.array_equal(p4[b], e1.pgd_get("phi"))


# ---------------------------------------------------------------------------------------
# full-size checks
# ---------------------------------------------------------------------------------------
def test_newton_512_vs_reference_history(V, O2):
    """Two full-size (512^2, dt=1e-3) Newton steps against the reference's residual-norm
    histories and sub-sampled fields (the round-off-limited regime: final norms sit at the
    ~6e-7 floor just under the 1e-6 tolerance)."""
    g = golden("g2d_newton_512.npz")
    N = 512
    e = V.Engine2D(Nx=N, Ny=N, max_steps=4)
    phi = O2.init_phi_random(N, N, 1e-2, amp=0.1, seed=42)
    assert relerr(phi[::8, ::8], g["sub"][0]) == 0.0
    w = np.zeros_like(phi)
    mu = e.initialize_mu(phi, w)
    for step in range(2):
        phi, mu, hist, st = e.newton_raphson(phi, mu, w, w, 1e-3)
        ref = g[f"hist{step}"]
        assert len(hist) == len(ref), (hist, ref, st)
        # norms above the evaluation floor (|L mu| eps ~ 6e-7 at 512^2, the level of the last entry) agree to solver
        # round-off relative to the FIRST norm (9e9 -> 2.0 is a reduction by 2e-10: the second entry is itself the
        # round-off residual of the first linear solve, hence rtol 1e-4 and not 1e-9)
        assert np.allclose(hist[:-1], ref[:-1], rtol=1e-4), ("norm history differs beyond the round-off floor", hist, ref)
        assert hist[-1] < 1e-6
        assert relerr(phi[::8, ::8], g["sub"][step + 1]) < SOLVE, st


def test_size_independent_properties_256(V, O2):
    """256^2, batch 2: linearity of the Schur operator, solve(apply(x)) == x, and the
    adjoint solve inverting the adjoint operator (no oracle solve needed at this size)."""
    N = 256
    rng = np.random.default_rng(5)
    e = V.Engine2D(Nx=N, Ny=N, batch=2, max_steps=4)
    phi = rng.uniform(-0.7, 0.7, (2, N + 1, N + 1))
    a, b_ = rng.standard_normal((2, N + 1, N + 1)), rng.standard_normal((2, N + 1, N + 1))
    dt = 2.5e-3
    lin = e.schur_apply(phi, dt, 2.0 * a - 3.0 * b_) - (2.0 * e.schur_apply(phi, dt, a) - 3.0 * e.schur_apply(phi, dt, b_))
    assert np.max(np.abs(lin)) < 1e-12 * np.max(np.abs(e.schur_apply(phi, dt, a)))
    # J [dphi; dmu] = [f; g]  ->  J applied to the solution returns [f; g]
    dphi, dmu, st = e.jacobian_solve(phi, dt, a, b_)
    top, bot = e.jacobian_apply(phi, dt, dphi, dmu)
    assert relerr(top, a) < 1e-9 and relerr(bot, b_) < 1e-9, st
    p, st = e.adjoint_solve(phi, dt, a)
    assert relerr(e.adjoint_apply("A", phi, dt, p), a) < 1e-9, st


def test_solves_on_other_fft_sizes(V, O2):
    """Newton-system and adjoint solves on grids that use the other FFT instantiations
    (128^2 default grid: run-time length; 1024^2: 2048-point image; 64x32: non-square)."""
    rng = np.random.default_rng(9)
    for (Nx, Ny, Ly) in ((128, 128, 1.0), (64, 32, 0.5), (1024, 1024, 1.0)):
        e = V.Engine2D(Nx=Nx, Ny=Ny, Ly=Ly, batch=1, max_steps=2)
        shp = (Nx + 1, Ny + 1)
        phi = rng.uniform(-0.7, 0.7, shp)
        xs, ys = np.linspace(0, 1, Nx + 1), np.linspace(0, 1, Ny + 1)
        a = np.cos(3 * np.pi * xs)[:, None] * np.cos(2 * np.pi * ys)[None, :] + 0.1 * rng.standard_normal(shp)
        b_ = np.cos(np.pi * xs)[:, None] * np.ones(Ny + 1)[None, :] + 0.1 * rng.standard_normal(shp)
        dt = 1e-3
        dphi, dmu, st = e.jacobian_solve(phi, dt, a, b_)
        top, bot = e.jacobian_apply(phi, dt, dphi, dmu)
        tol = 1e-8 if Nx <= 128 else 1e-5       # the forward check amplifies round-off by cond(A) ~ N^4
        assert relerr(top, a) < tol and relerr(bot, b_) < tol, (Nx, Ny, st)
        assert st["max_lin_relres"] < 1e-14
        p, st = e.adjoint_solve(phi, dt, a)
        assert relerr(e.adjoint_apply("A", phi, dt, p), a) < tol, (Nx, Ny, st)
        e.close()


def test_pgd_eight_iterations_vs_oracle(V, O2):
    """8 PGD iterations at 32^2 (alpha growth x1.2 capped at alpha_max, backtracking when it
    overshoots) against the CPU oracle run here on the same inputs; batch of 2 seeds."""
    N, T, dt = 32, 0.05, 1e-2
    P = O2.Params2D(Nx=N, Ny=N, T=T, dt_initial=dt)
    Op = O2.OptParams(alpha_max=200.0)
    t, dts = V.time_grid(T, dt)
    e = V.Engine2D(Nx=N, Ny=N, batch=2, max_steps=len(dts))
    seeds = (42, 43)
    phi0 = np.stack([O2.init_phi_random(N, N, 1e-2, amp=0.1, seed=s) for s in seeds])
    refs = [O2.pgd(P, Op, n_iter=8, seed=s) for s in seeds]
    J0 = e.pgd_init(phi0, np.stack([r.phi_T for r in refs]), t, V.make_opt(Op), ramp=True, T=T)
    out = e.pgd_iterate(8)
    u = e.pgd_get("u")
    for b, r in enumerate(refs):
        assert abs(J0[b, 4] / r.costs[0] - 1) < 1e-10
        assert np.allclose(out["cost"][b], r.costs[1:], rtol=1e-7), (out["cost"][b], r.costs)
        assert np.allclose(out["alpha"][b], r.alphas, rtol=1e-12), (out["alpha"][b], r.alphas)
        assert list(out["attempts"][b]) == list(r.attempts)
        assert relerr(u[b], r.u) < 1e-6
    assert np.all(np.diff(np.concatenate([J0[:, 4:5], out["cost"]], axis=1), axis=1) < 0)      # monotone descent


def test_full_size_pgd_iteration_invariants_512x1000(V, O2):
    """BASELINE config 4 at full size for one trajectory (512^2, 1000 steps of 1e-3): size-independent
    properties of one device-resident PGD iteration -- the cost decreases, the controlled and the
    uncontrolled state conserve mass (phi_t = Lap mu whatever the control) and stay inside the clip
    band, the uncontrolled free energy does not increase, the adjoint vanishes at t = T (B2:187) and
    the accepted control respects the box and the prox fixed-point structure."""
    F2 = V.module("Vch_control_2D.Forward2_solver")
    N, M = 512, 1000
    t, dts = V.time_grid(1.0, 1e-3)
    assert len(dts) == M
    e = V.Engine2D(Nx=N, Ny=N, batch=1, max_steps=M)
    xs = np.linspace(0, 1, N + 1)
    phi_T = 0.7 * np.sin(2 * np.pi * xs)[:, None] * np.cos(np.pi * xs)[None, :]
    phi0 = F2.init_phi_random(N, N, 1e-2, amp=0.1, seed=42)
    opt = V.make_opt()
    J0 = e.pgd_init(phi0, phi_T, t, opt, ramp=True, T=1.0)
    wts = np.outer(F2.trapz_weights(N + 1), F2.trapz_weights(N + 1))
    ph = e.pgd_get("phi")                     # uncontrolled march
    assert ph.shape == (M + 1, N + 1, N + 1) and np.array_equal(ph[0], phi0)
    lv = list(range(0, M + 1, 50))
    mass = np.array([np.sum(wts * ph[k]) for k in lv])
    assert np.abs(mass - mass[0]).max() <= 1e-10 * np.sum(wts)
    assert np.abs(ph).max() <= 0.99 + 1e-15 and np.isfinite(ph).all()
    E = e.free_energy_resident(M + 1)[lv]          # device reduction over the resident history (vch2d_free_energy)
    assert abs(E[3] - O2.free_energy(ph[lv[3]], 1e-4, 0.75, 1.0, 1 / N, 1 / N)) < 1e-12 * abs(E[3])
    assert np.all(np.diff(E) <= 1e-9) and E[-1] < E[0]
    del ph
    res = e.pgd_iterate(1)
    assert res["iters"] == 1 and res["cost"][0, 0] < J0[0, 4]
    assert 0 <= res["attempts"][0, 0] <= 10 and res["alpha"][0, 0] <= 50.0
    r = e.pgd_get("r")
    assert np.isfinite(r).all() and not r[M].any() and np.abs(r[0]).max() > 0
    u = e.pgd_get("u")
    assert u.min() >= -1.0 and u.max() <= 1.0
    # prox structure of u = clip(soft(0 - alpha (r + b3*0), alpha*kappa_s)): zero exactly where |alpha r| <= alpha kappa_s
    a = res["alpha"][0, 0]
    zero = np.abs(a * r) <= a * 1e-4
    assert not u[zero].any() and np.all(u[~zero] != 0)
    del r, u
    ph = e.pgd_get("phi")                     # controlled state of the accepted step
    mass = np.array([np.sum(wts * ph[k]) for k in lv])
    assert np.abs(mass - mass[0]).max() <= 1e-10 * np.sum(wts)
    assert np.abs(ph).max() <= 0.99 + 1e-15



def test_config4_shape_two_contexts_of_four_512(V, O2):
    """The benched shape of BASELINE config 4 on one GPU: 512^2, 8 trajectories (seeds 42..49) as two contexts of 4 running
    concurrently (one host thread each, as bench.py drives them), a short horizon (40 steps of 1e-3) and two PGD
    iterations; every trajectory reproduces its own single-trajectory run BIT FOR BIT -- costs, step lengths, line-search
    attempts, control and state -- so a seed gives the same iterates whatever the batch, context or GPU it lands in."""
    import threading
    F2 = V.module("Vch_control_2D.Forward2_solver")
    N, M = 512, 40
    t, dts = V.time_grid(M * 1e-3, 1e-3)
    xs = np.linspace(0, 1, N + 1)
    phi_T = 0.7 * np.sin(2 * np.pi * xs)[:, None] * np.cos(np.pi * xs)[None, :]
    phi0 = np.stack([F2.init_phi_random(N, N, 1e-2, amp=0.1, seed=42 + i) for i in range(8)])
    opt = V.make_opt()
    engs = [V.Engine2D(Nx=N, Ny=N, batch=4, max_steps=M) for _ in range(2)]
    out = [None, None]

    def work(k):
        e = engs[k]
        J0 = e.pgd_init(phi0[4 * k:4 * k + 4], np.stack([phi_T] * 4), t, opt, ramp=True, T=M * 1e-3)
        r = e.pgd_iterate(2)
        out[k] = (J0, r, e.pgd_get("u"), e.pgd_get("phi"))
    ths = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    for e in engs:
        e.close()
    assert all(o is not None for o in out)
    e1 = V.Engine2D(Nx=N, Ny=N, batch=1, max_steps=M)
    for i in range(8):
        J0, r, u, ph = out[i // 4]
        b = i % 4
        J1 = e1.pgd_init(phi0[i], phi_T, t, opt, ramp=True, T=M * 1e-3)
        r1 = e1.pgd_iterate(2)
        assert np.array_equal(J0[b], J1[0]), i
        assert np.array_equal(r["attempts"][b], r1["attempts"][0]) and np.array_equal(r["alpha"][b], r1["alpha"][0]), i
        assert np.array_equal(r["cost"][b], r1["cost"][0]), (i, r["cost"][b], r1["cost"][0])
        assert np.array_equal(u[b], e1.pgd_get("u")) and np.array_equal(ph[b], e1.pgd_get("phi")), i
        assert np.all(np.diff(np.concatenate([J1[0, 4:5], r1["cost"][0]])) < 0)
    e1.close()


def test_config4_full_size_eight_trajectories_512x1000(V, O2):
    """BASELINE config 4's per-GPU share AT FULL SIZE and in the benched shape: 512^2, 1000 steps of 1e-3, 8 trajectories
    (seeds 42..49) as two concurrent contexts of 4, one device-resident PGD iteration (adjoint sweep, optimistic march, line
    search).  Size-independent properties for every trajectory -- the cost decreases, the controlled state conserves mass
    and stays inside the clip band, the control respects the box -- and two of the eight (one per context) reproduce
    their own single-trajectory run of the same 1000 steps bit for bit: cost, step length, attempts, control, state."""
    import threading
    F2 = V.module("Vch_control_2D.Forward2_solver")
    N, M = 512, 1000
    t, dts = V.time_grid(1.0, 1e-3)
    xs = np.linspace(0, 1, N + 1)
    phi_T = 0.7 * np.sin(2 * np.pi * xs)[:, None] * np.cos(np.pi * xs)[None, :]
    phi0 = np.stack([F2.init_phi_random(N, N, 1e-2, amp=0.1, seed=42 + i) for i in range(8)])
    opt = V.make_opt()
    wts = np.outer(F2.trapz_weights(N + 1), F2.trapz_weights(N + 1))
    lv = list(range(0, M + 1, 100))
    engs = [V.Engine2D(Nx=N, Ny=N, batch=4, max_steps=M) for _ in range(2)]
    out, err = [None, None], []
    probe = {0: 1, 1: 2}                                   # context -> the trajectory of it that is compared with a single run

    def work(k):
        try:
            e = engs[k]
            J0 = e.pgd_init(phi0[4 * k:4 * k + 4], np.stack([phi_T] * 4), t, opt, ramp=True, T=1.0)
            r = e.pgd_iterate(1)
            u, ph = e.pgd_get("u"), e.pgd_get("phi")
            checks = []
            for b in range(4):
                mass = np.array([np.sum(wts * ph[b, j]) for j in lv])
                checks.append((float(np.abs(mass - mass[0]).max()), float(np.abs(ph[b]).max()), float(u[b].min()), float(u[b].max()),
                               bool(np.isfinite(ph[b]).all())))
            b = probe[k]
            out[k] = (J0, r, u[b].copy(), ph[b].copy(), checks)
        except BaseException as exc:          # surfaces in the main thread
            err.append(exc)
    ths = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    for e in engs:
        e.close()
    assert not err, err
    for k in range(2):
        J0, r, _, _, checks = out[k]
        assert np.all(r["cost"][:, 0] < J0[:, 4]) and np.all(r["attempts"][:, 0] <= 10) and np.all(r["alpha"][:, 0] <= 50.0)
        for dm, amax, umin, umax, fin in checks:
            assert dm <= 1e-10 * np.sum(wts) and amax <= 0.99 + 1e-15 and umin >= -1.0 and umax <= 1.0 and fin
    e1 = V.Engine2D(Nx=N, Ny=N, batch=1, max_steps=M)
    for k in range(2):
        J0, r, u, ph, _ = out[k]
        b = probe[k]
        J1 = e1.pgd_init(phi0[4 * k + b], phi_T, t, opt, ramp=True, T=1.0)
        r1 = e1.pgd_iterate(1)
        assert np.array_equal(J0[b], J1[0]) and np.array_equal(r["cost"][b], r1["cost"][0]), (k, r["cost"][b], r1["cost"][0])
        assert np.array_equal(r["attempts"][b], r1["attempts"][0]) and np.array_equal(r["alpha"][b], r1["alpha"][0])
        assert np.array_equal(u, e1.pgd_get("u")) and np.array_equal(ph, e1.pgd_get("phi")), k
    e1.close()


def test_bitwise_reproducibility(V, O2):
    """All reductions run in a fixed order (per-workgroup partials summed by index, no atomics): two runs of the
    same march / PGD iteration give bit-identical histories, controls and costs, also from different contexts."""
    N, T, dt = 64, 0.05, 1e-2
    t, dts = V.time_grid(T, dt)
    xs = np.linspace(0, 1, N + 1)
    phi_T = 0.7 * np.sin(2 * np.pi * xs)[:, None] * np.cos(np.pi * xs)[None, :]
    phi0 = np.stack([O2.init_phi_random(N, N, 1e-2, amp=0.1, seed=42 + i) for i in range(3)])
    outs = []
    for rep in range(2):
        e = V.Engine2D(Nx=N, Ny=N, batch=3, max_steps=len(dts))
        ph, _ = e.forward(phi0, dts)
        J0 = e.pgd_init(phi0, np.stack([phi_T] * 3), t, V.make_opt(), ramp=True, T=T)
        r = e.pgd_iterate(2)
        outs.append((ph, J0, r["cost"], e.pgd_get("u"), e.pgd_get("r")))
        e.close()
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
