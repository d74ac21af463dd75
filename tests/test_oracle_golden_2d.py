"""Pins the 2D CPU oracle (oracle/vch2d_oracle.py) to golden vectors produced by the
reference itself (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import golden, relerr
from oracle import vch2d_oracle as O2

TIGHT = 5e-13      # same arithmetic, different summation order
SOLVE = 1e-9       # results that passed through a SuperLU solve


def _P(g, **kw):
    return O2.Params2D(Nx=int(g["Nx"]), Ny=int(g["Ny"]), Lx=float(g["Lx"]), Ly=float(g["Ly"]),
                       tau=float(g["tau"]), gamma=float(g["gamma"]), c1=float(g["c1"]),
                       c2=float(g["c2"]), kappa=float(g["kappa"]), **kw)


@pytest.mark.parametrize("tag", ["16", "12x9"])
def test_operators(tag):
    g = golden(f"g2d_ops_{tag}.npz")
    P = _P(g)
    hx, hy = P.Lx / P.Nx, P.Ly / P.Ny
    dt = float(g["dt"])
    assert relerr(O2.lap(g["v"], hx, hy), g["Lv"]) < TIGHT
    assert relerr(O2.lap(O2.lap(g["v"], hx, hy), hx, hy), g["LLv"]) < TIGHT
    assert relerr(O2.lap(O2.lap(g["v"], hx, hy), hx, hy), g["L2v"]) < 1e-11
    L = O2.lap_matrix(P.Nx, P.Ny, hx, hy)
    assert relerr((L @ g["v"].ravel()).reshape(g["v"].shape), g["Lv"]) < TIGHT
    assert relerr(O2.reg_log(g["phi_new"]), g["reglog"]) < TIGHT
    assert relerr(O2.mu_init(g["phi_old"], g["w_new"], P, hx, hy), g["mu0"]) < TIGHT
    assert relerr(O2.w_filter(g["w_old"], dt, P.gamma, g["u_n"], g["u_np1"]), g["w_filt"]) < 1e-15
    Rphi = O2.residual_phi(g["phi_new"], g["phi_old"], g["mu_new"], g["mu_old"], g["w_new"],
                           g["w_old"], dt, P, hx, hy)
    Rmu = O2.residual_mu(g["phi_new"], g["phi_old"], g["mu_new"], g["mu_old"], dt, hx, hy)
    assert relerr(Rphi, g["Rphi"]) < TIGHT and relerr(Rmu, g["Rmu"]) < TIGHT
    n = g["v"].size
    d = g["dvec"]
    top, bot = O2.jac_apply(g["phi_new"], d[:n].reshape(g["v"].shape), d[n:].reshape(g["v"].shape),
                            dt, P, hx, hy)
    assert relerr(np.concatenate([top.ravel(), bot.ravel()]), g["Jd"]) < TIGHT
    J = O2.jac_matrix(g["phi_new"], dt, P, L)
    assert relerr(J @ d, g["Jd"]) < TIGHT
    # Schur reduction: (I/dt + M K) dphi = -R_mu + L R_phi, dmu = 2(K dphi + R_phi)
    rhs = g["rhs"]
    sol = g["Jsol"]
    dphi = sol[:n].reshape(g["v"].shape)
    Rp, Rm = -rhs[:n].reshape(g["v"].shape), -rhs[n:].reshape(g["v"].shape)
    lhs = O2.schur_apply(g["phi_new"], dphi, dt, P, hx, hy)
    assert relerr(lhs, -Rm + O2.lap(Rp, hx, hy)) < 1e-9
    Kd = -0.5 * P.kappa * O2.lap(dphi, hx, hy) + O2.jac_diag(g["phi_new"], dt, P) * dphi
    assert relerr(2.0 * (Kd + Rp), sol[n:].reshape(g["v"].shape)) < 1e-9
    assert relerr(O2.fpp(g["phi_old"], P.c1, P.c2), g["fpp"]) < TIGHT
    assert relerr(O2.adjoint_A_apply(g["phi_new"], g["v"], dt, P, hx, hy), g["Av"]) < 1e-11
    assert relerr(O2.adjoint_B_apply(g["phi_new"], g["v"], dt, P, hx, hy), g["Bv"]) < 1e-11
    assert relerr(O2.adjoint_A_apply(g["phi_new"], g["Asol"], dt, P, hx, hy), g["v"]) < 1e-9
    E = O2.free_energy(g["phi_old"], P.kappa, P.c1, P.c2, hx, hy, w=g["w_old"], eps=0.5 * 1e-2)
    assert abs(E - float(g["free_energy"])) < 1e-12 * max(1.0, abs(float(g["free_energy"])))
    assert np.array_equal(O2.trapz_weights(P.Nx + 1), g["trapz_w"])


def test_init_phi_random():
    g = golden("g2d_init_phi.npz")
    for key in g.files:
        parts = key.split("_")
        n = parts[0][1:]
        Nx, Ny = (int(n), int(n)) if "x" not in n else tuple(int(v) for v in n.split("x"))
        seed, amp = int(parts[1][1:]), float(parts[2][1:])
        got = O2.init_phi_random(Nx, Ny, 1e-2, amp=amp, seed=seed)
        assert np.array_equal(got, g[key]), key          # same RNG stream and arithmetic


def test_newton_step():
    g = golden("g2d_newton_32.npz")
    P = O2.Params2D(Nx=32, Ny=32)
    h = 1.0 / 32
    for tag, dt in (("dt1e-2", 1e-2), ("dt1e-3", 1e-3)):
        pn, mn, hist = O2.newton_step(g["phi0"], g["mu_init"], g["w0"], g["w1"], dt, P, h, h,
                                      return_history=True)
        assert len(hist) == len(g[f"hist_{tag}"])
        assert relerr(pn, g[f"phi_new_{tag}"]) < SOLVE and relerr(mn, g[f"mu_new_{tag}"]) < SOLVE
        assert np.allclose(hist[:-1], g[f"hist_{tag}"][:-1], rtol=1e-6)
        assert hist[-1] < 1e-6
    pn, mn, hist = O2.newton_step(g["phi_stress"], g["mu_stress"], g["w0"], g["w0"], 1e-3, P, h, h,
                                  return_history=True)
    assert len(hist) == len(g["hist_stress"])
    assert relerr(pn, g["phi_new_stress"]) < SOLVE


@pytest.mark.parametrize("tag", ["16", "16_ragged", "32", "14x11", "64_fine"])
def test_forward_backward_cost(tag):
    g = golden(f"g2d_forward_{tag}.npz")
    P = O2.Params2D(Nx=int(g["Nx"]), Ny=int(g["Ny"]), Lx=float(g["Lx"]), Ly=float(g["Ly"]),
                    T=float(g["T"]), dt_initial=float(g["dt"]))
    Op = O2.OptParams()
    phi, (x, y), t = O2.forward(P)
    assert np.array_equal(t, g["t_hist"])
    assert relerr(phi, g["phi_nat"]) < SOLVE
    phi_u, _, _ = O2.forward(P, control=g["u"])
    assert relerr(phi_u, g["phi_u"]) < SOLVE
    if "phi_ushort" in g.files:
        ps, _, _ = O2.forward(P, control=g["u"][:4])
        assert relerr(ps, g["phi_ushort"]) < SOLVE
    for ct, cq in ((1, 1), (2, 2)):
        if f"r_{ct}{cq}" not in g.files:
            continue
        phi_T, phi_Q = O2.build_targets(x, y, t, g["phi_nat"][0], P.Lx, P.Ly, P.T, ct, cq)
        if f"phi_T_{ct}{cq}" in g.files:
            assert relerr(phi_T, g[f"phi_T_{ct}{cq}"]) < 1e-15
        if f"phi_Q_{ct}{cq}" in g.files:
            assert relerr(phi_Q, g[f"phi_Q_{ct}{cq}"]) < 1e-15
        p, q, r = O2.backward(g["phi_u"], x, y, t, P, Op.b1, Op.b2, phi_Q, phi_T)
        assert relerr(r, g[f"r_{ct}{cq}"]) < SOLVE
        if f"p_{ct}{cq}" in g.files:
            assert relerr(p, g[f"p_{ct}{cq}"]) < SOLVE
        if f"q_{ct}{cq}" in g.files:
            assert relerr(q, g[f"q_{ct}{cq}"]) < SOLVE
        if f"J_{ct}{cq}" in g.files:
            J = O2.cost(g["phi_u"], g["u"], phi_Q, phi_T, x, y, t, Op)
            assert abs(J - float(g[f"J_{ct}{cq}"])) < 1e-12 * abs(float(g[f"J_{ct}{cq}"]))
        if (ct, cq) == (1, 1) and "Jparts" in g.files:
            parts = O2.cost_parts(g["phi_u"], g["u"], phi_Q, phi_T, x, y, t, Op)
            assert np.allclose(parts, g["Jparts"], rtol=1e-12)
            if "grad" in g.files:
                gr = O2.gradient(g["r_11"], g["u"], Op)
                assert np.array_equal(gr, g["grad"])
                for a in (0.5, 50.0):
                    if f"prox_a{a}" in g.files:
                        assert np.array_equal(O2.prox_step(g["u"], gr, a, Op), g[f"prox_a{a}"])
    if "r_none" in g.files:
        _, _, r0 = O2.backward(g["phi_u"], x, y, t, P, 1.3, 0.7, None, None)
        assert relerr(r0, g["r_none"]) < SOLVE


def test_forward_stress():
    g = golden("g2d_forward_32_stress.npz")
    P = O2.Params2D(Nx=32, Ny=32, T=float(g["T"]), dt_initial=float(g["dt"]))
    phi, _, t = O2.forward(P, amp=1.0)
    assert np.array_equal(t, g["t_hist"])
    assert relerr(phi, g["phi_nat"]) < 1e-8


@pytest.mark.parametrize("tag", ["16", "16_bt"])
def test_pgd(tag):
    g = golden(f"g2d_pgd_{tag}.npz")
    N = int(g["N"])
    P = O2.Params2D(Nx=N, Ny=N, T=float(g["T"]), dt_initial=float(g["dt"]))
    Op = O2.OptParams(alpha_max=float(g["alpha_max"]), b3=float(g["b3"]))
    res = O2.pgd(P, Op, n_iter=int(g["n_iter"]))
    assert np.allclose(res.costs, g["costs"], rtol=1e-9)
    assert np.allclose(res.alphas, g["alphas"], rtol=1e-14)
    assert list(res.attempts) == list(g["attempts"])
    assert relerr(res.u, g["u_final"]) < 1e-8
    assert relerr(res.phi, g["phi_final"]) < 1e-8
    if tag == "16_bt":
        assert max(res.attempts) >= 1            # the backtracking branch was exercised


# ---------------------------------------------------------------------------------------
# round-2 goldens: error metrics, plateau boost, stop rule (G2:336-381) -- the ORACLE against the reference-made runs
# (smoke() and the GPU tests use these oracle functions as their checker)
# ---------------------------------------------------------------------------------------
def test_oracle_error_metrics_and_rms_fallback():
    g = golden("g2d_pgd_16_err.npz")
    res = O2.pgd(O2.Params2D(Nx=16, Ny=16, T=0.1, dt_initial=1e-2), O2.OptParams(), n_iter=4)
    assert np.allclose(res.costs, g["costs"], rtol=1e-9)
    assert np.allclose(res.tracking, g["tracking"], rtol=1e-8) and np.allclose(res.terminal, g["terminal"], rtol=1e-8)
    # phi_Q = 0: the denominator falls back to sqrt(|Omega| T) (G2:353-354)
    rz = O2.pgd(O2.Params2D(Nx=16, Ny=16, T=float(g["zq_T"]), dt_initial=float(g["zq_dt"])), O2.OptParams(), n_iter=2,
                zero_target_q=True)
    assert np.allclose(rz.costs, g["zq_costs"], rtol=1e-9) and np.allclose(rz.alphas, g["zq_alphas"], rtol=1e-13)
    assert np.allclose(rz.tracking, g["zq_tracking"], rtol=1e-8) and np.allclose(rz.terminal, g["zq_terminal"], rtol=1e-8)


def test_oracle_plateau_boost():
    """alpha_max = 4e4, 9 iterations: deep backtracking incl. 'return last try' (G2:144-146) and the x1.5 boost after five
    plateau iterations (G2:365-371), as the reference ran them."""
    g = golden("g2d_pgd_16_plateau.npz")
    res = O2.pgd(O2.Params2D(Nx=16, Ny=16, T=float(g["T"]), dt_initial=float(g["dt"])),
                 O2.OptParams(alpha_max=float(g["alpha_max"])), n_iter=int(g["n_iter"]))
    assert list(res.attempts) == list(g["attempts"])
    assert np.allclose(res.alphas, g["alphas"], rtol=1e-12) and np.allclose(res.costs, g["costs"], rtol=1e-8)
    assert np.allclose(res.tracking, g["tracking"], rtol=1e-7) and np.allclose(res.terminal, g["terminal"], rtol=1e-7)
    assert relerr(res.u, g["u_final"]) < 1e-7 and relerr(res.phi, g["phi_final"]) < 1e-7


def test_oracle_stop_rule():
    """kappa_sparsity = 10: the zero control is the fixed point, every line search exhausts its 10 attempts, the boost comes
    every 5 iterations and the loop leaves through `change < 1e-5 and k > 20` at k = 21 (G2:375-381)."""
    g = golden("g2d_pgd_16_stop.npz")
    res = O2.pgd(O2.Params2D(Nx=16, Ny=16, T=float(g["T"]), dt_initial=float(g["dt"])),
                 O2.OptParams(kappa_sparsity=float(g["kappa_sparsity"])), n_iter=int(g["n_iter"]))
    assert res.converged and len(res.alphas) == int(g["stopped_at"]) + 1 == 22
    assert list(res.attempts) == list(g["attempts"]) and np.array_equal(np.asarray(res.costs), g["costs"])
    assert np.allclose(res.alphas, g["alphas"], rtol=1e-13) and np.allclose(res.changes, g["changes"], atol=1e-300)
    assert np.allclose(res.tracking, g["tracking"], rtol=1e-10) and np.allclose(res.terminal, g["terminal"], rtol=1e-10)
    assert not np.any(res.u)
