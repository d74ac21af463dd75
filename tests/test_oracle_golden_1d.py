"""Pins the 1D CPU oracle (oracle/vch1d_oracle.py) to golden vectors produced by the
reference itself (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import golden, relerr
from oracle import vch1d_oracle as O1

TIGHT = 5e-13
SOLVE = 1e-9


def test_operators():
    g = golden("g1d_ops_24.npz")
    N, h, dt = int(g["N"]), float(g["Lx"]) / int(g["N"]), float(g["dt"])
    P = O1.Params1D(N=N, tau=float(g["tau"]), gamma=float(g["gamma"]), c1=float(g["c1"]),
                    c2=float(g["c2"]), kappa=float(g["kappa"]))
    assert relerr(O1.lap(g["v"], h), g["Lv"]) < TIGHT
    Eg = float(g["free_energy"])
    assert abs(O1.free_energy(g["phi_old"], P.kappa, P.c1, P.c2, h, w=g["w_old"]) - Eg) < 1e-13 * max(1.0, abs(Eg))
    assert relerr(O1.lap(O1.lap(g["v"], h), h), g["LLv"]) < TIGHT
    assert relerr(O1.lap_dense(N, h) @ g["v"], g["Lv"]) < TIGHT
    assert relerr(O1.mu_init(g["phi_old"], g["w_new"], P, h), g["mu0"]) < TIGHT
    Rp = O1.residual_phi(g["phi_new"], g["phi_old"], g["mu_new"], g["mu_old"], g["w_new"],
                         g["w_old"], dt, P, h)
    Rm = O1.residual_mu(g["phi_new"], g["phi_old"], g["mu_new"], g["mu_old"], dt, h)
    assert relerr(Rp, g["Rphi"]) < TIGHT and relerr(Rm, g["Rmu"]) < TIGHT
    n = N + 1
    d = g["dvec"]
    top, bot = O1.jac_apply(g["phi_new"], d[:n], d[n:], dt, P, h)
    assert relerr(np.concatenate([top, bot]), g["Jd"]) < TIGHT
    J = O1.jac_dense(g["phi_new"], dt, P, O1.lap_dense(N, h))
    assert relerr(J @ d, g["Jd"]) < TIGHT
    assert relerr(np.linalg.solve(J, d), g["Jsol"]) < 1e-10
    assert relerr(O1._solve_newton_banded(g["phi_new"], dt, P, h, -d), g["Jsol"]) < 1e-9
    assert relerr(O1.fpp(g["phi_old"]), g["fpp"]) < TIGHT
    assert relerr(O1.adjoint_A_apply(g["phi_new"], g["v"], dt, h), g["Av"]) < 1e-11
    assert relerr(O1.adjoint_B_apply(g["phi_new"], g["v"], dt, h), g["Bv"]) < 1e-11
    assert relerr(O1.adjoint_A_apply(g["phi_new"], g["Asol"], dt, h), g["v"]) < 1e-9


def test_init_phi_random():
    g = golden("g1d_init_phi.npz")
    for key in g.files:
        N, seed = int(key.split("_")[0][1:]), int(key.split("_")[1][1:])
        assert np.array_equal(O1.init_phi_random(N, 1e-2, amp=0.01, seed=seed), g[key])


def test_newton_step():
    g = golden("g1d_newton_64.npz")
    P = O1.Params1D(N=64)
    for solver in ("dense", "banded"):
        pn, mn, hist = O1.newton_step(g["phi0"], g["mu0"], g["w0"], g["w1"], 1e-2, P, 1.0 / 64,
                                      solver=solver, return_history=True)
        assert len(hist) == len(g["hist"])
        assert relerr(pn, g["phi_new"]) < SOLVE and relerr(mn, g["mu_new"]) < SOLVE


@pytest.mark.parametrize("tag", ["32", "64", "64_ragged", "64_ic"])
@pytest.mark.parametrize("solver", ["dense", "banded"])
def test_forward_backward_cost(tag, solver):
    g = golden(f"g1d_forward_{tag}.npz")
    P = O1.Params1D(N=int(g["N"]), T=float(g["T"]), dt_initial=float(g["dt"]))
    Op = O1.OptParams1D()
    ic = g["initial_phi"] if "initial_phi" in g.files else None
    phi, x, t = O1.forward(P, initial_phi=ic, solver=solver)
    assert np.array_equal(t, g["t_hist"]) and t[0] == t[1] == 0.0       # duplicated t=0 row
    assert relerr(phi, g["phi_nat"]) < SOLVE
    phi_u, _, _ = O1.forward(P, control=g["u"], initial_phi=ic, solver=solver)
    assert relerr(phi_u, g["phi_u"]) < SOLVE
    nrow = phi.shape[0] - 2
    phi_s, _, _ = O1.forward(P, control=g["u"][:nrow], initial_phi=ic, solver=solver)
    assert relerr(phi_s, g["phi_ushort"]) < SOLVE
    with pytest.raises(IndexError):
        O1.forward(P, control=g["u"][:3], initial_phi=ic, solver=solver)
    for ct in (1, 2, 3):
        phi_T, phi_Q = O1.build_targets(x, t, g["phi_nat"][0], P.Lx, P.T, ct, 1)
        assert relerr(phi_T, g[f"phi_T_{ct}"]) < 1e-15 and relerr(phi_Q, g[f"phi_Q_{ct}"]) < 1e-15
    phi_T, phi_Q = g["phi_T_1"], g["phi_Q_1"]
    p, q, r = O1.backward(g["phi_u"], x, t, Op.b1, Op.b2, phi_Q, phi_T, solver=solver)
    assert relerr(p, g["p"]) < SOLVE and relerr(q, g["q"]) < SOLVE and relerr(r, g["r"]) < SOLVE
    assert not r[0].any() and not p[0].any()                              # B1:110 quirk
    J = O1.cost(g["phi_u"], g["u"], phi_Q, phi_T, x, t, Op.b1, Op.b2, Op.b3, Op.kappa_sparsity)
    assert abs(J - float(g["J"])) < 1e-12 * abs(float(g["J"]))
    gr = O1.gradient(g["r"], g["u"], Op.b3)
    assert np.array_equal(gr, g["grad"])
    ut = O1.gradient_step(g["u"], gr, 7.0)
    assert np.array_equal(ut, g["gstep"])
    assert np.array_equal(O1.prox_project(ut, 7.0, Op.kappa_sparsity, Op.u_min, Op.u_max), g["prox"])
    _, _, r0 = O1.backward(g["phi_u"], x, t, 1.3, 0.7, None, None, solver=solver)
    assert relerr(r0, g["r_none"]) < SOLVE


@pytest.mark.parametrize("tag", ["32", "32_bt"])
def test_pgd(tag):
    g = golden(f"g1d_pgd_{tag}.npz")
    P = O1.Params1D(N=int(g["N"]), T=float(g["T"]), dt_initial=float(g["dt"]))
    Op = O1.OptParams1D(alpha_max=float(g["alpha_max"]))
    res = O1.pgd(P, Op, n_iter=int(g["n_iter"]))
    assert np.allclose(res.costs, g["costs"], rtol=1e-9)
    assert np.allclose(res.alphas, g["alphas"], rtol=1e-14)
    assert list(res.trials) == list(g["trials"])
    assert relerr(res.u, g["u_final"]) < 1e-8 and relerr(res.phi, g["phi_final"]) < 1e-8
    if tag == "32_bt":
        assert max(res.trials) > 1


def test_newton_4096_roundoff_floor():
    """Config 2 exercises a noise-dominated branch: with a white-noise IC at
    h=1/4096 Newton stalls above tol and leaves through the line-search-failure return.
    Only the regime is pinned (norm sequence order of magnitude), not bits."""
    g = golden("g1d_newton_4096_norms.npz")
    N = 4096
    P = O1.Params1D(N=N)
    h = 1.0 / N
    phi0 = O1.init_phi_random(N, 1e-2, amp=0.01, seed=42)
    w0 = np.zeros(N + 1)
    mu0 = O1.mu_init(phi0, w0, P, h)
    pn, mn, hist = O1.newton_step(phi0, mu0, w0, w0, 1e-3, P, h, solver="banded",
                                  return_history=True)
    ref = g["norms"]
    assert abs(hist[0] / ref[0] - 1) < 1e-9                 # initial residual ~1e12
    assert hist[-1] > 1e-6 and hist[-1] < 1e-2              # stalled above tol, as the reference
    assert relerr(pn[::16], g["phi_new_sub"]) < 1e-6


def test_config1_full_size():
    """BASELINE config 1 at its full size (N = 256, T = 1, dt = 5e-3: 200 steps, 202 history rows, default weights)
    against the reference's own run (tests/golden/make_golden_r3.py): natural march, adjoint sweep, and three iterations of
    the PGD loop G1:353-480 including the 'return last try' line search of its second iteration (5 trials)."""
    g = golden("g1d_config1_256.npz")
    P = O1.Params1D(N=int(g["N"]), T=float(g["T"]), dt_initial=float(g["dt"]))
    Op = O1.OptParams1D()
    phi, x, t = O1.forward(P, solver="banded")
    assert phi.shape == (202, 257) and np.array_equal(t, g["t_hist"])
    assert relerr(phi[::2], g["phi_nat_sub"]) < SOLVE
    assert np.allclose(np.sqrt((phi ** 2).sum(axis=1)), g["nrm_phi_nat"], rtol=1e-9)
    phi_T, phi_Q = O1.build_targets(x, t, phi[0].copy(), P.Lx, P.T, 1, 1)
    assert relerr(phi_T, g["phi_T"]) < 1e-15
    p, q, r = O1.backward(phi, x, t, Op.b1, Op.b2, phi_Q, phi_T, solver="banded")
    assert relerr(r[::2], g["r_nat_sub"]) < 1e-8 and relerr(p[::4], g["p_nat_sub"]) < 1e-8
    res = O1.pgd(P, Op, n_iter=int(g["n_iter"]), solver="banded")
    assert np.allclose(res.costs, g["costs"], rtol=1e-9), (res.costs, g["costs"])
    assert np.allclose(res.alphas, g["alphas"], rtol=1e-13) and list(res.trials) == list(g["trials"])
    assert max(res.trials) == 5                                   # the search that returns its last try
    assert relerr(res.u[::2], g["u_final_sub"]) < 1e-7 and relerr(res.phi[::2], g["phi_final_sub"]) < 1e-7
    trk, trm = O1.error_metrics(res.phi, phi_Q, phi_T, x, t)
    assert abs(trk / g["tracking"][-1] - 1) < 1e-8 and abs(trm / g["terminal"][-1] - 1) < 1e-8


def test_oracle_1d_error_metrics_plateau_and_stop():
    """G1:425-450 error metrics, the x2.0 boost after 10 plateau iterations (G1:453-463) and the stop at k = 11
    (G1:466-473, where the state is NOT advanced), oracle against the reference-made runs."""
    g = golden("g1d_pgd_32_err.npz")
    res = O1.pgd(O1.Params1D(N=32, T=0.1, dt_initial=1e-2), O1.OptParams1D(alpha_max=100.0), n_iter=4)
    assert np.allclose(res.costs, g["costs"], rtol=1e-9)
    assert np.allclose(res.tracking, g["tracking"], rtol=1e-8) and np.allclose(res.terminal, g["terminal"], rtol=1e-8)
    g = golden("g1d_pgd_32_stop.npz")
    res = O1.pgd(O1.Params1D(N=int(g["N"]), T=float(g["T"]), dt_initial=float(g["dt"])),
                 O1.OptParams1D(kappa_sparsity=float(g["kappa_sparsity"])), n_iter=int(g["n_iter"]))
    assert res.converged and len(res.alphas) == int(g["stopped_at"]) + 1 == 12
    assert list(res.trials) == list(g["trials"]) and np.allclose(res.alphas, g["alphas"], rtol=1e-13)
    assert np.array_equal(np.asarray(res.costs), g["costs"])
    assert np.allclose(res.tracking, g["tracking"], rtol=1e-10) and np.allclose(res.terminal, g["terminal"], rtol=1e-10)
    assert not np.any(res.u) and relerr(res.phi, g["phi_final"]) < 1e-9
