"""CPU oracle for the 1D hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

From-scratch numpy restatement of the reference's 1D algorithm; same role and
same import rule as oracle/vch2d_oracle.py (only tests/, smoke() and the
cpu_baseline leg of bench.py may import it).  Pinned against golden vectors made
by the reference itself (tests/golden/g1d_*.npz, tests/test_oracle_golden_1d.py).

The reference solves dense systems with numpy/LAPACK `dgesv` (Forward_solver.py:185,
backward_solver.py:94,116).  `solver="dense"` follows it literally; `solver="banded"`
solves the same matrices in LAPACK banded storage (`scipy.linalg.solve_banded`) so that
the N=4096 checks finish in seconds — same matrix, different elimination order.

Reference abbreviations: F1 = src/1D/Vch_control_1D/Forward_solver.py,
B1 = backward_solver.py, C1 = cost_and_function.py, G1 = GD_1D.py, K1 = config.py.
"""
from __future__ import annotations

from dataclasses import dataclass, field
import numpy as np
from scipy.linalg import solve_banded

DELTA_SEP = 1e-2       # F1:42
NEWTON_TOL = 1e-6      # F1:143
NEWTON_MAXIT = 50      # F1:144
ARMIJO_ETA = 1e-3      # F1:215


@dataclass
class Params1D:
    """K1:93-102."""
    N: int = 128
    Lx: float = 1.0
    T: float = 1.0
    dt_initial: float = 1e-2
    tau: float = 0.05
    gamma: float = 10.0
    c1: float = 0.75
    c2: float = 1.0
    kappa: float = 0.03 ** 2


@dataclass
class OptParams1D:
    """K1:115-123."""
    b1: float = 0.3
    b2: float = 13.0
    b3: float = 0.0019
    kappa_sparsity: float = 0.00009
    alpha_max: float = 100.0
    max_iter: int = 1000
    u_min: float = -1.0
    u_max: float = 1.0


# The reference's backward solver freezes c1, c2, tau, gamma from a DEFAULT config at
# import time and ignores the run-time one (B1:29-33).  Reproduced.
_FROZEN = Params1D()


def trapz_weights(n):
    w = np.ones(n)
    w[0] = w[-1] = 0.5
    return w


def lap(v, h):
    """Mirrored-Neumann second difference (F1:64-80), last axis."""
    a = 1.0 / (h * h)
    out = np.empty_like(v)
    out[..., 1:-1] = a * v[..., :-2] + (-2.0 * a) * v[..., 1:-1] + a * v[..., 2:]
    out[..., 0] = (-2.0 * a) * v[..., 0] + (2.0 * a) * v[..., 1]
    out[..., -1] = (2.0 * a) * v[..., -2] + (-2.0 * a) * v[..., -1]
    return out


def lap_dense(N, h):
    a = 1.0 / (h * h)
    L = np.zeros((N + 1, N + 1))
    i = np.arange(1, N)
    L[i, i - 1] = a; L[i, i] = -2 * a; L[i, i + 1] = a
    L[0, 0], L[0, 1] = -2 * a, 2 * a
    L[N, N - 1], L[N, N] = 2 * a, -2 * a
    return L


def reg_log(phi, eps=None):
    """F1:57-62."""
    if eps is None:
        eps = max(1e-8, 0.5 * DELTA_SEP)
    p = np.clip(phi, -1 + eps, 1 - eps)
    return np.log((1 + p) / (1 - p))


def fpp(phi, eps=1e-8):
    """B1:36-46 with the frozen default c1, c2."""
    p = np.clip(phi, -1 + eps, 1 - eps)
    return 2.0 * _FROZEN.c1 / (1.0 - p ** 2) - 2.0 * _FROZEN.c2


def mu_init(phi, w, P, h):
    """F1:82-86."""
    return -P.kappa * lap(phi, h) + (P.c1 * reg_log(phi) - 2.0 * P.c2 * phi) - w


def w_filter(w_old, dt, gamma, u_n, u_np1):
    g = gamma / dt
    return ((g - 0.5) * w_old + 0.5 * (u_np1 + u_n)) / (g + 0.5)


def residual_mu(pn, po, mn, mo, dt, h):
    """F1:93-97."""
    return (pn - po) / dt - 0.5 * (lap(mn, h) + lap(mo, h))


def residual_phi(pn, po, mn, mo, wn, wo, dt, P, h):
    """F1:99-109."""
    return ((P.tau * (pn - po) / dt) - 0.5 * P.kappa * (lap(pn, h) + lap(po, h))
            + (P.c1 * reg_log(pn) + (-2.0 * P.c2 * po)) - 0.5 * (mn + mo) - 0.5 * (wn + wo))


def jac_diag(pn, dt, P):
    """tau/dt + 2c1/(1-phi^2), NOT clipped in 1D (F1:122-124)."""
    return P.tau / dt + 2.0 * P.c1 / (1.0 - pn ** 2)


def jac_apply(pn, dphi, dmu, dt, P, h):
    top = -0.5 * P.kappa * lap(dphi, h) + jac_diag(pn, dt, P) * dphi - 0.5 * dmu
    bot = dphi / dt - 0.5 * lap(dmu, h)
    return top, bot


def jac_dense(pn, dt, P, L):
    """F1:111-137, unknown order [dphi; dmu]."""
    n = pn.size
    J = np.zeros((2 * n, 2 * n))
    Kpp = -0.5 * P.kappa * L.copy()
    np.fill_diagonal(Kpp, np.diag(Kpp) + jac_diag(pn, dt, P))
    J[:n, :n] = Kpp
    J[:n, n:] = -0.5 * np.eye(n)
    J[n:, :n] = (1.0 / dt) * np.eye(n)
    J[n:, n:] = -0.5 * L
    return J


def _solve_newton_banded(pn, dt, P, h, R):
    """Same matrix as jac_dense with unknowns interleaved (phi_i, mu_i): bandwidth 3."""
    n = pn.size
    a = 1.0 / (h * h)
    lo = np.full(n, a); lo[-1] = 2 * a; lo[0] = 0.0       # coefficient of v[i-1] in (Lv)[i]
    up = np.full(n, a); up[0] = 2 * a; up[-1] = 0.0       # coefficient of v[i+1]
    dg = -2.0 * a
    ab = np.zeros((7, 2 * n))                              # l = u = 3
    D = jac_diag(pn, dt, P)
    ev, od = np.arange(0, 2 * n, 2), np.arange(1, 2 * n, 2)

    def put(r, c, val):
        ab[3 + r - c, c] = val
    put(ev, ev, -0.5 * P.kappa * dg + D)
    put(ev, od, -0.5)
    put(od, ev, 1.0 / dt)
    put(od, od, -0.5 * dg)
    put(ev[1:], ev[:-1], -0.5 * P.kappa * lo[1:]); put(ev[:-1], ev[1:], -0.5 * P.kappa * up[:-1])
    put(od[1:], od[:-1], -0.5 * lo[1:]); put(od[:-1], od[1:], -0.5 * up[:-1])
    rhs = np.empty(2 * n)
    rhs[0::2], rhs[1::2] = -R[:n], -R[n:]
    sol = solve_banded((3, 3), ab, rhs)
    return np.concatenate([sol[0::2], sol[1::2]])


def free_energy(phi, kappa, c1, c2, h, w=None, eps=None):
    """Discrete free energy with forward-difference gradient (F1:243-262)."""
    phi = np.asarray(phi)
    wts = trapz_weights(len(phi))
    E = (kappa / (2.0 * h)) * np.sum(np.diff(phi) ** 2)
    eps = 1e-8 if eps is None else eps
    p = np.clip(phi, -1 + eps, 1 - eps)
    E += h * np.dot(wts, c1 * ((1 + p) * np.log(1 + p) + (1 - p) * np.log(1 - p)) - c2 * p ** 2)
    if w is not None:
        E -= h * np.dot(wts, np.asarray(w) * phi)
    return E


def init_phi_random(N, delta_sep=DELTA_SEP, amp=0.1, seed=42, enforce_zero_mean=True):
    """F1:264-277."""
    rng = np.random.default_rng(seed)
    phi0 = amp * rng.standard_normal(N + 1)
    if enforce_zero_mean:
        wts = trapz_weights(N + 1)
        phi0 -= np.dot(wts, phi0) / wts.sum()
    return np.clip(phi0, -1 + delta_sep, 1 - delta_sep)


def newton_step(phi_old, mu_old, w_old, w_new, dt, P, h, solver="dense", return_history=False,
                stats=None):
    """F1:139-235.  Initial guess (phi_old, mu_old) (F1:141-142); step
    alpha = min(1, 0.9*alpha_max) (F1:198-212); Armijo eta=1e-3 with the extra admissibility
    test all|phi_t| < 1-delta (F1:219); 12 failed halvings END the whole Newton loop with
    the current iterate (F1:227-229)."""
    pn, mn = phi_old.copy(), mu_old.copy()
    n = pn.size
    wts_h = h * trapz_weights(n)
    L = lap_dense(n - 1, h) if solver == "dense" else None
    hist = []
    nsolve = 0
    for k in range(NEWTON_MAXIT):
        R = np.concatenate([residual_phi(pn, phi_old, mn, mu_old, w_new, w_old, dt, P, h),
                            residual_mu(pn, phi_old, mn, mu_old, dt, h)])
        nR = np.linalg.norm(R)
        hist.append(nR)
        if k % 10 == 0 and not np.isfinite(np.dot(wts_h, R[n:])):
            raise RuntimeError("Non-finite mass_defect; check phi bounds/log regularization.")
        if nR < NEWTON_TOL:
            break
        if solver == "dense":
            J = jac_dense(pn, dt, P, L)
            try:
                delta = np.linalg.solve(J, -R)
            except np.linalg.LinAlgError:
                delta = np.linalg.solve(J + 1e-10 * np.eye(2 * n), -R)
        else:
            delta = _solve_newton_banded(pn, dt, P, h, R)
        nsolve += 1
        dphi, dmu = delta[:n], delta[n:]
        with np.errstate(divide="ignore", invalid="ignore"):
            pos, neg = dphi > 0, dphi < 0
            a_pos = np.min((1 - DELTA_SEP - pn[pos]) / dphi[pos]) if np.any(pos) else np.inf
            a_neg = np.min((-1 + DELTA_SEP - pn[neg]) / dphi[neg]) if np.any(neg) else np.inf
            amax = min(a_pos, a_neg)
        if not np.isfinite(amax) or amax <= 0:
            amax = 1.0
        alpha = min(1.0, 0.9 * amax)
        ok = False
        for _ in range(12):
            pt, mt = pn + alpha * dphi, mn + alpha * dmu
            if np.all(np.abs(pt) < 1 - DELTA_SEP):
                Rt = np.concatenate([residual_phi(pt, phi_old, mt, mu_old, w_new, w_old, dt, P, h),
                                     residual_mu(pt, phi_old, mt, mu_old, dt, h)])
                if np.linalg.norm(Rt) <= (1 - ARMIJO_ETA * alpha) * nR:
                    pn, mn, ok = pt, mt, True
                    break
            alpha *= 0.5
        if not ok:
            break
    if stats is not None:
        stats["solves"] = stats.get("solves", 0) + nsolve
    return (pn, mn, hist) if return_history else (pn, mn)


def forward(P: Params1D, control=None, initial_phi=None, seed=42, amp=0.01, solver="dense",
            max_steps=None, stats=None):
    """F1:286-386.  History has M+2 rows: t=0 appears twice (F1:329-336).  Control rows
    (u[step], u[step+1]) while step < len(u)-1, else hold u[step] (F1:347-353; an array
    shorter than that raises IndexError as in the reference).  Post-step: clip, then a
    uniform shift by mass_error/Lx, always (F1:361-366)."""
    N, h = int(P.N), P.Lx / int(P.N)
    x = np.linspace(0, P.Lx, N + 1)
    if initial_phi is not None and initial_phi.shape == (N + 1,):
        phi = initial_phi.copy()
    else:
        phi = init_phi_random(N, DELTA_SEP, amp=amp, seed=seed)
    w = np.zeros(N + 1)
    wts_h = h * trapz_weights(N + 1)
    mass0 = np.dot(wts_h, phi)
    mu = mu_init(phi, w, P, h)
    t, step = 0.0, 0
    ts = [0.0, 0.0]
    hist = [phi.copy(), phi.copy()]
    zero = np.zeros(N + 1)
    while t < P.T - 1e-10:
        if max_steps is not None and step >= max_steps:
            break
        dts = min(P.dt_initial, P.T - t)
        if control is not None:
            if step < control.shape[0] - 1:
                u_n, u_np1 = control[step, :], control[step + 1, :]
            else:
                u_n = u_np1 = control[step, :]
        else:
            u_n = u_np1 = zero
        w_new = w_filter(w, dts, P.gamma, u_n, u_np1)
        pn, mn = newton_step(phi, mu, w, w_new, dts, P, h, solver=solver, stats=stats)
        phi = np.clip(pn, -1 + DELTA_SEP, 1 - DELTA_SEP)
        mu, w = mn, w_new
        phi -= (np.dot(wts_h, phi) - mass0) / P.Lx
        t += dts
        step += 1
        hist.append(phi.copy())
        ts.append(min(t, P.T))
    return np.array(hist), x, np.array(ts)


def _lap_rows(n, h):
    """Row-wise tridiagonal coefficients of the mirrored-Neumann operator:
    (Lv)[i] = lo[i] v[i-1] + dg[i] v[i] + up[i] v[i+1]."""
    a = 1.0 / (h * h)
    lo = np.full(n, a); lo[-1] = 2 * a; lo[0] = 0.0
    up = np.full(n, a); up[0] = 2 * a; up[-1] = 0.0
    dg = np.full(n, -2.0 * a)
    return lo, dg, up


def _penta_rows(n, h):
    """Row-wise coefficients (offsets -2..2) of L@L from the tridiagonal rows of L."""
    lo, dg, up = _lap_rows(n, h)
    sh = lambda v, k: np.concatenate([v[k:], np.zeros(k)]) if k > 0 else np.concatenate([np.zeros(-k), v[:k]])
    m2 = lo * sh(lo, -1)
    m1 = lo * sh(dg, -1) + dg * lo
    d0 = lo * sh(up, -1) + dg * dg + up * sh(lo, 1)
    p1 = dg * up + up * sh(dg, 1)
    p2 = up * sh(up, 1)
    return m2, m1, d0, p1, p2


def _rows_to_banded(rows):
    """Row-wise offset coefficients -> LAPACK banded storage ab[u + i - j, j] = A[i, j]."""
    k = (len(rows) - 1) // 2
    n = rows[0].size
    ab = np.zeros((2 * k + 1, n))
    for off, v in zip(range(-k, k + 1), rows):
        i = np.arange(max(0, -off), min(n, n - off))
        ab[k - off, i + off] = v[i]
    return ab


def _rows_matvec(rows, x):
    k = (len(rows) - 1) // 2
    n = x.size
    y = np.zeros(n)
    for off, v in zip(range(-k, k + 1), rows):
        i = np.arange(max(0, -off), min(n, n - off))
        y[i] += v[i] * x[i + off]
    return y


def backward(phi_hist, x, t_hist, b1, b2, phi_Q=None, phi_T=None, solver="dense", max_steps=None):
    """B1:48-126.  Physical parameters are the frozen defaults (B1:29-33); dt_n <= 0
    rows are skipped and stay zero (B1:110), so row 0 of p, q, r is zero because of the
    duplicated t=0 entry."""
    M1, n = phi_hist.shape
    tau, gamma = _FROZEN.tau, _FROZEN.gamma
    if phi_Q is None:
        phi_Q = np.zeros_like(phi_hist)
    if phi_T is None:
        phi_T = np.zeros(n)
    h = x[1] - x[0]
    p = np.zeros_like(phi_hist); q = np.zeros_like(phi_hist); r = np.zeros_like(phi_hist)
    last = 0 if max_steps is None else max(0, M1 - 1 - max_steps)
    if solver == "dense":
        L = lap_dense(n - 1, h)
        I = np.eye(n)
        L2 = L @ L
        p[-1] = np.linalg.solve(I - tau * L, b2 * (phi_hist[-1] - phi_T))
        q[-1] = -(L @ p[-1])
    else:
        lo, dg, up = _lap_rows(n, h)
        z = np.zeros(n)
        one = np.ones(n)
        Lr = (z, lo, dg, up, z)
        L2r = _penta_rows(n, h)
        comb = lambda cI, cL, cL2, D: tuple(cI * e + (cL + (0.0 if D is None else D)) * l + cL2 * l2
                                            for e, l, l2 in zip((z, z, one, z, z), Lr, L2r))
        p[-1] = solve_banded((2, 2), _rows_to_banded(comb(1.0, -tau, 0.0, None)),
                             b2 * (phi_hist[-1] - phi_T))
        q[-1] = -lap(p[-1], h)
    for k in range(M1 - 2, last - 1, -1):
        dt = t_hist[k + 1] - t_hist[k]
        if dt <= 0:
            continue
        src = 0.5 * dt * b1 * ((phi_hist[k] - phi_Q[k]) + (phi_hist[k + 1] - phi_Q[k + 1]))
        if solver == "dense":
            A = I - tau * L + 0.5 * dt * L2 - 0.5 * dt * (fpp(phi_hist[k])[:, None] * L)
            Bm = I - tau * L - 0.5 * dt * L2 + 0.5 * dt * (fpp(phi_hist[k + 1])[:, None] * L)
            rhs = Bm @ p[k + 1] + src
            try:
                p[k] = np.linalg.solve(A, rhs)
            except np.linalg.LinAlgError:
                p[k] = np.linalg.solve(A + 1e-10 * I, rhs)
            q[k] = -(L @ p[k])
        else:
            Ar = comb(1.0, -tau, 0.5 * dt, -0.5 * dt * fpp(phi_hist[k]))
            Br = comb(1.0, -tau, -0.5 * dt, 0.5 * dt * fpp(phi_hist[k + 1]))
            rhs = _rows_matvec(Br, p[k + 1]) + src
            p[k] = solve_banded((2, 2), _rows_to_banded(Ar), rhs)
            q[k] = -lap(p[k], h)
        r[k] = ((gamma - 0.5 * dt) / (gamma + 0.5 * dt)) * r[k + 1] \
            + ((dt * 0.5) / (gamma + 0.5 * dt)) * (q[k] + q[k + 1])
    return p, q, r


def adjoint_A_apply(phi_n, v, dt, h):
    Lv = lap(v, h)
    return v - _FROZEN.tau * Lv + 0.5 * dt * lap(Lv, h) - 0.5 * dt * fpp(phi_n) * Lv


def adjoint_B_apply(phi_np1, v, dt, h):
    Lv = lap(v, h)
    return v - _FROZEN.tau * Lv - 0.5 * dt * lap(Lv, h) + 0.5 * dt * fpp(phi_np1) * Lv


def _trapz(f, x, axis=-1):
    f = np.moveaxis(np.asarray(f), axis, -1)
    d = np.diff(np.asarray(x))
    return np.sum(d * (f[..., 1:] + f[..., :-1]) / 2.0, axis=-1)


def cost_parts(phi_hist, u, phi_Q, phi_T, x, t_hist, b1, b2, b3, kappa):
    """C1:55-73."""
    J1 = (b1 / 2.0) * _trapz(_trapz((phi_hist - phi_Q) ** 2, x, 1), t_hist)
    J2 = (b2 / 2.0) * _trapz((phi_hist[-1] - phi_T) ** 2, x)
    J3 = (b3 / 2.0) * _trapz(_trapz(u ** 2, x, 1), t_hist)
    J4 = kappa * _trapz(_trapz(np.abs(u), x, 1), t_hist)
    return np.array([J1, J2, J3, J4])


def cost(phi_hist, u, phi_Q, phi_T, x, t_hist, b1, b2, b3, kappa):
    return float(np.sum(cost_parts(phi_hist, u, phi_Q, phi_T, x, t_hist, b1, b2, b3, kappa)))


def gradient(r, u, b3):
    """C1:99."""
    return r + b3 * u


def gradient_step(u, g, alpha):
    """C1:111."""
    return u - alpha * g


def prox_project(u_temp, alpha, kappa, u_min, u_max):
    """G1:56-71."""
    s = np.sign(u_temp) * np.maximum(np.abs(u_temp) - alpha * kappa, 0)
    return np.clip(s, u_min, u_max)


def build_targets(x, t_hist, phi_initial, Lx, T, choice_t=1, choice_q=1, A_T=0.7, k_tan=0.45):
    """G1:210-252 (ramp uses t_hist/t_hist[-1], not the config T)."""
    if choice_t == 1:
        phi_T = A_T * np.sin(2.0 * np.pi * x / Lx)
    elif choice_t == 2:
        phi_T = A_T * np.cos(2.0 * np.pi * x / Lx)
    else:
        tr = np.tan(2.0 * np.pi * k_tan * (x / Lx - 0.5))
        sc = np.max(np.abs(tr))
        phi_T = A_T * (tr / (sc if sc > 1e-12 else 1.0))
    if choice_q == 1:
        tp = (t_hist / (t_hist[-1] if t_hist[-1] > 0 else 1.0))[:, None]
        phi_Q = (1.0 - tp) * phi_initial + tp * phi_T
    else:
        phi_Q = np.zeros((len(t_hist), len(x)))
    return phi_T, phi_Q


def error_metrics(phi_hist, phi_Q, phi_T, x, t_hist):
    """Relative tracking / terminal errors of the driver loop, G1:425-450: space-time L2 of phi - phi_Q (trapezoid in x
    per level, then in t) over that of phi_Q -- replaced by the RMS scale sqrt(|Omega| T) when the target is ~0 -- and
    the L2 of phi(T) - phi_T over that of phi_T, each denominator + 1e-12."""
    l2xt = lambda a: float(np.sqrt(_trapz(_trapz(a ** 2, x, 1), t_hist)))
    l2x = lambda a: float(np.sqrt(_trapz(a ** 2, x)))
    rms = float(np.sqrt(max(float(x[-1] - x[0]), 1e-30) * max(float(t_hist[-1] - t_hist[0]), 1e-30)))
    den = l2xt(phi_Q)
    if den < 1e-9 * rms:
        den = rms
    return l2xt(phi_hist - phi_Q) / (den + 1e-12), l2x(phi_hist[-1] - phi_T) / (l2x(phi_T) + 1e-12)


@dataclass
class PGDResult:
    costs: list = field(default_factory=list)
    alphas: list = field(default_factory=list)
    trials: list = field(default_factory=list)
    tracking: list = field(default_factory=list)
    terminal: list = field(default_factory=list)
    u: np.ndarray = None
    phi: np.ndarray = None
    r: np.ndarray = None
    converged: bool = False


def pgd(P: Params1D, O: OptParams1D, n_iter=None, choice_t=1, choice_q=1, solver="dense"):
    """G1:333-477: optimistic step with alpha_prev; on failure backtracking from
    alpha_prev (beta 0.8, <=5 trials, G1:73-113); alpha growth 1.2, plateau (10 its,
    |dJ|<1e-7) -> 2.0; stop when the relative control change < 1e-5 and k > 10."""
    fwd = lambda u: forward(P, control=u, solver=solver)
    phi_k, x, t_hist = fwd(None)
    u_k = np.zeros_like(phi_k)
    phi_T, phi_Q = build_targets(x, t_hist, phi_k[0].copy(), P.Lx, P.T, choice_t, choice_q)
    cargs = (O.b1, O.b2, O.b3, O.kappa_sparsity)
    cost_k = cost(phi_k, u_k, phi_Q, phi_T, x, t_hist, *cargs)
    res = PGDResult(costs=[cost_k])
    alpha_prev, plateau = O.alpha_max, 0
    r_k = None
    for k in range(O.max_iter if n_iter is None else n_iter):
        _, _, r_k = backward(phi_k, x, t_hist, O.b1, O.b2, phi_Q, phi_T, solver=solver)
        g = gradient(r_k, u_k, O.b3)
        u_o = prox_project(gradient_step(u_k, g, alpha_prev), alpha_prev, O.kappa_sparsity,
                           O.u_min, O.u_max)
        phi_o, _, _ = fwd(u_o)
        c_o = cost(phi_o, u_o, phi_Q, phi_T, x, t_hist, *cargs)
        if c_o < cost_k:
            a_k, u_n, c_n, phi_n, nt = alpha_prev, u_o, c_o, phi_o, 1
        else:
            alpha, nt = alpha_prev, 0
            for _ in range(5):
                nt += 1
                u_n = prox_project(gradient_step(u_k, g, alpha), alpha, O.kappa_sparsity,
                                   O.u_min, O.u_max)
                phi_n, _, _ = fwd(u_n)
                c_n = cost(phi_n, u_n, phi_Q, phi_T, x, t_hist, *cargs)
                if c_n < cost_k:
                    break
                alpha *= 0.8
            a_k = alpha
        res.costs.append(c_n); res.alphas.append(a_k); res.trials.append(nt)
        e1, e2 = error_metrics(phi_n, phi_Q, phi_T, x, t_hist)
        res.tracking.append(e1); res.terminal.append(e2)
        if k > 0 and abs(res.costs[-1] - res.costs[-2]) < 1e-7:
            plateau += 1
        else:
            plateau = 0
        if plateau >= 10:
            alpha_prev, plateau = min(O.alpha_max, a_k * 2.0), 0
        else:
            alpha_prev = min(O.alpha_max, a_k * 1.2)
        change = np.linalg.norm(u_n - u_k) / (np.linalg.norm(u_k) + 1e-9)
        if change < 1e-5 and k > 10:
            u_k = u_n.copy()
            res.converged = True
            break
        u_k, cost_k, phi_k = u_n.copy(), c_n, phi_n
    res.u, res.phi, res.r = u_k, phi_k, r_k
    res.phi_T, res.phi_Q, res.t_hist, res.x = phi_T, phi_Q, t_hist, x
    return res
