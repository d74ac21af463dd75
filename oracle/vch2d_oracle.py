"""CPU oracle for the 2D hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A from-scratch numpy/scipy restatement of the reference's 2D algorithm
(forward Crank-Nicolson/Newton march, adjoint sweep, cost, gradient, prox, PGD
loop).  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import this module; the product path (the HIP engine behind
`include/vch.h`) never does.

Parity pin: every public function here is checked against golden vectors that
were produced by importing the reference itself (tests/golden/make_golden.py ->
tests/golden/g2d_*.npz; tests/test_oracle_golden.py).  Pinned, not "unpinned".

The reference solves its linear systems with SciPy's SuperLU (`spsolve`,
Forward2_solver.py:370, backward2_solver.py:185,229); SciPy is a third-party
dependency present in this image (scipy 1.15.3) and this oracle calls it at the
same seam, so the oracle has the reference's cost profile and is what
`bench.py` times as the CPU baseline (kind "port").

Reference citations use the abbreviations of SURVEY.md:
  F2 = src/2D/Vch_control_2D/Forward2_solver.py     B2 = .../backward2_solver.py
  C2 = .../cost2_and_function.py                     G2 = .../GD2_configured.py
"""
from __future__ import annotations

from dataclasses import dataclass, field
import time as _time

import numpy as np
import scipy.sparse as sp
from scipy.sparse.linalg import spsolve, splu

DELTA_SEP = 1e-2          # F2:510
NEWTON_TOL = 1e-6         # F2:353
NEWTON_MAXIT = 500        # F2:353
ARMIJO_ETA = 1e-4         # F2:394
ARMIJO_TRIALS = 12        # F2:398


@dataclass
class Params2D:
    """Field names and defaults of the reference's ForwardSolverConfig (config.py:103-113)."""
    Nx: int = 128
    Ny: int = 128
    Lx: float = 1.0
    Ly: float = 1.0
    T: float = 1.0
    dt_initial: float = 1e-2
    tau: float = 0.05
    gamma: float = 10.0
    c1: float = 0.75
    c2: float = 1.0
    kappa: float = 0.01 ** 2


@dataclass
class OptParams:
    """Field names and defaults of the reference's OptimizationConfig (config.py:137-144)."""
    b1: float = 5.0
    b2: float = 10.0
    b3: float = 1e-4
    kappa_sparsity: float = 1e-4
    alpha_max: float = 50.0
    max_iter: int = 500
    u_min: float = -1.0
    u_max: float = 1.0


# ----------------------------------------------------------------------------
# discrete operators
# ----------------------------------------------------------------------------
def trapz_weights(n):
    """[1/2, 1, ..., 1, 1/2] (F2:430-441)."""
    w = np.ones(n)
    w[0] = w[-1] = 0.5
    return w


def _second_diff_neumann(g, axis, h):
    """Mirrored-ghost second difference along `axis` (F2:115-122): interior
    (v[i-1]-2v[i]+v[i+1])/h^2, ends 2(v[1]-v[0])/h^2 and 2(v[N-1]-v[N])/h^2."""
    g = np.moveaxis(g, axis, 0)
    out = np.empty_like(g)
    a = 1.0 / (h * h)
    out[1:-1] = a * g[:-2] + (-2.0 * a) * g[1:-1] + a * g[2:]
    out[0] = (-2.0 * a) * g[0] + (2.0 * a) * g[1]
    out[-1] = (2.0 * a) * g[-2] + (-2.0 * a) * g[-1]
    return np.moveaxis(out, 0, axis)


def lap(v, hx, hy):
    """Matrix-free action of the reference's Laplacian on an (Nx+1, Ny+1) field.

    F2:125-152: the matrix is kron(I_{Ny+1}, Lx) + kron(Ly, I_{Nx+1}) but fields are
    raveled y-fastest, so the operator actually acts on the flat memory
    reinterpreted as (Ny+1) rows of (Nx+1) entries: the hx-stencil runs along the
    fast axis and the hy-stencil along the slow axis (identical
    to the textbook operator when Nx==Ny and hx==hy).  Reproduced as is."""
    nx1, ny1 = v.shape
    g = np.ascontiguousarray(v).reshape(ny1, nx1)
    out = _second_diff_neumann(g, 1, hx) + _second_diff_neumann(g, 0, hy)
    return out.reshape(nx1, ny1)


def lap_matrix(Nx, Ny, hx, hy):
    """Assembled CSR form of `lap` (same flat layout), built from stencil
    index arithmetic; used only for the direct solves."""
    nf, ns = Nx + 1, Ny + 1                 # fast-axis length, slow-axis length
    idx = np.arange(nf * ns).reshape(ns, nf)
    rows, cols, vals = [], [], []

    def add(r, c, val):
        rows.append(r.ravel()); cols.append(c.ravel()); vals.append(np.full(r.size, val))

    ax, ay = 1.0 / (hx * hx), 1.0 / (hy * hy)
    add(idx, idx, -2.0 * ax - 2.0 * ay)
    # fast axis
    add(idx[:, 1:-1], idx[:, :-2], ax); add(idx[:, 1:-1], idx[:, 2:], ax)
    add(idx[:, 0], idx[:, 1], 2.0 * ax); add(idx[:, -1], idx[:, -2], 2.0 * ax)
    # slow axis
    add(idx[1:-1, :], idx[:-2, :], ay); add(idx[1:-1, :], idx[2:, :], ay)
    add(idx[0, :], idx[1, :], 2.0 * ay); add(idx[-1, :], idx[-2, :], 2.0 * ay)
    n = nf * ns
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                         shape=(n, n))


def reg_log(phi, delta_sep=DELTA_SEP):
    """ln((1+p)/(1-p)), p = clip(phi, +-(1-eps)), eps = max(1e-8, delta_sep/2) (F2:86-102)."""
    eps = max(1e-8, 0.5 * delta_sep)
    p = np.clip(phi, -1.0 + eps, 1.0 - eps)
    return np.log((1.0 + p) / (1.0 - p))


def fpp(phi, c1, c2, eps=1e-8):
    """f''(phi) = 2c1/(1-p^2) - 2c2 with p = clip(phi, +-(1-1e-8)) (B2:41-72)."""
    p = np.clip(phi, -1.0 + eps, 1.0 - eps)
    return 2.0 * c1 / (1.0 - p * p) - 2.0 * c2


def mu_init(phi, w, P: Params2D, hx, hy):
    """mu = -kappa*Lap(phi) + c1*reglog(phi) - 2 c2 phi - w (F2:155-167)."""
    return -P.kappa * lap(phi, hx, hy) + (P.c1 * reg_log(phi) - 2.0 * P.c2 * phi) - w


def w_filter(w_old, dt, gamma, u_n, u_np1):
    """Crank-Nicolson step of gamma w' + w = u (F2:170-181)."""
    g = gamma / dt
    return ((g - 0.5) * w_old + 0.5 * (u_np1 + u_n)) / (g + 0.5)


def residual_mu(phi_new, phi_old, mu_new, mu_old, dt, hx, hy):
    """F2:184-196."""
    return (phi_new - phi_old) / dt - 0.5 * (lap(mu_new, hx, hy) + lap(mu_old, hx, hy))


def residual_phi(phi_new, phi_old, mu_new, mu_old, w_new, w_old, dt, P: Params2D, hx, hy):
    """F2:199-221 (convex part implicit, concave part explicit)."""
    return ((P.tau * (phi_new - phi_old) / dt)
            - 0.5 * P.kappa * (lap(phi_new, hx, hy) + lap(phi_old, hx, hy))
            + (P.c1 * reg_log(phi_new) + (-2.0 * P.c2 * phi_old))
            - 0.5 * (mu_new + mu_old) - 0.5 * (w_new + w_old))


def jac_diag(phi_new, dt, P: Params2D, delta_sep=DELTA_SEP):
    """tau/dt + 2c1/(1 - clip(phi^2, 0, 1-delta^2)) (F2:243-244)."""
    psq = np.clip(phi_new * phi_new, 0.0, 1.0 - delta_sep ** 2)
    return P.tau / dt + 2.0 * P.c1 / (1.0 - psq)


def jac_apply(phi_new, dphi, dmu, dt, P: Params2D, hx, hy):
    """Matrix-free action of the 2x2-block Newton matrix (F2:241-253)."""
    D = jac_diag(phi_new, dt, P)
    top = -0.5 * P.kappa * lap(dphi, hx, hy) + D * dphi - 0.5 * dmu
    bot = dphi / dt - 0.5 * lap(dmu, hx, hy)
    return top, bot


def jac_matrix(phi_new, dt, P: Params2D, L):
    """Assembled form of jac_apply, unknown order [dphi; dmu] (F2:224-253)."""
    n = phi_new.size
    Kpp = (-0.5 * P.kappa) * L + sp.diags(jac_diag(phi_new, dt, P).ravel(), 0, format="csr")
    I = sp.identity(n, format="csr")
    return sp.bmat([[Kpp, -0.5 * I], [(1.0 / dt) * I, -0.5 * L]], format="csr")


def schur_apply(phi_new, x, dt, P: Params2D, hx, hy):
    """(I/dt + M K) x with M=-Lap, K = kappa/2 M + D: the scalar 13-point system the
    HIP engine solves.  Here only as a checker."""
    D = jac_diag(phi_new, dt, P)
    Kx = -0.5 * P.kappa * lap(x, hx, hy) + D * x
    return x / dt - lap(Kx, hx, hy)


def adjoint_A_apply(phi_n, v, dt, P: Params2D, hx, hy):
    """(I - tau L + dt/2 L^2 - dt/2 D L) v, D = f''(phi_n) (B2:195-198)."""
    Lv = lap(v, hx, hy)
    return v - P.tau * Lv + 0.5 * dt * lap(Lv, hx, hy) - 0.5 * dt * fpp(phi_n, P.c1, P.c2) * Lv


def adjoint_B_apply(phi_np1, v, dt, P: Params2D, hx, hy):
    """(I - tau L - dt/2 L^2 + dt/2 D L) v, D = f''(phi_{n+1}) (B2:200-203)."""
    Lv = lap(v, hx, hy)
    return v - P.tau * Lv - 0.5 * dt * lap(Lv, hx, hy) + 0.5 * dt * fpp(phi_np1, P.c1, P.c2) * Lv


def free_energy(phi, kappa, c1, c2, hx, hy, w=None, eps=None):
    """Discrete free energy with forward-difference gradient (F2:256-319)."""
    eps = 1e-8 if eps is None else eps
    a = np.asarray(phi)
    wts = np.outer(trapz_weights(a.shape[0]), trapz_weights(a.shape[1]))
    d0, d1 = np.diff(a, axis=0), np.diff(a, axis=1)
    Eg = (kappa / (2.0 * hx)) * np.sum(d1 ** 2) * hy + (kappa / (2.0 * hy)) * np.sum(d0 ** 2) * hx
    p = np.clip(a, -1.0 + eps, 1.0 - eps)
    psi = c1 * ((1.0 + p) * np.log(1.0 + p) + (1.0 - p) * np.log(1.0 - p)) - c2 * p ** 2
    E = Eg + hx * hy * np.sum(wts * psi)
    if w is not None:
        E -= hx * hy * np.sum(wts * np.asarray(w) * a)
    return E


# ----------------------------------------------------------------------------
# initial condition
# ----------------------------------------------------------------------------
def init_phi_random(Nx, Ny, delta_sep=DELTA_SEP, amp=0.5, seed=42, enforce_zero_mean=True):
    """amp*N(0,1) from PCG64(seed), weighted zero mean, clip, <=8 interior mass-fix
    passes (F2:444-486)."""
    rng = np.random.default_rng(seed)
    phi0 = amp * rng.standard_normal((Nx + 1, Ny + 1))
    wts = np.outer(trapz_weights(Nx + 1), trapz_weights(Ny + 1))
    Wtot = np.sum(wts)
    if enforce_zero_mean:
        phi0 -= np.sum(wts * phi0) / Wtot
    lo, hi = -1.0 + delta_sep, 1.0 - delta_sep
    phi0 = np.clip(phi0, lo, hi)
    if enforce_zero_mean:
        margin = 5e-3
        for _ in range(8):
            Mass = np.sum(wts * phi0)
            if abs(Mass) <= 1e-14 * Wtot:
                break
            interior = np.abs(phi0) < (hi - margin)
            Wint = float(np.sum(wts[interior]))
            if Wint <= 0:
                phi0 -= Mass / Wtot
                phi0 = np.clip(phi0, lo, hi)
                break
            phi0[interior] -= Mass / Wint
    return phi0


# ----------------------------------------------------------------------------
# Newton step
# ----------------------------------------------------------------------------
def newton_step(phi_old, mu_old, w_old, w_new, dt, P: Params2D, hx, hy, L=None,
                return_history=False, stats=None):
    """One implicit time level (F2:323-427): Newton on [R_phi; R_mu] with initial
    guess (phi_old, mu_init(phi_old, w_new)) (F2:350-351), absolute stop
    ||R||_2 < 1e-6, step ceiling alpha_max = min(2, 0.9*min ratio) with
    alpha = min(1, alpha_max) (F2:377-391), Armijo (eta=1e-4, <=12 halvings) with
    best-trial fallback (F2:394-423)."""
    Nx, Ny = phi_old.shape[0] - 1, phi_old.shape[1] - 1
    if L is None:
        L = lap_matrix(Nx, Ny, hx, hy)
    phi_new = phi_old.copy()
    mu_new = mu_init(phi_old, w_new, P, hx, hy)
    n = phi_old.size
    hist = []
    nsolve = 0

    def resid(ph, mu):
        return np.concatenate([
            residual_phi(ph, phi_old, mu, mu_old, w_new, w_old, dt, P, hx, hy).ravel(),
            residual_mu(ph, phi_old, mu, mu_old, dt, hx, hy).ravel()])

    for _ in range(NEWTON_MAXIT):
        R = resid(phi_new, mu_new)
        nR = np.linalg.norm(R)
        hist.append(nR)
        if nR < NEWTON_TOL:
            break
        J = jac_matrix(phi_new, dt, P, L)
        try:
            delta = spsolve(J.tocsc(), -R)
        except Exception:                       # F2:371-372
            delta = spsolve((J + 1e-10 * sp.identity(2 * n, format="csr")).tocsc(), -R)
        nsolve += 1
        dphi, dmu = delta[:n], delta[n:]
        pf = phi_new.ravel()
        with np.errstate(divide="ignore", invalid="ignore"):
            amax = 2.0
            pos, neg = dphi > 0, dphi < 0
            if np.any(pos):
                amax = min(amax, 0.9 * np.min((1.0 - DELTA_SEP - pf[pos]) / dphi[pos]))
            if np.any(neg):
                amax = min(amax, 0.9 * np.min((-1.0 + DELTA_SEP - pf[neg]) / dphi[neg]))
        if not np.isfinite(amax) or amax <= 0:
            amax = 1
        alpha = min(1.0, amax)
        best = (np.inf, phi_new, mu_new)
        accepted = False
        for _t in range(ARMIJO_TRIALS):
            ph_t = phi_new + alpha * dphi.reshape(phi_new.shape)
            mu_t = mu_new + alpha * dmu.reshape(mu_new.shape)
            nRt = np.linalg.norm(resid(ph_t, mu_t))
            if nRt < best[0]:
                best = (nRt, ph_t, mu_t)
            if nRt <= (1.0 - ARMIJO_ETA * alpha) * nR:
                phi_new, mu_new = ph_t, mu_t
                accepted = True
                break
            alpha *= 0.5
        if not accepted and best[0] < nR:
            phi_new, mu_new = best[1], best[2]
    if stats is not None:
        stats["solves"] = stats.get("solves", 0) + nsolve
        stats["newton_its"] = stats.get("newton_its", 0) + len(hist)
    return (phi_new, mu_new, hist) if return_history else (phi_new, mu_new)


# ----------------------------------------------------------------------------
# forward march
# ----------------------------------------------------------------------------
def time_grid(T, dt, time_tol=1e-10):
    """The accumulated-time grid of F2:539-585: t += min(dt, T - t) while t < T - 1e-10;
    stored value min(t, T)."""
    t, ts, dts = 0.0, [0.0], []
    while t < T - time_tol:
        d = min(dt, T - t)
        dts.append(d)
        t += d
        ts.append(min(t, T))
    return np.array(ts), np.array(dts)


def forward(P: Params2D, control=None, phi0=None, seed=42, amp=0.1, max_steps=None, stats=None):
    """Time march (F2:489-596).  Returns (phi_hist (M+1,Nx+1,Ny+1), (x,y), t_hist).

    `phi0`/`seed`/`amp` generalise the hard-coded init_phi_random(amp=0.1, seed=42)
    of F2:517 (the reference's tests monkey-patch it instead).  `control` rows are
    used as (u[step], u[step+1]) while step < len(u)-1, else zeros (F2:545-548).
    mu and w are carried un-recomputed after the clip/mass fix (F2:579)."""
    Nx, Ny = int(P.Nx), int(P.Ny)
    hx, hy = P.Lx / Nx, P.Ly / Ny
    x = np.linspace(0.0, P.Lx, Nx + 1)
    y = np.linspace(0.0, P.Ly, Ny + 1)
    phi = init_phi_random(Nx, Ny, DELTA_SEP, amp=amp, seed=seed) if phi0 is None else phi0.copy()
    w = np.zeros_like(phi)
    L = lap_matrix(Nx, Ny, hx, hy)
    mu = mu_init(phi, w, P, hx, hy)
    if control is not None and (control.ndim != 3 or control.shape[1:] != phi.shape):
        raise ValueError(f"control_input must have shape (M, {Nx+1}, {Ny+1})")
    wts_h = hx * hy * np.outer(trapz_weights(Nx + 1), trapz_weights(Ny + 1))
    mass0 = np.sum(wts_h * phi)
    hist, ts = [phi.copy()], [0.0]
    t, step = 0.0, 0
    lo, hi = -1.0 + DELTA_SEP, 1.0 - DELTA_SEP
    while t < P.T - 1e-10:
        if max_steps is not None and step >= max_steps:
            break
        dts = min(P.dt_initial, P.T - t)
        if control is not None and step < control.shape[0] - 1:
            u_n, u_np1 = control[step], control[step + 1]
        else:
            u_n = u_np1 = np.zeros_like(phi)
        _t_step = _time.perf_counter()
        w_new = w_filter(w, dts, P.gamma, u_n, u_np1)
        phi_new, mu_new = newton_step(phi, mu, w, w_new, dts, P, hx, hy, L=L, stats=stats)
        phi = np.clip(phi_new, lo, hi)
        err = np.sum(wts_h * phi) - mass0
        if abs(err) > 1e-16:                                   # F2:567-577
            interior = np.abs(phi) < (1.0 - DELTA_SEP - 5e-3)
            Wint = float(np.sum(wts_h[interior]))
            if Wint > 0.0:
                phi[interior] -= err / Wint
            else:
                phi -= err / (P.Lx * P.Ly)
                phi = np.clip(phi, lo, hi)
        mu, w = mu_new, w_new
        t += dts
        step += 1
        hist.append(phi.copy())
        ts.append(min(t, P.T))
        if stats is not None:           # per-step wall time, operator assembly (once per march) excluded
            stats.setdefault("step_seconds", []).append(_time.perf_counter() - _t_step)
    return np.array(hist), (x, y), np.array(ts)


# ----------------------------------------------------------------------------
# adjoint sweep
# ----------------------------------------------------------------------------
def backward(phi_hist, x, y, t_hist, P: Params2D, b1, b2, phi_Q=None, phi_T=None,
             max_steps=None, step_seconds=None):
    """Adjoint sweep (B2:75-246): terminal (I - tau L) p_M = b2 (phi_M - phi_T), q = -L p,
    r_M = 0; per step A(phi_n) p_n = B(phi_{n+1}) p_{n+1} + src with the trapezoid
    source (B2:222), CN filter for r (B2:239-242); dt_n <= 1e-14 copies level n+1."""
    assert phi_hist.ndim == 3
    M1, nx1, ny1 = phi_hist.shape
    assert x.ndim == 1 and y.ndim == 1 and x.size >= 2 and y.size >= 2
    assert t_hist.ndim == 1 and t_hist.shape[0] == M1
    Nx, Ny = nx1 - 1, ny1 - 1
    hx, hy = float(x[1] - x[0]), float(y[1] - y[0])
    n = nx1 * ny1
    L = lap_matrix(Nx, Ny, hx, hy)
    LL = (L @ L).tocsr()
    I = sp.identity(n, format="csr")
    ph = phi_hist.reshape(M1, n)
    pq = np.zeros_like(ph) if phi_Q is None else phi_Q.reshape(M1, n)
    pt = np.zeros(n) if phi_T is None else phi_T.reshape(n)
    p = np.zeros((M1, n)); q = np.zeros((M1, n)); r = np.zeros((M1, n))
    p[-1] = spsolve((I - P.tau * L).tocsc(), b2 * (ph[-1] - pt))
    q[-1] = -(L @ p[-1])
    last = 0 if max_steps is None else max(0, M1 - 1 - max_steps)
    for k in range(M1 - 2, last - 1, -1):
        dt = float(t_hist[k + 1] - t_hist[k])
        if dt <= 1e-14:
            p[k], q[k], r[k] = p[k + 1], q[k + 1], r[k + 1]
            continue
        _t_step = _time.perf_counter()
        src = 0.5 * dt * b1 * ((ph[k] - pq[k]) + (ph[k + 1] - pq[k + 1]))
        Dn = sp.diags(fpp(ph[k], P.c1, P.c2), 0, format="csr")
        Dp = sp.diags(fpp(ph[k + 1], P.c1, P.c2), 0, format="csr")
        A = (I - P.tau * L + 0.5 * dt * LL - 0.5 * dt * (Dn @ L)).tocsc()
        Bm = (I - P.tau * L - 0.5 * dt * LL + 0.5 * dt * (Dp @ L)).tocsr()
        rhs = Bm @ p[k + 1] + src
        try:
            p[k] = spsolve(A, rhs)
        except Exception:
            p[k] = spsolve(A + 1e-10 * sp.identity(n, format="csc"), rhs)
        q[k] = -(L @ p[k])
        den = P.gamma + 0.5 * dt
        r[k] = ((P.gamma - 0.5 * dt) / den) * r[k + 1] + ((0.5 * dt) / den) * (q[k] + q[k + 1])
        if step_seconds is not None:    # per-step wall time (terminal solve and L, L@L assembly excluded)
            step_seconds.append(_time.perf_counter() - _t_step)
    shp = (M1, nx1, ny1)
    return p.reshape(shp), q.reshape(shp), r.reshape(shp)


# ----------------------------------------------------------------------------
# cost, gradient, prox, targets
# ----------------------------------------------------------------------------
def _trapz(f, x, axis=-1):
    """np.trapz arithmetic: sum(diff(x) * (f[1:] + f[:-1]) / 2) along axis."""
    f = np.moveaxis(np.asarray(f), axis, -1)
    d = np.diff(np.asarray(x))
    return np.sum(d * (f[..., 1:] + f[..., :-1]) / 2.0, axis=-1)


def cost_parts(phi_hist, u, phi_Q, phi_T, x, y, t_hist, O: OptParams):
    """J1..J4 by nested trapezoid in y, x, then t (C2:80-106)."""
    sp_int = lambda f: _trapz(_trapz(f, y, -1), x, -1)
    J1 = (O.b1 / 2.0) * _trapz(sp_int((phi_hist - phi_Q) ** 2), t_hist)
    J2 = (O.b2 / 2.0) * sp_int((phi_hist[-1] - phi_T) ** 2)
    J3 = (O.b3 / 2.0) * _trapz(sp_int(u ** 2), t_hist)
    J4 = O.kappa_sparsity * _trapz(sp_int(np.abs(u)), t_hist)
    return np.array([J1, J2, J3, J4])


def cost(phi_hist, u, phi_Q, phi_T, x, y, t_hist, O: OptParams):
    """C2:19-120 (without the prints)."""
    return float(np.sum(cost_parts(phi_hist, u, phi_Q, phi_T, x, y, t_hist, O)))


def gradient(r, u, O: OptParams):
    """r + b3 u (C2:150)."""
    return r + O.b3 * u


def prox_step(u, g, alpha, O: OptParams):
    """Gradient step, soft threshold alpha*kappa_s, box clip (C2:191-198)."""
    v = u - alpha * g
    s = np.sign(v) * np.maximum(np.abs(v) - alpha * O.kappa_sparsity, 0)
    return np.clip(s, O.u_min, O.u_max)


def build_targets(x, y, t_hist, phi_initial, Lx, Ly, T, choice_t=1, choice_q=1):
    """phi_T: 0.7 sin(2 pi x/Lx) cos(pi y/Ly) | centred disc of radius Lx/3.5;
    phi_Q: ramp (1-t/T) phi_0 + (t/T) phi_T with the CONFIG T | zeros (G2:184-226)."""
    xx, yy = np.meshgrid(x, y, indexing="ij")
    if choice_t == 1:
        phi_T = 0.7 * np.sin(2 * np.pi * xx / Lx) * np.cos(np.pi * yy / Ly)
    else:
        phi_T = -np.ones_like(xx)
        phi_T[(xx - Lx / 2) ** 2 + (yy - Ly / 2) ** 2 < (Lx / 3.5) ** 2] = 1.0
    if choice_q == 1:
        tp = (t_hist / T)[:, None, None]
        phi_Q = (1 - tp) * phi_initial + tp * phi_T
    else:
        phi_Q = np.zeros((len(t_hist), len(x), len(y)))
    return phi_T, phi_Q


def error_metrics(phi_hist, phi_Q, phi_T, x, y, t_hist):
    """Relative tracking / terminal errors of G2:336-363."""
    l2xy = lambda a: float(np.sqrt(max(_trapz(_trapz(a ** 2, y, 1), x), 0.0)))
    def l2xyt(a):
        s = np.array([l2xy(a[k]) ** 2 for k in range(a.shape[0])])
        return float(np.sqrt(max(_trapz(s, t_hist), 0.0)))
    area = float((x[-1] - x[0]) * (y[-1] - y[0]))
    tl = float(t_hist[-1] - t_hist[0])
    rms = float(np.sqrt(max(area, 1e-30) * max(tl, 1e-30)))
    den = l2xyt(phi_Q)
    if den < 1e-9 * rms:
        den = rms
    track = l2xyt(phi_hist - phi_Q) / (den + 1e-12)
    term = l2xy(phi_hist[-1] - phi_T) / (l2xy(phi_T) + 1e-12)
    return track, term


# ----------------------------------------------------------------------------
# PGD loop
# ----------------------------------------------------------------------------
@dataclass
class PGDResult:
    costs: list = field(default_factory=list)
    alphas: list = field(default_factory=list)
    attempts: list = field(default_factory=list)
    changes: list = field(default_factory=list)
    tracking: list = field(default_factory=list)
    terminal: list = field(default_factory=list)
    u: np.ndarray = None
    phi: np.ndarray = None
    r: np.ndarray = None
    converged: bool = False


def line_search(u_k, cost_k, g, phi_Q, phi_T, x, y, P, O, alpha_init, fwd, beta=0.8, max_ls=10):
    """G2:71-146: try alpha, alpha*beta, ...; accept the first cost < cost_k, else
    return the LAST try (alpha already multiplied by beta once more, G2:143-146)."""
    alpha = alpha_init
    u_n, phi_n, t_n, c_n, att = u_k, None, None, cost_k, 0
    for _ in range(max_ls):
        att += 1
        u_n = prox_step(u_k, g, alpha, O)
        phi_n, _, t_n = fwd(u_n)
        c_n = cost(phi_n, u_n, phi_Q, phi_T, x, y, t_n, O)
        if c_n < cost_k:
            return alpha, u_n, c_n, phi_n, t_n, att
        alpha *= beta
    return alpha, u_n, c_n, phi_n, t_n, att


def pgd(P: Params2D, O: OptParams, n_iter=None, seed=42, amp=0.1, choice_t=1, choice_q=1, zero_target_q=False):
    """The loop of G2:291-382 (optimistic step with alpha_prev, backtracking from
    0.8*alpha_prev, alpha growth 1.2 / plateau 1.5 after 5 its |dJ|<1e-5, stop when the
    relative control change < 1e-5 and k > 20)."""
    fwd = lambda u: forward(P, control=u, seed=seed, amp=amp)
    phi_k, (x, y), t_k = fwd(None)
    u_k = np.zeros_like(phi_k)
    phi_T, phi_Q = build_targets(x, y, t_k, phi_k[0].copy(), P.Lx, P.Ly, P.T, choice_t, choice_q)
    if zero_target_q:                  # tests: the RMS fallback of the tracking error (G2:353-354)
        phi_Q = np.zeros_like(phi_Q)
    cost_k = cost(phi_k, u_k, phi_Q, phi_T, x, y, t_k, O)
    res = PGDResult(costs=[cost_k])
    alpha_prev, plateau = O.alpha_max, 0
    r_k = None
    for k in range(O.max_iter if n_iter is None else n_iter):
        _, _, r_k = backward(phi_k, x, y, t_k, P, O.b1, O.b2, phi_Q, phi_T)
        g = gradient(r_k, u_k, O)
        u_o = prox_step(u_k, g, alpha_prev, O)
        phi_o, _, t_o = fwd(u_o)
        c_o = cost(phi_o, u_o, phi_Q, phi_T, x, y, t_o, O)
        if c_o < cost_k:
            a_k, u_n, c_n, phi_n, t_n, att = alpha_prev, u_o, c_o, phi_o, t_o, 0
        else:
            a_k, u_n, c_n, phi_n, t_n, att = line_search(u_k, cost_k, g, phi_Q, phi_T, x, y, P, O,
                                                         alpha_prev * 0.8, fwd)
        res.costs.append(c_n); res.alphas.append(a_k); res.attempts.append(att)
        e1, e2 = error_metrics(phi_n, phi_Q, phi_T, x, y, t_k)
        res.tracking.append(e1); res.terminal.append(e2)
        if k > 0 and abs(res.costs[-1] - res.costs[-2]) < 1e-5:
            plateau += 1
        else:
            plateau = 0
        if plateau >= 5:
            alpha_prev, plateau = min(O.alpha_max, a_k * 1.5), 0
        else:
            alpha_prev = min(O.alpha_max, a_k * 1.2)
        change = np.linalg.norm(u_n - u_k) / (np.linalg.norm(u_k) + 1e-9)
        res.changes.append(change)
        if change < 1e-5 and k > 20:
            u_k, phi_k = u_n, phi_n
            res.converged = True
            break
        u_k, cost_k, phi_k, t_k = u_n, c_n, phi_n, t_n
    res.u, res.phi, res.r = u_k, phi_k, r_k
    res.phi_T, res.phi_Q, res.t_hist, res.x, res.y = phi_T, phi_Q, t_k, x, y
    return res


# ----------------------------------------------------------------------------
# timing helper for bench.py's cpu_baseline leg
# ----------------------------------------------------------------------------
def time_one_step(N, dt, seed=42, amp=0.1, n_fwd_steps=3, n_bwd_steps=3):
    """Wall times of `n_fwd_steps` forward time steps and `n_bwd_steps` adjoint steps at full spatial size N x N
    with the default parameters, one sample per step (the per-step cost is flat in the step index);
    the once-per-march operator assembly (L, L@L) and the terminal adjoint solve are outside the samples."""
    P = Params2D(Nx=N, Ny=N, T=dt * max(n_fwd_steps, n_bwd_steps), dt_initial=dt)
    O = OptParams()
    st = {}
    phi, (x, y), t = forward(P, seed=seed, amp=amp, max_steps=n_fwd_steps, stats=st)
    phi_T, phi_Q = build_targets(x, y, t, phi[0], P.Lx, P.Ly, P.T)
    bsec = []
    backward(phi, x, y, t, P, O.b1, O.b2, phi_Q, phi_T, max_steps=n_bwd_steps, step_seconds=bsec)
    fsec = list(st.get("step_seconds", []))
    return dict(fwd_step_seconds=fsec, bwd_step_seconds=bsec, fwd_s_per_step=float(np.mean(fsec)),
                bwd_s_per_step=float(np.mean(bsec)), fwd_steps=len(fsec), bwd_steps=len(bsec),
                solves=st.get("solves", 0))
