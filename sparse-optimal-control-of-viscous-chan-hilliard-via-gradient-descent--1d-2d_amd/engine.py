"""Thin numpy-facing wrapper of the C ABI (include/vch.h): one object per GPU context.

No arithmetic happens here: every method marshals numpy buffers into one engine call.
Batched arguments have a leading axis B (the context's batch); with B == 1 the leading
axis may be omitted and is then omitted from the results as well.
"""
from __future__ import annotations

import ctypes as C
import numpy as np

from . import _lib
from ._lib import OptParams, Params1D, Params2D, Stats, VchError, check


def _dp(a):
    return None if a is None else a.ctypes.data_as(_lib._D)


def time_grid(T: float, dt: float, time_tol: float = 1e-10):
    """The reference's accumulated-time rule (Forward2_solver.py:539-585, Forward_solver.py:
    326-373): `t += min(dt, T - t)` while `t < T - 1e-10`, stored as `min(t, T)`.
    Returns (t_hist with a leading 0, per-step dt)."""
    t, ts, dts = 0.0, [0.0], []
    while t < T - time_tol:
        d = min(dt, T - t)
        dts.append(d)
        t += d
        ts.append(min(t, T))
    return np.array(ts), np.array(dts)


def make_opt(opt=None, **kw) -> OptParams:
    """OptParams from an object with the reference's OptimizationConfig field names."""
    names = [f for f, _ in OptParams._fields_]
    defaults = dict(b1=5.0, b2=10.0, b3=1e-4, kappa_sparsity=1e-4, alpha_max=50.0, max_iter=500,
                    u_min=-1.0, u_max=1.0)
    vals = dict(defaults)
    if opt is not None:
        for n in names:
            if hasattr(opt, n):
                vals[n] = getattr(opt, n)
    vals.update(kw)
    o = OptParams()
    for n in names:
        setattr(o, n, int(vals[n]) if n == "max_iter" else float(vals[n]))
    return o


class Engine2D:
    """One GPU context for `batch` trajectories on an (Nx+1) x (Ny+1) grid."""

    def __init__(self, Nx, Ny, Lx=1.0, Ly=1.0, tau=0.05, gamma=10.0, c1=0.75, c2=1.0, kappa=1e-4,
                 batch=1, max_steps=128, device=0):
        self.lib = _lib.load()
        n = self.lib.vch_device_count()
        if n <= 0:
            raise VchError("no HIP device visible: the engine has no CPU path")
        self.p = Params2D(int(Nx), int(Ny), float(Lx), float(Ly), float(tau), float(gamma), float(c1),
                          float(c2), float(kappa))
        self.B, self.max_steps, self.device = int(batch), int(max_steps), int(device)
        self.Nx, self.Ny = int(Nx), int(Ny)
        self.shape = (self.Nx + 1, self.Ny + 1)
        self.ctx = self.lib.vch2d_create(C.byref(self.p), self.B, self.max_steps, self.device)
        if not self.ctx:
            raise ValueError("vch2d_create failed: " + _lib.last_error())
        self.uses_fft = bool(self.lib.vch2d_uses_fft(self.ctx))
        self.x = np.linspace(0.0, float(Lx), self.Nx + 1)
        self.y = np.linspace(0.0, float(Ly), self.Ny + 1)

    @classmethod
    def from_config(cls, cfg, **kw):
        return cls(cfg.Nx, cfg.Ny, cfg.Lx, cfg.Ly, cfg.tau, cfg.gamma, cfg.c1, cfg.c2, cfg.kappa, **kw)

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.vch2d_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- marshalling ----------------------------------------------------------------
    def _fld(self, a, name="field"):
        """(B, Nx+1, Ny+1) float64 C-contiguous view/copy; ValueError on a wrong shape
        (Forward2_solver.py:148-149)."""
        a = np.ascontiguousarray(a, dtype=np.float64)
        if a.shape == self.shape and self.B == 1:
            a = a.reshape((1,) + self.shape)
        if a.shape != (self.B,) + self.shape:
            raise ValueError(f"{name} must have shape ({self.Nx + 1}, {self.Ny + 1}) "
                             f"[batch {self.B}], got {a.shape}")
        return a

    def _hist(self, a, rows, name="history"):
        a = np.ascontiguousarray(a, dtype=np.float64)
        if a.ndim == 3 and self.B == 1:
            a = a.reshape((1,) + a.shape)
        if a.ndim != 4 or a.shape[0] != self.B or a.shape[2:] != self.shape or (rows and a.shape[1] != rows):
            raise ValueError(f"{name} must have shape (M, {self.Nx + 1}, {self.Ny + 1}), got {a.shape}")
        return a

    def _out(self, squeeze=True, rows=None):
        shp = (self.B,) + ((rows,) if rows else ()) + self.shape
        return np.empty(shp, dtype=np.float64)

    def _sq(self, a):
        return a[0] if self.B == 1 else a

    # -- operators --------------------------------------------------------------------
    def apply_laplacian(self, v):
        v = self._fld(v)
        out = self._out()
        check(self.lib.vch2d_apply_laplacian(self.ctx, _dp(v), _dp(out)))
        return self._sq(out)

    def initialize_mu(self, phi, w):
        phi, w = self._fld(phi), self._fld(w)
        out = self._out()
        check(self.lib.vch2d_initialize_mu(self.ctx, _dp(phi), _dp(w), _dp(out)))
        return self._sq(out)

    def solve_w(self, w_old, dt, u_n=None, u_np1=None):
        w_old = self._fld(w_old)
        u_n = None if u_n is None else self._fld(u_n)
        u_np1 = None if u_np1 is None else self._fld(u_np1)
        out = self._out()
        check(self.lib.vch2d_solve_w(self.ctx, _dp(w_old), float(dt), _dp(u_n), _dp(u_np1), _dp(out)))
        return self._sq(out)

    def residuals(self, phi_new, phi_old, mu_new, mu_old, w_new, w_old, dt):
        a = [self._fld(v) for v in (phi_new, phi_old, mu_new, mu_old, w_new, w_old)]
        Rp, Rm = self._out(), self._out()
        nrm = np.empty(self.B)
        check(self.lib.vch2d_residuals(self.ctx, *[_dp(v) for v in a], float(dt), _dp(Rp), _dp(Rm), _dp(nrm)))
        return self._sq(Rp), self._sq(Rm), (nrm[0] if self.B == 1 else nrm)

    def jacobian_apply(self, phi_new, dt, dphi, dmu):
        a = [self._fld(v) for v in (phi_new, dphi, dmu)]
        o1, o2 = self._out(), self._out()
        check(self.lib.vch2d_jacobian_apply(self.ctx, _dp(a[0]), float(dt), _dp(a[1]), _dp(a[2]), _dp(o1), _dp(o2)))
        return self._sq(o1), self._sq(o2)

    def jacobian_solve(self, phi_new, dt, rhs_phi, rhs_mu):
        a = [self._fld(v) for v in (phi_new, rhs_phi, rhs_mu)]
        o1, o2 = self._out(), self._out()
        st = Stats()
        check(self.lib.vch2d_jacobian_solve(self.ctx, _dp(a[0]), float(dt), _dp(a[1]), _dp(a[2]), _dp(o1), _dp(o2),
                                            C.byref(st)))
        return self._sq(o1), self._sq(o2), st.as_dict()

    def schur_apply(self, phi_new, dt, x):
        a = [self._fld(v) for v in (phi_new, x)]
        out = self._out()
        check(self.lib.vch2d_schur_apply(self.ctx, _dp(a[0]), float(dt), _dp(a[1]), _dp(out)))
        return self._sq(out)

    def adjoint_apply(self, which, phi, dt, v):
        a = [self._fld(z) for z in (phi, v)]
        out = self._out()
        check(self.lib.vch2d_adjoint_apply(self.ctx, {"A": 0, "B": 1}[which], _dp(a[0]), float(dt), _dp(a[1]), _dp(out)))
        return self._sq(out)

    def adjoint_solve(self, phi_n, dt, rhs):
        phi_n = None if phi_n is None else self._fld(phi_n)
        rhs = self._fld(rhs)
        out = self._out()
        st = Stats()
        check(self.lib.vch2d_adjoint_solve(self.ctx, _dp(phi_n), float(dt), _dp(rhs), _dp(out), C.byref(st)))
        return self._sq(out), st.as_dict()

    def spectral_solve(self, c0, c1, c2, v):
        v = self._fld(v)
        out = self._out()
        check(self.lib.vch2d_spectral_solve(self.ctx, float(c0), float(c1), float(c2), _dp(v), _dp(out)))
        return self._sq(out)

    # -- Newton / march / sweep --------------------------------------------------------
    def newton_raphson(self, phi_old, mu_old, w_old, w_new, dt, hist_cap=512):
        a = [self._fld(v) for v in (phi_old, mu_old, w_old, w_new)]
        pn, mn = self._out(), self._out()
        hist = np.zeros((self.B, hist_cap))
        nh = np.zeros(self.B, dtype=np.int32)
        st = Stats()
        check(self.lib.vch2d_newton_raphson(self.ctx, *[_dp(v) for v in a], float(dt), _dp(pn), _dp(mn), _dp(hist),
                                            int(hist_cap), nh.ctypes.data_as(_lib._I32), C.byref(st)))
        hists = [hist[b, :nh[b]].copy() for b in range(self.B)]
        return self._sq(pn), self._sq(mn), (hists[0] if self.B == 1 else hists), st.as_dict()

    def forward(self, phi0, dt, u=None, store=True):
        """March len(dt) steps.  u: (B, rows, ...) control, None, or "resident".
        Returns (phi_hist or None, stats)."""
        phi0 = self._fld(phi0, "phi0")
        dt = np.ascontiguousarray(dt, dtype=np.float64)
        M = int(dt.size)
        rows = 0
        if isinstance(u, str) and u == "resident":
            up = C.cast(C.c_void_p(1), _lib._D)
        elif u is None:
            up = None
        else:
            if u.ndim == 3 and self.B == 1:
                u = u.reshape((1,) + u.shape)
            if u.ndim != 4 or u.shape[0] != self.B or u.shape[2:] != self.shape:
                raise ValueError(f"control_input must have shape (M, {self.Nx + 1}, {self.Ny + 1})")
            u = np.ascontiguousarray(u, dtype=np.float64)
            rows = int(u.shape[1])
            up = _dp(u)
        out = self._out(rows=M + 1) if store else None
        st = Stats()
        check(self.lib.vch2d_forward(self.ctx, _dp(phi0), up, rows, _dp(dt), M, _dp(out), C.byref(st)))
        return (self._sq(out) if store else None), st.as_dict()

    def backward(self, phi_hist, t_hist, b1, b2, phi_Q=None, phi_T=None, hx=None, hy=None,
                 want=("p", "q", "r")):
        """Adjoint sweep; phi_hist None = resident history of the last forward call."""
        t_hist = np.ascontiguousarray(t_hist, dtype=np.float64)
        M = int(t_hist.size) - 1
        ph = None if phi_hist is None else self._hist(phi_hist, M + 1, "phi_hist")
        pq = None if phi_Q is None else self._hist(phi_Q, M + 1, "phi_Q")
        pt = None if phi_T is None else self._fld(phi_T, "phi_T_target")
        outs = {k: (self._out(rows=M + 1) if k in want else None) for k in ("p", "q", "r")}
        st = Stats()
        hx = float(self.x[1] - self.x[0]) if hx is None else float(hx)
        hy = float(self.y[1] - self.y[0]) if hy is None else float(hy)
        check(self.lib.vch2d_backward(self.ctx, _dp(ph), M, _dp(t_hist), hx, hy, float(b1), float(b2), _dp(pq),
                                      _dp(pt), _dp(outs["p"]), _dp(outs["q"]), _dp(outs["r"]), C.byref(st)))
        return tuple(None if outs[k] is None else self._sq(outs[k]) for k in ("p", "q", "r")) + (st.as_dict(),)

    def cost(self, phi_hist, u, phi_Q, phi_T, t_hist, opt, x=None, y=None):
        t_hist = np.ascontiguousarray(t_hist, dtype=np.float64)
        M = int(t_hist.size) - 1
        ph = None if phi_hist is None else self._hist(phi_hist, M + 1, "phi_hist")
        uu = None if u is None else self._hist(u, M + 1, "u")
        pq = None if phi_Q is None else self._hist(phi_Q, M + 1, "phi_Q_target")
        pt = None if phi_T is None else self._fld(phi_T, "phi_T_target")
        x = np.ascontiguousarray(self.x if x is None else x, dtype=np.float64)
        y = np.ascontiguousarray(self.y if y is None else y, dtype=np.float64)
        if x.shape != (self.Nx + 1,) or y.shape != (self.Ny + 1,):
            raise ValueError("x, y must have Nx+1, Ny+1 entries")
        J = np.empty((self.B, 5))
        o = opt if isinstance(opt, OptParams) else make_opt(opt)
        check(self.lib.vch2d_cost(self.ctx, _dp(ph), _dp(uu), _dp(pq), _dp(pt), M, _dp(x), _dp(y), _dp(t_hist),
                                  C.byref(o), _dp(J)))
        return J[0] if self.B == 1 else J

    def grad_prox(self, u, r, alpha, opt):
        u = self._hist(u, 0, "u")
        r = self._hist(r, u.shape[1], "r")
        al = np.ascontiguousarray(np.broadcast_to(np.asarray(alpha, dtype=np.float64), (self.B,)))
        out = np.empty_like(u)
        o = opt if isinstance(opt, OptParams) else make_opt(opt)
        check(self.lib.vch2d_grad_prox(self.ctx, _dp(u), _dp(r), int(u.shape[1]), _dp(al), C.byref(o), _dp(out)))
        return self._sq(out)

    # -- device-resident PGD -------------------------------------------------------------
    def pgd_init(self, phi0, phi_T, t_hist, opt, phi_Q=None, ramp=True, T=None, x=None, y=None):
        phi0, phi_T = self._fld(phi0, "phi0"), self._fld(phi_T, "phi_T_target")
        t_hist = np.ascontiguousarray(t_hist, dtype=np.float64)
        M = int(t_hist.size) - 1
        pq = None if phi_Q is None else self._hist(phi_Q, M + 1, "phi_Q_target")
        x = np.ascontiguousarray(self.x if x is None else x, dtype=np.float64)
        y = np.ascontiguousarray(self.y if y is None else y, dtype=np.float64)
        o = opt if isinstance(opt, OptParams) else make_opt(opt)
        J0 = np.empty((self.B, 5))
        check(self.lib.vch2d_pgd_init(self.ctx, _dp(phi0), _dp(phi_T), _dp(pq), int(bool(ramp)),
                                      float(t_hist[-1] if T is None else T), _dp(t_hist), M, _dp(x), _dp(y),
                                      C.byref(o), _dp(J0)))
        self._pgd_M = M
        return J0

    def pgd_iterate(self, n_iters):
        n = int(n_iters)
        cost = np.full((self.B, n), np.nan)
        alpha = np.full((self.B, n), np.nan)
        att = np.zeros((self.B, n), dtype=np.int32)
        chg = np.full((self.B, n), np.nan)
        sec = np.zeros(5)
        done = check(self.lib.vch2d_pgd_iterate(self.ctx, n, _dp(cost), _dp(alpha), att.ctypes.data_as(_lib._I32),
                                                _dp(chg), _dp(sec)))
        trk, trm = np.full((self.B, n), np.nan), np.full((self.B, n), np.nan)
        check(self.lib.vch2d_pgd_errors(self.ctx, n, _dp(trk), _dp(trm)))
        return dict(iters=done, cost=cost, alpha=alpha, attempts=att, change=chg, tracking_error=trk, terminal_error=trm,
                    seconds=dict(zip(("backward", "gradprox", "optimistic_forward", "cost", "backtracking"), sec)))

    def pgd_get(self, what):
        out = self._out(rows=self._pgd_M + 1)
        check(self.lib.vch2d_pgd_get(self.ctx, {"u": 0, "phi": 1, "r": 2, "phi_Q": 3}[what], _dp(out)))
        return self._sq(out)

    def pgd_cost_device_ptr(self):
        p = C.c_void_p()
        check(self.lib.vch2d_pgd_cost_dev(self.ctx, C.byref(p)))
        return p.value

    def free_energy(self, phi_hist, hx=None, hy=None, w_hist=None, eps=None):
        """Free energy of every level (F2:256-319): phi_hist (B, rows, Nx+1, Ny+1) or "resident".
        Returns (B, rows) (or (rows,) for one trajectory)."""
        hx = float(self.x[1] - self.x[0]) if hx is None else float(hx)
        hy = float(self.y[1] - self.y[0]) if hy is None else float(hy)
        if isinstance(phi_hist, str) and phi_hist == "resident":
            if w_hist is None:
                raise ValueError("free_energy('resident') needs the number of levels: pass w_hist or use free_energy_resident(rows)")
            ph, rows = C.cast(C.c_void_p(1), _lib._D), None
        else:
            ph = self._hist(phi_hist, 0, "phi_hist")
            rows = int(ph.shape[1])
        wh = None if w_hist is None else self._hist(w_hist, rows or 0, "w_hist")
        if rows is None:
            rows = int(wh.shape[1])
        E = np.empty((self.B, rows))
        check(self.lib.vch2d_free_energy(self.ctx, ph if not isinstance(ph, np.ndarray) else _dp(ph), rows, _dp(wh), hx, hy,
                                         float(eps or 0.0), _dp(E)))
        return self._sq(E)

    def free_energy_resident(self, rows, hx=None, hy=None, eps=None):
        """Free energy of the first `rows` levels of the state history left by the last march."""
        hx = float(self.x[1] - self.x[0]) if hx is None else float(hx)
        hy = float(self.y[1] - self.y[0]) if hy is None else float(hy)
        E = np.empty((self.B, int(rows)))
        check(self.lib.vch2d_free_energy(self.ctx, C.cast(C.c_void_p(1), _lib._D), int(rows), None, hx, hy, float(eps or 0.0), _dp(E)))
        return self._sq(E)

    # -- in-situ kernel timing -------------------------------------------------------------
    PROF_CLASSES = ("schur_p", "dct_gemm", "residual", "adj_q", "cg_update", "adj_rhs", "cost", "prox", "dct_rows_fwd",
                    "dct_cols", "dct_rows_inv", "schur_p_first", "cg_rows_fwd", "cg_rows_fwd_first", "event_pair_noop", "guess",
                    "adj_guess", "cheb_rows", "cheb_rows_first", "residual_first")

    def counters(self):
        """(kernel launches, blocking looks of the host at the device state) since the context was created."""
        out = np.zeros(2, dtype=np.int64)
        check(self.lib.vch2d_counters(self.ctx, out.ctypes.data_as(C.POINTER(C.c_int64))))
        return int(out[0]), int(out[1])

    def prof_begin(self, max_launches=200000):
        check(self.lib.vch2d_prof_begin(self.ctx, int(max_launches)))

    def prof_end(self):
        n = len(self.PROF_CLASSES)
        ms = np.zeros(n)
        cnt = np.zeros(n, dtype=np.int64)
        check(self.lib.vch2d_prof_end(self.ctx, _dp(ms), cnt.ctypes.data_as(C.POINTER(C.c_int64)), n))
        return {k: dict(ms=float(ms[i]), launches=int(cnt[i])) for i, k in enumerate(self.PROF_CLASSES)}

    def prof_spans(self):
        """{class: float32 array of the event-pair spans (us) of every launch recorded between prof_begin and prof_end}."""
        cap = 1 << 20
        cls = np.zeros(cap, dtype=np.int32)
        ms = np.zeros(cap, dtype=np.float32)
        n = check(self.lib.vch2d_prof_spans(self.ctx, cls.ctypes.data_as(C.POINTER(C.c_int32)),
                                            ms.ctypes.data_as(C.POINTER(C.c_float)), cap))
        n = min(int(n), cap)
        return {k: 1e3 * ms[:n][cls[:n] == i] for i, k in enumerate(self.PROF_CLASSES)}


class Engine1D:
    """One GPU context for `batch` 1D trajectories on N+1 nodes (N <= 4096)."""

    def __init__(self, N, Lx=1.0, tau=0.05, gamma=10.0, c1=0.75, c2=1.0, kappa=0.03 ** 2, batch=1,
                 max_steps=128, device=0):
        self.lib = _lib.load()
        if self.lib.vch_device_count() <= 0:
            raise VchError("no HIP device visible: the engine has no CPU path")
        self.p = Params1D(int(N), float(Lx), float(tau), float(gamma), float(c1), float(c2), float(kappa))
        self.B, self.max_steps, self.N, self.n = int(batch), int(max_steps), int(N), int(N) + 1
        self.ctx = self.lib.vch1d_create(C.byref(self.p), self.B, self.max_steps, int(device))
        if not self.ctx:
            raise ValueError("vch1d_create failed: " + _lib.last_error())
        self.x = np.linspace(0, float(Lx), self.n)

    @classmethod
    def from_config(cls, cfg, **kw):
        return cls(cfg.N, cfg.Lx, cfg.tau, cfg.gamma, cfg.c1, cfg.c2, cfg.kappa, **kw)

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.vch1d_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _fld(self, a, name="field"):
        a = np.ascontiguousarray(a, dtype=np.float64)
        if a.shape == (self.n,) and self.B == 1:
            a = a.reshape(1, self.n)
        if a.shape != (self.B, self.n):
            raise ValueError(f"{name} must have shape ({self.n},) [batch {self.B}], got {a.shape}")
        return a

    def _hist(self, a, rows=0, name="history"):
        a = np.ascontiguousarray(a, dtype=np.float64)
        if a.ndim == 2 and self.B == 1:
            a = a.reshape((1,) + a.shape)
        if a.ndim != 3 or a.shape[0] != self.B or a.shape[2] != self.n or (rows and a.shape[1] != rows):
            raise ValueError(f"{name} must have shape (rows, {self.n}), got {a.shape}")
        return a

    def _sq(self, a):
        return a[0] if self.B == 1 else a

    def apply_laplacian(self, v):
        v = self._fld(v)
        out = np.empty_like(v)
        check(self.lib.vch1d_apply_laplacian(self.ctx, _dp(v), _dp(out)))
        return self._sq(out)

    def residuals(self, phi_new, phi_old, mu_new, mu_old, w_new, w_old, dt):
        a = [self._fld(v) for v in (phi_new, phi_old, mu_new, mu_old, w_new, w_old)]
        Rp, Rm = np.empty_like(a[0]), np.empty_like(a[0])
        check(self.lib.vch1d_residuals(self.ctx, *[_dp(v) for v in a], float(dt), _dp(Rp), _dp(Rm)))
        return self._sq(Rp), self._sq(Rm)

    def jacobian_solve(self, phi_new, dt, rhs_phi, rhs_mu):
        a = [self._fld(v) for v in (phi_new, rhs_phi, rhs_mu)]
        o1, o2 = np.empty_like(a[0]), np.empty_like(a[0])
        check(self.lib.vch1d_jacobian_solve(self.ctx, _dp(a[0]), float(dt), _dp(a[1]), _dp(a[2]), _dp(o1), _dp(o2)))
        return self._sq(o1), self._sq(o2)

    def adjoint_solve(self, phi_n, dt, rhs):
        phi_n = None if phi_n is None else self._fld(phi_n)
        rhs = self._fld(rhs)
        out = np.empty_like(rhs)
        check(self.lib.vch1d_adjoint_solve(self.ctx, _dp(phi_n), float(dt), _dp(rhs), _dp(out)))
        return self._sq(out)

    def newton_raphson(self, phi_old, mu_old, w_old, w_new, dt, hist_cap=64):
        a = [self._fld(v) for v in (phi_old, mu_old, w_old, w_new)]
        pn, mn = np.empty_like(a[0]), np.empty_like(a[0])
        hist = np.zeros((self.B, hist_cap))
        nh = np.zeros(self.B, dtype=np.int32)
        check(self.lib.vch1d_newton_raphson(self.ctx, *[_dp(v) for v in a], float(dt), _dp(pn), _dp(mn), _dp(hist),
                                            int(hist_cap), nh.ctypes.data_as(_lib._I32)))
        hists = [hist[b, :nh[b]].copy() for b in range(self.B)]
        return self._sq(pn), self._sq(mn), (hists[0] if self.B == 1 else hists)

    def forward(self, phi0, dt, u=None, store=True):
        """Returns (phi_hist with M+2 rows or None, stats)."""
        phi0 = self._fld(phi0, "phi0")
        dt = np.ascontiguousarray(dt, dtype=np.float64)
        M = int(dt.size)
        rows = 0
        if u is not None:
            u = self._hist(u, 0, "control_input")
            rows = int(u.shape[1])
            if rows < M:          # the reference indexes control_input[step] for every step (F1:347-353)
                raise IndexError(f"index {rows} is out of bounds for axis 0 with size {rows}")
            if rows > M + 2:
                u = np.ascontiguousarray(u[:, :M + 2])
                rows = M + 2
        out = np.empty((self.B, M + 2, self.n)) if store else None
        st = Stats()
        check(self.lib.vch1d_forward(self.ctx, _dp(phi0), _dp(u), rows, _dp(dt), M, _dp(out), C.byref(st)))
        return (self._sq(out) if store else None), st.as_dict()

    def backward(self, phi_hist, t_hist, b1, b2, phi_Q=None, phi_T=None, h=None):
        t_hist = np.ascontiguousarray(t_hist, dtype=np.float64)
        rows = int(t_hist.size)
        ph = None if phi_hist is None else self._hist(phi_hist, rows, "phi_hist")
        pq = None if phi_Q is None else self._hist(phi_Q, rows, "phi_Q")
        pt = None if phi_T is None else self._fld(phi_T, "phi_T_target")
        p, q, r = (np.empty((self.B, rows, self.n)) for _ in range(3))
        h = float(self.x[1] - self.x[0]) if h is None else float(h)
        check(self.lib.vch1d_backward(self.ctx, _dp(ph), rows, _dp(t_hist), h, float(b1), float(b2), _dp(pq), _dp(pt),
                                      _dp(p), _dp(q), _dp(r)))
        return self._sq(p), self._sq(q), self._sq(r)

    def cost(self, phi_hist, u, phi_Q, phi_T, x, t_hist, opt):
        t_hist = np.ascontiguousarray(t_hist, dtype=np.float64)
        rows = int(t_hist.size)
        ph = self._hist(phi_hist, rows, "phi_hist")
        uu = None if u is None else self._hist(u, rows, "u")
        pq = None if phi_Q is None else self._hist(phi_Q, rows, "phi_Q_target")
        pt = None if phi_T is None else self._fld(phi_T, "phi_T_target")
        x = np.ascontiguousarray(x, dtype=np.float64)
        J = np.empty((self.B, 5))
        o = opt if isinstance(opt, OptParams) else make_opt(opt)
        check(self.lib.vch1d_cost(self.ctx, _dp(ph), _dp(uu), _dp(pq), _dp(pt), rows, _dp(x), _dp(t_hist), C.byref(o), _dp(J)))
        return J[0] if self.B == 1 else J

    def grad_prox(self, u, r, alpha, opt):
        u = self._hist(u, 0, "u")
        r = self._hist(r, u.shape[1], "r")
        al = np.ascontiguousarray(np.broadcast_to(np.asarray(alpha, dtype=np.float64), (self.B,)))
        out = np.empty_like(u)
        o = opt if isinstance(opt, OptParams) else make_opt(opt)
        check(self.lib.vch1d_grad_prox(self.ctx, _dp(u), _dp(r), int(u.shape[1]), _dp(al), C.byref(o), _dp(out)))
        return self._sq(out)

    def free_energy(self, phi_hist, h=None, w_hist=None, eps=None):
        """Free energy of every level (F1:243-262): phi_hist (B, rows, N+1) -> (B, rows)."""
        ph = self._hist(phi_hist, 0, "phi_hist")
        rows = int(ph.shape[1])
        wh = None if w_hist is None else self._hist(w_hist, rows, "w_hist")
        h = float(self.x[1] - self.x[0]) if h is None else float(h)
        E = np.empty((self.B, rows))
        check(self.lib.vch1d_free_energy(self.ctx, _dp(ph), rows, _dp(wh), h, float(eps or 0.0), _dp(E)))
        return self._sq(E)

    # -- device-resident PGD (G1:333-477) ----------------------------------------------------
    def pgd_init(self, phi0, phi_T, t_hist, dt, opt, phi_Q=None, x=None):
        """Uncontrolled march + targets + J0; t_hist has M+2 entries, dt M.  Returns J0 (B, 5)."""
        phi0, phi_T = self._fld(phi0, "phi0"), self._fld(phi_T, "phi_T_target")
        t_hist = np.ascontiguousarray(t_hist, dtype=np.float64)
        dt = np.ascontiguousarray(dt, dtype=np.float64)
        rows = int(t_hist.size)
        if dt.size != rows - 2:
            raise ValueError(f"dt must have len(t_hist) - 2 = {rows - 2} entries, got {dt.size}")
        pq = None if phi_Q is None else self._hist(phi_Q, rows, "phi_Q_target")
        x = np.ascontiguousarray(self.x if x is None else x, dtype=np.float64)
        o = opt if isinstance(opt, OptParams) else make_opt(opt)
        J0 = np.empty((self.B, 5))
        check(self.lib.vch1d_pgd_init(self.ctx, _dp(phi0), _dp(phi_T), _dp(pq), _dp(x), _dp(t_hist), rows, _dp(dt),
                                      C.byref(o), _dp(J0)))
        self._pgd_rows = rows
        return J0

    def pgd_iterate(self, n_iters):
        n = int(n_iters)
        cost = np.full((self.B, n), np.nan)
        alpha = np.full((self.B, n), np.nan)
        trials = np.zeros((self.B, n), dtype=np.int32)
        chg = np.full((self.B, n), np.nan)
        sec = np.zeros(3)
        done = check(self.lib.vch1d_pgd_iterate(self.ctx, n, _dp(cost), _dp(alpha), trials.ctypes.data_as(_lib._I32),
                                                _dp(chg), _dp(sec)))
        trk, trm = np.full((self.B, n), np.nan), np.full((self.B, n), np.nan)
        check(self.lib.vch1d_pgd_errors(self.ctx, n, _dp(trk), _dp(trm)))
        return dict(iters=done, cost=cost, alpha=alpha, trials=trials, change=chg, tracking_error=trk, terminal_error=trm,
                    seconds=dict(zip(("backward", "optimistic", "backtracking"), sec)))

    def pgd_get(self, what):
        out = np.empty((self.B, self._pgd_rows, self.n))
        check(self.lib.vch1d_pgd_get(self.ctx, {"u": 0, "phi": 1, "r": 2, "phi_Q": 3}[what], _dp(out)))
        return self._sq(out)
