"""Host-side mirror of the reference's 2D forward module (src/2D/Vch_control_2D/
Forward2_solver.py): same function names, positional orders, return tuples, array shapes and
error behaviour, with every field operation executed by the HIP engine through the C ABI.

Differences that are deliberate and documented:
  * `laplacian_matrix_neumann` returns a matrix-free operator handle (supports `L @ v`), not a
    scipy CSR matrix: nothing is assembled on the GPU path.
  * `run_main_simulation` accepts the extra keyword arguments `initial_phi`, `seed`, `amp`
    (the reference hard-codes init_phi_random(amp=0.1, seed=42), F2:517, and its tests
    monkey-patch it); with `store_history=False` it returns None like the reference but does
    not open a matplotlib window (plots are out of scope).
"""
from __future__ import annotations

import numpy as np

from ..engine import time_grid
from ._ctx import engine_for, engine_for_config
from .config import ForwardSolverConfig, load_params, get_user_input_for_config   # noqa: F401

DEBUG = True
COMPUTE_ENERGY = False
DELTA_SEP = 1e-2               # F2:510


class NeumannLaplacian:
    """Matrix-free stand-in for the CSR matrix of F2:125-137 (mirrored-Neumann 5-point
    operator incl. the reference's Kronecker-order behaviour for Nx != Ny)."""

    def __init__(self, Nx, Ny, hx, hy):
        self.Nx, self.Ny, self.hx, self.hy = int(Nx), int(Ny), float(hx), float(hy)
        self.shape = ((Nx + 1) * (Ny + 1),) * 2

    def _engine(self, tau=0.05, gamma=10.0, c1=0.75, c2=1.0, kappa=1e-4):
        return engine_for(self.Nx, self.Ny, self.hx * self.Nx, self.hy * self.Ny, tau, gamma, c1, c2, kappa)

    def __matmul__(self, v):
        v = np.asarray(v, dtype=np.float64)
        return self._engine().apply_laplacian(v.reshape(self.Nx + 1, self.Ny + 1)).ravel()


def regularized_log(phi, delta_sep):
    """ln((1+p)/(1-p)) with p clipped to +-(1-eps), eps = max(1e-8, delta_sep/2) (F2:86-102).
    Element-wise helper kept on the host for API parity; the engine evaluates the same
    expression inside its residual kernels."""
    eps = max(1e-8, 0.5 * delta_sep)
    p = np.clip(phi, -1.0 + eps, 1.0 - eps)
    return np.log((1.0 + p) / (1.0 - p))


def _check_delta_sep(delta_sep):
    """The engine compiles the separation margin of F2:510 in (DELTA_SEP = 1e-2, the only value the reference's
    march ever passes); any other value would silently give different clips, so it is refused."""
    if abs(float(delta_sep) - DELTA_SEP) > 1e-15:
        raise ValueError(f"delta_sep = {delta_sep!r}: the GPU engine is built for delta_sep = {DELTA_SEP} (F2:510)")


class NeumannLaplacian1D:
    """Matrix-free stand-in for the (N+1)x(N+1) CSR factor of F2:105-122 (second difference with the mirrored
    Neumann rows L[0,1] = L[N,N-1] = 2/h^2): `L1 @ v` for a vector or for the columns of an (N+1, k) array, applied
    by the 1D engine's Laplacian kernel (the same stencil, F1:64-76).  The 2D operator never needs the factor as a
    matrix: `laplacian_matrix_neumann` applies kron(I, Lx) + kron(Ly, I) directly."""

    def __init__(self, N, h):
        self.N, self.h = int(N), float(h)
        self.shape = (self.N + 1, self.N + 1)

    def __matmul__(self, v):
        from ..Vch_control_1D._ctx import engine_for as engine1d_for
        v = np.asarray(v, dtype=np.float64)
        if v.shape[0] != self.N + 1 or v.ndim > 2:
            raise ValueError(f"dimension mismatch: operator is {self.shape}, operand has shape {v.shape}")
        cols = v.reshape(self.N + 1, -1).T                      # one engine call per column (test-size operands)
        eng = engine1d_for(self.N, Lx=self.h * self.N)
        out = np.stack([np.asarray(eng.apply_laplacian(np.ascontiguousarray(c))) for c in cols], axis=1)
        return out.reshape(v.shape)


def laplacian_matrix_neumann_1d(N, h):
    """F2:105-122 as a matrix-free handle."""
    return NeumannLaplacian1D(N, h)


def laplacian_matrix_neumann(Nx, Ny, hx, hy):
    return NeumannLaplacian(Nx, Ny, hx, hy)


def apply_laplacian(L, v, Nx, Ny):
    """F2:140-152 (ValueError on a wrong shape, F2:148-149)."""
    v = np.asarray(v)
    if v.ndim != 2 or v.shape != (Nx + 1, Ny + 1):
        raise ValueError(f"Input field must have shape ({Nx+1}, {Ny+1})")
    return L._engine().apply_laplacian(v)


def initialize_mu(phi, w, c1, c2, kappa, L, Nx, Ny, delta_sep):
    """F2:155-167."""
    _check_delta_sep(delta_sep)
    return L._engine(c1=c1, c2=c2, kappa=kappa).initialize_mu(phi, w)


def solve_w(w_old, dt, gamma, u_n, u_np1):
    """F2:170-181."""
    w_old = np.asarray(w_old, dtype=np.float64)
    nx1, ny1 = w_old.shape
    return engine_for(nx1 - 1, ny1 - 1, 1.0, 1.0, 0.05, gamma, 0.75, 1.0, 1e-4).solve_w(w_old, dt, u_n, u_np1)


def solve_mu_residual(phi_new, phi_old, mu_new, mu_old, dt, L, Nx, Ny):
    """F2:184-196."""
    z = np.zeros_like(np.asarray(phi_new, dtype=np.float64))
    return L._engine().residuals(phi_new, phi_old, mu_new, mu_old, z, z, dt)[1]


def solve_phi_residual(phi_new, phi_old, mu_new, mu_old, w_new, w_old, dt, tau, c1, c2, kappa, L, Nx, Ny,
                       delta_sep):
    """F2:199-221."""
    _check_delta_sep(delta_sep)
    return L._engine(tau=tau, c1=c1, c2=c2, kappa=kappa).residuals(phi_new, phi_old, mu_new, mu_old, w_new,
                                                                  w_old, dt)[0]


class JacobianOperator:
    """Matrix-free stand-in for assemble_jacobian's 2x2 block CSR matrix (F2:224-253):
    `J @ v` applies it, `J.solve(rhs)` replaces spsolve(J.tocsc(), rhs) (F2:370)."""

    def __init__(self, eng, phi_new, dt):
        self.eng, self.phi, self.dt = eng, np.asarray(phi_new, dtype=np.float64), float(dt)
        n = self.phi.size
        self.shape = (2 * n, 2 * n)

    def __matmul__(self, v):
        v = np.asarray(v, dtype=np.float64)
        n, shp = self.phi.size, self.phi.shape
        a, b = self.eng.jacobian_apply(self.phi, self.dt, v[:n].reshape(shp), v[n:].reshape(shp))
        return np.concatenate([a.ravel(), b.ravel()])

    def solve(self, rhs):
        rhs = np.asarray(rhs, dtype=np.float64)
        n, shp = self.phi.size, self.phi.shape
        a, b, _ = self.eng.jacobian_solve(self.phi, self.dt, rhs[:n].reshape(shp), rhs[n:].reshape(shp))
        return np.concatenate([a.ravel(), b.ravel()])


def assemble_jacobian(phi_new, dt, tau, c1, kappa, L, delta_sep):
    _check_delta_sep(delta_sep)
    return JacobianOperator(L._engine(tau=tau, c1=c1, kappa=kappa), phi_new, dt)


def trapz_weights(n_nodes):
    """F2:430-441."""
    w = np.ones(n_nodes)
    w[0], w[-1] = 0.5, 0.5
    return w


def free_energy(phi, kappa, c1, c2, hx, hy, w=None, eps=None):
    """Discrete free energy (F2:256-319) of one field, as a device reduction (`vch2d_free_energy`).
    The array is taken as the reference takes it: axis 0 with hy, axis 1 with hx."""
    return float(free_energy_history(np.asarray(phi)[None], kappa, c1, c2, hx, hy,
                                     w_hist=None if w is None else np.asarray(w)[None], eps=eps)[0])


def free_energy_history(phi_hist, kappa, c1, c2, hx, hy, w_hist=None, eps=None):
    """free_energy of every level of a (rows, A0, A1) history in one launch -> (rows,)."""
    a = np.asarray(phi_hist, dtype=np.float64)
    rows, A0, A1 = a.shape
    eng = engine_for(A0 - 1, A1 - 1, 1.0, 1.0, 0.05, 10.0, c1, c2, kappa, max_steps=max(rows - 1, 1))
    return np.atleast_1d(eng.free_energy(a, hx=hx, hy=hy, w_hist=w_hist, eps=eps))


def instability_report(c1, c2, kappa, tau, Lx, Nmodes=12):
    """Linear growth rates (F2:53-83)."""
    a = 2 * (c1 - c2)
    q = (np.pi * np.arange(1, Nmodes + 1) / Lx) ** 2
    lam = (-kappa * q ** 2 - a * q) / (1 + tau * q)
    print(f"a={a:.3g},  max λ={lam.max():.3g} at mode n={lam.argmax()+1},  unstable modes={(lam>0).sum()}")
    return lam


def init_phi_random(Nx, Ny, delta_sep, amp=0.5, seed=42, enforce_zero_mean=True):
    """Initial data (F2:444-486): host-side by nature (NumPy's PCG64 stream defines 'identical
    initial data'); amp*N(0,1), weighted zero mean, clip, <= 8 interior mass-fix passes."""
    rng = np.random.default_rng(seed)
    phi0 = amp * rng.standard_normal((Nx + 1, Ny + 1))
    wts = np.outer(trapz_weights(Nx + 1), trapz_weights(Ny + 1))
    Wtot = np.sum(wts)
    if enforce_zero_mean:
        phi0 -= np.sum(wts * phi0) / Wtot
    lo, hi = -1.0 + delta_sep, 1.0 - delta_sep
    phi0 = np.clip(phi0, lo, hi)
    if enforce_zero_mean:
        for _ in range(8):
            Mass = np.sum(wts * phi0)
            if abs(Mass) <= 1e-14 * Wtot:
                break
            interior = np.abs(phi0) < (hi - 5e-3)
            Wint = float(np.sum(wts[interior]))
            if Wint <= 0:
                phi0 = np.clip(phi0 - Mass / Wtot, lo, hi)
                break
            phi0[interior] -= Mass / Wint
    return phi0


def newton_raphson(phi_old, mu_old, w_old, w_new, dt, tau, c1, c2, kappa, delta_sep, L, Nx, Ny, hx, hy,
                   return_residual_history=False):
    """One implicit time level on the GPU (F2:323-427); no exception on non-convergence."""
    _check_delta_sep(delta_sep)
    eng = L._engine(tau=tau, c1=c1, c2=c2, kappa=kappa)
    pn, mn, hist, _ = eng.newton_raphson(phi_old, mu_old, w_old, w_new, dt)
    return (pn, mn, list(hist)) if return_residual_history else (pn, mn)


def run_main_simulation(config: ForwardSolverConfig, store_history: bool = False, control_input=None,
                        verbose: bool = True, initial_phi=None, seed: int = 42, amp: float = 0.1):
    """Time march (F2:489-596) -> (phi_hist (M+1,Nx+1,Ny+1), (x, y), t_hist) when
    store_history, else None.  ValueError for a control of the wrong shape (F2:523-525)."""
    Nx, Ny = int(config.Nx), int(config.Ny)
    t_hist, dts = time_grid(float(config.T), float(config.dt_initial))
    if control_input is not None:
        control_input = np.asarray(control_input)
        if control_input.ndim != 3 or control_input.shape[1:] != (Nx + 1, Ny + 1):
            raise ValueError(f"control_input must have shape (M, {Nx+1}, {Ny+1})")
    rows = 0 if control_input is None else control_input.shape[0]
    eng = engine_for_config(config, max_steps=max(len(dts), rows - 1, 1))
    phi0 = init_phi_random(Nx, Ny, DELTA_SEP, amp=amp, seed=seed) if initial_phi is None else initial_phi
    u = control_input
    if u is not None and rows > len(dts) + 1:
        u = u[:len(dts) + 1]                    # rows beyond the march are never read (F2:545-548)
    phi_hist, st = eng.forward(phi0, dts, u=u, store=True)
    if verbose:
        print(f"Simulation complete. ({st['newton_iters']} Newton residuals, {st['linear_solves']} solves, "
              f"{st['seconds']:.3f} s on device)")
    if store_history:
        return phi_hist, (eng.x.copy(), eng.y.copy()), t_hist
    return None
