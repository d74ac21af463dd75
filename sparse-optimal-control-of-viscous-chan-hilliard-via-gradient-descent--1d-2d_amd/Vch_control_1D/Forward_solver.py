"""Host-side mirror of src/1D/Vch_control_1D/Forward_solver.py: same function names and
signatures, evaluated by the HIP engine (one persistent workgroup per trajectory, block cyclic
reduction for the Newton systems).  `laplacian_matrix_neumann` returns a matrix-free handle."""
from __future__ import annotations

import numpy as np

from ..engine import time_grid
from ._ctx import engine_for, engine_for_config
from .config import ForwardSolverConfig

delta_sep = 1e-2          # F1:42
DEBUG = True
COMPUTE_ENERGY = True


class NeumannLaplacian1D:
    """Matrix-free stand-in for the dense (N+1)^2 matrix of F1:64-76 (`L @ v`)."""

    def __init__(self, N, h):
        self.N, self.h = int(N), float(h)
        self.shape = (N + 1, N + 1)

    def _engine(self, **kw):
        return engine_for(self.N, self.h * self.N, **kw)

    def __matmul__(self, v):
        return self._engine().apply_laplacian(np.asarray(v, dtype=np.float64))


def regularized_log(phi, eps=None):
    """F1:57-62 (element-wise host helper; the engine evaluates it inside its kernels)."""
    if eps is None:
        eps = max(1e-8, 0.5 * delta_sep)
    p = np.clip(phi, -1 + eps, 1 - eps)
    return np.log((1 + p) / (1 - p))


def laplacian_matrix_neumann(N, h):
    return NeumannLaplacian1D(N, h)


def apply_laplacian(L, v):
    """F1:78-80."""
    return L @ v


def initialize_mu(phi, w, c1, c2, L, kappa):
    """F1:82-86: one stencil + element-wise pass on the GPU Laplacian."""
    return -kappa * (L @ phi) + (c1 * regularized_log(phi) - 2.0 * c2 * np.asarray(phi)) - w


def solve_w(w_old, dt, gamma, u_n, u_np1):
    """F1:88-91 (a four-flop element-wise filter: inside the march it is fused into the persistent
    kernel; the stand-alone form is evaluated on the host)."""
    g = gamma / dt
    return ((g - 0.5) * w_old + 0.5 * (u_np1 + u_n)) / (g + 0.5)


def solve_mu_residual(phi_new, phi_old, mu_new, mu_old, dt, L):
    """F1:93-97."""
    z = np.zeros_like(np.asarray(phi_new, dtype=np.float64))
    return L._engine().residuals(phi_new, phi_old, mu_new, mu_old, z, z, dt)[1]


def solve_phi_residual(phi_new, phi_old, mu_new, mu_old, w_new, w_old, dt, tau, c1, c2, L, kappa):
    """F1:99-109."""
    return L._engine(tau=tau, c1=c1, c2=c2, kappa=kappa).residuals(phi_new, phi_old, mu_new, mu_old, w_new, w_old, dt)[0]


class JacobianOperator1D:
    """Stand-in for the dense Jacobian of F1:111-137: `.solve(rhs)` replaces np.linalg.solve(J, rhs)."""

    def __init__(self, eng, phi_new, dt):
        self.eng, self.phi, self.dt = eng, np.asarray(phi_new, dtype=np.float64), float(dt)

    def solve(self, rhs):
        n = self.phi.size
        a, b = self.eng.jacobian_solve(self.phi, self.dt, rhs[:n], rhs[n:])
        return np.concatenate([a, b])


def assemble_jacobian(phi_new, dt, tau, c1, L, kappa):
    return JacobianOperator1D(L._engine(tau=tau, c1=c1, kappa=kappa), phi_new, dt)


def newton_raphson(phi_old, mu_old, w_old, w_new, dt, tau, c1, c2, h, delta_sep, L, kappa,
                   return_residual_history=False):
    """F1:139-235 on the GPU.  RuntimeError for a non-finite mass defect like the reference."""
    if abs(float(delta_sep) - 1e-2) > 1e-15:        # the engine compiles the module constant of F1:42 in
        raise ValueError(f"delta_sep = {delta_sep!r}: the GPU engine is built for delta_sep = 0.01 (F1:42)")
    eng = L._engine(tau=tau, c1=c1, c2=c2, kappa=kappa)
    try:
        pn, mn, hist = eng.newton_raphson(phi_old, mu_old, w_old, w_new, dt)
    except Exception as e:          # engine state error -> the reference's RuntimeError (F1:170)
        if "mass_defect" in str(e):
            raise RuntimeError(str(e)) from None
        raise
    return (pn, mn, list(hist)) if return_residual_history else (pn, mn)


def trapz_weights(n_nodes: int):
    w = np.ones(n_nodes)
    w[0] = w[-1] = 0.5
    return w


def free_energy(phi, kappa, c1, c2, h, w=None, eps=None):
    """F1:243-262 for one field, as a device reduction (`vch1d_free_energy`)."""
    return float(free_energy_history(np.asarray(phi)[None], kappa, c1, c2, h,
                                     w_hist=None if w is None else np.asarray(w)[None], eps=eps)[0])


def free_energy_history(phi_hist, kappa, c1, c2, h, w_hist=None, eps=None):
    """free_energy of every row of a (rows, N+1) history in one launch -> (rows,)."""
    a = np.asarray(phi_hist, dtype=np.float64)
    rows, n = a.shape
    eng = engine_for(n - 1, h * (n - 1), c1=c1, c2=c2, kappa=kappa, max_steps=max(rows, 1))
    return np.atleast_1d(eng.free_energy(a, h=h, w_hist=w_hist, eps=eps))


def init_phi_random(N, delta_sep, amp=0.1, seed=42, enforce_zero_mean=True):
    """F1:264-277 (NumPy PCG64 stream = 'identical initial data')."""
    rng = np.random.default_rng(seed)
    phi0 = amp * rng.standard_normal(N + 1)
    if enforce_zero_mean:
        wts = trapz_weights(N + 1)
        phi0 -= np.dot(wts, phi0) / wts.sum()
    return np.clip(phi0, -1 + delta_sep, 1 - delta_sep)


def run_main_simulation(fwd_config: ForwardSolverConfig | None = None, store_history=False, control_input=None,
                        verbose=True, initial_phi=None):
    """F1:286-397 -> (phi_hist (M+2, N+1), x, t_hist) with the duplicated t = 0 row; with
    store_history=False returns (phi_T, x, t_hist) like the reference (no plot)."""
    if fwd_config is None:
        fwd_config = ForwardSolverConfig()
    N = int(fwd_config.N)
    tg, dts = time_grid(float(fwd_config.T), float(fwd_config.dt_initial))
    t_hist = np.concatenate([[0.0], tg])
    eng = engine_for_config(fwd_config, max_steps=max(len(dts), 1))
    if initial_phi is not None and np.shape(initial_phi) == (N + 1,):
        phi0 = np.asarray(initial_phi, dtype=np.float64)
    else:
        phi0 = init_phi_random(N, delta_sep, amp=0.01, seed=42, enforce_zero_mean=True)
    try:
        phi_hist, st = eng.forward(phi0, dts, u=control_input, store=True)
    except Exception as e:
        if "mass_defect" in str(e):
            raise RuntimeError(str(e)) from None
        raise
    if verbose:
        print("Simulation complete.")
    if store_history:
        return phi_hist, eng.x.copy(), t_hist
    return phi_hist[-1].copy(), eng.x.copy(), np.array([0.0, 0.0])     # F1:329-334: only the two t=0 entries
