#!/usr/bin/env python3
"""How good a starting guess for the first Newton solve of step n is an extrapolation of the previous steps' increments?
(GPU box)  d_n = phi_{n+1} - phi_n along a 512^2 march with u = 0; the measure is the one the solver sees, the Schur
residual of the guess relative to the right-hand side: ||A (d_n - guess)|| / ||A d_n||, A = Schur operator at phi_n."""
import sys
import numpy as np
sys.path.insert(0, ".")
import vch_amd
O2 = vch_amd.module("Vch_control_2D.Forward2_solver")
N, M = 512, 1000
t, dts = vch_amd.time_grid(M / 1000.0, 1e-3)
eng = vch_amd.Engine2D(Nx=N, Ny=N, batch=1, max_steps=M)
phi0 = np.stack([O2.init_phi_random(N, N, 1e-2, amp=0.1, seed=42)])
hist, st = eng.forward(phi0, dts, u=None, store=True)
hist = np.asarray(hist).reshape(M + 1, N + 1, N + 1)
d = np.diff(hist, axis=0)
nrm = lambda v: float(np.sqrt((v * v).sum()))
print("step  ||A d||   order0(d1)  linear  quadratic  cubic")
for n in (5, 20, 50, 100, 200, 400, 600, 800, 990):
    A = lambda v: np.asarray(eng.schur_apply(hist[n][None], 1e-3, v[None])).reshape(N + 1, N + 1)
    ref = nrm(A(d[n]))
    g = [d[n - 1], 2 * d[n - 1] - d[n - 2], 3 * d[n - 1] - 3 * d[n - 2] + d[n - 3],
         4 * d[n - 1] - 6 * d[n - 2] + 4 * d[n - 3] - d[n - 4]]
    print("%4d  %.3e  " % (n, ref) + "  ".join("%.3e" % (nrm(A(d[n] - x)) / ref) for x in g))
