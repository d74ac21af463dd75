#!/usr/bin/env python3
"""Starting guesses for the adjoint solve of level n from the levels already solved (GPU box): ||A (p_n - guess)|| / ||A p_n||
(A = A(phi_n), A p_n = rhs_n) along a 512^2 sweep with the bench's targets.  Crank-Nicolson does not damp the highest
modes (amplification -> -1), so p carries a component that alternates from level to level; guesses that fit a smooth part
a and an alternating part (-1)^n b separately are listed next to the plain polynomial ones."""
import sys
import numpy as np
sys.path.insert(0, ".")
import vch_amd
O2 = vch_amd.module("Vch_control_2D.Forward2_solver")
N, M = 512, 400
t, dts = vch_amd.time_grid(M / 1000.0, 1e-3)
eng = vch_amd.Engine2D(Nx=N, Ny=N, batch=1, max_steps=M)
phi0 = np.stack([O2.init_phi_random(N, N, 1e-2, amp=0.1, seed=42)])
xs = np.linspace(0, 1, N + 1)
phi_T = (0.7 * np.sin(2 * np.pi * xs)[:, None] * np.cos(np.pi * xs)[None, :])[None].copy()
hist, _ = eng.forward(phi0, dts, u=None, store=True)
hist = np.asarray(hist).reshape(M + 1, N + 1, N + 1)
frac = (t / t[-1])[:, None, None]
phi_Q = ((1 - frac) * hist[0][None] + frac * phi_T[0][None])[None]
p, q, r, st = eng.backward(None, t, 5.0, 10.0, phi_Q, phi_T, want=("p",))[0:1] + (None, None, None)
p = np.asarray(p).reshape(M + 1, N + 1, N + 1)
nrm = lambda v: float(np.sqrt((v * v).sum()))
names = ["p1", "2p1-p2", "3p1-3p2+p3", "p2", "p1+p2-p3", "2p2-p4", "a quad b lin (6 lv)"]
print("level  " + "  ".join("%-12s" % s for s in names))
for n in (390, 350, 300, 200, 100, 50, 20, 5):
    A = lambda v: np.asarray(eng.adjoint_apply("A", hist[n][None], 1e-3, v[None])).reshape(N + 1, N + 1)
    P = [None] + [p[n + j] for j in range(1, 8)]
    ref = nrm(A(p[n]))
    # a quadratic, b linear through 6 levels: least squares on s_k = a0 + a1 k + a2 k^2 + (-1)^k (b0 + b1 k), k = 1..6 -> value at 0
    ks = np.arange(1, 7)
    V = np.stack([np.ones(6), ks, ks ** 2, (-1.0) ** ks, (-1.0) ** ks * ks], axis=1)
    w = np.linalg.pinv(V)          # coefficients = w @ s ; value at k=0: a0 + b0
    c6 = w[0] + w[3]
    g = [P[1], 2 * P[1] - P[2], 3 * P[1] - 3 * P[2] + P[3], P[2], P[1] + P[2] - P[3], 2 * P[2] - P[4],
         sum(c6[k - 1] * P[k] for k in range(1, 7))]
    print("%5d  " % n + "  ".join("%-12.3e" % (nrm(A(p[n] - x)) / ref) for x in g))
