#!/usr/bin/env python3
"""Quality of starting guesses for the first Newton solve of a step along a march under a PGD control (GPU box): the
bench problem at 512^2 x 400 steps, a few PGD iterations, then ||A (d_n - guess)|| / ||A d_n|| for polynomial
extrapolations over consecutive steps, over steps of equal parity, and least-squares fits with an alternating part."""
import sys
import numpy as np
sys.path.insert(0, ".")
import vch_amd
O2 = vch_amd.module("Vch_control_2D.Forward2_solver")
N, M, ITS = 512, 400, int(sys.argv[1]) if len(sys.argv) > 1 else 4
T = M / 1000.0
t, dts = vch_amd.time_grid(T, 1e-3)
eng = vch_amd.Engine2D(Nx=N, Ny=N, batch=1, max_steps=M)
phi0 = np.stack([O2.init_phi_random(N, N, 1e-2, amp=0.1, seed=42)])
xs = np.linspace(0, 1, N + 1)
phi_T = (0.7 * np.sin(2 * np.pi * xs)[:, None] * np.cos(np.pi * xs)[None, :])[None].copy()
eng.pgd_init(phi0, phi_T, t, vch_amd.make_opt(), ramp=True, T=T)
out = eng.pgd_iterate(ITS)
print("attempts", out["attempts"], "cost", out["cost"])
hist = np.asarray(eng.pgd_get("phi")).reshape(M + 1, N + 1, N + 1)
u = np.asarray(eng.pgd_get("u")).reshape(M + 1, N + 1, N + 1)
print("|u| max", np.abs(u).max(), "nonzero share", (u != 0).mean())
d = np.diff(hist, axis=0)
nrm = lambda v: float(np.sqrt((v * v).sum()))
def lagr(nodes):            # weights of the polynomial through values at `nodes` (steps back), evaluated at 0
    w = []
    for j in nodes:
        x = 1.0
        for k in nodes:
            if k != j: x *= (0.0 - k) / (j - k)
        w.append(x)
    return w
def ls(nodes, da, db):      # least squares: smooth part degree da + alternating part degree db
    ks = np.array(nodes, float)
    cols = [ks ** i for i in range(da + 1)] + [(-1.0) ** ks * ks ** i for i in range(db + 1)]
    w = np.linalg.pinv(np.stack(cols, axis=1))
    return list(w[0] + w[da + 1])
cands = {"poly1": ([1], None), "poly2": ([1, 2], None), "poly3": ([1, 2, 3], None), "poly4": ([1, 2, 3, 4], None),
         "poly6": ([1, 2, 3, 4, 5, 6], None), "par1": ([2], None), "par2": ([2, 4], None), "par3": ([2, 4, 6], None),
         "par4": ([2, 4, 6, 8], None), "a2b0/6": ([1, 2, 3, 4, 5, 6], (2, 0)), "a2b1/6": ([1, 2, 3, 4, 5, 6], (2, 1)),
         "a3b1/8": ([1, 2, 3, 4, 5, 6, 7, 8], (3, 1)), "a3b2/8": ([1, 2, 3, 4, 5, 6, 7, 8], (3, 2))}
print("step   " + " ".join("%-9s" % k for k in cands))
for n in (30, 60, 100, 150, 200, 300, 390):
    A = lambda v: np.asarray(eng.schur_apply(hist[n][None], 1e-3, v[None])).reshape(N + 1, N + 1)
    ref = nrm(A(d[n]))
    row = []
    for k, (nodes, fit) in cands.items():
        w = lagr(nodes) if fit is None else ls(nodes, *fit)
        g = sum(wj * d[n - j] for wj, j in zip(w, nodes))
        row.append(nrm(A(d[n] - g)) / ref)
    print("%4d   " % n + " ".join("%-9.2e" % x for x in row))
