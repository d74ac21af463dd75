#!/usr/bin/env python3
"""When does a REJECTED line-search candidate become hopeless?  J = J1 + J2 + J3 + J4 with J3, J4 known before the march and
J1's integrand non-negative: the candidate is certain to be rejected from the first level n at which
J3 + J4 + b1/2 int_0^{t_n} |phi - phi_Q|^2 >= cost_k.  Replays PGD iterations through the function-level API and reports
that level for every rejected candidate.  python scripts/r3_reject_probe.py [N] [M] [iters]   (GPU box)"""
import sys
import numpy as np
sys.path.insert(0, ".")
import vch_amd

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
M = int(sys.argv[2]) if len(sys.argv) > 2 else 400
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
F2 = vch_amd.module("Vch_control_2D.Forward2_solver")
T = 1.0
t, dts = vch_amd.time_grid(T, T / M)
e = vch_amd.Engine2D(Nx=N, Ny=N, batch=1, max_steps=M)
phi0 = F2.init_phi_random(N, N, 1e-2, amp=0.1, seed=42)
xs = np.linspace(0, 1, N + 1)
phi_T = 0.7 * np.sin(2 * np.pi * xs)[:, None] * np.cos(np.pi * xs)[None, :]
tf = (t / T)[:, None, None]
phi_Q = (1 - tf) * phi0[None] + tf * phi_T[None]
opt = vch_amd.make_opt()
b1, b2, b3, ks = 5.0, 10.0, 1e-4, 1e-4
w1 = np.ones(N + 1); w1[0] = w1[-1] = 0.5
W = np.outer(w1, w1) / (N * N)
dtv = np.diff(t)


def costs(phi, u):
    q = np.einsum("kij,ij->k", (phi - phi_Q) ** 2, W)
    J1c = 0.5 * b1 * np.concatenate([[0.0], np.cumsum(dtv * (q[1:] + q[:-1]) / 2)])      # J1 up to level n
    J2 = 0.5 * b2 * np.sum(W * (phi[-1] - phi_T) ** 2)
    qu = np.einsum("kij,ij->k", u ** 2, W); qa = np.einsum("kij,ij->k", np.abs(u), W)
    J3 = 0.5 * b3 * np.sum(dtv * (qu[1:] + qu[:-1]) / 2); J4 = ks * np.sum(dtv * (qa[1:] + qa[:-1]) / 2)
    return J1c, J2, J3, J4


u = np.zeros((M + 1, N + 1, N + 1))
phi, _ = e.forward(phi0, dts)
J1c, J2, J3, J4 = costs(phi, u)
cost_k = J1c[-1] + J2 + J3 + J4
alpha_prev = 50.0
print(f"J0 = {cost_k:.9f}")
for k in range(iters):
    _, _, r, _ = e.backward(phi, t, b1, b2, phi_Q, phi_T, want=("r",))
    alpha = alpha_prev
    for rnd in range(11):
        ut = e.grad_prox(u, r, alpha, opt)
        pt, _ = e.forward(phi0, dts, u=ut)
        J1c, J2, J3, J4 = costs(pt, ut)
        Jt = J1c[-1] + J2 + J3 + J4
        ok = Jt < cost_k
        if not ok:
            part = J3 + J4 + J1c
            hit = np.nonzero(part >= cost_k)[0]
            lvl = int(hit[0]) if len(hit) else -1
            print(f"  it {k} round {rnd} alpha {alpha:.4g}: REJECTED J {Jt:.9f} vs {cost_k:.9f} (J1 {J1c[-1]:.6f} J2 {J2:.6f} J3+J4 {J3 + J4:.6f}); "
                  f"hopeless from level {lvl} of {M}" + ("" if lvl >= 0 else " (only the terminal term tips it)"))
        if ok or rnd == 10:
            u, phi, cost_k = ut, pt, Jt
            alpha_prev = min(50.0, (alpha if ok else alpha * 0.8) * 1.2)
            print(f"it {k}: accepted round {rnd} alpha {alpha:.4g} J {Jt:.9f}")
            break
        alpha = alpha_prev * 0.8 if rnd == 0 else alpha * 0.8
