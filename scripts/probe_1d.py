#!/usr/bin/env python3
"""1D kernel-level timings at N = 4096, one trajectory (run under rocprofv3 --kernel-trace --stats)."""
import sys
import numpy as np
sys.path.insert(0, ".")
import vch_amd
F1 = vch_amd.module("Vch_control_1D.Forward_solver")
N = 4096
e = vch_amd.Engine1D(N=N, batch=1, max_steps=16)
phi = F1.init_phi_random(N, 1e-2, amp=0.01, seed=42)
rng = np.random.default_rng(0)
a, b = rng.standard_normal(N + 1), rng.standard_normal(N + 1)
for _ in range(50):
    e.jacobian_solve(phi, 1e-3, a, b)
for _ in range(50):
    e.adjoint_solve(phi, 1e-3, a)
w = np.zeros(N + 1)
mu = np.zeros(N + 1)
for _ in range(10):
    pn, mn, hist = e.newton_raphson(phi, mu, w, w, 1e-3)
print(len(hist), hist)
