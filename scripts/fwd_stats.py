#!/usr/bin/env python3
"""Solver statistics of one forward march + one adjoint sweep at the headline size (GPU box)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import vch_amd
O2 = vch_amd.module("Vch_control_2D.Forward2_solver")

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
M = int(sys.argv[2]) if len(sys.argv) > 2 else 200
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
T = 1.0 * M / 1000.0
t, dts = vch_amd.time_grid(T, 1e-3)
eng = vch_amd.Engine2D(Nx=N, Ny=N, batch=B, max_steps=len(dts))
phi0 = np.stack([O2.init_phi_random(N, N, 1e-2, amp=0.1, seed=42 + i) for i in range(B)])
for rep in range(2):
    t0 = time.perf_counter()
    _, st = eng.forward(phi0, dts, store=False)
    wall = time.perf_counter() - t0
    print("forward", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in st.items()}, "wall", round(wall, 3),
          "| per step: newton res %.2f solves %.2f cg/solve %.2f ms/step %.3f" % (
              st["newton_iters"] / B / M, st["linear_solves"] / B / M, st["linear_iters"] / max(st["linear_solves"], 1),
              1e3 * st["seconds"] / M))
xs = np.linspace(0, 1, N + 1)
phi_T = np.broadcast_to(0.7 * np.sin(2 * np.pi * xs)[:, None] * np.cos(np.pi * xs)[None, :], phi0.shape).copy()
t0 = time.perf_counter()
_, _, _, st = eng.backward(None, t, 5.0, 10.0, None, phi_T, want=())
print("backward", st, "wall", round(time.perf_counter() - t0, 3), "| cg/solve %.2f ms/step %.3f" % (
    st["linear_iters"] / max(st["linear_solves"], 1), 1e3 * st["seconds"] / M))
