#!/usr/bin/env python3
"""1D configs of BASELINE.json on the GPU: forward march + adjoint sweep timings.
   python scripts/bench_1d.py N steps dt batch"""
import sys, time, json
import numpy as np
sys.path.insert(0, ".")
import vch_amd
F1 = vch_amd.module("Vch_control_1D.Forward_solver")

N = int(sys.argv[1]); M = int(sys.argv[2]); dt = float(sys.argv[3]); B = int(sys.argv[4])
t, dts = vch_amd.time_grid(M * dt, dt)
tt = np.concatenate([[0.0], t])
eng = vch_amd.Engine1D(N=N, batch=B, max_steps=len(dts))
phi0 = np.stack([F1.init_phi_random(N, 1e-2, amp=0.01, seed=42 + i) for i in range(B)])
out = {}
for rep in range(2):
    t0 = time.perf_counter(); ph, st = eng.forward(phi0, dts, store=True); w = time.perf_counter() - t0
    out["forward"] = dict(device_s=st["seconds"], wall_s=w, newton_res_per_step=st["newton_iters"] / B / len(dts),
                          solves_per_step=st["linear_solves"] / B / len(dts), ls_failures=st["linear_iters"])
x = np.linspace(0, 1, N + 1)
phiT = np.broadcast_to(0.7 * np.sin(2 * np.pi * x), (B, N + 1)).copy()
t0 = time.perf_counter(); p, q, r = eng.backward(ph, tt, 0.3, 13.0, None, phiT); out["backward_wall_s"] = time.perf_counter() - t0
out["cfg"] = dict(N=N, steps=len(dts), dt=dt, batch=B)
out["traj_steps_per_s_forward"] = B * len(dts) / out["forward"]["device_s"]
print(json.dumps(out))
