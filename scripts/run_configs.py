#!/usr/bin/env python3
"""BASELINE.json configs 2, 3 and 5 (the ones that are not the headline bench line) run once on the
GPU box: solver statistics and wall times.  Usage: python scripts/run_configs.py [2] [3] [5]"""
import sys, time, json
import numpy as np
sys.path.insert(0, ".")
import vch_amd

which = [int(a) for a in sys.argv[1:]] or [2, 3, 5]
out = {}

if 3 in which:      # 2D 256x256, 400 time steps, single trajectory: one PGD iteration
    F2 = vch_amd.module("Vch_control_2D.Forward2_solver")
    N, M = 256, 400
    t, dts = vch_amd.time_grid(1.0, 1.0 / M)
    e = vch_amd.Engine2D(Nx=N, Ny=N, batch=1, max_steps=len(dts))
    xs = np.linspace(0, 1, N + 1)
    phi_T = 0.7 * np.sin(2 * np.pi * xs)[:, None] * np.cos(np.pi * xs)[None, :]
    J0 = e.pgd_init(F2.init_phi_random(N, N, 1e-2, amp=0.1, seed=42), phi_T, t, vch_amd.make_opt(), ramp=True, T=1.0)
    t0 = time.perf_counter(); r = e.pgd_iterate(3); w = time.perf_counter() - t0
    out["config3_2d_256_400"] = dict(J0=float(J0[0, 4]), cost=r["cost"][0].tolist(), attempts=r["attempts"][0].tolist(),
                                     s_per_pgd_iteration=w / 3, seconds=r["seconds"])
    e.close()

if 5 in which:      # 2D 1024x1024, amp = 1.0 (about a third of the nodes start clipped), 100 steps of 1e-3
    F2 = vch_amd.module("Vch_control_2D.Forward2_solver")
    N, M, B = 1024, 100, 2
    t, dts = vch_amd.time_grid(0.1, 1e-3)
    e = vch_amd.Engine2D(Nx=N, Ny=N, batch=B, max_steps=len(dts))
    phi0 = np.stack([F2.init_phi_random(N, N, 1e-2, amp=1.0, seed=42 + i) for i in range(B)])
    clipped = float(np.mean(np.abs(phi0) >= 0.99 - 1e-12))
    t0 = time.perf_counter(); ph, st = e.forward(phi0, dts, store=True); w = time.perf_counter() - t0
    wts = np.outer(F2.trapz_weights(N + 1), F2.trapz_weights(N + 1))
    mass = np.array([[np.sum(wts * ph[b, k]) for k in (0, len(dts))] for b in range(B)])
    xs = np.linspace(0, 1, N + 1)
    phi_T = np.broadcast_to(0.7 * np.sin(2 * np.pi * xs)[:, None] * np.cos(np.pi * xs)[None, :], phi0.shape).copy()
    t1 = time.perf_counter(); _, _, r, sb = e.backward(None, t, 5.0, 10.0, None, phi_T, want=("r",)); wb = time.perf_counter() - t1
    out["config5_2d_1024_stress"] = dict(clipped_fraction=clipped, forward_s=w, stats={k: (float(v) if isinstance(v, float) else int(v)) for k, v in st.items()},
                                         newton_res_per_step=st["newton_iters"] / B / len(dts), armijo_trials=int(st["armijo_trials"]),
                                         cg_per_solve=st["linear_iters"] / max(st["linear_solves"], 1),
                                         max_abs_phi=float(np.abs(ph).max()), finite=bool(np.isfinite(ph).all()),
                                         mass_drift=float(np.abs(mass[:, 1] - mass[:, 0]).max()),
                                         backward_s=wb, backward_cg_per_solve=sb["linear_iters"] / max(sb["linear_solves"], 1),
                                         r_finite=bool(np.isfinite(r).all()))
    e.close()

if 2 in which:      # 1D N = 4096, 1000 time steps, single trajectory: forward + backward + cost + prox
    F1 = vch_amd.module("Vch_control_1D.Forward_solver")
    K1 = vch_amd.module("Vch_control_1D.config")
    G1 = vch_amd.module("Vch_control_1D.GD_1D")
    cfg = K1.ForwardSolverConfig(N=4096, T=1.0, dt_initial=1e-3)
    t0 = time.perf_counter()
    ph, x, t = F1.run_main_simulation(cfg, store_history=True, verbose=False)
    w = time.perf_counter() - t0
    B1 = vch_amd.module("Vch_control_1D.backward_solver")
    phi_T = 0.7 * np.cos(2 * np.pi * x)
    phi_Q = np.linspace(0, 1, len(t))[:, None] * phi_T[None, :] + (1 - np.linspace(0, 1, len(t)))[:, None] * ph[0][None, :]
    t1 = time.perf_counter(); p, q, r = B1.run_backward(ph, x, t, 5.0, 10.0, phi_Q, phi_T); wb = time.perf_counter() - t1
    out["config2_1d_4096_1000"] = dict(rows=int(ph.shape[0]), forward_s=w, backward_s=wb, finite=bool(np.isfinite(ph).all() and np.isfinite(r).all()),
                                       mass_drift=float(np.abs(ph @ F1.trapz_weights(4097) / 4096 - (ph[0] @ F1.trapz_weights(4097)) / 4096).max()))
    # device-resident PGD iterations at the same size: one trajectory, then a batch of 64 seeds
    O1 = K1.OptimizationConfig()
    for seeds in ([42], list(range(42, 106))):
        t0 = time.perf_counter()
        res = G1.run_optimization_resident(cfg, O1, n_iter=2, seeds=seeds)
        w = time.perf_counter() - t0
        out[f"config2_pgd_resident_batch{len(seeds)}"] = dict(
            wall_s_incl_initial_march=w, seconds=res["seconds"], trials=np.asarray(res["trials"]).tolist()[:4],
            costs=np.asarray(res["costs"]).tolist()[:2])
print(json.dumps(out, indent=1))
