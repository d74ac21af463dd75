#!/usr/bin/env python3
"""bench.py — PGD iterations/sec of the MI355X-native viscous Cahn–Hilliard optimal-control
engine on the headline workload of BASELINE.json: 2D 512x512 grid, 1000 Crank–Nicolson time
steps, batch of 64 random initial conditions sharded over 8 GPUs = 8 trajectories per GPU
(weak scaling: the per-GPU batch is fixed as N grows).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one proximal-gradient iteration for every trajectory of the batch = one pass of
the reference's loop body (GD2_configured.py:295-382): adjoint sweep (1000 linear solves) +
gradient/soft-threshold prox + forward march (1000 Newton solves) + cost (+ backtracking
forwards when the optimistic step fails).  Inputs are resident in HBM when the timed region
starts.  `value` = trajectory-iterations per second over all ranks.

The one collective of the data path is a single RCCL all-reduce of the cost scalars per step
(10*B doubles); trajectories are independent.

One JSON line is printed by rank 0.  Extra objects:
  roofline      in-situ HIP-event timing of the dominant kernel class (one extra, untimed PGD
                iteration with an event pair around each launch of the profiled kernels)
  cpu_baseline  the CPU oracle (numpy/scipy restatement of the reference, SuperLU solves) timed
                on this box's host cores on a bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_MFMA_PEAK_TF = 78.6     # = 1/2 of the 157.3 TF FP32 vector/matrix peak of the same table
                             # (v_mfma_f64_16x16x4_f64 issues at the FP64 vector rate on gfx950)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)       # K = 5 timed iterations from u0 = 0
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--grid", type=int, default=512, help="Nx = Ny")
    ap.add_argument("--time-steps", type=int, default=1000)
    ap.add_argument("--batch-per-gpu", type=int, default=8)
    ap.add_argument("--contexts", type=int, default=2,
                    help="engine contexts (HIP streams, one host thread each) the per-GPU batch is split over")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-grid", type=int, default=None, help="grid of the CPU sample (default: --grid)")
    return ap.parse_args()


def cpu_baseline(N, dt, M):
    """Bounded CPU sample with the oracle (kind 'port'): at the full spatial size, one adjoint
    time step (assemble + SuperLU solve of A(phi_n)) and one Newton linear solve of a forward
    step (assemble + SuperLU solve of the 2x2-block Jacobian), single trajectory; a PGD
    iteration is extrapolated linearly in the step count with 2 Newton solves per forward step
    (the reference's own count at 512^2, tests/golden/g2d_newton_512.npz)."""
    import scipy.sparse as sp
    from scipy.sparse.linalg import spsolve
    from oracle import vch2d_oracle as O2
    P = O2.Params2D(Nx=N, Ny=N, T=dt * M, dt_initial=dt)
    h = 1.0 / N
    phi = O2.init_phi_random(N, N, 1e-2, amp=0.1, seed=42)
    w = np.zeros_like(phi)
    t0 = time.perf_counter()
    L = O2.lap_matrix(N, N, h, h)
    mu = O2.mu_init(phi, w, P, h, h)
    Rp = O2.residual_phi(phi, phi, mu, mu, w, w, dt, P, h, h)
    Rm = O2.residual_mu(phi, phi, mu, mu, dt, h, h)
    J = O2.jac_matrix(phi, dt, P, L)
    spsolve(J.tocsc(), -np.concatenate([Rp.ravel(), Rm.ravel()]))
    t_newton = time.perf_counter() - t0
    t0 = time.perf_counter()
    n = phi.size
    I = sp.identity(n, format="csr")
    LL = (L @ L).tocsr()
    Dn = sp.diags(O2.fpp(phi.ravel(), P.c1, P.c2), 0, format="csr")
    A = (I - P.tau * L + 0.5 * dt * LL - 0.5 * dt * (Dn @ L)).tocsc()
    Bm = (I - P.tau * L - 0.5 * dt * LL + 0.5 * dt * (Dn @ L)).tocsr()
    spsolve(A, Bm @ phi.ravel())
    t_adj = time.perf_counter() - t0
    solves_per_step = 2.0
    per_iter = M * (t_adj + solves_per_step * t_newton)
    return dict(value=1.0 / per_iter, unit="PGD iterations/s", cores=1, kind="port",
                sample=(f"oracle (scipy SuperLU, serial) at {N}x{N}: 1 Newton linear solve {t_newton:.1f} s + "
                        f"1 adjoint step {t_adj:.1f} s, 1 trajectory; extrapolated to {M} steps x "
                        f"(1 adjoint + {solves_per_step:g} Newton solves)"),
                newton_solve_s=t_newton, adjoint_step_s=t_adj)


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if a.gpus > 1 or world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    import vch_amd
    F2 = vch_amd.module("Vch_control_2D.Forward2_solver")      # init_phi_random: the package's own host function

    if dist is None:
        vch_amd.build()
    else:                       # one builder per node, the other ranks wait (no concurrent hipcc into one .so)
        if local == 0:
            vch_amd.build()
        dist.barrier()
    N, M, B = a.grid, a.time_steps, a.batch_per_gpu
    T = 1.0
    dt = T / M
    t_hist, dts = vch_amd.time_grid(T, dt)
    M = len(dts)
    K = max(1, a.contexts)
    assert B % K == 0, "--batch-per-gpu must be divisible by --contexts"
    Bc = B // K
    engs = [vch_amd.Engine2D(Nx=N, Ny=N, batch=Bc, max_steps=M, device=local) for _ in range(K)]
    eng = engs[0]
    par = vch_amd.parallel
    dev = f"cuda:{local}" if dist is not None else "cpu"
    seeds = par.shard_seeds(rank, max(world, 1), B)
    phi0 = np.stack([F2.init_phi_random(N, N, 1e-2, amp=0.1, seed=s) for s in seeds])
    xs = np.linspace(0.0, 1.0, N + 1)
    phi_T = 0.7 * np.sin(2 * np.pi * xs)[:, None] * np.cos(np.pi * xs)[None, :]       # G2:199
    opt = vch_amd.make_opt()
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=K)

    def on_all(fn):
        """run fn(k, engine) for every context concurrently (ctypes releases the GIL)"""
        return list(pool.map(lambda ke: fn(*ke), enumerate(engs)))

    t_init = time.perf_counter()
    J0 = np.concatenate(on_all(lambda k, e: e.pgd_init(
        phi0[k * Bc:(k + 1) * Bc], np.broadcast_to(phi_T, (Bc,) + phi_T.shape).copy(), t_hist, opt, ramp=True, T=T)))
    t_init = time.perf_counter() - t_init

    import queue
    import threading

    def run_iterations(n):
        """n PGD iterations of every context.  The trajectories of different contexts are independent
        problems, so each context advances at its own pace (one worker thread each; a line search that
        ends early in one context does not wait for the other); the main thread takes the costs of
        iteration k from every context as they arrive and issues the k-th all-reduce -- one RCCL
        collective per iteration, in iteration order, none of them on a context's critical path."""
        qs = [queue.Queue() for _ in engs]

        def worker(k):
            try:
                for _ in range(n):
                    qs[k].put(engs[k].pgd_iterate(1))
            except BaseException as exc:          # surfaces in the main thread
                qs[k].put(exc)

        ths = [threading.Thread(target=worker, args=(k,), daemon=True) for k in range(K)]
        for th in ths:
            th.start()
        outs = []
        for _ in range(n):
            rs = [q.get() for q in qs]
            for r in rs:
                if isinstance(r, BaseException):
                    raise r
            J = np.zeros((B, 5))
            J[:, 4] = np.concatenate([x["cost"] for x in rs])[:, 0]
            Jsum = par.allreduce_cost(J, dist, dev)      # the single RCCL collective of an iteration
            outs.append(dict(cost_sum=float(Jsum[4]), attempts=int(sum(int(x["attempts"].sum()) for x in rs)),
                             seconds={kk: max(x["seconds"][kk] for x in rs) for kk in rs[0]["seconds"]}))
        for th in ths:
            th.join()
        return outs

    costs = []
    if a.warmup:
        run_iterations(a.warmup)
    par.barrier(dist, dev)
    t0 = time.perf_counter()
    buckets = {}
    attempts = 0
    for r in run_iterations(a.steps):               # synchronous: returns when the devices are done
        costs.append(r["cost_sum"])
        attempts += r["attempts"]
        for k, v in r["seconds"].items():
            buckets[k] = buckets.get(k, 0.0) + float(v)
    par.barrier(dist, dev)
    el = par.max_over_ranks(time.perf_counter() - t0, dist, dev)
    total_traj = B * max(world, 1)
    value = a.steps * total_traj / el

    roof = None
    extra = {}
    if rank == 0 and not a.no_roofline:
        eng.prof_begin(400000)
        eng.pgd_iterate(1)
        prof = eng.prof_end()
        nodes = (N + 1) * (N + 1) * Bc
        alg = {  # algorithmic bytes / flops per launch
            "schur_p": ("hbm", 72.0 * nodes), "adj_q": ("hbm", 32.0 * nodes), "residual": ("hbm", 88.0 * nodes),
            "cg_update": ("hbm", 48.0 * nodes), "adj_rhs": ("hbm", 72.0 * nodes),
            # DCT preconditioner: three FFT passes (field in, field out) or four MFMA f64 GEMMs
            "dct": ("hbm", 16.0 * nodes) if eng.uses_fft else ("mfma", 2.0 * (N + 1) ** 3 * Bc),
        }
        tot = {k: v["ms"] for k, v in prof.items()}
        dom = max(tot, key=tot.get)
        extra["kernel_time_ms"] = {k: round(v["ms"], 3) for k, v in prof.items()}
        extra["kernel_launches"] = {k: v["launches"] for k, v in prof.items()}

        # HBM-side traffic per launch from the committed PMC passes (scripts/pmc_traffic.sh: separate
        # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this bench at B = 8 per launch, gfx950 read
        # correction calibrated on k_grad_prox), scaled to this run's trajectories per launch
        pmc = {}
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")) as fh:
                pk = json.load(fh)["kernels"]
            tr = lambda names: sum(pk[n]["fetch_corrected"] + pk[n]["write_raw"] for n in names) / len(names) * Bc / 8.0
            pmc = {"dct": tr(["k_dct_rows<0, 1024, 10>", "k_dct_cols<1024, 10>", "k_dct_rows<3, 1024, 10>"]),
                   "schur_p": tr(["k_schur_p<0>"]), "residual": tr(["k_residual<1>"]), "adj_q": tr(["k_adj_q"])}
        except (OSError, KeyError, ValueError):
            pmc = {}
        use_pmc = (N == 512 and eng.uses_fft)

        def roof_of(k):
            kind, per = alg[k]
            avg_s = prof[k]["ms"] * 1e-3 / max(prof[k]["launches"], 1)
            if kind == "hbm":
                ach = per / avg_s / 1e9
                return dict(kernel=k, bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s",
                            frac=ach / HBM_PEAK_GBS, traffic=(pmc.get(k) if use_pmc else None), avg_us=avg_s * 1e6,
                            launches=prof[k]["launches"], algorithmic_bytes_per_launch=per)
            ach = per / avg_s / 1e12
            return dict(kernel=k, bound="mfma", achieved=ach, peak=FP64_MFMA_PEAK_TF, unit="TFLOP/s",
                        frac=ach / FP64_MFMA_PEAK_TF, traffic=None, avg_us=avg_s * 1e6,
                        launches=prof[k]["launches"], algorithmic_flops_per_launch=per)
        roof = roof_of(dom if dom in alg else "schur_p")
        extra["roofline_newton_stencil"] = roof_of("schur_p")
        extra["dct_path"] = "fft (in-LDS radix-4 Stockham)" if eng.uses_fft else "gemm (MFMA f64 16x16x4)"

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(a.cpu_grid or N, dt, M)

    if rank == 0:
        out = {
            "metric": "PGD iterations/sec (fwd Newton + adjoint + prox), 2D 512^2 grid",
            "value": value, "unit": "PGD iterations/s", "n_gpus": max(world, 1), "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": 1e3 * el / a.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"2D {N}x{N}, {M} time steps, batch {B} trajectories per GPU "
                                   f"({total_traj} total), seeds 42+i, targets build_targets 1/1, u0 = 0",
                       "grid": N, "time_steps": M, "batch_per_gpu": B, "contexts_per_gpu": K, "parallelism": f"batch-shard x{max(world, 1)}"},
            "roofline": roof, "cpu_baseline": cpu,
            "init_s": t_init, "J0_sum": float(J0[:, 4].sum()), "cost_sum_per_step": costs,
            "backtracking_forwards": attempts, "time_buckets_s": buckets,
        }
        out.update(extra)
        print(json.dumps(out))
    for e in engs:
        e.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
