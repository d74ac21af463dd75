#!/usr/bin/env python3
"""bench.py — PGD iterations/sec of the MI355X-native viscous Cahn–Hilliard optimal-control
engine on the headline workload of BASELINE.json: 2D 512x512 grid, 1000 Crank–Nicolson time
steps, batch of 64 random initial conditions sharded over 8 GPUs = 8 trajectories per GPU
(weak scaling: the per-GPU batch is fixed as N grows).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: either under a launcher -- python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
     -- or bare, in which case this process starts the N ranks itself as child processes, spawn_ranks)

A "step" is one proximal-gradient iteration for every trajectory of the batch = one pass of
the reference's loop body (GD2_configured.py:295-382): adjoint sweep (1000 linear solves) +
gradient/soft-threshold prox + forward march (1000 Newton solves) + cost + error metrics
(+ backtracking forwards when the optimistic step fails).  Inputs are resident in HBM when the
timed region starts.  `value` = trajectory-iterations per second over all ranks (batch x
iterations / s; the batch advances in lock step, so batch-iterations/s = value / batch).

The one collective of the data path is a single RCCL all-reduce of the cost scalars per step
(SURVEY 8e), issued through the C ABI on the device-resident scalars (vch_comm_allreduce_cost;
`--collective torch` uses torch.distributed on a host copy instead).

One JSON line is printed by rank 0.  Extra objects:
  roofline      the dominant kernel: algorithmic bytes per launch / mean duration of its LIVE launches, measured with
                HIP event pairs on the engine's stream(s) during ONE extra PGD iteration in which every context
                of the rank runs (the same contention as the timed region), the pair's own cost calibrated on an
                empty kernel and taken off; `roofline_kernels` lists every profiled kernel the same way (with the
                committed rocprofv3 device-side durations next to it), `roofline_newton_stencil` is the north-star kernel
  cpu_baseline  the CPU oracle (numpy/scipy restatement of the reference, SuperLU solves) timed
                on this box's host cores on a bounded sample of the same workload (rank 0, N = 1)
"""
import argparse
import hashlib
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

# Algorithmic bytes per node per launch (fp64 fields read + written once).
ALG_BYTES = {
    "cheb_rows": 56.0,       # row kernel of a reduction-free sweep, j >= 2: E_cols output, b~, y_j, y_{j-1}, D -> y_{j+1}, E_rows(Delta y_{j+1})
    "cheb_rows_first": 40.0,  # j = 0: E_cols output, D -> b~, y_1, E_rows(Delta y_1)
    "cg_rows_fwd": 72.0,     # CG form (wide spectra only): z, q, p_old, D, x -> x, z', p', E_rows(Delta p')
    "cg_rows_fwd_first": 40.0,   # z, D -> x (zeroed), p, E_rows(Delta p)
    "schur_p": 72.0,         # GEMM-DCT grids only: z, q, p_old, D, x -> x, z', p, v
    "schur_p_first": 40.0,   # z, D -> x (zeroed), p, v
    "dct_rows_fwd": 16.0,    # field -> field
    "dct_cols": 16.0,        # field -> field (forward transform, multiplier, inverse transform in LDS)
    "dct_rows_inv": 32.0,    # field, D, other -> field (+ the CG dot products)
    "residual": 96.0,        # Armijo trial (k_eval<2>): phi, dphi, mu, D, R_phi, c_phi, c_mu -> phi_t, mu_t, R_phi, rhs, D
                             # (+ 8 per plane of a folded-in starting guess and 8 for x0: not counted, frac is a lower bound)
    "residual_first": 104.0,  # start of a step (k_eval<0>): phi, mu, w, u_n, u_n+1 -> w+, c_phi, c_mu, phi, mu0, R_phi, rhs, D
    "adj_q": 72.0,           # first pass of an adjoint CG sweep (k_adj_rows_fwd): r, q, ph_old, y, D_n -> y, r', ph', E_rows(ph')
    "cg_update": 24.0, "adj_rhs": 64.0,   # p, q, phi_n, phi_n+1, phiQ_n, phiQ_n+1 -> rhs, D_n
    "guess": 80.0,           # k_guess at order 6 (CG-form steps only): six increments, D, rhs -> rhs, x0
    "adj_guess": 56.0,       # k_adj_guess at order 4: four levels, p -> ring, x0
}
PMC_NAMES = {"cheb_rows": "k_cheb_rows<1024, 10, 0>", "cheb_rows_first": "k_cheb_rows<1024, 10, 1>", "cg_rows_fwd": "k_cg_rows_fwd<0, 1024, 10>",
             "schur_p": "k_schur_p<0>", "dct_rows_fwd": "k_dct_rows<0, 1024, 10>", "dct_cols": "k_dct_cols<1024, 10>",
             "dct_rows_inv": "k_dct_rows<5, 1024, 10>", "residual": "k_eval<2, false>", "residual_first": "k_eval<0, false>", "adj_q": "k_adj_rows_fwd<0, 1024, 10>",
             "guess": "k_guess", "adj_guess": "k_adj_guess", "adj_rhs": "k_adj_rhs"}


# kernel classes of the in-bench timing -> kernel names in the committed rocprofv3 summaries of the same command
# (profiles/live_stats_contexts{K}.txt = scripts/r3_profile.sh on the final build, device-side durations over LIVE launches;
# profiles/live_stats_march_b8.txt = scripts/r3_prof_march.sh, one context of 8 trajectories alone on the chip)
ROCPROF_NAMES = {"dct_cols": ("k_dct_cols<",), "dct_rows_inv": ("k_dct_rows<3", "k_dct_rows<4", "k_dct_rows<5"),
                 "dct_rows_fwd": ("k_dct_rows<0",), "cg_rows_fwd": ("k_cg_rows_fwd<0",), "cg_rows_fwd_first": ("k_cg_rows_fwd<1",),
                 "cheb_rows": ("k_cheb_rows<1024, 10, 0>",), "cheb_rows_first": ("k_cheb_rows<1024, 10, 1>",),
                 "residual": ("k_eval<2", "k_residual2", "k_residual<1>"), "residual_first": ("k_eval<0", "k_residual<0>"),
                 "adj_q": ("k_adj_rows_fwd<",), "guess": ("k_guess",), "adj_guess": ("k_adj_guess",),
                 "adj_rhs": ("k_adj_rhs",), "cg_update": ("k_cg_finish",), "cost": ("k_cost",), "prox": ("k_grad_prox",)}


def live_stats_table(name):
    """{kernel name: (live calls, live mean us)} from a committed scripts/r3_live_stats.py table, or {}."""
    path = os.path.join(ROOT, "profiles", name)
    out = {}
    try:
        with open(path) as fh:
            for ln in fh:
                if ln.startswith("#") or ln.startswith("kernel"):
                    continue
                f = ln.split()
                if len(f) < 7:
                    continue
                out[" ".join(f[:-6])] = (float(f[-5]), float(f[-3]))
    except OSError:
        return {}, None
    return out, os.path.relpath(path, ROOT)


def rocprof_reference(table):
    """{class: mean device-side duration in us over live launches} from a live_stats_table, call-weighted over the class's kernels."""
    out = {}
    for cls, prefixes in ROCPROF_NAMES.items():
        calls = tot = 0.0
        for name, (n, mean) in table.items():
            if any(name.startswith(p) for p in prefixes):
                calls += n
                tot += n * mean
        if calls:
            out[cls] = tot / calls
    return out


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
    ap.add_argument("--collective", choices=["cabi", "torch"], default="cabi")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-grid", type=int, default=None, help="grid of the CPU sample (default: --grid)")
    ap.add_argument("--cpu-steps", type=int, default=3, help="forward and adjoint time steps of the CPU sample")
    ap.add_argument("--dry-run", action="store_true",
                    help="tests only: gloo + a stand-in engine, exercises the multi-rank skeleton without a GPU")
    ap.add_argument("--dry-run-fail-rank", type=int, default=-1, help="tests only: this rank's stand-in engine raises")
    return ap.parse_args()


def cpu_baseline(N, dt, M, n_steps):
    """Bounded CPU sample with the oracle (kind 'port'): `n_steps` forward time steps (Newton with SuperLU solves of
    the 2x2-block Jacobian, F2:323-427) and `n_steps` adjoint steps (assemble + SuperLU solve of A(phi_n), B2:212-242)
    at the full spatial size, one trajectory, one wall-time sample per step; the once-per-march assembly of L and L@L
    is outside the samples.  A PGD iteration (no backtracking) is extrapolated linearly in the step count."""
    from oracle import vch2d_oracle as O2
    r = O2.time_one_step(N, dt, n_fwd_steps=n_steps, n_bwd_steps=n_steps)
    per_iter = M * (r["fwd_s_per_step"] + r["bwd_s_per_step"])
    return dict(value=1.0 / per_iter, unit="PGD iterations/s", cores=1, kind="port",
                sample=(f"oracle (scipy SuperLU, serial) at {N}x{N}, 1 trajectory: {r['fwd_steps']} forward steps "
                        f"({r['solves']} Newton solves) + {r['bwd_steps']} adjoint steps, one sample per step; "
                        f"extrapolated to {M} x (1 forward + 1 adjoint step), no backtracking"),
                fwd_step_seconds=r["fwd_step_seconds"], bwd_step_seconds=r["bwd_step_seconds"])


def source_hash():
    import importlib
    import vch_amd
    L = importlib.import_module(vch_amd.PKG_NAME + "._lib")
    return L.source_hash()


class DryEngine:
    """Stand-in for Engine2D in --dry-run (tests of the multi-rank skeleton on CPU): same methods, no arithmetic."""
    uses_fft = True
    PROF_CLASSES = ()

    def __init__(self, batch, fail=False):
        self.B, self.k, self.ctx, self.fail = batch, 0, None, fail

    def pgd_init(self, phi0, *a, **k):
        return np.ones((self.B, 5))

    def pgd_iterate(self, n):
        time.sleep(0.02)
        self.k += 1
        if self.fail and self.k >= 2:
            raise RuntimeError("injected failure (--dry-run-fail-rank)")
        c = np.full((self.B, n), 1.0 / self.k)
        return dict(iters=n, cost=c, alpha=c, attempts=np.zeros((self.B, n), dtype=np.int32), change=c, tracking_error=c,
                    terminal_error=c, seconds=dict(backward=0.0, gradprox=0.0, optimistic_forward=0.02, cost=0.0, backtracking=0.0))

    def prof_begin(self, n):
        pass

    def prof_end(self):
        return {}

    def counters(self):
        return 0, 0

    def close(self):
        pass


RESULT_OUT = None      # the process's real stdout, kept for the one JSON line (see main)


def spawn_ranks(a):
    """`python bench.py --gpus N` started WITHOUT a launcher (no RANK / WORLD_SIZE in the environment): start the N ranks
    as child processes of this one -- which has not imported torch or the engine and has made no HIP call, and which is
    never replaced by another program -- with the env:// rendezvous variables torch.distributed.run would set, pass rank 0's
    single JSON line through and exit with the worst child status.  A rank that fails ends the others."""
    import socket
    import subprocess
    n = a.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=(subprocess.PIPE if r == 0 else subprocess.DEVNULL), text=True))
    line = None
    worst = 0
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            rc = procs[r].poll()
            if rc is None:
                continue
            pending.discard(r)
            if rc != 0:
                worst = worst or rc
                for q in pending:                 # the failed rank's peers would wait in the next collective
                    procs[q].terminate()
        if pending:
            time.sleep(0.05)
    out0 = procs[0].stdout.read() if procs[0].stdout else ""
    for ln in out0.splitlines():
        if ln.startswith("{"):
            line = ln
    if worst == 0 and line is None:
        print("[bench] rank 0 printed no result line", file=sys.stderr)
        worst = 1
    if line is not None and worst == 0:
        print(line, flush=True)
    sys.exit(worst if worst >= 0 else 1)


def main():
    global RESULT_OUT
    a = parse()
    if a.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a)
    # stdout carries exactly one line, the JSON result: libraries that print banners on file descriptor 1 (RCCL
    # prints its version block there when a communicator is created) are sent to stderr instead
    sys.stdout.flush()
    RESULT_OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if a.gpus > 1 or world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.dry_run:
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    try:
        run(a, world, rank, local, dist)
    except BaseException:
        # a rank that fails must not leave its peers inside the next collective: tear the group down and exit
        # non-zero (torchrun then ends the other ranks) instead of unwinding through a half-finished iteration
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(1)


def run(a, world, rank, local, dist):
    import vch_amd
    par = vch_amd.parallel
    dev = "cpu" if (dist is None or a.dry_run) else f"cuda:{local}"
    if not a.dry_run:
        if dist is None:
            vch_amd.build()
        else:                   # one builder per node, the other ranks wait (no concurrent hipcc into one .so)
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
    if a.dry_run:
        engs = [DryEngine(Bc, fail=(rank == a.dry_run_fail_rank)) for _ in range(K)]
        phi0 = np.zeros((B, 2, 2))
        phi_T = np.zeros((2, 2))
    else:
        F2 = vch_amd.module("Vch_control_2D.Forward2_solver")      # init_phi_random: the package's own host function
        engs = [vch_amd.Engine2D(Nx=N, Ny=N, batch=Bc, max_steps=M, device=local) for _ in range(K)]
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

    # the collective: through the C ABI on the device-resident scalars, or torch.distributed on a host copy
    comm, collective = None, "torch"
    if a.collective == "cabi" and not a.dry_run:
        try:
            comm = par.CostComm(rank, max(world, 1), local, dist)
            collective = "cabi-rccl"
        except Exception as exc:          # reported, not hidden: the JSON line names the path that ran
            print(f"[bench] C-ABI collective unavailable ({exc}); using torch.distributed", file=sys.stderr)
            comm = None
    if dist is not None:                  # every rank must take the same path
        import torch
        flag = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0 and comm is not None:
            comm.close()
            comm, collective = None, "torch"
    if comm is None:
        collective = "torch-rccl" if (dist is not None and not a.dry_run) else ("gloo" if dist is not None else "local")

    import queue
    import threading
    it_done = [0]
    last_cost = J0[:, 4].copy()          # a stopped trajectory reports no new cost: its last one stands

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
            if comm is not None:
                Jsum = comm.allreduce(engs, it_done[0])      # the single RCCL collective of an iteration
            else:
                J = np.zeros((B, 5))
                c = np.concatenate([x["cost"] for x in rs])[:, 0]
                last_cost[:] = np.where(np.isnan(c), last_cost, c)
                J[:, 4] = last_cost
                Jsum = par.allreduce_cost(J, dist, dev)
            it_done[0] += 1
            outs.append(dict(cost_sum=float(Jsum[4]), attempts=int(sum(int(x["attempts"].sum()) for x in rs)),
                             attempts_list=[int(v) for x in rs for v in x["attempts"][:, 0]],
                             iters=[int(x["iters"]) for x in rs],
                             seconds={kk: max(x["seconds"][kk] for x in rs) for kk in rs[0]["seconds"]}))
        for th in ths:
            th.join()
        return outs

    costs = []
    if a.warmup:
        run_iterations(a.warmup)
    c0 = [e.counters() for e in engs]
    par.barrier(dist, dev)
    t0 = time.perf_counter()
    buckets = {}
    attempts = 0
    attempts_per_step = []
    for r in run_iterations(a.steps):               # synchronous: returns when the devices are done
        assert all(i == 1 for i in r["iters"]), "a context ran no iteration (all of its trajectories have stopped)"
        costs.append(r["cost_sum"])
        attempts += r["attempts"]
        attempts_per_step.append(r["attempts_list"])
        for k, v in r["seconds"].items():
            buckets[k] = buckets.get(k, 0.0) + float(v)
    par.barrier(dist, dev)
    el = par.max_over_ranks(time.perf_counter() - t0, dist, dev)
    c1 = [e.counters() for e in engs]
    total_traj = B * max(world, 1)
    value = a.steps * total_traj / el

    roof = None
    extra = {}
    if rank == 0 and not a.no_roofline and not a.dry_run:
        # one extra iteration with an event pair around every profiled launch, ALL contexts of this rank running
        # (the contention of the timed region); the other ranks wait at the barrier below
        on_all(lambda k, e: e.prof_begin(400000))
        on_all(lambda k, e: e.pgd_iterate(1))
        profs = on_all(lambda k, e: e.prof_end())
        spans_c = on_all(lambda k, e: e.prof_spans())
        prof = {k: dict(ms=sum(p[k]["ms"] for p in profs), launches=sum(p[k]["launches"] for p in profs)) for k in profs[0]}
        spans = {k: np.concatenate([sp[k] for sp in spans_c]) for k in spans_c[0]}
        nodes = (N + 1) * (N + 1) * Bc
        alg = {k: ("hbm", v * nodes) for k, v in ALG_BYTES.items()}
        alg["dct_gemm"] = ("mfma", 2.0 * (N + 1) ** 3 * Bc)
        tot = {k: v["ms"] for k, v in prof.items() if v["launches"] and k != "event_pair_noop"}
        extra["kernel_time_ms"] = {k: round(v["ms"], 3) for k, v in prof.items()}
        extra["kernel_launches"] = {k: v["launches"] for k, v in prof.items()}

        # HBM-side traffic per launch: NOT measured in this run -- taken from the committed PMC pass
        # (scripts/pmc_traffic.sh -> profiles/pmc_traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs
        # of this bench, 8 trajectories per launch), scaled to this run's trajectories per launch
        pmc, pmc_src = {}, None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
                pj = json.load(fh)
            pk, pb = pj["kernels"], float(pj.get("trajectories_per_launch", 8))
            for k, name in PMC_NAMES.items():
                if name in pk and pk[name].get("fetch_corrected") is not None:
                    pmc[k] = (pk[name]["fetch_corrected"] + pk[name]["write_raw"]) * Bc / pb
            pmc_src = f"profiles/pmc_traffic.json (committed rocprofv3 --pmc pass at {pb:g} trajectories per launch, scaled x{Bc / pb:g})"
        except (OSError, KeyError, ValueError):
            pmc = {}
        use_pmc = (N == 512 and engs[0].uses_fft)

        # What an event pair spans around an EMPTY kernel (256 pairs per context, recorded by prof_begin while the other
        # context runs): the pair's own cost -- two barrier packets, dispatch latency, queueing behind the other context.
        # A launch whose trajectories are all gated off lasts no longer than that: launches within 2.5 us of it are counted as
        # no-ops and left out of the LIVE mean.  The kernel's own time is estimated as live span - (empty span - 1 us, the
        # empty kernel itself); `frac` uses that estimate, `frac_span` the plain span (a lower bound).
        noop_sp = spans.get("event_pair_noop", np.zeros(0))
        noop_us = float(np.median(noop_sp)) if len(noop_sp) else 0.0
        extra["event_pair_span_of_empty_kernel_us"] = noop_us
        pair_cost = max(noop_us - 1.0, 0.0)

        # device-side durations of the same kernels over live launches from the committed rocprofv3 traces: the same command
        # at the same number of contexts, and one context of 8 trajectories alone on the chip (a march + an adjoint sweep)
        tab_c, src_c = live_stats_table(f"live_stats_contexts{K}.txt") if use_pmc else ({}, None)
        tab_a, src_a = live_stats_table("live_stats_march_b8.txt") if use_pmc else ({}, None)
        rp, rp_alone = rocprof_reference(tab_c), rocprof_reference(tab_a)

        def roof_of(k):
            kind, per = alg[k]
            sp = spans.get(k, np.zeros(0))
            live = sp[sp > noop_us + 2.5] if len(sp) else sp
            span_us = float(live.mean()) if len(live) else prof[k]["ms"] * 1e3 / max(prof[k]["launches"], 1)
            avg_s = max(span_us - pair_cost, 0.5) * 1e-6
            share = prof[k]["ms"] / max(sum(tot.values()), 1e-30)
            if kind == "hbm":
                ach = per / avg_s / 1e9
                d = dict(kernel=k, bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s",
                         frac=ach / HBM_PEAK_GBS, traffic=(pmc.get(k) if use_pmc else None),
                         traffic_source=(pmc_src if (use_pmc and k in pmc) else None), avg_us=avg_s * 1e6,
                         live_span_us=span_us, frac_span=per / (span_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                         launches=prof[k]["launches"], live_launches=int(len(live)),
                         noop_share=(1.0 - len(live) / len(sp)) if len(sp) else None,
                         algorithmic_bytes_per_launch=per, share_of_profiled_kernel_time=share)
                if k in rp:
                    d["rocprof_avg_us"] = rp[k]
                    d["rocprof_frac"] = per / (rp[k] * 1e-6) / 1e9 / HBM_PEAK_GBS
                    d["rocprof_source"] = src_c
                if k in rp_alone:      # 8 trajectories per launch, nothing else on the chip
                    per8 = per / Bc * 8
                    d["alone_b8_avg_us"] = rp_alone[k]
                    d["alone_b8_frac"] = per8 / (rp_alone[k] * 1e-6) / 1e9 / HBM_PEAK_GBS
                    d["alone_b8_source"] = src_a
                return d
            ach = per / avg_s / 1e12
            return dict(kernel=k, bound="mfma", achieved=ach, peak=FP64_MFMA_PEAK_TF, unit="TFLOP/s",
                        frac=ach / FP64_MFMA_PEAK_TF, traffic=None, traffic_source=None, avg_us=avg_s * 1e6,
                        launches=prof[k]["launches"], algorithmic_flops_per_launch=per,
                        share_of_profiled_kernel_time=share)
        ranked = sorted((k for k in tot if k in alg), key=lambda k: -tot[k])
        roof = roof_of(ranked[0]) if ranked else None
        extra["roofline_kernels"] = [roof_of(k) for k in ranked]
        # the Newton stencil kernel of the path: the residual / Schur right-hand side evaluation (the CG sweeps of
        # power-of-two grids apply the operator in the DCT basis and contain no stencil; GEMM-DCT grids use k_schur_p)
        for k in ("schur_p", "residual"):
            if k in tot and k in alg:
                extra["roofline_newton_stencil"] = roof_of(k)
                break
        extra["dct_path"] = ("in-LDS Stockham FFT of the even extension, plan 8x8x4x4 at 512^2 (8x8x8 / 8x8x8x4 at 256^2 / 1024^2)"
                             if engs[0].uses_fft else "MFMA f64 16x16x4 GEMM with DCT matrices")
        extra["roofline_note"] = ("one extra PGD iteration after the timed region, all contexts of the rank running; "
                                  f"{Bc} trajectories per launch")
    par.barrier(dist, dev)        # every rank leaves the measurement together (rank 0 has done its profiled iteration)

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and not a.dry_run:
        cpu = cpu_baseline(a.cpu_grid or N, dt, M, a.cpu_steps)

    if rank == 0:
        nl = sum(x[0] - y[0] for x, y in zip(c1, c0))
        ns = sum(x[1] - y[1] for x, y in zip(c1, c0))
        out = {
            "metric": "PGD iterations/sec (fwd Newton + adjoint + prox), 2D 512^2 grid",
            "value": value, "unit": "PGD iterations/s",
            "value_definition": "trajectory-iterations per second over all ranks (batch x iterations / s)",
            "n_gpus": max(world, 1), "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": 1e3 * el / a.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"2D {N}x{N}, {M} time steps, batch {B} trajectories per GPU "
                                   f"({total_traj} total), seeds 42+i, targets build_targets 1/1, u0 = 0",
                       "grid": N, "time_steps": M, "batch_per_gpu": B, "contexts_per_gpu": K,
                       "parallelism": f"batch-shard x{max(world, 1)}", "collective": collective},
            "roofline": roof, "cpu_baseline": cpu,
            "init_s": t_init, "J0_sum": float(J0[:, 4].sum()), "cost_sum_per_step": costs,
            "backtracking_forwards": attempts, "backtracking_forwards_per_step": attempts_per_step, "time_buckets_s": buckets,
            "kernel_launches_per_step": nl / max(a.steps, 1), "host_looks_per_step": ns / max(a.steps, 1),
            "source_hash": (None if a.dry_run else source_hash()),
        }
        out.update(extra)
        print(json.dumps(out), file=RESULT_OUT, flush=True)
        sys.stdout.flush()
    if comm is not None:
        comm.close()
    for e in engs:
        e.close()
    pool.shutdown()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
