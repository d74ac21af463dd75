"""The N > 1 path on CPU: two gloo ranks exercise the batch sharding, the single cost
all-reduce and the max-over-ranks timing used by bench.py (no GPU work)."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, q):
    import torch.distributed as dist
    import vch_amd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    par = vch_amd.parallel
    seeds = par.shard_seeds(rank, world, B)
    J = np.array([[1.0 * s, 2.0, 3.0, 4.0, s + 9.0] for s in seeds])     # per-trajectory {J1..J4, J}
    tot = par.allreduce_cost(J, dist, "cpu")
    par.barrier(dist, "cpu")
    tmax = par.max_over_ranks(1.0 + rank, dist, "cpu")
    q.put((rank, seeds, tot.tolist(), tmax))
    dist.destroy_process_group()


def test_two_rank_sharding_and_cost_allreduce():
    import torch.multiprocessing as mp
    world, B = 2, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    all_seeds = res[0][1] + res[1][1]
    assert all_seeds == list(range(42, 42 + world * B))                  # disjoint, block-contiguous, seeds 42+i
    expect = [float(sum(all_seeds)), 2.0 * 6, 3.0 * 6, 4.0 * 6, float(sum(all_seeds)) + 9.0 * 6]
    for r in res:
        assert np.allclose(r[2], expect) and r[3] == 2.0                 # same global sums on every rank; MAX time


def test_single_rank_is_a_no_op():
    import vch_amd
    J = np.arange(10.0).reshape(2, 5)
    assert np.allclose(vch_amd.parallel.allreduce_cost(J), J.sum(axis=0))
    assert vch_amd.parallel.max_over_ranks(3.5) == 3.5


def _run_bench(args, nproc, timeout=300):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    if nproc > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py")] + args
    else:
        cmd = [sys.executable, os.path.join(root, "bench.py")] + args
    return subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_skeleton_two_ranks_dry_run():
    """bench.py's multi-rank skeleton under torchrun with two gloo ranks and a stand-in engine (--dry-run): rendezvous,
    barriers around the timed region, one all-reduce per step in step order with two contexts per rank, MAX over
    ranks, rank 0's extra (roofline) leg before the closing barrier, and a teardown every rank reaches -- one JSON
    line, exit code 0, nobody left inside a collective."""
    import json
    r = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run", "--contexts", "2", "--batch-per-gpu", "4"], 2)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak"
    assert out["config"]["collective"] == "gloo" and out["config"]["batch_per_gpu"] == 4
    # the stand-in engine reports cost 1/k per trajectory at its k-th iteration: 8 trajectories over both ranks
    assert np.allclose(out["cost_sum_per_step"], [8.0 / k for k in (2, 3, 4)])
    assert out["value"] > 0 and out["ms_per_step"] > 0


def test_bench_failing_rank_does_not_hang_its_peer():
    """A rank whose engine raises inside an iteration exits non-zero at once (bench.py main), so torchrun ends the
    job instead of leaving the other rank blocked in the next all-reduce."""
    r = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "0", "--dry-run", "--contexts", "1", "--batch-per-gpu", "2",
                    "--dry-run-fail-rank", "1"], 2, timeout=240)
    assert r.returncode != 0
    assert "injected failure" in (r.stdout + r.stderr)


def _run_bench_bare(args, timeout=300):
    """`python bench.py --gpus N ...` with NO launcher and no rendezvous variables in the environment: bench.py starts
    its own ranks (spawn_ranks)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, cwd=root, env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_bench_bare_command_spawns_its_own_ranks():
    """The driver's command form, `python bench.py --gpus 2 ...` without torchrun: the parent starts two child ranks
    (gloo here), relays rank 0's single JSON line and exits 0."""
    import json
    r = _run_bench_bare(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run", "--contexts", "2", "--batch-per-gpu", "4"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak"
    assert out["config"]["collective"] == "gloo" and out["config"]["parallelism"] == "batch-shard x2"
    assert np.allclose(out["cost_sum_per_step"], [8.0 / k for k in (2, 3, 4)])


def test_bench_bare_command_failing_rank_exits_nonzero():
    r = _run_bench_bare(["--gpus", "2", "--steps", "2", "--warmup", "0", "--dry-run", "--contexts", "1", "--batch-per-gpu", "2",
                         "--dry-run-fail-rank", "1"], timeout=240)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_source_hash_drives_rebuild(tmp_path):
    """Build staleness is decided by a content hash of the sources recorded beside the library, not by mtimes."""
    import importlib
    import vch_amd
    L = importlib.import_module(vch_amd.PKG_NAME + "._lib")
    vch_amd.build()
    assert not L._stale()
    h = L.source_hash()
    assert len(h) == 64 and open(L.HASH_PATH).read().strip() == h
    # touching a source without changing it does not make the library stale; a recorded hash that differs does
    src = os.path.join(L.CSRC, "vch_common.h")
    os.utime(src, None)
    assert not L._stale()
    saved = open(L.HASH_PATH).read()
    try:
        open(L.HASH_PATH, "w").write("0" * 64 + "\n")
        assert L._stale()
    finally:
        open(L.HASH_PATH, "w").write(saved)
    assert not L._stale()
