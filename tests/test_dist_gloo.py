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
