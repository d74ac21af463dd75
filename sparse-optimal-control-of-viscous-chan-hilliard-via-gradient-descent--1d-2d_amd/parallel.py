"""Multi-GPU plumbing of the batch-sharded PGD run: trajectories are independent, so
ranks share nothing on the data path; per PGD iteration ONE all-reduce (sum) of the cost scalars
and, for the timing contract, one MAX-reduce of the elapsed time.  Backend-agnostic (RCCL on the
GPUs, gloo in the CPU tests): tensors live on `device`."""
from __future__ import annotations

import numpy as np


def shard_seeds(rank: int, world: int, batch_per_rank: int, base_seed: int = 42):
    """Block-contiguous partition of the global trajectory list (seed 42+i) over ranks."""
    start = base_seed + rank * batch_per_rank
    return list(range(start, start + batch_per_rank))


def allreduce_cost(J, dist=None, device="cpu"):
    """Global {J1,J2,J3,J4,J} sums from the per-trajectory costs [B][5] of this rank."""
    local = np.ascontiguousarray(np.asarray(J, dtype=np.float64).reshape(-1, 5).sum(axis=0))
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    import torch
    t = torch.from_numpy(local).to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def max_over_ranks(seconds: float, dist=None, device="cpu") -> float:
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier(dist=None, device="cpu"):
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
        if str(device).startswith("cuda"):
            import torch
            torch.cuda.synchronize()
