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


class CostComm:
    """The collective behind the C ABI (include/vch.h, vch_comm_*): RCCL communicator owned by the engine library,
    all-reduce of the device-resident cost scalars of one PGD iteration.  The 128-byte unique id travels from rank
    0 to the others over the process group the caller already has (`dist`); with one rank no group is needed."""

    def __init__(self, rank: int, world: int, device: int, dist=None):
        import ctypes as C
        from . import _lib
        self.lib = _lib.load()
        ident = (C.c_ubyte * 128)()
        err = None
        if rank == 0:
            try:
                _lib.check(self.lib.vch_comm_unique_id(ident))
            except Exception as exc:          # the other ranks must still get their broadcast
                err = str(exc)
        if world > 1:
            box = [None if err else bytes(ident)]
            dist.broadcast_object_list(box, src=0)
            if box[0] is None:
                raise _lib.VchError("rank 0 could not create an RCCL unique id" + (": " + err if err else ""))
            ident = (C.c_ubyte * 128).from_buffer_copy(box[0])
        elif err:
            raise _lib.VchError(err)
        self.h = self.lib.vch_comm_create(ident, int(rank), int(world), int(device))
        if not self.h:
            raise _lib.VchError("vch_comm_create failed: " + _lib.last_error())

    def allreduce(self, engines, iteration: int = -1):
        """Global {J1,J2,J3,J4,J} sums over the trajectories of `engines` (this rank) and over all ranks, for PGD
        iteration `iteration` (0-based count since pgd_init; -1 = the current iterate)."""
        import ctypes as C
        from . import _lib
        arr = (C.c_void_p * len(engines))(*[e.ctx for e in engines])
        out = np.zeros(5)
        _lib.check(self.lib.vch_comm_allreduce_cost(self.h, arr, len(engines), int(iteration), out.ctypes.data_as(_lib._D)))
        return out

    def close(self):
        if getattr(self, "h", None):
            self.lib.vch_comm_destroy(self.h)
            self.h = None


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
