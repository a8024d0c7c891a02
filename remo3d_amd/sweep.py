"""Depth-sweep partitioning across GPUs: replaces the MPI task farm of remo3d.py:592-599, 809-865
and workers/worker.py:20-145.

Batches are independent (each has its own mesh, matrix and right-hand sides; worker.py:74-138), so
the path shards with no data-path collective: rank r of W takes batches r, r+W, r+2W, ...
(block-cyclic: neighbouring batches have similar cost, so the shares are balanced without the
reference's pull scheduling, remo3d.py:847-852).  The only exchange is the final gather of the
[n_depths, n_tools] apparent-resistivity slab, done as ONE all-reduce(sum) over RCCL / xGMI of a
slab every rank fills only at its own entries (16 kB at 1000 depths x 2 tools: pure latency).
NaN entries (failed batches, worker.py:135-138) survive the sum.
"""
from __future__ import annotations

import os
from typing import Iterable, Optional

import numpy as np


def _dist():
    try:
        import torch.distributed as dist
    except Exception:
        return None
    return dist if dist.is_available() and dist.is_initialized() else None


def rank() -> int:
    d = _dist()
    return d.get_rank() if d else 0


def world_size() -> int:
    d = _dist()
    return d.get_world_size() if d else 1


def my_share(n_batches: int, r: Optional[int] = None, w: Optional[int] = None) -> Iterable[int]:
    """Batch indices of this rank (block-cyclic)."""
    r = rank() if r is None else r
    w = world_size() if w is None else w
    return range(r, n_batches, w)


_QUEUE_SEQ = 0


class BatchQueue:
    """Iterator over the batch indices this rank should take.

    schedule "static": the block-cyclic share (my_share).  schedule "dynamic": the reference's pull scheduling
    (remo3d.py:843-860: a worker asks the master for the next task index whenever it is free) - here every free rank
    draws the next index from ONE shared counter, an atomic fetch-add on the process group's rendezvous store (the
    TCPStore torchrun already runs on the node: a pull is one local round trip, ~0.1 ms against ~20 ms of a 3D batch).
    Without a process group both schedules are range(n).  Every rank must construct its queues in the same order (the
    counter's key is a per-process sequence number)."""

    def __init__(self, n_batches: int, schedule: str = "static"):
        global _QUEUE_SEQ
        if schedule not in ("static", "dynamic"):
            raise ValueError("schedule must be 'static' or 'dynamic'")
        self.n = int(n_batches)
        self.schedule = schedule
        self.taken = []
        _QUEUE_SEQ += 1
        self._key = "remo3d_batch_queue_%d" % _QUEUE_SEQ
        self._store = None
        d = _dist()
        if schedule == "dynamic" and d is not None and d.get_world_size() > 1:
            from torch.distributed import distributed_c10d
            self._store = distributed_c10d._get_default_store()

    def __iter__(self):
        if self._store is None:
            for i in (my_share(self.n) if self.schedule == "static" else range(rank(), self.n, world_size())):
                self.taken.append(i)
                yield i
            return
        while True:
            i = int(self._store.add(self._key, 1)) - 1      # fetch-add: every index is handed out exactly once
            if i >= self.n:
                return
            self.taken.append(i)
            yield i


def gather_floats(x) -> list:
    """One small float vector per rank -> list over ranks (per-rank busy times: how balanced was the sweep)."""
    d = _dist()
    x = [float(v) for v in np.atleast_1d(x)]
    if d is None or d.get_world_size() == 1:
        return [x]
    import torch
    t = torch.tensor(x, dtype=torch.float64)
    if d.get_backend() == "nccl":
        t = t.cuda()
    out = [torch.empty_like(t) for _ in range(d.get_world_size())]
    d.all_gather(out, t)
    return [o.cpu().tolist() for o in out]


def init_from_env(backend: Optional[str] = None) -> bool:
    """Initialise torch.distributed from the torchrun environment (RANK / WORLD_SIZE / MASTER_*).
    backend: "nccl" (= RCCL on ROCm) when a GPU is visible, else "gloo".  Returns True if a
    process group is active afterwards."""
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        return False
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return True
    if backend is None:
        backend = os.environ.get("REMO_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("REMO_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    dist.init_process_group(backend=backend)
    return True


def combine(slab: np.ndarray) -> np.ndarray:
    """Sum of the per-rank slabs (each rank has zeros outside its own entries)."""
    d = _dist()
    if d is None or d.get_world_size() == 1:
        return slab
    import torch
    t = torch.from_numpy(np.ascontiguousarray(slab, dtype=np.float64))
    if d.get_backend() == "nccl":
        t = t.cuda()
    d.all_reduce(t, op=d.ReduceOp.SUM)
    return t.cpu().numpy()


def barrier():
    d = _dist()
    if d is not None:
        d.barrier()


def max_over_ranks(x: float) -> float:
    d = _dist()
    if d is None or d.get_world_size() == 1:
        return x
    import torch
    t = torch.tensor([x], dtype=torch.float64)
    if d.get_backend() == "nccl":
        t = t.cuda()
    d.all_reduce(t, op=d.ReduceOp.MAX)
    return float(t.item())
