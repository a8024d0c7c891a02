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
