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


_STORE = None          # the TCPStore the sweep owns (dynamic schedule); created once per process group, collectively
_STORE_GROUP = None    # the process group it was made for
_SWEEP_SEQ = 0         # rank 0's count of dynamic queues: the id it broadcasts names the shared counter


def _own_store():
    """A TCPStore of the sweep's own for the shared batch counter: rank 0 hosts it on a free local port and tells the
    others through one broadcast (collective: every rank of the group must get here).  The process group's private
    rendezvous store is not touched.  Rank 0 tries a few ports (another process can take a probed port before the store binds
    it) and broadcasts a (port | error) record, so that a failure raises on EVERY rank instead of leaving the others in the
    broadcast; the store is kept per process group (a new group makes a new store)."""
    global _STORE, _STORE_GROUP
    import datetime
    import socket
    import torch.distributed as dist
    group = dist.distributed_c10d._get_default_group() if hasattr(dist, "distributed_c10d") else None
    if _STORE is not None and _STORE_GROUP is group:
        return _STORE
    _STORE = None
    host = os.environ.get("REMO_STORE_ADDR", os.environ.get("MASTER_ADDR", "127.0.0.1"))
    box = [None]
    if dist.get_rank() == 0:
        err = None
        for _ in range(8):
            s = socket.socket()
            s.bind(("", 0))
            port = s.getsockname()[1]
            s.close()
            try:
                _STORE = dist.TCPStore(host, port, world_size=dist.get_world_size(), is_master=True, wait_for_workers=False,
                                       timeout=datetime.timedelta(seconds=300))
                box[0] = ("port", port)
                break
            except Exception as ex:      # the port went to somebody else in between (or MASTER_ADDR is not this host): next port
                err = "%s: %s" % (type(ex).__name__, ex)
        if box[0] is None:
            box[0] = ("error", "rank 0 could not host the sweep's TCPStore on %s: %s" % (host, err))
    dist.broadcast_object_list(box, src=0)
    kind, payload = box[0]
    if kind == "error":
        raise RuntimeError(payload)
    if dist.get_rank() != 0:
        _STORE = dist.TCPStore(host, int(payload), world_size=dist.get_world_size(), is_master=False,
                               timeout=datetime.timedelta(seconds=300))
    _STORE_GROUP = group
    return _STORE


class BatchQueue:
    """Iterator over the batch indices this rank should take.

    schedule "static": the block-cyclic share (my_share).  schedule "dynamic": the reference's pull scheduling
    (remo3d.py:843-860: a worker asks the master for the next task index whenever it is free) - here every free rank
    draws the next index from ONE shared counter, an atomic fetch-add on a TCPStore the sweep owns (rank 0 hosts it: a
    pull is one local round trip, ~0.1 ms against ~20 ms of a 3D batch).  Without a process group both schedules are
    range(n).  Constructing a dynamic queue is COLLECTIVE: rank 0 broadcasts the id that names the counter, so all ranks
    draw from the same one by construction (a rank that builds a queue the others do not build blocks in the broadcast
    instead of silently computing every batch itself); `check_complete()` after the sweep confirms through a collective
    that the ranks together took every batch exactly once."""

    def __init__(self, n_batches: int, schedule: str = "static"):
        global _SWEEP_SEQ
        if schedule not in ("static", "dynamic"):
            raise ValueError("schedule must be 'static' or 'dynamic'")
        self.n = int(n_batches)
        self.schedule = schedule
        self.taken = []
        self.last_drawn = -1
        self._key = None
        self._store = None
        d = _dist()
        if schedule == "dynamic" and d is not None and d.get_world_size() > 1:
            self._store = _own_store()
            box = [None]
            if d.get_rank() == 0:
                _SWEEP_SEQ += 1
                box[0] = "remo3d_batch_queue_%d_%d" % (os.getpid(), _SWEEP_SEQ)
            d.broadcast_object_list(box, src=0)
            self._key = box[0]

    def remaining_hint(self) -> int:
        """Batches not yet handed out (as of this rank's last draw; exact for the static share)."""
        if self._store is None:
            return max(0, len(range(rank(), self.n, world_size())) - len(self.taken))
        return max(0, self.n - 1 - self.last_drawn)

    def __iter__(self):
        if self._store is None:
            for i in (my_share(self.n) if self.schedule == "static" else range(rank(), self.n, world_size())):
                self.taken.append(i)
                yield i
            return
        while True:
            i = int(self._store.add(self._key, 1)) - 1      # fetch-add: every index is handed out exactly once
            self.last_drawn = i
            if i >= self.n:
                return
            self.taken.append(i)
            yield i

    def check_complete(self) -> int:
        """Collective: the ranks together took n batches (a rank that drew nothing takes part with 0).  Returns the total;
        raises if batches were lost or taken twice (diverged counters would otherwise SUM into multiplied logs)."""
        d = _dist()
        mine = len(self.taken)
        if d is None or d.get_world_size() == 1:
            total = mine
        else:
            import torch
            t = torch.tensor([float(mine)], dtype=torch.float64)
            if d.get_backend() == "nccl":
                t = t.cuda()
            d.all_reduce(t, op=d.ReduceOp.SUM)
            total = int(round(float(t.item())))
        if total != self.n:
            raise RuntimeError("batch queue: the ranks took %d batches in all, the sweep has %d" % (total, self.n))
        return total


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
