#!/usr/bin/env python3
"""Exercise the RCCL code path of remo3d_amd.sweep (backend "nccl", CUDA tensors) with a one-rank process group on the
GPU box: the 8-GPU run is the driver's, this only proves that combine / barrier / max_over_ranks execute on the device."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from remo3d_amd import sweep  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", world_size=1, rank=0)
slab = np.arange(12, dtype=np.float64).reshape(6, 2)
slab[3, 1] = np.nan
out = sweep.combine(slab)
assert np.array_equal(np.isnan(out), np.isnan(slab)) and np.allclose(out[~np.isnan(out)], slab[~np.isnan(slab)])
sweep.barrier()
assert sweep.max_over_ranks(3.5) == 3.5
print("nccl single-rank path ok:", dist.get_backend(), sweep.rank(), sweep.world_size())
dist.destroy_process_group()
