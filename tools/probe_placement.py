#!/usr/bin/env python3
"""Does the SpMM's time depend on WHERE its arrays land in device memory?  One batch of the bench workload (size S), the SpMM alone
(remo_batch_spmv, 50 launches), with dummy allocations of different sizes made before the context's arena exists.
The boxes of the pool run this kernel at 49 or at 59 us while every generic bandwidth probe reads the same on both kinds."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from remo3d_amd import solver  # noqa: E402

hip = C.CDLL("libamdhip64.so")
wl = bench.build_workload(0, 1, 10, bench.SIZES["S"], max_batches=1)
w = wl["work"][0]
rng = np.random.default_rng(0)
pads = []
for pad_mb in [0, 64, 200, 512, 1000, 1536, 3000, 7000]:
    p = C.c_void_p()
    if pad_mb:
        assert hip.hipMalloc(C.byref(p), C.c_size_t(pad_mb << 20)) == 0
        pads.append(p)
    with solver.Context(0) as ctx:
        b = ctx.batch(w["mesh"], w["sigma"], w["sources"], w["evals"])
        b.run(solver.make_opts(rtol=1e-1, maxsteps=20))
        n = b.stats["n_free"]
        x = rng.standard_normal((n, 5))
        res = []
        for rep in range(3):
            y, ms = b.spmv(x, reps=50)
            res.append(1e3 * ms)
        print("after a further dummy allocation of %5d MB (%d held): SpMM %s us" % (pad_mb, len(pads), " ".join("%.1f" % v for v in res)), flush=True)
        b.close()
