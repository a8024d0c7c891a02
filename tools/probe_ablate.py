#!/usr/bin/env python3
"""Ablations of the edge-pair SpMM at k = 5 (results are wrong on purpose): where does the time go?
mode 0 full kernel, 1 no x gather, 2 x gather confined to 256 rows (L1 hits), 3 no reduction / store."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from remo3d_amd import _lib, solver  # noqa: E402

size = sys.argv[1] if len(sys.argv) > 1 else "S"
wl = bench.build_workload(0, 1, 10, bench.SIZES[size])
w = wl["work"][0]
L = _lib.load()
ctx = solver.Context(0)
b = ctx.batch(w["mesh"], w["sigma"], w["sources"], w["evals"])
b.run(solver.make_opts(rtol=1e-2))
n, nnz = b.stats["n_free"], b.stats["nnz"]
x = np.random.default_rng(0).standard_normal((n, 5))
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 3
L.remo_debug_tune(0, variant)
for grid in (1024, 768):
    L.remo_debug_tune(4, grid)
    for mode in (0, 1, 2, 3):
        L.remo_debug_tune(5, mode)
        ts = [b.spmv(x, reps=40)[1] for _ in range(3)]
        print(f"grid {grid} mode {mode}: {np.median(ts)*1e3:7.1f} us")
