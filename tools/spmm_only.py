#!/usr/bin/env python3
"""Assemble one batch of the bench workload and launch only the SpMM kernel `reps` times with a
fixed variant - the target of rocprofv3 --pmc passes.  python tools/spmm_only.py S 5 variant lpr threads mapping grid reps"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from remo3d_amd import _lib, solver  # noqa: E402

size, k = sys.argv[1], int(sys.argv[2])
variant, lpr, threads, mapping, grid, reps = (int(v) for v in sys.argv[3:9])
wl = bench.build_workload(0, 1, 10, bench.SIZES[size])
w = wl["work"][0]
L = _lib.load()
ctx = solver.Context(0)
b = ctx.batch(w["mesh"], w["sigma"], w["sources"], w["evals"])
b.run(solver.make_opts(rtol=1e-1, maxsteps=20))
for key, v in enumerate((variant, lpr, threads, mapping, grid)):
    L.remo_debug_tune(key, v)
n = b.stats["n_free"]
x = np.random.default_rng(0).standard_normal((n, k))
y, ms = b.spmv(x if k > 1 else x[:, 0], reps=reps)
print(f"n={n} nnz={b.stats['nnz']} k={k} avg_ms={ms:.4f}")
