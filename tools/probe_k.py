#!/usr/bin/env python3
"""SpMM time as a function of the number of interleaved right-hand sides k (one assembled batch).
Separates the per-entry streaming cost from the per-column gather / reduce cost.
Usage on the GPU box:  python tools/probe_k.py [S|M]"""
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
rng = np.random.default_rng(0)
print(f"n={n} nnz={nnz}")
ref = {}
for variant in (1, 3):
    L.remo_debug_tune(0, variant)
    for k in (1, 2, 3, 4, 5, 6, 7, 8):
        x = np.random.default_rng(k).standard_normal((n, k))
        ts = []
        for rnd in range(3):
            y, ms = b.spmv(x if k > 1 else x[:, 0], reps=40)
            ts.append(ms)
        y = np.asarray(y).reshape(n, -1)
        if k in ref:
            err = float(np.max(np.abs(y - ref[k])) / np.max(np.abs(ref[k])))
            assert err < 1e-12, (variant, k, err)
        else:
            ref[k] = y
        by = 12.0 * nnz + 4.0 * n + 16.0 * k * n
        print(f"variant {variant} k={k}: {np.median(ts)*1e3:7.1f} us  {by/1e9/(np.median(ts)/1e3):7.0f} GB/s")
