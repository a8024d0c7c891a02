#!/usr/bin/env python3
"""A/B timing of the row schedules of the pair SpMM on one assembled batch: XCD windows (mapping 1) against XCD
regions with nc chunks per XCD (mapping 16 * nc).  Usage on the GPU box:  python tools/probe_regions.py [S|M|L] [k]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from remo3d_amd import _lib, solver  # noqa: E402

if __name__ == "__main__":
    size = sys.argv[1] if len(sys.argv) > 1 else "S"
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    wl = bench.build_workload(0, 1, 5, bench.SIZES[size])
    w = wl["work"][0]
    L = _lib.load()
    ctx = solver.Context(0)
    b = ctx.batch(w["mesh"], w["sigma"], w["sources"], w["evals"])
    b.run(solver.make_opts(rtol=1e-2))
    n, nnz = b.stats["n_free"], b.stats["nnz"]
    bytes_alg = 12.0 * nnz + 4.0 * n + 16.0 * k * n
    x = np.random.default_rng(0).standard_normal((n, k))
    mappings = [1, 0, 16, 32, 64, 128, 256, 512, 1024]
    ref = None
    res = {m: [] for m in mappings}
    for rnd in range(3):
        for m in mappings:
            L.remo_debug_tune(3, m)
            y, ms = b.spmv(x, reps=30)
            if ref is None:
                ref = y
            assert np.array_equal(y, ref), (m, float(np.max(np.abs(y - ref))))
            res[m].append(ms)
    print(f"size {size} n={n} nnz={nnz} k={k} algorithmic MB/launch={bytes_alg / 1e6:.1f}", flush=True)
    for m in mappings:
        ms = np.array(res[m])
        print(f"mapping {m:5d} (chunks per XCD {m >> 4:3d}): median {np.median(ms) * 1e3:8.1f} us  min {ms.min() * 1e3:8.1f} us -> "
              f"{bytes_alg / 1e9 / (np.median(ms) / 1e3):6.0f} GB/s", flush=True)
    L.remo_debug_tune(3, -1)
