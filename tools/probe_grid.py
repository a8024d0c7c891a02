#!/usr/bin/env python3
"""Grid / workgroup-size sweep of the pair SpMM (probe path: no partial sums, so grids above 1024 are allowed).
Usage on the GPU box:  python tools/probe_grid.py [S|M|L]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from remo3d_amd import _lib, solver  # noqa: E402

if __name__ == "__main__":
    size = sys.argv[1] if len(sys.argv) > 1 else "S"
    k = 5
    wl = bench.build_workload(0, 1, 5, bench.SIZES[size])
    w = wl["work"][0]
    L = _lib.load()
    ctx = solver.Context(0)
    b = ctx.batch(w["mesh"], w["sigma"], w["sources"], w["evals"])
    b.run(solver.make_opts(rtol=1e-2))
    n, nnz = b.stats["n_free"], b.stats["nnz"]
    bytes_alg = 12.0 * nnz + 4.0 * n + 16.0 * k * n
    x = np.random.default_rng(0).standard_normal((n, k))
    configs = [(256, 1024), (256, 1280), (256, 1536), (256, 2048), (256, 4096), (512, 512), (512, 640), (128, 2048), (128, 2560)]
    ref = None
    res = {c: [] for c in configs}
    for rnd in range(3):
        for c in configs:
            L.remo_debug_tune(2, c[0]); L.remo_debug_tune(4, c[1])
            y, ms = b.spmv(x, reps=30)
            if ref is None:
                ref = y
            assert np.array_equal(y, ref), (c, float(np.max(np.abs(y - ref))))
            res[c].append(ms)
    print(f"size {size} n={n} nnz={nnz} k={k} algorithmic MB/launch={bytes_alg / 1e6:.1f}", flush=True)
    for c in configs:
        ms = np.array(res[c])
        print(f"threads {c[0]:4d} grid {c[1]:5d}: median {np.median(ms) * 1e3:8.1f} us  min {ms.min() * 1e3:8.1f} us -> {bytes_alg / 1e9 / (np.median(ms) / 1e3):6.0f} GB/s", flush=True)
