#!/usr/bin/env python3
"""Interleaved A/B timing of the SpMM variants on one assembled batch (one process, several rounds;
cdna_hip_programming.md rule 24).  Usage on the GPU box:  python tools/probe_spmm.py [S|M] [k]"""
import json
import sys
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from remo3d_amd import _lib, solver  # noqa: E402

size = sys.argv[1] if len(sys.argv) > 1 else "S"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 5
wl = bench.build_workload(0, 1, 10, bench.SIZES[size])
w = wl["work"][0]
L = _lib.load()
ctx = solver.Context(0)
b = ctx.batch(w["mesh"], w["sigma"], w["sources"], w["evals"])
b.run(solver.make_opts(rtol=1e-2))
n, nnz = b.stats["n_free"], b.stats["nnz"]
bytes_alg = 12.0 * nnz + 4.0 * n + 16.0 * k * n
rng = np.random.default_rng(0)
x = rng.standard_normal((n, k))


def tune(variant, lpr, threads, mapping, grid):
    for key, v in enumerate((variant, lpr, threads, mapping, grid)):
        L.remo_debug_tune(key, v)


configs = []
for variant, lpr, threads, mapping, grid in [
    (1, 16, 256, 0, 1024), (3, 16, 256, 0, 1024), (3, 16, 256, 1, 1024), (3, 16, 256, 1, 896), (3, 16, 256, 1, 768), (3, 16, 256, 1, 640), (3, 16, 256, 1, 512), (3, 16, 256, 1, 256),
]:
    configs.append(dict(variant=variant, lpr=lpr, threads=threads, mapping=mapping, grid=grid))

ref = None
res = {i: [] for i in range(len(configs))}
for rnd in range(3):
    for i, c in enumerate(configs):
        tune(c["variant"], c["lpr"], c["threads"], c["mapping"], c["grid"])
        y, ms = b.spmv(x if k > 1 else x[:, 0], reps=40)
        if ref is None:
            ref = y
        err = float(np.max(np.abs(y - ref)) / np.max(np.abs(ref)))
        assert err < 1e-12, (c, err)
        res[i].append(ms)
print(f"n={n} nnz={nnz} k={k} algorithmic MB/launch={bytes_alg/1e6:.1f}")
for i, c in enumerate(configs):
    ms = np.array(res[i])
    print(f"{json.dumps(c):80s} median {np.median(ms)*1e3:7.1f} us  min {ms.min()*1e3:7.1f} us  -> {bytes_alg/1e9/(np.median(ms)/1e3):7.0f} GB/s")
# full solve timing: Jacobi vs two-level with several Chebyshev degrees / intervals
tune(0, 0, 0, -1, 0)
for pre, deg, ratio in [("local", 0, 0), ("multigrid", 4, 8), ("multigrid", 6, 15), ("multigrid", 8, 20)]:
    for rnd in range(2):
        b.run(solver.make_opts(preconditioner=pre, rtol=1e-8, time_kernels=True, coarse_degree=deg, coarse_ratio=ratio))
    st = b.stats
    print("solve %s deg=%d ratio=%d: steps" % (pre, deg, ratio), st["pcg_steps"], "max its", st["max_iterations"], "ms_solve %.2f" % st["ms_solve"],
          "us/step %.1f" % (1e3 * st["ms_solve"] / st["pcg_steps"]), "spmv us %.1f" % (1e3 * st["spmv_ms"] / st["spmv_launches"]))
