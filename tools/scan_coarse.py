#!/usr/bin/env python3
"""Chebyshev degree / interval ratio of the two-level preconditioner: total PCG steps and solve time on the
2D benchmark (BM1, 100 depths) and on one 3D bench batch.  python tools/scan_coarse.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from remo3d_amd import solver  # noqa: E402
from remo3d_amd.model import Model  # noqa: E402

orig = solver.make_opts
ex = os.path.join(ROOT, "tests", "golden", "examples", "Benchmark models", "Benchmark model 1")
depths = np.linspace(5, 55, 100)
wl = bench.build_workload(0, 1, 10, bench.SIZES["S"])
ctx = solver.Context(0)
b3 = [ctx.batch(w["mesh"], w["sigma"], w["sources"], w["evals"]) for w in wl["work"]]
for deg, ratio in [(6, 15), (6, 30), (6, 45), (6, 60), (6, 90), (6, 150), (5, 45), (5, 90), (8, 60), (8, 120), (4, 45)]:
    solver.make_opts = lambda **kw: orig(**{**kw, "coarse_degree": deg, "coarse_ratio": ratio})
    t0 = time.time()
    m = Model.compute_synthetic_logs(["A0.4M6.0N"], depths, os.path.join(ex, "Formation_BM1.txt"), os.path.join(ex, "Borehole_BM1.txt"),
                                     gpu_workers=1, verbose=False)
    t2d = m.timing["solve_s"]
    steps = 0; ms = 0.0
    for b in b3:
        b.run(orig(coarse_degree=deg, coarse_ratio=ratio))
        steps += b.stats["pcg_steps"]; ms += b.stats["ms_solve"]
    print(f"deg {deg} ratio {ratio:3d}: 2D BM1 solve {t2d:.3f} s | 3D S ({len(b3)} batches) steps {steps} solve {ms:.1f} ms", flush=True)
