#!/usr/bin/env python3
"""Chebyshev degree / interval ratio of the two-level preconditioner on 3D bench batches (20 depths = 8 batches), two rounds."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from remo3d_amd import solver  # noqa: E402

conf = len(sys.argv) > 2 and sys.argv[2] == "conforming"
wl = bench.build_workload(0, 1, 10, bench.SIZES[sys.argv[1] if len(sys.argv) > 1 else "S"], mesh_3d="conforming" if conf else "lattice")
ctx = solver.Context(0)
b3 = [ctx.batch(w["mesh"], w["sigma"], w["sources"], w["evals"]) for w in wl["work"]]
for rnd in range(1):
    for deg, ratio in ([(12, 300), (14, 450), (16, 600), (20, 900), (24, 1200), (16, 400)] if (len(sys.argv) > 2 and not conf) else [(5, 90), (6, 120), (8, 150), (10, 200), (12, 300), (14, 400), (8, 300), (10, 400)]):
        steps = 0; ms = 0.0
        for b in b3:
            b.run(solver.make_opts(coarse_degree=deg, coarse_ratio=ratio))
            steps += b.stats["pcg_steps"]; ms += b.stats["ms_solve"]
        print(f"round {rnd} deg {deg} ratio {ratio:3d}: steps {steps} solve {ms:.1f} ms (n {b3[0].stats['n_free']})", flush=True)
