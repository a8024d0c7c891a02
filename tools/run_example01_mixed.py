#!/usr/bin/env python3
"""Example_01 (every 5th depth, 6 tools) solved in fp64 and in mixed precision at the reference's CG tolerance (1e-8) and
in fp64 at 1e-12: how far the logs move with the precision mode and with the tolerance itself; all three sit at the
same distance from the reference's committed log."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from remo3d_amd.model import Model  # noqa: E402

ex = os.path.join(ROOT, "tests", "golden", "examples", "Example_01")
tools = ["B5.7A0.4M", "B4.48A1.62M", "M1.0A0.1B", "A2.0M0.5N", "N0.5M2.0A", "M4.0A0.5B"]
depths = np.arange(0, 25.1, 0.1)[::5]
gold = np.loadtxt(os.path.join(ex, "Output/Results_2024_08_17__18_59_29/Results_1.txt"), skiprows=2)[::5]
res = {}
for prec, rtol in (("fp64", 1e-8), ("mixed", 1e-8), ("fp64 rtol 1e-12", 1e-12)):
    m = Model.compute_synthetic_logs(tools, depths, os.path.join(ex, "Input/Formation.txt"), os.path.join(ex, "Input/Borehole.txt"),
                                     gpu_workers=1, verbose=False, precision=prec.split()[0], rtol=rtol, maxsteps=5000)
    res[prec] = np.array([m.logs[t][:, 1] for t in tools])
    rel = np.abs(res[prec] - gold[:, 1:7].T) / gold[:, 1:7].T
    print(f"{prec:16s}: solve {m.timing['solve_s']:.2f} s, vs reference log median {np.nanmedian(rel):.2e} max {np.nanmax(rel):.2e}, NaN {int(np.isnan(res[prec]).sum())}")
for a, b in (("mixed", "fp64"), ("fp64", "fp64 rtol 1e-12"), ("mixed", "fp64 rtol 1e-12")):
    d = np.abs(res[a] - res[b]) / np.abs(res[b])
    print(f"{a} vs {b}: median {np.median(d):.2e}, p99 {np.percentile(d, 99):.2e}, max {np.max(d):.2e}")
