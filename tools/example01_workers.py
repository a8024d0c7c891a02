#!/usr/bin/env python3
"""Example_01 (1506 points, 2D) end to end with 1 / 2 / 3 GPU contexts per rank (Model's gpu_workers): wall time."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from remo3d_amd.model import Model  # noqa: E402
if __name__ == "__main__":
    ex = os.path.join(ROOT, "tests", "golden", "examples", "Example_01")
    tools = ["B5.7A0.4M", "B4.48A1.62M", "M1.0A0.1B", "A2.0M0.5N", "N0.5M2.0A", "M4.0A0.5B"]
    depths = np.arange(0, 25.1, 0.1)
    ref = None
    for gw in (1, 2, 3, 1, 2):
        t0 = time.time()
        m = Model.compute_synthetic_logs(tools, depths, os.path.join(ex, "Input/Formation.txt"), os.path.join(ex, "Input/Borehole.txt"),
                                         gpu_workers=gw, verbose=False, mesh_workers=12)
        dt = time.time() - t0
        logs = np.array([m.logs[t][:, 1] for t in tools])
        if ref is None:
            ref = logs
        print("gpu_workers %d: %.2f s wall, solve %.2f s (summed over threads), mesh wait %.2f s, max rel diff to first run %.1e" %
              (gw, dt, m.timing["solve_s"], m.timing["mesh_s"], np.max(np.abs(logs - ref) / ref)), flush=True)
