#!/usr/bin/env python3
"""End-to-end wall clock of Model.compute_synthetic_logs (windowing + meshing + solve): inline meshing vs
mesh worker processes, one vs several GPU contexts (gpu_workers): 2D BM1 (100 depths, one tool) and 3D BM3 dip 30 (40 depths, two tools), conforming meshes.
usage: python tools/run_end_to_end.py [workers]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from remo3d_amd.model import Model  # noqa: E402

if __name__ == "__main__":
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    bm = os.path.join(ROOT, "tests", "golden", "examples", "Benchmark models")
    cases = [("2D BM1, 100 depths x 1 tool", ["A0.4M6.0N"], np.linspace(5, 55, 100), os.path.join(bm, "Benchmark model 1", "Formation_BM1.txt"),
              os.path.join(bm, "Benchmark model 1", "Borehole_BM1.txt"), 0),
             ("3D BM3 dip 30, 40 depths x 2 tools", ["A0.4M6.0N", "A2.0M0.5N"], np.linspace(5, 20, 40, endpoint=False),
              os.path.join(bm, "Benchmark model 3", "Formation_BM3_30.txt"), os.path.join(bm, "Benchmark model 3", "Borehole_BM3.txt"), 30)]
    for name, tools, depths, form, bore, dip in cases:
        ref = None
        for w, g in ((0, 1), (workers, 1), (workers, 2), (workers, 3), (0, 2), (workers, 1), (workers, 2)):
            t0 = time.time()
            m = Model.compute_synthetic_logs(tools, depths, form, bore, dip=dip, gpu_workers=g, verbose=False, mesh_workers=w)
            dt = time.time() - t0
            logs = np.stack([m.logs[k][:, 1] for k in tools])
            if ref is None:
                ref = logs
            n = logs.size
            print(f"{name}: mesh_workers={w} gpu_workers={g}: {dt:6.2f} s -> {n / dt:6.1f} points/s end to end (solve {m.timing['solve_s']:.2f} s, "
                  f"mesh wait {m.timing['mesh_s']:.2f} s), identical logs: {bool(np.array_equal(logs, ref))}", flush=True)
