#!/usr/bin/env python3
"""Example_01 (all 251 depths x 6 tools) against the reference's committed log at several mesh_scale values: per-point
relative differences (JSON) and percentiles.  usage: example01_scales.py OUT.json [scale ...]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from remo3d_amd.model import Model  # noqa: E402

if __name__ == "__main__":
    ex = os.path.join(ROOT, "tests", "golden", "examples", "Example_01")
    tools = ["B5.7A0.4M", "B4.48A1.62M", "M1.0A0.1B", "A2.0M0.5N", "N0.5M2.0A", "M4.0A0.5B"]
    depths = np.arange(0, 25.1, 0.1)
    gold = np.loadtxt(os.path.join(ex, "Output/Results_2024_08_17__18_59_29/Results_1.txt"), skiprows=2)
    scales = [float(s) for s in sys.argv[2:]] or [1.0, 0.7, 0.5]
    out = dict(depths=depths.tolist(), tools=tools, gold=gold[:, 1:7].T.tolist(), runs={})
    for scale in scales:
        t0 = time.time()
        m = Model.compute_synthetic_logs(tools, depths, os.path.join(ex, "Input/Formation.txt"), os.path.join(ex, "Input/Borehole.txt"),
                                         gpu_workers=1, verbose=False, mesh_scale=scale, mesh_workers=12)
        ours = np.array([m.logs[t][:, 1] for t in tools])
        rel = np.abs(ours - gold[:, 1:7].T) / gold[:, 1:7].T
        out["runs"][str(scale)] = dict(seconds=time.time() - t0, solve_s=m.timing["solve_s"], mesh_s=m.timing["mesh_s"], ours=ours.tolist())
        print(scale, "median %.2e p90 %.2e p99 %.2e max %.2e  (%.1f s, solve %.1f s)" % (np.nanmedian(rel), np.nanpercentile(rel, 90), np.nanpercentile(rel, 99),
                                                                                       np.nanmax(rel), time.time() - t0, m.timing["solve_s"]), flush=True)
    json.dump(out, open(sys.argv[1], "w"))
