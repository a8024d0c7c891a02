#!/usr/bin/env python3
"""Example_02 of the reference (Example_02.py: domain_radius 25, batch_size 10) against its committed log under a few
settings: which ones does the log agree with best?  Writes per-point signed differences.  usage: ... OUT.json"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from remo3d_amd.model import Model  # noqa: E402

if __name__ == "__main__":
    ex = os.path.join(ROOT, "tests", "golden", "examples", "Example_02")
    tools = ["B5.7A0.4M", "B4.48A1.62M", "M1.0A0.1B", "A2.0M0.5N", "N0.5M2.0A", "M4.0A0.5B"]
    depths = np.arange(0, 25.1, 0.1)
    gold = np.loadtxt(os.path.join(ex, "Output/Results_2024_08_17__19_03_42/Results_1.txt"), skiprows=2)
    gold1 = np.loadtxt(os.path.join(ROOT, "tests", "golden", "examples", "Example_01", "Output/Results_2024_08_17__18_59_29/Results_1.txt"), skiprows=2)
    out = dict(depths=depths.tolist(), tools=tools, runs={})
    for R, bs in ((25, 10), (25, 5), (50, 10), (50, 5), (20, 10), (30, 10)):
        t0 = time.time()
        m = Model.compute_synthetic_logs(tools, depths, os.path.join(ex, "Input/Formation.txt"), os.path.join(ex, "Input/Borehole.txt"),
                                         borehole_geometry_type="diameter", gpu_workers=1, verbose=False, domain_radius=R, batch_size=bs, mesh_workers=12)
        ours = np.array([m.logs[t][:, 1] for t in tools])
        for name, g in (("Example_02 log", gold), ("Example_01 log", gold1)):
            rel = (ours - g[:, 1:7].T) / g[:, 1:7].T
            print("R=%g batch=%d vs %s: median %.2e p90 %.2e p99 %.2e max %.2e | per tool median %s" %
                  (R, bs, name, np.nanmedian(np.abs(rel)), np.nanpercentile(np.abs(rel), 90), np.nanpercentile(np.abs(rel), 99), np.nanmax(np.abs(rel)),
                   np.round(np.nanmedian(np.abs(rel), axis=1), 5)), flush=True)
        out["runs"]["R=%g batch=%d" % (R, bs)] = dict(seconds=time.time() - t0, signed_vs_example02=((ours - gold[:, 1:7].T) / gold[:, 1:7].T).tolist())
    json.dump(out, open(sys.argv[1], "w"))
