#!/usr/bin/env python3
"""Example_01 against the reference's committed log as the in-repo mesh is refined (mesh_scale 1, 0.7, 0.5) on every 5th
depth: does the difference behave like a discretisation error of OUR mesh (falls with h) or level off at the reference's?"""
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
    sel = np.arange(0, len(depths), 5)
    gold = np.loadtxt(os.path.join(ex, "Output/Results_2024_08_17__18_59_29/Results_1.txt"), skiprows=2)[sel]
    out = {}
    prev = None
    for scale in (1.0, 0.7, 0.5):
        t0 = time.time()
        m = Model.compute_synthetic_logs(tools, depths[sel], os.path.join(ex, "Input/Formation.txt"), os.path.join(ex, "Input/Borehole.txt"),
                                         gpu_workers=1, verbose=False, mesh_scale=scale, mesh_workers=8)
        ours = np.array([m.logs[t][:, 1] for t in tools])
        rel = np.abs(ours - gold[:, 1:7].T) / gold[:, 1:7].T
        out[str(scale)] = dict(seconds=time.time() - t0, median=float(np.nanmedian(rel)), p90=float(np.nanpercentile(rel, 90)), max=float(np.nanmax(rel)),
                               self_change_median=None if prev is None else float(np.nanmedian(np.abs(ours - prev) / np.abs(prev))),
                               self_change_max=None if prev is None else float(np.nanmax(np.abs(ours - prev) / np.abs(prev))))
        prev = ours
        print(scale, json.dumps(out[str(scale)]), flush=True)
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)
