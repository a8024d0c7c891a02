#!/usr/bin/env python3
"""End-to-end parity on the reference's thin-bedded benchmark (140 / 201 layers of ~0.125 m, 81 depths
x 4 tools): Model.compute_synthetic_logs on the GPU vs the reference's committed logs
(Examples/Benchmark models/Thin-bedded model/Logs/Logs {1,2}/Results_1.txt; settings not recorded by
the reference, defaults assumed).  Logs 1 = formation model 1, Logs 2 = formation model 2, aligned depths."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from remo3d_amd.model import Model  # noqa: E402

base = os.path.join(ROOT, "tests", "golden", "examples", "Benchmark models", "Thin-bedded model")
tools = ["A0.4M6.0N", "A1.62M6.0N", "A4.0M0.5N", "A8.0M1.0N"]
depths = np.arange(0, 20.01, 0.25)
out = {}
for logs, formation in (("Logs 1", "Formation_model_1.txt"), ("Logs 2", "Formation_model_2.txt")):
    gold = np.loadtxt(os.path.join(base, "Logs", logs, "Results_1.txt"), skiprows=2)
    t0 = time.time()
    m = Model.compute_synthetic_logs(tools, depths, os.path.join(base, "Formation", formation),
                                     os.path.join(base, "Borehole", "Borehole_model_correct_rm.txt"), gpu_workers=1, verbose=False)
    rel = np.array([np.abs(m.logs[t][:, 1] - gold[:, 1 + i]) / gold[:, 1 + i] for i, t in enumerate(tools)])
    out[logs] = dict(points=int(rel.size), seconds=time.time() - t0, mesh_s=m.timing["mesh_s"], solve_s=m.timing["solve_s"],
                     nan=int(np.isnan(rel).sum()), median_rel_diff=float(np.nanmedian(rel)), p90=float(np.nanpercentile(rel, 90)),
                     max_rel_diff=float(np.nanmax(rel)), per_tool_median={t: float(np.nanmedian(rel[i])) for i, t in enumerate(tools)})
print(json.dumps(out))
