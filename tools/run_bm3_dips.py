#!/usr/bin/env python3
"""Benchmark model 3 at every dip the reference ships (0 ... 60 degrees) through Model.compute_synthetic_logs
on the default meshes (2D conforming at dip 0, revolved conforming 3D otherwise): robustness of the whole
path (NaN count, PCG behaviour) and the physical trend of the logs with dip.
usage: python tools/run_bm3_dips.py [mesh_scale] [n_depths] [out.json]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from remo3d_amd.model import Model  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
nd = int(sys.argv[2]) if len(sys.argv) > 2 else 20
out = sys.argv[3] if len(sys.argv) > 3 else None
ex = os.path.join(ROOT, "tests", "golden", "examples", "Benchmark models", "Benchmark model 3")
tools = ["A0.4M6.0N", "A2.0M0.5N"]
depths = np.linspace(5.0, 20.0, nd, endpoint=False)
res = {}
for dip in (0, 15, 30, 45, 60):
    t0 = time.time()
    m = Model.compute_synthetic_logs(tools, depths, os.path.join(ex, "Formation_BM3_%02d.txt" % dip), os.path.join(ex, "Borehole_BM3.txt"), dip=dip,
                                     gpu_workers=1, verbose=False, mesh_scale=scale)
    logs = {k: m.logs[k][:, 1] for k in tools}
    res[dip] = {k: v.tolist() for k, v in logs.items()}
    print("dip %2d: %6.1f s (mesh %.1f s, solve %.1f s), NaN %d, %s" % (
        dip, time.time() - t0, m.timing["mesh_s"], m.timing["solve_s"], sum(int(np.isnan(v).sum()) for v in logs.values()),
        "  ".join("%s %.2f..%.2f" % (k, np.nanmin(v), np.nanmax(v)) for k, v in logs.items())), flush=True)
if out:
    json.dump(dict(command="python tools/run_bm3_dips.py %g %d" % (scale, nd), depths=depths.tolist(), tools=tools, mesh_scale=scale, logs=res), open(out, "w"), indent=1)
