#!/usr/bin/env python3
"""The reference did not record the settings of its thin-bedded logs.  The signed difference of the 9 m lateral A8.0M1.0N
between our default run and the reference's Logs 1 is a sawtooth in depth, linear in the position of the current electrode
inside a batch of TEN depths - the signature of a grounded boundary close to the tool (image of an off-centre source) and of
batch_size = 10.  This scan re-runs all 81 depths x 4 tools for domain_radius x batch_size and prints the agreement."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from remo3d_amd.model import Model  # noqa: E402

base = os.path.join(ROOT, "tests", "golden", "examples", "Benchmark models", "Thin-bedded model")
tools = ["A0.4M6.0N", "A1.62M6.0N", "A4.0M0.5N", "A8.0M1.0N"]
depths = np.arange(0, 20.01, 0.25)
gold = np.loadtxt(os.path.join(base, "Logs", "Logs 1", "Results_1.txt"), skiprows=2)
out = {}
radii = [float(r) for r in sys.argv[2].split(",")] if len(sys.argv) > 2 else [15, 17.5, 20, 25, 50]
batches = [int(b) for b in sys.argv[3].split(",")] if len(sys.argv) > 3 else [10, 5]
for bs in batches:
    for R in radii:
        m = Model.compute_synthetic_logs(tools, depths, os.path.join(base, "Formation", "Formation_model_1.txt"),
                                         os.path.join(base, "Borehole", "Borehole_model_correct_rm.txt"), gpu_workers=1, verbose=False,
                                         domain_radius=R, batch_size=bs, mesh_workers=12)
        rel = np.array([(m.logs[t][:, 1] - gold[:, 1 + i]) / gold[:, 1 + i] for i, t in enumerate(tools)])
        out["R=%g batch=%d" % (R, bs)] = dict(median_abs=[float(np.nanmedian(np.abs(r))) for r in rel], max_abs=[float(np.nanmax(np.abs(r))) for r in rel],
                                              signed_mean=[float(np.nanmean(r)) for r in rel])
        print("R=%-5g batch=%-3d median |rel| %s  max %s  A8 signed mean %+.4f" % (R, bs, np.round(np.nanmedian(np.abs(rel), axis=1), 5),
                                                                                 np.round(np.nanmax(np.abs(rel), axis=1), 4), np.nanmean(rel[3])), flush=True)
json.dump(out, open(sys.argv[1], "w"), indent=1)
