#!/usr/bin/env python3
"""Which formation model produced which committed thin-bedded log?  The reference's README says Logs 1 are "unaffected by
boundary effects" and that formation model 2 is the one with the thick layers at top and bottom "to prevent the occurrence
of boundary effects"; the file numbering suggests Logs 1 <-> model 1.  Both pairings are computed on every 4th depth."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from remo3d_amd.model import Model  # noqa: E402

if __name__ == "__main__":
    base = os.path.join(ROOT, "tests", "golden", "examples", "Benchmark models", "Thin-bedded model")
    tools = ["A0.4M6.0N", "A1.62M6.0N", "A4.0M0.5N", "A8.0M1.0N"]
    depths = np.arange(0, 20.01, 0.25)[::4]
    ours = {}
    for fm in ("Formation_model_1.txt", "Formation_model_2.txt"):
        m = Model.compute_synthetic_logs(tools, depths, os.path.join(base, "Formation", fm), os.path.join(base, "Borehole", "Borehole_model_correct_rm.txt"),
                                         gpu_workers=1, verbose=False)
        ours[fm] = np.array([m.logs[t][:, 1] for t in tools])
    out = {}
    for logs in ("Logs 1", "Logs 2"):
        gold = np.loadtxt(os.path.join(base, "Logs", logs, "Results_1.txt"), skiprows=2)[::4, 1:5].T
        for fm in ours:
            rel = np.abs(ours[fm] - gold) / gold
            out[f"{logs} vs {fm}"] = dict(median=float(np.median(rel)), per_tool_median={t: float(np.median(rel[i])) for i, t in enumerate(tools)}, max=float(rel.max()))
            print(f"{logs} vs {fm}: median {np.median(rel):.2e}  per tool {[float('%.2e' % np.median(rel[i])) for i in range(4)]}  max {rel.max():.2e}", flush=True)
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)
