#!/usr/bin/env python3
"""Which unrecorded setting of the reference's thin-bedded run explains the systematic offset of the 9 m lateral A8.0M1.0N
(+0.3 ... +3.4 % in our logs against the reference's Logs 1)?  The reference's README does not give the
settings; the HIP path and the CPU oracle agree with each other, and our result does not move under refinement.  This
script re-runs a sample of depths under one changed setting at a time and prints the signed relative differences."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from remo3d_amd.model import Model  # noqa: E402

base = os.path.join(ROOT, "tests", "golden", "examples", "Benchmark models", "Thin-bedded model")
tools = ["A0.4M6.0N", "A1.62M6.0N", "A4.0M0.5N", "A8.0M1.0N"]
depths = np.arange(0, 20.01, 1.25)
rows = np.rint(depths / 0.25).astype(int)
gold = np.loadtxt(os.path.join(base, "Logs", "Logs 1", "Results_1.txt"), skiprows=2)
bore = lambda n: os.path.join(base, "Borehole", n)
cases = [("defaults (R=50, batch 5)", dict()),
         ("R=100", dict(domain_radius=100)), ("R=25", dict(domain_radius=25)), ("R=20", dict(domain_radius=20)),
         ("R=15", dict(domain_radius=15)), ("R=12", dict(domain_radius=12)), ("R=10", dict(domain_radius=10)),
         ("batch_size=1", dict(batch_size=1)), ("batch_size=10", dict(batch_size=10)),
         ("gmsh windowing", dict(mesh_generator="gmsh")),
         ("condense=False", dict(condense=False)),
         ("two-electrode configuration", dict(force_single_electrode_configuration=False)),
         ("mud 0.2", dict(_bore="Borehole_model_low_rm.txt")), ("mud 0.5", dict(_bore="Borehole_model_high_rm.txt")),
         ("formation model 2", dict(_form="Formation_model_2.txt")),
         ("3D path, dip 1e-6", dict(dip=1e-6))]
out = {}
for label, kw in cases:
    kw = dict(kw)
    b = bore(kw.pop("_bore", "Borehole_model_correct_rm.txt"))
    f = os.path.join(base, "Formation", kw.pop("_form", "Formation_model_1.txt"))
    try:
        m = Model.compute_synthetic_logs(tools, depths, f, b, gpu_workers=1, verbose=False, **kw)
    except Exception as ex:
        print(label, "FAILED", ex, flush=True)
        continue
    rel = np.array([(m.logs[t][:, 1] - gold[rows, 1 + i]) / gold[rows, 1 + i] for i, t in enumerate(tools)])
    out[label] = dict(median_abs=[float(np.nanmedian(np.abs(r))) for r in rel], signed_mean=[float(np.nanmean(r)) for r in rel],
                      A8_signed=[float(v) for v in rel[3]])
    print("%-30s median |rel| per tool %s   A8.0M1.0N signed mean %+.4f  min %+.4f max %+.4f" %
          (label, np.round(np.nanmedian(np.abs(rel), axis=1), 5), np.nanmean(rel[3]), np.nanmin(rel[3]), np.nanmax(rel[3])), flush=True)
if len(sys.argv) > 1:
    json.dump(dict(depths=depths.tolist(), tools=tools, cases=out), open(sys.argv[1], "w"), indent=1)
