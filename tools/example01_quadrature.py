#!/usr/bin/env python3
"""Which 2D rule do the reference's logs prefer?  The complete Example_01 (1506 points) through the GPU path with the reference
tensors of `2 pi x sigma grad(u) grad(v)` (ngsolve_functions.py:34) integrated exactly and by the 6-point degree-4 rule
(remo_opts_t.quadrature), against the reference's committed log.  SURVEY.md 7.3-2 / VERDICT r2 item 2c."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from remo3d_amd.model import Model

ex = os.path.join(ROOT, "tests", "golden", "examples", "Example_01")
tools = ["B5.7A0.4M", "B4.48A1.62M", "M1.0A0.1B", "A2.0M0.5N", "N0.5M2.0A", "M4.0A0.5B"]
depths = np.arange(0, 25.1, 0.1)
gold = np.loadtxt(os.path.join(ex, "Output/Results_2024_08_17__18_59_29/Results_1.txt"), skiprows=2)
out, logs = {}, {}
for rule in ("exact", "degree4"):
    m = Model(tools)
    m.set_model_parameters(os.path.join(ex, "Input/Formation.txt"), os.path.join(ex, "Input/Borehole.txt"))
    m.initialize_workers(cpu_workers=8, gpu_workers=1)
    m.simulate_logs(depths, verbose=False, solver_options=dict(quadrature=rule))
    m.shutdown_workers()
    logs[rule] = np.array([m.logs[t][:, 1] for t in tools])
    rel = np.array([np.abs(m.logs[t][:, 1] - gold[:, 1 + i]) / gold[:, 1 + i] for i, t in enumerate(tools)])
    out[rule] = dict(points=int(rel.size), median=float(np.median(rel)), p90=float(np.percentile(rel, 90)), p99=float(np.percentile(rel, 99)), max=float(rel.max()),
                     failed_batches=m.timing["failed_batches"])
    print(rule, out[rule], flush=True)
d = np.abs(logs["exact"] - logs["degree4"]) / np.abs(logs["exact"])
out["exact_vs_degree4"] = dict(median=float(np.median(d)), p99=float(np.percentile(d, 99)), max=float(d.max()))
print("difference between the two rules:", out["exact_vs_degree4"])
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r03_example01_quadrature.json"), "w"), indent=1)
