#!/usr/bin/env python3
"""Cross-check of the 3D path against the 2D path on the same physics: Benchmark model 3 without dip
run (a) as the axisymmetric 2D model and (b) as a 3D model (dip = 1e-6 degrees switches the 3D code
path on: half-ball tetrahedral mesh, 3D assembly, Ra / 2) on interface-conforming meshes of both
kinds.  The 2D path is pinned against the reference's own logs (Example_01, thin-bedded benchmark);
this pins the 3D path against the 2D path.  Also runs the dipping model (30 degrees) once for sanity.
usage: python tools/run_3d_vs_2d.py [mesh_scale] [n_depths] [out.json] [sectors]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from remo3d_amd.model import Model, default_mesh_provider  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
nd = int(sys.argv[2]) if len(sys.argv) > 2 else 20
out = sys.argv[3] if len(sys.argv) > 3 else None
sectors = int(sys.argv[4]) if len(sys.argv) > 4 else 6
ex = os.path.join(ROOT, "tests", "golden", "examples", "Benchmark models", "Benchmark model 3")
tools = ["A0.4M6.0N", "A2.0M0.5N"]
depths = np.linspace(5.0, 20.0, nd, endpoint=False)
res = {}
for label, dip, kw in [("2d", 0.0, {}), ("3d_conforming", 1e-6, dict(mesh_provider=default_mesh_provider(scale=scale, sectors=sectors))),
                       ("3d_lattice", 1e-6, dict(mesh_provider=default_mesh_provider(scale=scale, mesh_3d="lattice")))]:
    t0 = time.time()
    m = Model.compute_synthetic_logs(tools, depths, os.path.join(ex, "Formation_BM3_00.txt"), os.path.join(ex, "Borehole_BM3.txt"), dip=dip,
                                     gpu_workers=1, verbose=False, mesh_generator="gmsh", mesh_scale=scale, **kw)
    res[label] = {k: m.logs[k][:, 1].tolist() for k in tools}
    print(f"{label:14s} {time.time() - t0:6.1f} s (mesh {m.timing['mesh_s']:.1f} s, solve {m.timing['solve_s']:.1f} s)  NaN {sum(int(np.isnan(v).sum()) for v in map(np.array, res[label].values()))}", flush=True)
summary = {}
for label in ("3d_conforming", "3d_lattice"):
    for k in tools:
        a, b = np.array(res[label][k]), np.array(res["2d"][k])
        d = np.abs(a - b) / np.abs(b)
        summary[f"{label}:{k}"] = dict(median=float(np.nanmedian(d)), p90=float(np.nanpercentile(d, 90)), max=float(np.nanmax(d)))
        print(f"{label:14s} {k:10s} vs 2d: median {np.nanmedian(d):.2e}  p90 {np.nanpercentile(d, 90):.2e}  max {np.nanmax(d):.2e}", flush=True)
t0 = time.time()
m = Model.compute_synthetic_logs(tools, depths[:10], os.path.join(ex, "Formation_BM3_30.txt"), os.path.join(ex, "Borehole_BM3.txt"), dip=30,
                                 gpu_workers=1, verbose=False, mesh_scale=scale)
ra = {k: m.logs[k][:, 1].tolist() for k in tools}
print("dip 30 conforming: %.1f s, NaN %d, Ra range %.3f .. %.3f" % (time.time() - t0, sum(int(np.isnan(np.array(v)).sum()) for v in ra.values()),
                                                                      min(np.nanmin(v) for v in ra.values()), max(np.nanmax(v) for v in ra.values())))
if out:
    json.dump(dict(command="python tools/run_3d_vs_2d.py %g %d out.json %d" % (scale, nd, sectors), sectors=sectors, model="Benchmark model 3, dip 0 (2D) vs dip 1e-6 deg (3D path)",
                   depths=depths.tolist(), tools=tools, mesh_scale=scale, relative_difference_vs_2d=summary, logs=res, dip30_first10=ra),
              open(out, "w"), indent=1)
