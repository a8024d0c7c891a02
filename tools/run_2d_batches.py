#!/usr/bin/env python3
"""A few batches of BASELINE configs[1] (Benchmark model 1, 2D, tool A0.4M6.0N) through the HIP path, meshes built in this
process (safe under rocprofv3): prints sizes, PCG steps and milliseconds per batch.  usage: run_2d_batches.py [n_batches] [mesh_scale]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from remo3d_amd import geometry, solver, tasks  # noqa: E402
from remo3d_amd.model import Model, default_mesh_provider  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4
scale = float(sys.argv[2]) if len(sys.argv) > 2 and float(sys.argv[2]) > 0 else None
coarse = [tuple(int(v) for v in a.split(",")) for a in sys.argv[3:]] or [(0, 0)]
ex = os.path.join(ROOT, "tests", "golden", "examples", "Benchmark models", "Benchmark model 1")
m = Model(["A0.4M6.0N"])
m.set_model_parameters(os.path.join(ex, "Formation_BM1.txt"), os.path.join(ex, "Borehole_BM1.txt"))
sim, batches = tasks.build_batches(m.tools, m.sec, np.linspace(5, 55, 100), 5)
mud = np.interp(sim, m.borehole_model[:, 0], m.borehole_model[:, 2])
bg = np.ascontiguousarray(m.borehole_model[:, :2])
provider = default_mesh_provider(scale=scale)
work = []
for bi in range(0, len(batches), max(1, len(batches) // nb))[:nb]:
    fg, bh, sigma = geometry.select_netgen_data_range(bg, m.formation_model, mud[bi], sim[bi], 50.0)
    work.append((provider(2, 50.0, batches[bi], fg, bh, 0.0), sigma) + tasks.batch_rhs(batches[bi], m.tools)[:2])
for kv in os.environ.get("REMO_TUNE", "").split():
    from remo3d_amd import _lib
    _lib.load().remo_debug_tune(*[int(v) for v in kv.split("=")])
with solver.Context(0) as ctx:
    for deg, ratio in coarse:
      tot = [0.0, 0]
      for rep in range(2):
        for mesh, sigma, sources, evals in work:
            t0 = time.time()
            outs, st, rc = ctx.solve_batch(mesh, sigma, sources, evals, solver.make_opts(coarse_degree=deg, coarse_ratio=ratio))
            if rep:
                tot[0] += st["ms_solve"]; tot[1] += st["pcg_steps"]
            if rep and len(coarse) == 1:
                print("T %d n %d nnz %d rhs %d: steps %d, solve %.2f ms (%.1f us per step), numbering %.2f, assembly %.2f, total %.2f ms" %
                      (mesh.n_elems, st["n_free"], st["nnz"], len(sources), st["pcg_steps"], st["ms_solve"], 1e3 * st["ms_solve"] / max(1, st["pcg_steps"]),
                       st["ms_symbolic"], st["ms_assemble"], 1e3 * (time.time() - t0)))
      print("coarse (%d, %d): %.2f ms of solve for %d batches, %d steps, %.1f us per step" % (deg, ratio, tot[0], len(work), tot[1], 1e3 * tot[0] / max(1, tot[1])))
