#!/usr/bin/env python3
"""BASELINE configs[1] batches (2D, Benchmark model 1) with the two vertex-block solvers of the "multigrid" preconditioner:
Chebyshev polynomial vs the smoothed-aggregation cycle (remo_opts_t.coarse).  Prints steps, solve / total time and the largest
relative difference of the potentials.  usage: probe_amg2d.py [n_batches] [mesh_scale] [precision]"""
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
precision = sys.argv[3] if len(sys.argv) > 3 else "fp64"
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
ref = {}
with solver.Context(0) as ctx:
    for coarse in ("chebyshev", "amg"):
        tot = [0.0, 0, 0.0]
        for rep in range(2):
            for wi, (mesh, sigma, sources, evals) in enumerate(work):
                t0 = time.time()
                outs, st, rc = ctx.solve_batch(mesh, sigma, sources, evals, solver.make_opts(coarse=coarse, rtol=float(os.environ.get("RTOL", "1e-10")), precision=precision, maxsteps=4000, check_every=int(os.environ.get("CHECK", "5"))))
                if rc != 0:
                    print(coarse, "rc", rc, ctx.last_error() if hasattr(ctx, "last_error") else "", flush=True)
                    continue
                u = np.concatenate([np.asarray(o, float) for o in outs])
                if coarse == "chebyshev":
                    ref[wi] = u
                if rep:
                    tot[0] += st["ms_solve"]; tot[1] += st["pcg_steps"]; tot[2] += 1e3 * (time.time() - t0)
                    print("%-9s T %d n %d: coarse_used %d steps %d (max it %d), solve %.2f ms (%.1f us per step), numbering %.2f, assembly %.2f, total %.2f ms, max rel diff %.2e" %
                          (coarse, mesh.n_elems, st["n_free"], st["coarse_used"], st["pcg_steps"], st["max_iterations"], st["ms_solve"],
                           1e3 * st["ms_solve"] / max(1, st["pcg_steps"]), st["ms_symbolic"], st["ms_assemble"], 1e3 * (time.time() - t0),
                           float(np.max(np.abs(u - ref[wi]) / np.abs(ref[wi])))), flush=True)
        print("%s: %.2f ms of solve, %.2f ms wall for %d batches, %d steps" % (coarse, tot[0], tot[2], len(work), tot[1]), flush=True)
