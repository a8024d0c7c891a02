#!/usr/bin/env python3
"""PCG steps on the revolved conforming meshes as a function of dip and sector count (one BM3 batch)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from remo3d_amd import geometry, meshgen, solver, tasks  # noqa: E402
from remo3d_amd.model import Model  # noqa: E402

ex = os.path.join(ROOT, "tests", "golden", "examples", "Benchmark models", "Benchmark model 3")
ctx = solver.Context(0)
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
for dip in (1e-6, 15, 30, 45):
    m = Model(["A0.4M6.0N", "A2.0M0.5N"])
    m.set_model_parameters(os.path.join(ex, "Formation_BM3_%02d.txt" % int(round(dip))), os.path.join(ex, "Borehole_BM3.txt"), dip=dip)
    m.borehole_model = m._add_points_to_borehole()
    sim, batches = tasks.build_batches(m.tools, m.sec, np.linspace(5.0, 20.0, 10, endpoint=False), 5)
    b = batches[0]
    bg = np.ascontiguousarray(m.borehole_model[:, :2])
    mud = np.interp(sim, m.borehole_model[:, 0], m.borehole_model[:, 2])
    fg, bh, sigma = geometry.select_data_range(bg, m.formation_model, m.dip_rad, mud[0], sim[0], 50.0)
    cur = b.electrodes[0, b.electrodes[1, :] != 0]; pot = b.electrodes[0, b.electrodes[1, :] == 0]
    sources, evals, readers = tasks.batch_rhs(b, m.tools)
    for sectors in (4, 6, 8):
        for er in (0.5,):
            mesh = meshgen.make_mesh_3d_conforming(50.0, fg, bh, m.dip_rad, sources_z=list(cur), snap_z=list(pot), scale=scale, sectors=sectors, exact_radius=er)
            for pre in ("local", "multigrid"):
                outs, st, rc = ctx.solve_batch(mesh, sigma, sources, evals, solver.make_opts(preconditioner=pre, maxsteps=3000))
                print(f"dip {dip:5.1f} sectors {sectors} {pre:9s}: tets {mesh.n_elems:7d} n {st['n_free']:8d} min q {mesh.meta['min_quality']:.3f} rc {rc} its {st['max_iterations']:4d} solve {st['ms_solve']:.1f} ms", flush=True)
