#!/usr/bin/env python3
"""Chebyshev degree / interval ratio of the two-level preconditioner on 2D BM1 batches (paired steps for even degrees)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from remo3d_amd import geometry, solver, tasks  # noqa: E402
from remo3d_amd.model import Model, default_mesh_provider  # noqa: E402

ex = os.path.join(ROOT, "tests", "golden", "examples", "Benchmark models", "Benchmark model 1")
m = Model(["A0.4M6.0N"])
m.set_model_parameters(os.path.join(ex, "Formation_BM1.txt"), os.path.join(ex, "Borehole_BM1.txt"))
sim, batches = tasks.build_batches(m.tools, m.sec, np.linspace(5, 55, 100), 5)
bg = np.ascontiguousarray(m.borehole_model[:, :2])
mud = np.interp(sim, m.borehole_model[:, 0], m.borehole_model[:, 2])
prov = default_mesh_provider()
ctx = solver.Context(0)
bs = []
for bi in range(0, 20, 3):
    fg, bh, sigma = geometry.select_netgen_data_range(bg, m.formation_model, mud[bi], sim[bi], 50)
    s_, e_, _ = tasks.batch_rhs(batches[bi], m.tools)
    bs.append(ctx.batch(prov(2, 50, batches[bi], fg, bh, 0.0), sigma, s_, e_))
for rnd in range(2):
    for deg, ratio in [(10, 200), (10, 300), (12, 300), (12, 450), (14, 400), (14, 600), (16, 600), (20, 1000)]:
        steps = 0; ms = 0.0
        for b in bs:
            b.run(solver.make_opts(coarse_degree=deg, coarse_ratio=ratio))
            steps += b.stats["pcg_steps"]; ms += b.stats["ms_solve"]
        print(f"round {rnd} deg {deg:2d} ratio {ratio:3d}: steps {steps} solve {ms:.1f} ms", flush=True)
