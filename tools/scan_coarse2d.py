#!/usr/bin/env python3
"""Chebyshev degree / interval ratio of the two-level preconditioner on 2D batches (paired launches for even degrees):
BM1 (default) or the thin-bedded benchmark (argument "thin": 5.5e5 dofs; optimum 20 / 1000 - 24 / 1500, 3 % ahead of 16 / 600)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from remo3d_amd import geometry, solver, tasks  # noqa: E402
from remo3d_amd.model import Model, default_mesh_provider  # noqa: E402

thin = len(sys.argv) > 1 and sys.argv[1] == "thin"
if thin:
    ex = os.path.join(ROOT, "tests", "golden", "examples", "Benchmark models", "Thin-bedded model")
    m = Model(["A0.4M6.0N", "A8.0M1.0N"])
    m.set_model_parameters(os.path.join(ex, "Formation", "Formation_model_1.txt"), os.path.join(ex, "Borehole", "Borehole_model_correct_rm.txt"))
    sim, batches = tasks.build_batches(m.tools, m.sec, np.arange(0, 20.01, 0.25), 5)
else:
    ex = os.path.join(ROOT, "tests", "golden", "examples", "Benchmark models", "Benchmark model 1")
    m = Model(["A0.4M6.0N"])
    m.set_model_parameters(os.path.join(ex, "Formation_BM1.txt"), os.path.join(ex, "Borehole_BM1.txt"))
    sim, batches = tasks.build_batches(m.tools, m.sec, np.linspace(5, 55, 100), 5)
bg = np.ascontiguousarray(m.borehole_model[:, :2])
mud = np.interp(sim, m.borehole_model[:, 0], m.borehole_model[:, 2])
prov = default_mesh_provider()
ctx = solver.Context(0)
bs = []
for bi in range(0, min(20, len(batches)), 3):
    fg, bh, sigma = geometry.select_netgen_data_range(bg, m.formation_model, mud[bi], sim[bi], 50)
    s_, e_, _ = tasks.batch_rhs(batches[bi], m.tools)
    bs.append(ctx.batch(prov(2, 50, batches[bi], fg, bh, 0.0), sigma, s_, e_))
for rnd in range(1):
    for deg, ratio in [(8, 120), (12, 300), (16, 600), (20, 1000), (24, 1500), (32, 2500)]:
        steps = 0; ms = 0.0
        for b in bs:
            b.run(solver.make_opts(coarse_degree=deg, coarse_ratio=ratio))
            steps += b.stats["pcg_steps"]; ms += b.stats["ms_solve"]
        print(f"round {rnd} deg {deg:2d} ratio {ratio:4d}: steps {steps} solve {ms:.1f} ms (n {bs[0].stats['n_free']})", flush=True)
