#!/usr/bin/env python3
"""Where a 2D batch spends its time: per-phase milliseconds and PCG steps of BM1 batches (one tool, 5 RHS)."""
import os
import sys
import time

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
res = []
for bi in (0, 5, 10):
    fg, bh, sigma = geometry.select_netgen_data_range(bg, m.formation_model, mud[bi], sim[bi], 50)
    mesh = prov(2, 50, batches[bi], fg, bh, 0.0)
    sources, evals, readers = tasks.batch_rhs(batches[bi], m.tools)
    b = ctx.batch(mesh, sigma, sources, evals)
    for check in (5, 20):
        for rep in range(3):
            t0 = time.time(); b.run(solver.make_opts(check_every=check)); wall = (time.time() - t0) * 1e3
        st = b.stats
        print(f"batch {bi} check_every {check}: T {mesh.n_elems} n {st['n_free']} nnz {st['nnz']} steps {st['pcg_steps']} wall {wall:.2f} ms = symbolic {st['ms_symbolic']:.2f} + assemble {st['ms_assemble']:.2f} "
              f"+ solve {st['ms_solve']:.2f} + eval {st['ms_eval']:.2f} (+h2d {st['ms_h2d']:.2f}); us per step {1e3 * st['ms_solve'] / max(1, st['pcg_steps']):.1f}", flush=True)
