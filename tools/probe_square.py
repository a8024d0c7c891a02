#!/usr/bin/env python3
"""A/B of the paired Chebyshev steps on the squared vertex block (remo_debug_tune key 6) against one launch per step:
solve time and PCG steps of bench batches (3D) and of 2D BM1 batches, interleaved in one process."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from remo3d_amd import _lib, geometry, solver, tasks  # noqa: E402
from remo3d_amd.model import Model, default_mesh_provider  # noqa: E402

L = _lib.load()
ctx = solver.Context(0)
wl = bench.build_workload(0, 1, 10, bench.SIZES[sys.argv[1] if len(sys.argv) > 1 else "S"])
sets = {"3D": [ctx.batch(w["mesh"], w["sigma"], w["sources"], w["evals"]) for w in wl["work"]]}
ex = os.path.join(ROOT, "tests", "golden", "examples", "Benchmark models", "Benchmark model 1")
m = Model(["A0.4M6.0N"])
m.set_model_parameters(os.path.join(ex, "Formation_BM1.txt"), os.path.join(ex, "Borehole_BM1.txt"))
sim, batches = tasks.build_batches(m.tools, m.sec, np.linspace(5, 55, 100), 5)
bg = np.ascontiguousarray(m.borehole_model[:, :2])
mud = np.interp(sim, m.borehole_model[:, 0], m.borehole_model[:, 2])
prov = default_mesh_provider()
b2 = []
for bi in (0, 5, 10, 15):
    fg, bh, sigma = geometry.select_netgen_data_range(bg, m.formation_model, mud[bi], sim[bi], 50)
    s_, e_, _ = tasks.batch_rhs(batches[bi], m.tools)
    b2.append(ctx.batch(prov(2, 50, batches[bi], fg, bh, 0.0), sigma, s_, e_))
sets["2D"] = b2
for name, bs in sets.items():
    ref = None
    res = {}
    for rnd in range(3):
        for sq, lanes in ((0, 0), (2, 8), (2, 16), (2, 32)):
            L.remo_debug_tune(6, sq); L.remo_debug_tune(7, lanes)
            ms = tot = 0.0; steps = 0; us = []
            for b in bs:
                b.run(solver.make_opts())
                st = b.stats
                ms += st["ms_solve"]; tot += st["ms_total"]; steps += st["pcg_steps"]
                us.append(np.concatenate([np.atleast_1d(x) for x in b.fetch()]))
            u = np.concatenate(us)
            if ref is None:
                ref = u
            err = float(np.max(np.abs(u - ref) / np.abs(ref)))
            assert err < 1e-6, err
            res.setdefault((sq, lanes), []).append(ms)
            print(f"{name} round {rnd} paired={sq} lanes={lanes}: solve {ms:.2f} ms, batch total {tot:.2f} ms, steps {steps}, us/step {1e3 * ms / steps:.1f}, max rel diff {err:.1e}", flush=True)
    print(name, "median solve ms:", {k: round(float(np.median(v)), 2) for k, v in res.items()})
