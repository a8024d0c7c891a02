#!/usr/bin/env python3
"""fp64 PCG vs mixed precision (fp32 inner PCG + fp64 refinement) on one batch of the bench workload:
solve time, total PCG steps, refinement cycles, difference of the potentials.  python tools/probe_mixed.py [S|M|L|XL]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from remo3d_amd import solver  # noqa: E402

size = sys.argv[1] if len(sys.argv) > 1 else "S"
wl = bench.build_workload(0, 1, 10, bench.SIZES[size])
w = wl["work"][0]
ctx = solver.Context(0)
b = ctx.batch(w["mesh"], w["sigma"], w["sources"], w["evals"])
ref = None
for label, kw in [("fp64", dict()), ("mixed d=2", dict(precision="mixed", inner_digits=2)), ("mixed d=3", dict(precision="mixed", inner_digits=3)),
                  ("mixed d=4", dict(precision="mixed", inner_digits=4)), ("mixed d=5", dict(precision="mixed", inner_digits=5)), ("fp64", dict())]:
    for rep in range(2):
        rc = b.run(solver.make_opts(rtol=1e-8, time_kernels=True, **kw))
    u = np.concatenate([np.atleast_1d(x) for x in b.fetch()])
    st = b.stats
    if ref is None:
        ref = u
    print(f"{label:10s} rc={rc} n={st['n_free']} ms_solve {st['ms_solve']:8.2f} steps {st['pcg_steps']:5d} cycles {st['refinement_cycles']:3d} "
          f"max its {st['max_iterations']:4d} relres {max(st['relres'][:5]):.2e} spmv us {1e3 * st['spmv_ms'] / max(1, st['spmv_launches']):6.1f} "
          f"max rel diff vs fp64 {np.max(np.abs(u - ref) / np.abs(ref)):.2e}")
