#!/usr/bin/env python3
"""A/B of the CSR pattern build: row by row through LDS (default) vs the global sort of all element pairs
(remo_debug_tune key 8 = 0).  Same matrices (pattern + values) required; numbering time per batch printed."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from remo3d_amd import _lib, solver  # noqa: E402

L = _lib.load()
ctx = solver.Context(0)
wl = bench.build_workload(0, 1, 10, bench.SIZES[sys.argv[1] if len(sys.argv) > 1 else "S"])
bs = [ctx.batch(w["mesh"], w["sigma"], w["sources"], w["evals"]) for w in wl["work"]]
ref = {}
for rnd in range(3):
    for mode in (0, 1):
        L.remo_debug_tune(8, mode)
        ms = []
        for i, b in enumerate(bs):
            b.run(solver.make_opts(maxsteps=3), raise_on_error=False)
            ms.append(b.stats["ms_symbolic"])
            rp, col, val = b.system()[:3]
            if i not in ref:
                ref[i] = (rp.copy(), col.copy(), val.copy())
            assert np.array_equal(rp, ref[i][0]) and np.array_equal(col, ref[i][1]) and np.array_equal(val, ref[i][2])
        print(f"round {rnd} row-by-row={mode}: numbering {np.mean(ms):.3f} ms per batch (n {b.stats['n_free']}, nnz {b.stats['nnz']})", flush=True)
