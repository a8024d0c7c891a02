#!/usr/bin/env python3
"""The SpMM alone on one batch of the bench workload (size S): isolated launches (one warm-up + one timed launch per call, milliseconds
of idle time between calls) against bursts of 50 and 400 back-to-back launches.  A box that runs the burst slower than the isolated
launch is holding its clock down under the kernel's load."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from remo3d_amd import solver  # noqa: E402

wl = bench.build_workload(0, 1, 10, bench.SIZES["S"], max_batches=1)
w = wl["work"][0]
rng = np.random.default_rng(0)
with solver.Context(0) as ctx:
    b = ctx.batch(w["mesh"], w["sigma"], w["sources"], w["evals"])
    b.run(solver.make_opts(rtol=1e-1, maxsteps=20))
    x = rng.standard_normal((b.stats["n_free"], 5))
    for label, reps, calls, pause in (("isolated", 1, 12, 0.05), ("burst of 50", 50, 4, 0.0), ("burst of 400", 400, 3, 0.0), ("isolated", 1, 6, 0.05)):
        res = []
        for _ in range(calls):
            time.sleep(pause)
            y, ms = b.spmv(x, reps=reps)
            res.append(1e3 * ms)
        print("%-13s SpMM us per launch: %s" % (label, " ".join("%.1f" % v for v in res)), flush=True)
    b.close()
