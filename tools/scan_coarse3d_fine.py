#!/usr/bin/env python3
"""Fine scan of the Chebyshev degree / interval ratio around the 3D defaults (20 depths = 4 batches of the bench
workload, two rounds), after the first and the last step lost their launches.  Usage: python tools/scan_coarse3d_fine.py [S|M|L]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from remo3d_amd import solver  # noqa: E402

if __name__ == "__main__":
    size = sys.argv[1] if len(sys.argv) > 1 else "S"
    import multiprocessing
    from concurrent.futures import ProcessPoolExecutor
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    with ProcessPoolExecutor(max_workers=4, mp_context=multiprocessing.get_context("spawn")) as pool:     # before the GPU is touched
        wl = bench.build_workload(0, 1, 20, bench.SIZES[size], max_batches=4, pool=pool)
    print("meshes done", flush=True)
    ctx = solver.Context(0)
    bs = [ctx.batch(w["mesh"], w["sigma"], w["sources"], w["evals"]) for w in wl["work"]]
    degs = {"S": (3, 4, 5, 6, 7, 8), "M": (6, 7, 8, 9, 10, 12)}.get(size, (8, 9, 10, 11, 12, 13, 14, 16))
    ratios = {"S": (40, 60, 90, 120, 160), "M": (90, 120, 160, 220, 300)}.get(size, (150, 220, 307, 420, 600))
    grid = [(0, 0)] + [(d, r) for d in degs for r in ratios]
    for rnd in range(2):
        for deg, ratio in grid:
            steps = 0; ms = 0.0
            for b in bs:
                b.run(solver.make_opts(coarse_degree=deg, coarse_ratio=ratio))
                steps += b.stats["pcg_steps"]; ms += b.stats["ms_solve"]
            print(f"round {rnd} size {size} deg {deg} ratio {ratio:3d}: steps {steps} solve {ms:.1f} ms (n {bs[0].stats['n_free']}, nv {bs[0].stats.get('nv_coarse', '?')})", flush=True)
