#!/usr/bin/env python3
"""(probe keys need REMO_LIB=remo3d_amd/libremo3d_hip_probes.so: make -C remo3d_amd/csrc probes)
A/B of one remo_debug_tune key on whole solves of bench batches, alternating the values on the same meshes in one process:
   python tools/ab_tune.py L 23 0 1 [--batches=2]   (key "deg": Chebyshev degree, 0 = default) [--rounds=3] [--precision=fp64]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench


def main():
    pos = [a for a in sys.argv[1:] if not a.startswith("--")]
    kw = dict(a[2:].split("=") for a in sys.argv[1:] if a.startswith("--"))
    size, key, values = pos[0], (pos[1] if pos[1] == "deg" else int(pos[1])), [int(v) for v in pos[2:]]     # "deg": remo_opts_t.coarse_degree instead of a tune key
    nb, rounds = int(kw.get("batches", 2)), int(kw.get("rounds", 3))
    import multiprocessing
    from concurrent.futures import ProcessPoolExecutor
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    if bench.under_profiler() or nb <= 1:      # behind a preloaded profiler library no child process may start: mesh here (or from the cache)
        wl = bench.build_workload(0, 1, int(kw.get("depths", 20)), bench.SIZES[size], max_batches=nb)
    else:
        with ProcessPoolExecutor(max_workers=min(nb, 8), mp_context=multiprocessing.get_context("spawn")) as pool:   # before the GPU is touched
            wl = bench.build_workload(0, 1, int(kw.get("depths", 20)), bench.SIZES[size], max_batches=nb, pool=pool)
    print("meshes done: T = %s" % [int(w["mesh"].n_elems) for w in wl["work"]], flush=True)
    from remo3d_amd import _lib, solver
    L = _lib.load()
    with solver.Context(0) as ctx:
        bs = [ctx.batch(w["mesh"], w["sigma"], w["sources"], w["evals"]) for w in wl["work"]]
        ref = None
        for rnd in range(rounds):
            for v in values:
                if key != "deg":
                    L.remo_debug_tune(key, v)
                ms = tot = 0.0
                steps = 0
                us = []
                for b in bs:
                    b.run(solver.make_opts(rtol=1e-8, precision=kw.get("precision", "fp64"), time_kernels=8, coarse_degree=(v if key == "deg" else 0)))
                    ms += b.stats["ms_solve"]; tot += b.stats["ms_total"]; steps += b.stats["pcg_steps"]
                    us.append(1e3 * b.stats["spmv_ms"] / max(1, b.stats["spmv_launches"]))
                u = np.concatenate(bs[0].fetch())
                if ref is None:
                    ref = u
                print("round %d key %s = %d: solve %.2f ms  total %.2f ms  steps %d  apply %s us  (potentials vs first run: %.1e)"
                      % (rnd, key, v, ms, tot, steps, ["%.1f" % x for x in us], float(np.max(np.abs(u - ref)) / np.max(np.abs(ref)))), flush=True)
        if key != "deg":
            L.remo_debug_tune(key, values[0])


if __name__ == "__main__":
    main()
