#!/usr/bin/env python3
"""BASELINE configs[2] through Model.compute_synthetic_logs (meshing included), with solver options given on the command line:
   python tools/run_end_to_end_3d.py [--reps=3] [--depths=100] [coarse=amg] [coarse_degree=16] ..."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    kw = dict(a[2:].split("=") for a in sys.argv[1:] if a.startswith("--"))
    opts = dict(a.split("=") for a in sys.argv[1:] if "=" in a and not a.startswith("--"))
    opts = {k: (int(v) if v.lstrip("-").isdigit() else v) for k, v in opts.items()}
    reps, nd = int(kw.get("reps", 3)), int(kw.get("depths", 100))
    from remo3d_amd.model import Model
    ex = os.path.join(ROOT, "tests", "golden", "examples", "Benchmark models", "Benchmark model 3")
    depths = np.linspace(5.0, 20.0, nd, endpoint=False)
    ref = None
    for variant, so in (("default", {}), ("options %s" % opts, opts)) * reps:
        t0 = time.time()
        m = Model.compute_synthetic_logs(["A0.4M6.0N", "A2.0M0.5N"], depths, os.path.join(ex, "Formation_BM3_30.txt"), os.path.join(ex, "Borehole_BM3.txt"), dip=30,
                                         cpu_workers=8, verbose=False, solver_options=so)
        dt = time.time() - t0
        logs = np.stack([m.logs[k][:, 1] for k in m.logs])
        ref = logs if ref is None else ref
        t = m.timing
        print("%-40s %.2f s  %.1f points/s  busy %.2f s  steps/batch %.1f  failed %d  max rel diff of the logs vs first run %.1e" %
              (variant, dt, 2 * nd / dt, t["busy_s"], t["pcg_steps"] / max(1, t["batches"]), t["failed_batches"], float(np.nanmax(np.abs(logs - ref) / np.abs(ref)))), flush=True)


if __name__ == "__main__":
    main()
