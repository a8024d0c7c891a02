#!/usr/bin/env python3
"""Vertex-block solver on the interface-conforming 3D meshes `Model` builds (BM3 dip 30, scale 1.0): PCG steps and solve time per batch
for Chebyshev degrees / intervals and for the multigrid cycle, one context, batches resident:
   python tools/scan_conforming.py [--batches=4] [--scale=1.0]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench


def main():
    kw = dict(a[2:].split("=") for a in sys.argv[1:] if a.startswith("--"))
    nb, scale = int(kw.get("batches", 4)), float(kw.get("scale", 1.0))
    wl = bench.build_workload(0, 1, 100, scale, mesh_3d="conforming", max_batches=None)
    work = wl["work"][:: max(1, len(wl["work"]) // nb)][:nb]
    from remo3d_amd import solver
    from remo3d_amd.model import tuned_coarse_for_conforming
    print("meshes: T = %s  nv = %s" % ([int(w["mesh"].n_elems) for w in work], [int(w["mesh"].n_nodes) for w in work]), flush=True)
    variants = [("tuned (Model default)", None), ("library default", {}), ("cheb 12,400", dict(coarse_degree=12, coarse_ratio=400)), ("cheb 16,650", dict(coarse_degree=16, coarse_ratio=650)),
                ("cheb 16,1200", dict(coarse_degree=16, coarse_ratio=1200)), ("cheb 20,1200", dict(coarse_degree=20, coarse_ratio=1200)), ("cheb 24,2000", dict(coarse_degree=24, coarse_ratio=2000)),
                ("multigrid cycle", dict(coarse="amg"))]
    with solver.Context(0) as ctx:
        bs = [ctx.batch(w["mesh"], w["sigma"], w["sources"], w["evals"]) for w in work]
        ref = None
        for rnd in range(2):
            for name, opt in variants:
                ms = 0.0; steps = 0; tot = 0.0
                for b, w in zip(bs, work):
                    o = tuned_coarse_for_conforming(w["mesh"].n_nodes) if opt is None else opt
                    rc = b.run(solver.make_opts(rtol=1e-8, **o), raise_on_error=False)
                    ms += b.stats["ms_solve"]; steps += b.stats["pcg_steps"]; tot += b.stats["ms_total"]
                u = np.concatenate(bs[0].fetch())
                ref = u if ref is None else ref
                print("round %d %-24s solve %.1f ms/batch  total %.1f ms/batch  steps %.1f/batch  coarse_used %d  (vs first %.1e)" %
                      (rnd, name, ms / len(bs), tot / len(bs), steps / len(bs), bs[0].stats["coarse_used"], float(np.max(np.abs(u - ref)) / np.max(np.abs(ref)))), flush=True)


if __name__ == "__main__":
    main()
