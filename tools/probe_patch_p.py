#!/usr/bin/env python3
"""Phase clocks of the persistent patch kernel (k_patch_apply_p) on headline batches; needs the probes build:
   make -C remo3d_amd/csrc probes && REMO_LIB=remo3d_amd/libremo3d_hip_probes.so python tools/probe_patch_p.py [L] [--wgs=N ...]"""
import ctypes as C
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench


def main():
    size = ([a for a in sys.argv[1:] if not a.startswith("--")] or ["L"])[0]
    wgs = [int(a[6:]) for a in sys.argv[1:] if a.startswith("--wgs=")] or [0]
    w = bench.build_workload(0, 1, 100 if size == "L" else 20, bench.SIZES[size], max_batches=1)["work"][0]
    from remo3d_amd import _lib, solver
    L = _lib.load()
    names = ["B0", "DMA issue", "x->regs", "B1", "clear+B2", "chains+accumulate", "wait DMA/stores", "B3", "rows out", "own lgkm before B0"]
    with solver.Context(0) as ctx:
        b = ctx.batch(w["mesh"], w["sigma"], w["sources"], w["evals"])
        b.run(solver.make_opts(rtol=1e-8))
        print("T=%d n=%d op_used=%d" % (w["mesh"].n_elems, b.stats["n_free"], b.stats["op_used"]), flush=True)
        names_r = ["park+B0", "chain 1 + issue of next x", "x->regs", "B1", "clear+B2", "chain 2 + accumulate", "issue element/table loads + B3", "read-out + B4", "store X(p+1) + rows out", "-"]
        for fp32 in (0, 1):
          for mode in (2, 1):
            for nw in wgs:
                L.remo_debug_tune(34, mode); L.remo_debug_tune(35, nw)
                ph = (C.c_double * 16)()
                rc = L.remo_debug_patch_phases_p(ctx._h, b._h, fp32, ph)
                if rc != 0:
                    print("remo_debug_patch_phases_p:", ctx.last_error()); return
                tot = sum(ph[i] for i in range(10))
                nm_ = names_r if mode == 2 else names
                print("%s %s wgs/xcd %s: per patch (ticks) " % ("fp32" if fp32 else "fp64", "REGISTER prefetch" if mode == 2 else "LDS-DMA prefetch", nw or "auto") + "  ".join("%s %.0f" % (nm, ph[i]) for i, nm in enumerate(nm_[:10]))
                      + "  | sum %.0f  patches %d  workgroups %d  busiest workgroup %.0f ticks  application %.1f us" % (tot, ph[10], ph[11], ph[12], ph[13]), flush=True)
            L.remo_debug_tune(34, 0)
            ph = (C.c_double * 16)()
            if L.remo_debug_patch_phases(ctx._h, b._h, fp32, ph) == 0:
                n2 = ["tables", "stage x", "x->regs", "zero", "arith+accumulate", "output", "partials", "workgroup"]
                print("%s one workgroup per patch: " % ("fp32" if fp32 else "fp64") + "  ".join("%s %.0f" % (nm, ph[i]) for i, nm in enumerate(n2)) + "  | application %.1f us" % ph[10], flush=True)
            L.remo_debug_tune(34, 1)
        b.close()


if __name__ == "__main__":
    main()
