#!/usr/bin/env python3
"""Operator application (k = 5) and whole solves through the two operators on bench meshes, ablations and phase clocks of the patch kernel
(needs the probes build: make -C remo3d_amd/csrc probes; REMO_LIB=remo3d_amd/libremo3d_hip_probes.so):
   python tools/probe_patch.py S M L      (CSR product | patch)"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    only = None
    coarse = "auto"
    precision = "fp64"
    block = 256
    tunes = []
    for a in sys.argv[1:]:
        if a.startswith("--ops="):
            only = a[6:].split(",")
        if a.startswith("--coarse="):
            coarse = a[9:]
        if a.startswith("--precision="):
            precision = a[12:]
        if a.startswith("--block="):
            block = int(a[8:])
        if a.startswith("--tune="):
            tunes.append(tuple(int(v) for v in a[7:].split("=")))
    sizes = args or ["S"]
    work = {}
    for sz in sizes:
        t0 = time.time()
        wl = bench.build_workload(0, 1, 20, bench.SIZES[sz], max_batches=1)
        work[sz] = wl["work"][0]
        print("mesh %s: T=%d built in %.1f s" % (sz, work[sz]["mesh"].n_elems, time.time() - t0), flush=True)
    from remo3d_amd import _lib, solver
    L = _lib.load()
    L.remo_debug_tune(19, block)
    for key, val in tunes:
        L.remo_debug_tune(key, val)
    out = []
    with solver.Context(0) as ctx:
        for sz in sizes:
            w = work[sz]
            b = ctx.batch(w["mesh"], w["sigma"], w["sources"], w["evals"])
            ref = None
            for name, op, cpl, mode in (("csr", "csr", 1, 0), ("patch", "patch", 1, 0),
                                        ("patch/no-atomics", "patch", 1, 1), ("patch/no-arithmetic", "patch", 1, 2), ("patch/no-output", "patch", 1, 3)):
                if only and name not in only:
                    continue
                if mode and not only:
                    continue
                L.remo_debug_tune(21, 0)
                rc = b.run(solver.make_opts(rtol=1e-8, op=op, coarse=coarse, precision=precision, assemble=("full" if op != "patch" else "auto")), raise_on_error=False)
                st = dict(b.stats)
                n = st["n_free"]
                x = np.random.default_rng(0).standard_normal((n, 5))
                L.remo_debug_tune(21, mode)
                y, ms = b.spmv(x, reps=30)
                L.remo_debug_tune(21, 0)
                if ref is None:
                    ref = y
                err = float(np.max(np.abs(y - ref)) / np.max(np.abs(ref)))
                u = np.concatenate(b.fetch())
                rec = dict(size=sz, op=name, rc=rc, op_used=st["op_used"], T=int(w["mesh"].n_elems), n=n, nnz=st["nnz"], apply_us=1e3 * ms, rel_diff_vs_csr=err,
                           pcg_steps=st["pcg_steps"], solve_ms=st["ms_solve"], symbolic_ms=st["ms_symbolic"], assemble_ms=st["ms_assemble"], total_ms=st["ms_total"],
                           u0=float(u[0]))
                out.append(rec)
                print(json.dumps(rec), flush=True)
                if name == "patch":
                    import ctypes as C
                    for fp32 in (0, 1):
                        ph = (C.c_double * 16)()
                        if L.remo_debug_patch_phases(ctx._h, b._h, fp32, ph) == 0:
                            names = ["tables", "stage x", "x->regs", "zero", "arith+accumulate", "output", "partials", "workgroup", "launch span", "workgroups"]
                            print("phases %s (clock ticks): " % ("fp32" if fp32 else "fp64") + "  ".join("%s %.0f" % (nm, ph[i]) for i, nm in enumerate(names))
                                  + "  | application us: full %.1f  no-atomics %.1f  no-arithmetic %.1f  no-output %.1f" % (ph[10], ph[11], ph[12], ph[13]), flush=True)
            b.close()
    return out

if __name__ == "__main__":
    main()
