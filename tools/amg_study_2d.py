#!/usr/bin/env python3
"""Offline (CPU, scipy): smoothed-aggregation multigrid on the P1 (vertex) block of a 2D batch of BASELINE configs[1] against
the Chebyshev polynomial the product uses: PCG steps of the whole P3 system (rtol 1e-8) and the work of one cycle in
fine-block sweeps.  Greedy aggregation on the matrix graph, prolongator smoothed by one damped Jacobi step, Galerkin products."""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.fem_oracle import Oracle  # noqa: E402
from remo3d_amd import geometry, tasks  # noqa: E402
from remo3d_amd.model import Model, default_mesh_provider  # noqa: E402
from tools.precond_study import pcg  # noqa: E402
from tools.amg_study import cheb_smoother  # noqa: E402


def aggregate(A):
    """Standard greedy aggregation (Vanek): pass 1 roots with all neighbours free, pass 2 attach leftovers."""
    n = A.shape[0]
    indptr, indices = A.indptr, A.indices
    agg = -np.ones(n, dtype=np.int64)
    na = 0
    for i in range(n):
        nb = indices[indptr[i]:indptr[i + 1]]
        if agg[i] < 0 and np.all(agg[nb] < 0):
            agg[nb] = na
            agg[i] = na
            na += 1
    for i in range(n):
        if agg[i] < 0:
            nb = indices[indptr[i]:indptr[i + 1]]
            cand = agg[nb][agg[nb] >= 0]
            if cand.size:
                agg[i] = cand[0]
            else:
                agg[i] = na
                na += 1
    return agg, na


def build(A, min_size=300, max_levels=10, smooth=True):
    levels = []
    while True:
        A = A.tocsr()
        d = A.diagonal()
        lv = dict(A=A, dinv=1.0 / d, lmax=float(np.max(np.abs(A).sum(1).A1 / d)))
        levels.append(lv)
        n = A.shape[0]
        if n <= min_size or len(levels) >= max_levels:
            break
        agg, na = aggregate(A)
        P = sp.csr_matrix((np.ones(n), (np.arange(n), agg)), shape=(n, na))
        if smooth:
            P = (P - (4.0 / (3.0 * lv["lmax"])) * (sp.diags(lv["dinv"]) @ (A @ P))).tocsr()
        lv["P"] = P
        A = (P.T @ A @ P).tocsr()
    levels[-1]["solve"] = spla.splu(levels[-1]["A"].tocsc()).solve
    return levels


def vcycle(levels, r, k=0, nu=1):
    L = levels[k]
    if k == len(levels) - 1:
        return L["solve"](r)
    S = cheb_smoother(L["A"], L["dinv"], L["lmax"], nu, 4.0)
    z = S(r)
    zc = vcycle(levels, L["P"].T @ (r - L["A"] @ z), k + 1, nu)
    z = z + L["P"] @ zc
    return z + S(r - L["A"] @ z)


def main3(size, kind):
    import bench
    w = bench.build_workload(0, 1, 5, bench.SIZES[size], mesh_3d=kind)["work"][0]
    o = Oracle(w["mesh"], np.asarray(w["sigma"], float), condense=True)
    rp, col, val = o.csr()
    n = o.nfree
    A = sp.csr_matrix((val, col, rp), shape=(n, n))
    fid = o.freeid()
    nvf = int((fid[:o.nv] >= 0).sum())
    print(f"3D {size} {kind}: n={n} nnz={o.nnz} nv_free={nvf}", flush=True)
    f, _, _ = o.rhs(*w["sources"][0])
    dinv = 1.0 / A.diagonal()
    Avv = A[:nvf, :nvf].tocsr()
    nv_rel = nvf / 12600.0
    deg = int(min(16.0, max(5.0, np.floor(5.0 * np.sqrt(nv_rel) + 0.9))))
    ratio = min(1200.0, max(60.0, 90.0 * nv_rel ** (2.0 / 3.0)))
    study(A, f, dinv, Avv, nvf, deg, ratio, f"{deg - 1} launches on A_vv")


def study(A, f, dinv, Avv, nvf, deg, ratio, note):
    def two_level(p1):
        def C(r):
            out = np.empty_like(r)
            out[:nvf] = p1(r[:nvf])
            out[nvf:] = dinv[nvf:] * r[nvf:]
            return out
        return C

    def run(name, C):
        t = time.time()
        _, it = pcg(A, f, C, maxit=3000)
        print(f"{name:86s} {it:5d} steps ({time.time() - t:.1f}s)", flush=True)

    run("exact P1 + Jacobi on edge dofs", two_level(spla.splu(Avv.tocsc()).solve))
    lmax = float(np.max(np.abs(Avv).sum(1).A1 * dinv[:nvf]))
    run(f"Chebyshev({deg}, {ratio:.0f}) on P1 [product default: {note}]", two_level(cheb_smoother(Avv, dinv[:nvf], lmax, deg, ratio)))
    for smooth in (True, False):
        t = time.time()
        levels = build(Avv, smooth=smooth)
        sizes = [lv["A"].shape[0] for lv in levels]
        nnzs = [lv["A"].nnz for lv in levels]
        work = sum(nnzs[:-1]) / nnzs[0]
        print(f"{'smoothed' if smooth else 'plain'} aggregation: levels {sizes}, nnz {nnzs}, operator complexity {work:.2f} (setup {time.time() - t:.1f}s)", flush=True)
        for nu in (1, 2):
            sweeps = (2 * nu + 1) * work
            run(f"  V({nu},{nu}) Chebyshev({nu}, 4) smoothing: {sweeps:.1f} fine-block sweeps per cycle", two_level(lambda r: vcycle(levels, r, 0, nu)))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "3d":
        return main3(sys.argv[2] if len(sys.argv) > 2 else "S", sys.argv[3] if len(sys.argv) > 3 else "lattice")
    ex = os.path.join(ROOT, "tests", "golden", "examples", "Benchmark models", "Benchmark model 1")
    m = Model(["A0.4M6.0N"])
    m.set_model_parameters(os.path.join(ex, "Formation_BM1.txt"), os.path.join(ex, "Borehole_BM1.txt"))
    sim, batches = tasks.build_batches(m.tools, m.sec, np.linspace(5, 55, 100), 5)
    mud = np.interp(sim, m.borehole_model[:, 0], m.borehole_model[:, 2])
    bg = np.ascontiguousarray(m.borehole_model[:, :2])
    bi = 5
    fg, bh, sigma = geometry.select_netgen_data_range(bg, m.formation_model, mud[bi], sim[bi], 50.0)
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else None
    mesh = default_mesh_provider(scale=scale)(2, 50.0, batches[bi], fg, bh, 0.0)
    sources, evals, _ = tasks.batch_rhs(batches[bi], m.tools)
    o = Oracle(mesh, np.asarray(sigma, float), condense=True)
    rp, col, val = o.csr()
    n = o.nfree
    A = sp.csr_matrix((val, col, rp), shape=(n, n))
    fid = o.freeid()
    nvf = int((fid[:o.nv] >= 0).sum())
    print(f"2D batch: T={o.nt} n={n} nnz={o.nnz} nv_free={nvf}", flush=True)
    f, _, _ = o.rhs(*sources[0])
    dinv = 1.0 / A.diagonal()
    Avv = A[:nvf, :nvf].tocsr()

    nv2 = nvf / 25000.0
    deg = 2 * int(min(16.0, max(8.0, np.floor(8.0 * np.sqrt(nv2) + 0.5))))
    ratio = min(2400.0, max(600.0, 750.0 * nv2))
    study(A, f, dinv, Avv, nvf, deg, ratio, f"{deg // 2} paired launches on B = {deg} sweeps of A_vv")


if __name__ == "__main__":
    main()
