#!/usr/bin/env python3
"""Offline (CPU, scipy): would an aggregation multigrid cycle on the P1 (vertex) block be a better coarse solver than the
Chebyshev polynomial?  Aggregates = runs of g consecutive free vertices (the vertices are in Morton order of the mesh).
Prints PCG step counts (rtol 1e-8) of the whole P3 system with  C = blockdiag(P1 solver, Jacobi on edge / face dofs).
  python tools/amg_study.py [S|M|L] [lattice|conforming]"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle.fem_oracle import Oracle  # noqa: E402
from tools.precond_study import pcg  # noqa: E402


def cheb_smoother(A, dinv, lmax, deg, ratio):
    lmin = lmax / ratio
    theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
    sig = theta / delta

    def apply(r, z0=None):
        # z ~ A^-1 r starting from z0 (None = 0): Chebyshev semi-iteration on the Jacobi-scaled operator
        z = np.zeros_like(r) if z0 is None else z0.copy()
        res = r.copy() if z0 is None else r - A @ z
        rho = 1.0 / sig
        d = dinv * res / theta
        for j in range(deg):
            z += d
            if j == deg - 1:
                break
            res -= A @ d
            rho_new = 1.0 / (2.0 * sig - rho)
            d = rho_new * rho * d + 2.0 * rho_new / delta * dinv * res
            rho = rho_new
        return z
    return apply


class Level:
    pass


def build_hierarchy(A, g, min_size=400, max_levels=8):
    levels = []
    while True:
        L = Level()
        L.A = A.tocsr()
        L.dinv = 1.0 / L.A.diagonal()
        L.lmax = float(np.max(np.abs(L.A).sum(1).A1 * L.dinv))
        levels.append(L)
        n = A.shape[0]
        if n <= min_size or len(levels) >= max_levels:
            break
        agg = np.arange(n) // g
        nc = int(agg.max()) + 1
        P = sp.csr_matrix((np.ones(n), (np.arange(n), agg)), shape=(n, nc))
        L.P = P
        A = (P.T @ A @ P).tocsr()
    levels[-1].solve = spla.splu(levels[-1].A.tocsc()).solve
    return levels


def vcycle(levels, r, k=0, smooth_deg=2, ratio=4.0, omega=1.0, gamma=1):
    L = levels[k]
    if k == len(levels) - 1:
        return L.solve(r)
    S = cheb_smoother(L.A, L.dinv, L.lmax, smooth_deg, ratio)
    z = S(r)
    rc = L.P.T @ (r - L.A @ z)
    zc = vcycle(levels, rc, k + 1, smooth_deg, ratio, omega, gamma)
    for _ in range(gamma - 1):
        zc = zc + vcycle(levels, rc - levels[k + 1].A @ zc, k + 1, smooth_deg, ratio, omega, gamma)
    z = z + omega * (L.P @ zc)
    # post-smoothing with the same polynomial, symmetric cycle
    z = z + S(r - L.A @ z)
    return z


def main():
    size = sys.argv[1] if len(sys.argv) > 1 else "M"
    kind = sys.argv[2] if len(sys.argv) > 2 else "lattice"
    w = bench.build_workload(0, 1, 5, bench.SIZES[size], mesh_3d=kind)["work"][0]
    o = Oracle(w["mesh"], np.asarray(w["sigma"], float), condense=True)
    rp, col, val = o.csr()
    n = o.nfree
    A = sp.csr_matrix((val, col, rp), shape=(n, n))
    fid = o.freeid()
    nvf = int((fid[:o.nv] >= 0).sum())
    print(f"{size} {kind}: n={n} nnz={o.nnz} nv_free={nvf}", flush=True)
    z, I = w["sources"][0]
    f, _, _ = o.rhs(z, I)
    dinv = 1.0 / A.diagonal()
    Avv = A[:nvf, :nvf].tocsr()

    def two_level(p1):
        def C(r):
            out = np.empty_like(r)
            out[:nvf] = p1(r[:nvf])
            out[nvf:] = dinv[nvf:] * r[nvf:]
            return out
        return C

    def run(name, C):
        t = time.time()
        _, it = pcg(A, f, C)
        print(f"{name:70s} {it:4d} steps ({time.time() - t:.1f}s)", flush=True)

    lu = spla.splu(Avv.tocsc())
    run("exact P1", two_level(lu.solve))
    lmax = float(np.max(np.abs(Avv).sum(1).A1 * dinv[:nvf]))
    nv_rel = nvf / 12600.0
    deg = int(min(16.0, max(5.0, np.floor(5.0 * np.sqrt(nv_rel) + 0.9))))
    ratio = min(1200.0, max(60.0, 90.0 * nv_rel ** (2.0 / 3.0)))
    run(f"cheb({deg},{ratio:.0f}) [product default: {deg - 1} launches]", two_level(cheb_smoother(Avv, dinv[:nvf], lmax, deg, ratio)))
    for g in (4, 8):
        levels = build_hierarchy(Avv, g)
        sizes = [L.A.shape[0] for L in levels]
        nnzs = [L.A.nnz for L in levels]
        for sd, rt, om, gm in ((1, 4.0, 1.0, 1), (2, 4.0, 1.0, 1), (2, 4.0, 1.5, 1), (3, 6.0, 1.0, 1), (2, 4.0, 1.0, 2), (3, 6.0, 1.5, 1)):
            run(f"AMG g={g} levels {sizes} nnz {nnzs} smoother cheb({sd},{rt:g}) omega {om} gamma {gm}",
                two_level(lambda r: vcycle(levels, r, 0, sd, rt, om, gm)))


if __name__ == "__main__":
    main()
