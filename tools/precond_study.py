#!/usr/bin/env python3
"""Offline (CPU, scipy) study of preconditioner variants for the P3 system of one bench batch: PCG step
counts to rtol 1e-8 (the reference's CGSolver default).  Test infrastructure: uses the oracle's assembly.

  python tools/precond_study.py [S|M|L] [lattice|conforming]
"""
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


def pcg(A, f, C, rtol=1e-8, maxit=2000):
    x = np.zeros_like(f)
    r = f.copy()
    z = C(r)
    p = z.copy()
    rz = float(r @ z)
    rz0 = rz
    for it in range(1, maxit + 1):
        q = A @ p
        a = rz / float(p @ q)
        x += a * p
        r -= a * q
        z = C(r)
        rzn = float(r @ z)
        if rzn <= rtol * rtol * rz0:
            return x, it
        p = z + (rzn / rz) * p
        rz = rzn
    return x, maxit


def main():
    size = sys.argv[1] if len(sys.argv) > 1 else "S"
    mesh_kind = sys.argv[2] if len(sys.argv) > 2 else "lattice"
    w = bench.build_workload(0, 1, 5, bench.SIZES[size], mesh_3d=mesh_kind)["work"][0]
    mesh, sigma = w["mesh"], np.asarray(w["sigma"], float)
    t0 = time.time()
    o = Oracle(mesh, sigma, condense=True)
    rp, col, val = o.csr()
    n = o.nfree
    A = sp.csr_matrix((val, col, rp), shape=(n, n))
    fid = o.freeid()
    nvf = int((fid[:o.nv] >= 0).sum())
    nef = int((fid[o.nv:o.nv + 2 * o.ne] >= 0).sum())
    print(f"size {size} {mesh_kind}: T={o.nt} n={n} nnz={o.nnz} nv_free={nvf} edge rows={nef} face rows={n - nvf - nef}  ({time.time() - t0:.1f}s)", flush=True)
    z, I = w["sources"][0]
    f, _, _ = o.rhs(z, I)
    d = A.diagonal()
    dinv = 1.0 / d
    Avv = A[:nvf, :nvf].tocsc()
    lu = spla.splu(Avv)
    results = {}

    def run(name, C):
        t = time.time()
        x, it = pcg(A, f, C)
        results[name] = it
        print(f"{name:60s} {it:5d} steps  ({time.time() - t:.1f}s)", flush=True)
        return x

    quick = "--quick" in sys.argv
    vcycle = "--vcycle" in sys.argv
    if not quick and not vcycle:
        run("jacobi", lambda r: dinv * r)

    # current product: Chebyshev(d) on the Jacobi-scaled vertex block, Gershgorin lmax, ratio
    Dv = dinv[:nvf]
    lmax = float(np.max(np.abs(Avv).sum(1).A1 * Dv))
    nv_rel = nvf / 12600.0
    deg = int(min(16.0, max(5.0, np.floor(5.0 * np.sqrt(nv_rel) + 0.9))))
    ratio = min(1200.0, max(60.0, 90.0 * nv_rel ** (2.0 / 3.0)))

    def cheb(rv, deg, ratio):
        lmin = lmax / ratio
        theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
        sig = theta / delta
        rho = 1.0 / sig
        res = rv.copy()
        dd = Dv * res / theta
        zz = np.zeros_like(rv)
        for j in range(deg):
            zz += dd
            if j == deg - 1:
                break
            res -= Avv @ dd
            rho_new = 1.0 / (2.0 * sig - rho)
            dd = rho_new * rho * dd + 2.0 * rho_new / delta * Dv * res
            rho = rho_new
        return zz

    def two_level(p1, ho):
        def C(r):
            z = np.empty_like(r)
            z[:nvf] = p1(r[:nvf])
            z[nvf:] = ho(r)
            return z
        return C

    ho_jac = lambda r: dinv[nvf:] * r[nvf:]
    run(f"cheb({deg},{ratio:.0f}) P1 + jacobi HO   [product default]", two_level(lambda rv: cheb(rv, deg, ratio), ho_jac))
    run("exact P1 + jacobi HO", two_level(lu.solve, ho_jac))
    for dg, rt in ((deg + 3, ratio * 2), (deg * 2, ratio * 4)):
        run(f"cheb({dg},{rt:.0f}) P1 + jacobi HO", two_level(lambda rv: cheb(rv, dg, rt), ho_jac))
    if quick:
        print(results)
        return

    if vcycle:
        # Round 4: a MULTIPLICATIVE symmetric cycle as the preconditioner of the same PCG, the smoother running on
        # the FULL matrix-free operator:  z = S r;  z_v += P1 (r - A z)_v;  z += S^T (r - A z)   with S = nu steps of Chebyshev-
        # accelerated Jacobi on [lmax_A / alpha, lmax_A] (nu = 1: damped Jacobi).  Operator applications per PCG step beside the
        # PCG's own: 2 nu (nu - 1 inside each smoother, one residual before the P1 correction, one after it).
        # Priced with the costs measured in round 3 at size L (profiles/r03_sizeL_20depths_kernel_stats_working.txt): operator
        # application 128 us, a vector pass over n x 5 ~50 us, the P1 chain 90 us, update + direction of the PCG 187 us.
        lmaxA = float(spla.eigsh(sp.diags(np.sqrt(dinv)) @ A @ sp.diags(np.sqrt(dinv)), k=1, which="LA", return_eigenvectors=False, tol=1e-3)[0]) * 1.02
        print(f"lmax(D^-1 A) = {lmaxA:.3f}", flush=True)

        def cheb_smooth(r, x0, nu, alpha):
            """nu steps of Chebyshev-Jacobi for A x = r from x0 (x0 None = 0: the first step needs no product)."""
            lmin = lmaxA / alpha
            theta, delta = 0.5 * (lmaxA + lmin), 0.5 * (lmaxA - lmin)
            sig = theta / delta
            rho = 1.0 / sig
            res = r.copy() if x0 is None else r - A @ x0
            x = np.zeros_like(r) if x0 is None else x0.copy()
            dd = dinv * res / theta
            for j in range(nu):
                x += dd
                if j == nu - 1:
                    break
                res -= A @ dd
                rho_new = 1.0 / (2.0 * sig - rho)
                dd = rho_new * rho * dd + 2.0 * rho_new / delta * dinv * res
                rho = rho_new
            return x

        def vcyc(p1, nu, alpha):
            def C(r):
                z = cheb_smooth(r, None, nu, alpha)                # pre-smoothing from zero: nu - 1 products
                r1 = r - A @ z                                      # 1 product
                z[:nvf] += p1(r1[:nvf])
                return cheb_smooth(r, z, nu, alpha)                # post-smoothing: 1 + (nu - 1) products
            return C

        base = results.get(f"cheb({deg},{ratio:.0f}) P1 + jacobi HO   [product default]")
        t_app, t_vec, t_chain, t_pcg_vec = 128.0, 50.0, 90.0, 187.0
        t_now = t_app + t_pcg_vec + t_chain
        rows = []
        for p1name, p1 in (("exact P1", lu.solve), (f"cheb({deg},{ratio:.0f}) P1", lambda rv: cheb(rv, deg, ratio))):
            for nu in (1, 2, 3):
                for alpha in ((4.0, 8.0) if nu == 1 else (4.0, 10.0, 30.0)):
                    name = f"V({nu},{nu}) Chebyshev-Jacobi on the full operator, alpha {alpha:.0f} + {p1name}"
                    run(name, vcyc(p1, nu, alpha))
                    it = results[name]
                    extra = 2 * nu
                    t_step = t_now + extra * (t_app + t_vec) + 2 * (nu - 1) * t_vec
                    rows.append((name, it, extra, t_step, it * t_step / (base * t_now)))
        print("\nsteps, extra applications per PCG step, priced step (us), solve time relative to the product default (%d steps x %.0f us):" % (base, t_now))
        for name, it, extra, t_step, rel in rows:
            print(f"  {name:75s} {it:4d} steps  +{extra} applications  {t_step:6.0f} us/step  x{rel:.2f}")
        print(results)
        return

    # 2x2 edge blocks
    ne2 = nef // 2
    e0 = nvf
    Aee = A[e0:e0 + nef, e0:e0 + nef].tocsr()
    a11 = d[e0:e0 + nef:2]; a22 = d[e0 + 1:e0 + nef:2]
    a12 = np.asarray(Aee[np.arange(0, nef, 2), np.arange(1, nef, 2)]).ravel()
    det = a11 * a22 - a12 * a12

    def ho_edge2(r):
        z = dinv[nvf:] * r[nvf:]
        r1 = r[e0:e0 + nef:2]; r2 = r[e0 + 1:e0 + nef:2]
        z[0:nef:2] = (a22 * r1 - a12 * r2) / det
        z[1:nef:2] = (a11 * r2 - a12 * r1) / det
        return z
    run("exact P1 + 2x2 edge blocks + jacobi faces", two_level(lu.solve, ho_edge2))

    # edge-star additive Schwarz: block of edge k = its 2 dofs + the free face dofs coupled to it
    Ah = A[nvf:, nvf:].tocsr()
    nh = n - nvf
    t = time.time()
    Aef = A[e0:e0 + nef:2, e0 + nef:].tocsr()   # first dof of each edge vs faces: pattern gives the faces of the edge
    blocks = []
    for k in range(ne2):
        faces = Aef.indices[Aef.indptr[k]:Aef.indptr[k + 1]] + nef
        idx = np.concatenate([[2 * k, 2 * k + 1], faces])
        blocks.append(idx)
    sizes = np.array([len(b) for b in blocks])
    print(f"edge-star blocks: {len(blocks)}, size min/mean/max {sizes.min()}/{sizes.mean():.1f}/{sizes.max()}  ({time.time() - t:.1f}s)", flush=True)
    # batched inverses grouped by block size
    inv_by_size = {}
    for s in np.unique(sizes):
        ids = np.nonzero(sizes == s)[0]
        I = np.stack([blocks[i] for i in ids])                     # [m, s]
        rows = np.repeat(I[:, :, None], s, 2); cols = np.repeat(I[:, None, :], s, 1)
        M = np.asarray(Ah[rows.ravel(), cols.ravel()]).reshape(len(ids), s, s)
        inv_by_size[s] = (I, np.linalg.inv(M))

    def ho_star(r):
        rh = r[nvf:]
        z = np.zeros(nh)
        for s, (I, Minv) in inv_by_size.items():
            zz = np.einsum("mij,mj->mi", Minv, rh[I])
            np.add.at(z, I.ravel(), zz.ravel())
        return z
    run("exact P1 + edge-star additive Schwarz (edge pair + its faces)", two_level(lu.solve, ho_star))
    run(f"cheb({deg},{ratio:.0f}) P1 + edge-star additive Schwarz", two_level(lambda rv: cheb(rv, deg, ratio), ho_star))

    # face count weighting variant (each face is in 3 blocks): scale face contributions by 1/3 symmetric: D^1/2 C D^1/2
    wgt = np.ones(nh); wgt[nef:] = 1.0 / np.sqrt(3.0)

    def ho_star_w(r):
        rh = r[nvf:] * wgt
        z = np.zeros(nh)
        for s, (I, Minv) in inv_by_size.items():
            zz = np.einsum("mij,mj->mi", Minv, rh[I])
            np.add.at(z, I.ravel(), zz.ravel())
        return z * wgt
    run("exact P1 + edge-star AS, faces weighted 1/3", two_level(lu.solve, ho_star_w))

    # symmetric Gauss-Seidel on the HO block
    Lh = sp.tril(Ah, format="csr"); Uh = sp.triu(Ah, format="csr"); dh = d[nvf:]

    def ho_sgs(r):
        y = spla.spsolve_triangular(Lh, r[nvf:], lower=True)
        return spla.spsolve_triangular(Uh, dh * y, lower=False)
    if n < 700000:
        run("exact P1 + SGS HO", two_level(lu.solve, ho_sgs))

    # multiplicative (symmetric) coupling P1 <-> HO with Jacobi HO:  z_v' = P1 r_v; z_h = D^-1 (r_h - A_hv z_v'); z_v = P1 (r_v - A_vh z_h)
    Ahv = A[nvf:, :nvf].tocsr(); Avh = A[:nvf, nvf:].tocsr()

    def mult(p1, ho_apply):
        def C(r):
            zv1 = p1(r[:nvf])
            rh = r.copy(); rh[nvf:] -= Ahv @ zv1
            zh = ho_apply(rh)
            zv = p1(r[:nvf] - Avh @ zh)
            return np.concatenate([zv, zh])
        return C
    run("multiplicative: exact P1 / jacobi HO / exact P1", mult(lu.solve, ho_jac))
    run("multiplicative: exact P1 / edge-star AS / exact P1", mult(lu.solve, ho_star))
    print(results)


if __name__ == "__main__":
    main()
