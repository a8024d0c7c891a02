"""The CPU oracle behind `Model`'s solver-context interface - TEST INFRASTRUCTURE ONLY (tests/ may use oracle/, the product
may not).  Lets the same sweep (tool tables, batching, windowing, conforming meshes, Ra) run with oracle/fem_oracle.c as the
solver, so that the ORACLE itself is checked against the reference's committed logs."""
import numpy as np

from oracle.fem_oracle import solve_batch as _oracle_batch


class OracleContext:
    def __init__(self, device=0, maxit=100000):
        self.maxit = maxit
        self.calls = 0

    def solve_batch(self, mesh, sigma, sources, evals, opts, raise_on_error=True):
        sp, sz, sI, ep, ez = [0], [], [], [0], []
        for (z, I), e in zip(sources, evals):
            sz += list(z); sI += list(I); sp.append(len(sz)); ez += list(e); ep.append(len(ez))
        out, rc, st = _oracle_batch(mesh, sigma, sp, sz, sI, ep, ez, condense=bool(opts.condense), rtol=float(opts.rtol), maxit=self.maxit)
        self.calls += 1
        if rc < 0:
            raise RuntimeError("oracle: point outside the mesh")
        return [np.asarray(out[ep[k]:ep[k + 1]]) for k in range(len(evals))], dict(pcg_steps=st["iterations"], n_free=st["n"]), (1 if rc > 0 else 0)

    def close(self):
        pass


class OracleDirectContext(OracleContext):
    """The oracle's SYSTEMS (numbering, quadrature assembly, condensed bubble, point sources, evaluation - oracle/fem_oracle.c)
    with the linear solve done by a sparse direct factorisation (SuperLU through scipy, symmetric mode) instead of the oracle's
    Jacobi-PCG: the same right-hand sides solved to rounding, ~10 x faster at the default 2D mesh scale (one factorisation per
    batch serves its 5-10 right-hand sides; the PCG needs ~2300 steps each).  Used where the subject of the test is everything
    AROUND the iteration - meshes, windowing, assembly, sources, evaluation, Ra - against the reference's logs; the oracle's PCG
    itself is pinned by the GPU-vs-oracle tests and by the PCG-backed sweep of test_oracle_reference_logs.py."""

    def solve_batch(self, mesh, sigma, sources, evals, opts, raise_on_error=True):
        import scipy.sparse as sp
        import scipy.sparse.linalg as spla
        from oracle.fem_oracle import Oracle
        o = Oracle(mesh, sigma, condense=bool(opts.condense))
        try:
            rp, col, val = o.csr()
            lu = spla.splu(sp.csr_matrix((val, col, rp), shape=(o.nfree, o.nfree)).tocsc(), permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0,
                           options=dict(SymmetricMode=True))
            outs = []
            for (z, I), e in zip(sources, evals):
                f, se, sf = o.rhs(list(z), list(I))
                outs.append(np.asarray(o.eval(lu.solve(f), list(e), (se, sf))))
            self.calls += 1
            return outs, dict(pcg_steps=0, n_free=o.nfree), 0
        finally:
            o.close()
