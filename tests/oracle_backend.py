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
