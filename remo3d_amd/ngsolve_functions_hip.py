"""Solver-module drop-in: the slot the reference fills with `ngsolve_functions` (CPU) or
`ngsolve_functions_gpu` (CUDA attempt) - workers/worker.py:32-35 picks the module per rank and
calls `ngsf.SolveBVP(...)` once per right-hand side (worker.py:110), then reads
`gfu(mesh(0.0, z))` / `gfu(mesh(0.0, 0.0, z))` (worker.py:124-131).

Same names, argument meaning and error behaviour (any failure raises, which the worker turns into
NaN for the batch, worker.py:135-138; non-convergence within `maxsteps` is silent like
ngsolve_functions.py:50).  The NGSolve objects are replaced by thin stand-ins that carry arrays:

    mesh  = Mesh(meshgen.Mesh)             # instead of ngs.Mesh(netgen mesh), worker.py:100
    sigma = CoefficientFunction([..])      # instead of ngs.CoefficientFunction(list), worker.py:101

This per-RHS interface re-assembles the matrix for every call exactly as the reference does; the
batch interface (solver.Context.solve_batch / Model.simulate_logs) assembles once per batch and is
what the benchmark measures.
"""
from __future__ import annotations

import os

import numpy as np

from . import solver

_CTX = {}


def _context():
    dev = int(os.environ.get("LOCAL_RANK", "0"))
    if dev not in _CTX:
        _CTX[dev] = solver.Context(dev)
    return _CTX[dev]


class MeshPoint:
    def __init__(self, z):
        self.z = float(z)


class Mesh:
    """Stand-in for ngs.Mesh: `.dim` and point lookup `mesh(0, z)` / `mesh(0, 0, z)` on the axis."""

    def __init__(self, arrays):
        self.arrays = arrays
        self.dim = arrays.dim

    def __call__(self, *xyz):
        if len(xyz) != self.dim:
            raise ValueError("mesh(...) takes %d coordinates" % self.dim)
        if any(float(c) != 0.0 for c in xyz[:-1]):
            raise ValueError("only points on the borehole axis are supported (the reference evaluates nowhere else)")
        return MeshPoint(xyz[-1])


class CoefficientFunction:
    def __init__(self, values):
        self.values = np.asarray(list(values), dtype=np.float64)


class FESpace:
    order = 3

    def __init__(self, stats):
        self.ndof = int(stats["n_dof"])
        self.nfree = int(stats["n_free"])


class GridFunction:
    def __init__(self, batch, stats):
        self._batch = batch
        self.stats = stats

    def __call__(self, point: MeshPoint) -> float:
        return float(self._batch.eval(0, [point.z])[0])


def SolveBVP(mesh, sigma, tool_geometry, source_terms, dirichlet_boundary, preconditioner, condense):
    """ngsolve_functions.py:23-58.  `dirichlet_boundary` is accepted for signature compatibility:
    the Dirichlet facets are flagged in the mesh arrays (bdirichlet)."""
    tool_geometry = np.asarray(tool_geometry, dtype=np.float64)
    source_terms = np.asarray(source_terms, dtype=np.float64)
    vals = sigma.values if isinstance(sigma, CoefficientFunction) else np.asarray(sigma, dtype=np.float64)
    live = source_terms != 0.0
    ctx = _context()
    batch = ctx.batch(mesh.arrays, vals, [(tool_geometry[live], source_terms[live])], [np.zeros(0)])
    rc = batch.run(solver.make_opts(preconditioner=preconditioner, condense=bool(condense)))   # raises on error codes
    return FESpace(batch.stats), GridFunction(batch, batch.stats)
