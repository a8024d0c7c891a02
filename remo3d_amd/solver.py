"""Python face of the C ABI: contexts, resident batches, one-shot batch solves.

Host-side mirror of what remo3d/workers/worker.py:100-134 does with NGSolve objects, expressed on
the arrays that define a batch.  Everything numerical happens in libremo3d_hip.so on the GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib
from ._lib import RemoOpts, RemoStats, ptr

REMO_OK = 0
REMO_NOT_CONVERGED = 1
REMO_ERR_ARG = -1


class RemoError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libremo3d_hip error {code}: {msg}")
        self.code = code


def make_opts(preconditioner="multigrid", condense=True, maxsteps=1000, rtol=1e-8, check_every=5,
              time_kernels=False, coarse_degree=0, coarse_ratio=0, precision="fp64", inner_digits=0,
              serialize_solves=False, op="auto", coarse="auto", quadrature="exact", assemble="auto") -> RemoOpts:
    """Options with the reference's names (remo3d.py:82-83, ngsolve_functions.py:46, 50).
    precision: "fp64" (default) or "mixed" = PCG in fp32 storage inside an fp64 residual-refinement loop
    (BASELINE config 5); inner_digits: decimal digits of <Cr,r> between two residual replacements (0 = library default 3)."""
    L = _lib.load()
    o = RemoOpts()
    L.remo_opts_default(C.byref(o))
    if preconditioner not in ("local", "multigrid"):
        raise ValueError("preconditioner must be 'local' or 'multigrid'")
    o.preconditioner = 0 if preconditioner == "local" else 1
    o.condense = 1 if condense else 0
    o.maxsteps = int(maxsteps)
    o.rtol = float(rtol)
    o.check_every = int(check_every)
    o.time_kernels = int(time_kernels)     # True / 1: every SpMV launch, k: every k-th
    o.coarse_degree = int(coarse_degree)
    o.coarse_ratio = int(coarse_ratio)
    if precision not in ("fp64", "mixed"):
        raise ValueError("precision must be 'fp64' or 'mixed'")
    o.precision = 1 if precision == "mixed" else 0
    o.inner_digits = int(inner_digits)
    o.serialize_solves = 1 if serialize_solves else 0   # several contexts: one PCG at a time, the others prepare (see the header)
    if op not in ("auto", "csr", "patch"):
        raise ValueError("op must be 'auto' (patch operator in 3D, CSR product in 2D), 'csr' (SpMM on the assembled matrix) or 'patch' (matrix-free, 3D)")
    o.op = {"auto": 0, "csr": 2, "patch": 3}[op]
    if coarse not in ("auto", "chebyshev", "amg", "amg_or_chebyshev"):
        raise ValueError("coarse must be 'auto' (multigrid cycle in 2D, Chebyshev polynomial in 3D), 'chebyshev', 'amg' or 'amg_or_chebyshev' (the cycle if its hierarchy can be built)")
    o.coarse = {"auto": 0, "chebyshev": 1, "amg": 2, "amg_or_chebyshev": 3}[coarse]   # solver of the P1 block inside "multigrid"
    if quadrature not in ("exact", "degree4"):
        raise ValueError("quadrature must be 'exact' or 'degree4' (2D reference tensors by the 6-point rule)")
    o.quadrature = 1 if quadrature == "degree4" else 0
    if assemble not in ("auto", "full", "vertex_block"):
        raise ValueError("assemble must be 'auto', 'full' or 'vertex_block' (diagonal + P1 block only: patch operator batches)")
    o.assemble = {"auto": 0, "full": 1, "vertex_block": 2}[assemble]
    return o


def _rhs_arrays(sources, evals):
    """sources: list (per RHS) of (z array, I array); evals: list (per RHS) of z arrays."""
    src_ptr = np.zeros(len(sources) + 1, dtype=np.int32)
    eval_ptr = np.zeros(len(evals) + 1, dtype=np.int32)
    sz, sI, ez = [], [], []
    for k, (z, I) in enumerate(sources):
        z = np.atleast_1d(np.asarray(z, dtype=np.float64)); I = np.atleast_1d(np.asarray(I, dtype=np.float64))
        if z.shape != I.shape:
            raise ValueError("source positions and strengths differ in length")
        sz.append(z); sI.append(I); src_ptr[k + 1] = src_ptr[k] + z.size
    for k, z in enumerate(evals):
        z = np.atleast_1d(np.asarray(z, dtype=np.float64))
        ez.append(z); eval_ptr[k + 1] = eval_ptr[k] + z.size
    cat = lambda xs: np.ascontiguousarray(np.concatenate(xs) if xs else np.zeros(0), dtype=np.float64)
    return src_ptr, cat(sz), cat(sI), eval_ptr, cat(ez)


class Context:
    """One per GPU (remo_ctx_create)."""

    def __init__(self, device_id: int = 0):
        self._L = _lib.load()
        self._h = self._L.remo_ctx_create(int(device_id))
        if not self._h:
            raise RemoError(-2, (self._L.remo_last_error(None) or b"").decode())
        self.device_id = device_id

    def close(self):
        if getattr(self, "_h", None):
            self._L.remo_ctx_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_error(self) -> str:
        return (self._L.remo_last_error(self._h) or b"").decode()

    def solve_batch(self, mesh, sigma, sources, evals, opts: Optional[RemoOpts] = None, raise_on_error=True):
        """One-shot remo_solve_batch.  Returns (list of per-RHS potential arrays, stats dict, rc)."""
        sigma = np.ascontiguousarray(sigma, dtype=np.float64)
        src_ptr, sz, sI, eval_ptr, ez = _rhs_arrays(sources, evals)
        ms, keep = _lib.mesh_struct(mesh)
        out = np.full(int(eval_ptr[-1]), np.nan)
        st = RemoStats()
        o = opts if opts is not None else make_opts()
        rc = self._L.remo_solve_batch(self._h, C.byref(ms), len(sigma), ptr(sigma, C.c_double), len(sources),
                                      ptr(src_ptr, C.c_int32), ptr(sz, C.c_double), ptr(sI, C.c_double),
                                      ptr(eval_ptr, C.c_int32), ptr(ez, C.c_double), ptr(out, C.c_double),
                                      C.byref(o), C.byref(st))
        if rc < 0 and raise_on_error:
            raise RemoError(rc, self.last_error())
        return [out[eval_ptr[k]:eval_ptr[k + 1]] for k in range(len(evals))], st.as_dict(), rc

    def batch(self, mesh, sigma, sources, evals) -> "Batch":
        return Batch(self, mesh, sigma, sources, evals)


class Batch:
    """Resident batch (remo_batch_create / run / fetch)."""

    def __init__(self, ctx: Context, mesh, sigma, sources, evals):
        self.ctx = ctx
        self._L = ctx._L
        sigma = np.ascontiguousarray(sigma, dtype=np.float64)
        self._arr = _rhs_arrays(sources, evals)
        src_ptr, sz, sI, eval_ptr, ez = self._arr
        ms, keep = _lib.mesh_struct(mesh)
        h = C.c_void_p()
        rc = self._L.remo_batch_create(ctx._h, C.byref(ms), len(sigma), ptr(sigma, C.c_double), len(sources),
                                       ptr(src_ptr, C.c_int32), ptr(sz, C.c_double), ptr(sI, C.c_double),
                                       ptr(eval_ptr, C.c_int32), ptr(ez, C.c_double), C.byref(h))
        if rc != 0:
            raise RemoError(rc, ctx.last_error())
        self._h = h
        self.n_rhs = len(sources)
        self.stats = None

    def run(self, opts: Optional[RemoOpts] = None, raise_on_error=True, ctx: Optional["Context"] = None):
        """ctx: run on ANOTHER context of the same GPU than the one that uploaded the batch (its arrays are plain device memory:
        an uploader context can bring the next batch in while the solver context works on this one).  The batch stays with
        that context from then on (system / solution of the run live in its arena)."""
        st = RemoStats()
        o = opts if opts is not None else make_opts()
        if ctx is not None:
            self.ctx = ctx
        rc = self._L.remo_batch_run(self.ctx._h, self._h, C.byref(o), C.byref(st))
        self.stats = st.as_dict()
        if rc < 0 and raise_on_error:
            raise RemoError(rc, self.ctx.last_error())
        return rc

    def fetch(self):
        eval_ptr = self._arr[3]
        out = np.full(int(eval_ptr[-1]), np.nan)
        rc = self._L.remo_batch_fetch(self.ctx._h, self._h, ptr(out, C.c_double))
        if rc != 0:
            raise RemoError(rc, self.ctx.last_error())
        return [out[eval_ptr[k]:eval_ptr[k + 1]] for k in range(self.n_rhs)]

    def eval(self, rhs: int, z):
        """u_h of right-hand side `rhs` of the last run at further axis points (remo_batch_eval)."""
        z = np.ascontiguousarray(np.atleast_1d(z), dtype=np.float64)
        out = np.full(z.size, np.nan)
        rc = self._L.remo_batch_eval(self.ctx._h, self._h, int(rhs), z.size, ptr(z, C.c_double), ptr(out, C.c_double))
        if rc != 0:
            raise RemoError(rc, self.ctx.last_error())
        return out

    def system(self):
        """CSR system of the last run: (rowptr, col, val, dinv, freeid)."""
        s = self.stats
        n, nnz, ndof = int(s["n_free"]), int(s["nnz"]), int(s["n_dof"])
        rowptr = np.zeros(n + 1, dtype=np.int32); col = np.zeros(nnz, dtype=np.int32)
        val = np.zeros(nnz); dinv = np.zeros(n); freeid = np.zeros(ndof, dtype=np.int32)
        rc = self._L.remo_batch_get_system(self.ctx._h, self._h, ptr(rowptr, C.c_int32), ptr(col, C.c_int32),
                                           ptr(val, C.c_double), ptr(dinv, C.c_double), ptr(freeid, C.c_int32))
        if rc != 0:
            raise RemoError(rc, self.ctx.last_error())
        return rowptr, col, val, dinv, freeid

    def jacobi(self):
        """1 / diag(A) of the last run (remo_batch_get_system, the other arrays skipped)."""
        n = int(self.stats["n_free"])
        dinv = np.zeros(n)
        rc = self._L.remo_batch_get_system(self.ctx._h, self._h, None, None, None, ptr(dinv, C.c_double), None)
        if rc != 0:
            raise RemoError(rc, self.ctx.last_error())
        return dinv

    def vectors(self):
        """(x, f) of the last chunk of right-hand sides of the last run, each [n_free, k] (remo_batch_get_vectors)."""
        n = int(self.stats["n_free"])
        k = C.c_int32(0)
        rc = self._L.remo_batch_get_vectors(self.ctx._h, self._h, None, None, C.byref(k))
        if rc != 0:
            raise RemoError(rc, self.ctx.last_error())
        x = np.zeros((n, k.value)); f = np.zeros((n, k.value))
        rc = self._L.remo_batch_get_vectors(self.ctx._h, self._h, ptr(x, C.c_double), ptr(f, C.c_double), C.byref(k))
        if rc != 0:
            raise RemoError(rc, self.ctx.last_error())
        return x, f

    def apply_vertex_solver(self, r, fp32=False):
        """z = one multigrid cycle of the last run's hierarchy applied to r [nv, k] (remo_batch_apply_coarse); nv = rows of the
        P1 block.  Raises when the last run used the Chebyshev polynomial."""
        nv = C.c_int64(0)
        rc = self._L.remo_batch_apply_coarse(self.ctx._h, self._h, 1, None, None, 1 if fp32 else 0, C.byref(nv))
        if rc != 0:
            raise RemoError(rc, self.ctx.last_error())
        r = np.ascontiguousarray(np.asarray(r, float).reshape(nv.value, -1))
        z = np.zeros_like(r)
        rc = self._L.remo_batch_apply_coarse(self.ctx._h, self._h, r.shape[1], ptr(r, C.c_double), ptr(z, C.c_double), 1 if fp32 else 0, C.byref(nv))
        if rc != 0:
            raise RemoError(rc, self.ctx.last_error())
        return z

    def true_relres(self):
        """sqrt(<C r, r> / <C f, f>) per column with r = f - A x recomputed from the solution (one device SpMM), C = Jacobi:
        what the recurrence residual of the PCG claims, measured."""
        x, f = self.vectors()
        y, _ = self.spmv(x if x.shape[1] > 1 else x[:, 0])
        r = f - y.reshape(f.shape)
        d = self.jacobi()[:, None]
        return np.sqrt((d * r * r).sum(0) / np.maximum((d * f * f).sum(0), 1e-300))

    def spmv(self, x, reps=1):
        """y = A x on the GPU (x: [n] or [n, k]); returns (y, average ms per launch)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        k = 1 if x.ndim == 1 else x.shape[1]
        y = np.zeros_like(x)
        ms = C.c_double(0)
        rc = self._L.remo_batch_spmv(self.ctx._h, self._h, k, ptr(x, C.c_double), ptr(y, C.c_double), int(reps), C.byref(ms))
        if rc != 0:
            raise RemoError(rc, self.ctx.last_error())
        return y, ms.value

    def close(self):
        if getattr(self, "_h", None):
            self._L.remo_batch_destroy(self.ctx._h, self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def host_element_matrix(dim: int, vertex_coords: np.ndarray, sigma: float) -> np.ndarray:
    """Element matrix from the library's reference tensors (host code path shared with the kernels)."""
    L = _lib.load()
    X = np.ascontiguousarray(vertex_coords, dtype=np.float64)
    n = 10 if dim == 2 else 20
    K = np.zeros((n, n))
    rc = L.remo_host_element_matrix(dim, ptr(X, C.c_double), float(sigma), ptr(K, C.c_double))
    if rc != 0:
        raise RemoError(rc, "remo_host_element_matrix")
    return K


def host_symbolic(mesh, condense=True):
    """Dof numbering + CSR pattern computed by the library (host part)."""
    L = _lib.load()
    ms, keep = _lib.mesh_struct(mesh)
    sizes = np.zeros(6, dtype=np.int64)
    rc = L.remo_host_symbolic(C.byref(ms), int(bool(condense)), ptr(sizes, C.c_int64), None, None, None)
    if rc != 0:
        raise RemoError(rc, (L.remo_last_error(None) or b"").decode())
    ndof, nfree, nnz = int(sizes[0]), int(sizes[1]), int(sizes[2])
    rowptr = np.zeros(nfree + 1, dtype=np.int32); col = np.zeros(nnz, dtype=np.int32); freeid = np.zeros(ndof, dtype=np.int32)
    rc = L.remo_host_symbolic(C.byref(ms), int(bool(condense)), ptr(sizes, C.c_int64), ptr(rowptr, C.c_int32),
                              ptr(col, C.c_int32), ptr(freeid, C.c_int32))
    if rc != 0:
        raise RemoError(rc, (L.remo_last_error(None) or b"").decode())
    return dict(n_dof=ndof, n_free=nfree, nnz=nnz, n_edges=int(sizes[3]), n_faces=int(sizes[4]), nld=int(sizes[5]),
                rowptr=rowptr, col=col, freeid=freeid)
