"""ctypes binding of oracle/fem_oracle.c — TEST INFRASTRUCTURE ONLY (see the C file's header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Parity against NGSolve itself is UNPINNED (NGSolve is not installable here); the oracle is
pinned by analytic solutions and the reference's committed example logs (DESIGN.md "Oracle").
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libfem_oracle.so")
    src = os.path.join(_HERE, "fem_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libfem_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        dp, ip, lp, bp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_long), C.POINTER(C.c_ubyte)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int, C.c_long, dp, C.c_long, ip, ip, C.c_long, ip, bp, C.c_int, dp, C.c_int]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_sizes.argtypes = [C.c_void_p, lp]
        L.orc_get_csr.argtypes = [C.c_void_p, lp, ip, dp]
        L.orc_get_freeid.argtypes = [C.c_void_p, ip]
        L.orc_get_eldof.argtypes = [C.c_void_p, ip]
        L.orc_spmv.argtypes = [C.c_void_p, dp, dp]
        L.orc_rhs.restype = C.c_int
        L.orc_rhs.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, lp, dp]
        L.orc_pcg.restype = C.c_int
        L.orc_pcg.argtypes = [C.c_void_p, dp, dp, C.c_double, C.c_int, ip, dp]
        L.orc_eval.restype = C.c_int
        L.orc_eval.argtypes = [C.c_void_p, dp, C.c_int, dp, dp, C.c_int, lp, dp]
        L.orc_solve_batch.restype = C.c_int
        L.orc_solve_batch.argtypes = [C.c_int, C.c_long, dp, C.c_long, ip, ip, C.c_long, ip, bp, C.c_int, dp,
                                      C.c_int, C.c_int, ip, dp, dp, ip, dp, C.c_double, C.c_int, dp, dp]
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _mesh_args(mesh):
    coords = np.ascontiguousarray(mesh.coords, dtype=np.float64)
    conn = np.ascontiguousarray(mesh.conn, dtype=np.int32)
    mat = np.ascontiguousarray(mesh.mat, dtype=np.int32)
    bconn = np.ascontiguousarray(mesh.bconn, dtype=np.int32)
    bdir = np.ascontiguousarray(mesh.bdirichlet, dtype=np.uint8)
    keep = (coords, conn, mat, bconn, bdir)
    args = [mesh.dim, coords.shape[0], _p(coords, C.c_double), conn.shape[0], _p(conn, C.c_int), _p(mat, C.c_int),
            bconn.shape[0], _p(bconn, C.c_int), _p(bdir, C.c_ubyte)]
    return args, keep


class Oracle:
    """Assembled system for one mesh / sigma (orc_create)."""

    def __init__(self, mesh, sigma, condense=True):
        L = lib()
        sigma = np.ascontiguousarray(sigma, dtype=np.float64)
        args, self._keep = _mesh_args(mesh)
        self._h = L.orc_create(*args, len(sigma), _p(sigma, C.c_double), int(bool(condense)))
        if not self._h:
            raise RuntimeError("orc_create failed")
        s = np.zeros(8, dtype=np.int64)
        L.orc_sizes(self._h, _p(s, C.c_long))
        self.nv, self.ne, self.nf, self.ndof, self.nfree, self.nnz, self.nt, self.nld = (int(x) for x in s)
        self.dim = mesh.dim

    def close(self):
        if self._h:
            lib().orc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def csr(self):
        rp = np.zeros(self.nfree + 1, dtype=np.int64)
        col = np.zeros(self.nnz, dtype=np.int32)
        val = np.zeros(self.nnz, dtype=np.float64)
        lib().orc_get_csr(self._h, _p(rp, C.c_long), _p(col, C.c_int), _p(val, C.c_double))
        return rp, col, val

    def freeid(self):
        f = np.zeros(self.ndof, dtype=np.int32)
        lib().orc_get_freeid(self._h, _p(f, C.c_int))
        return f

    def eldof(self):
        n = 10 if self.dim == 2 else 20
        e = np.zeros((self.nt, n), dtype=np.int32)
        lib().orc_get_eldof(self._h, _p(e, C.c_int))
        return e

    def spmv(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.zeros(self.nfree)
        lib().orc_spmv(self._h, _p(x, C.c_double), _p(y, C.c_double))
        return y

    def rhs(self, z, I):
        z = np.ascontiguousarray(z, dtype=np.float64); I = np.ascontiguousarray(I, dtype=np.float64)
        f = np.zeros(self.nfree)
        se = np.zeros(len(z), dtype=np.int64); sf = np.zeros(len(z))
        rc = lib().orc_rhs(self._h, len(z), _p(z, C.c_double), _p(I, C.c_double), _p(f, C.c_double),
                           _p(se, C.c_long), _p(sf, C.c_double))
        if rc < 0:
            raise RuntimeError("source point outside the mesh")
        return f, se, sf

    def pcg(self, f, rtol=1e-12, maxit=20000):
        f = np.ascontiguousarray(f, dtype=np.float64)
        u = np.zeros(self.nfree)
        it = C.c_int(0); rr = C.c_double(0)
        rc = lib().orc_pcg(self._h, _p(f, C.c_double), _p(u, C.c_double), rtol, maxit, C.byref(it), C.byref(rr))
        return u, it.value, rr.value, rc

    def eval(self, u, z, src=None):
        z = np.ascontiguousarray(z, dtype=np.float64); u = np.ascontiguousarray(u, dtype=np.float64)
        out = np.zeros(len(z))
        if src is None:
            se = np.zeros(0, dtype=np.int64); sf = np.zeros(0)
        else:
            se, sf = src
        rc = lib().orc_eval(self._h, _p(u, C.c_double), len(z), _p(z, C.c_double), _p(out, C.c_double),
                            len(se), _p(se, C.c_long), _p(sf, C.c_double))
        if rc < 0:
            raise RuntimeError("evaluation point outside the mesh")
        return out


def solve_batch(mesh, sigma, src_ptr, src_z, src_I, eval_ptr, eval_z, condense=True, rtol=1e-10, maxit=20000):
    """CPU restatement of one batch (same argument meaning as remo_solve_batch)."""
    L = lib()
    sigma = np.ascontiguousarray(sigma, dtype=np.float64)
    src_ptr = np.ascontiguousarray(src_ptr, dtype=np.int32); eval_ptr = np.ascontiguousarray(eval_ptr, dtype=np.int32)
    src_z = np.ascontiguousarray(src_z, dtype=np.float64); src_I = np.ascontiguousarray(src_I, dtype=np.float64)
    eval_z = np.ascontiguousarray(eval_z, dtype=np.float64)
    nrhs = len(src_ptr) - 1
    out = np.zeros(int(eval_ptr[-1]))
    stats = np.zeros(4)
    args, keep = _mesh_args(mesh)
    rc = L.orc_solve_batch(*args, len(sigma), _p(sigma, C.c_double), int(bool(condense)), nrhs,
                           _p(src_ptr, C.c_int), _p(src_z, C.c_double), _p(src_I, C.c_double),
                           _p(eval_ptr, C.c_int), _p(eval_z, C.c_double), rtol, maxit,
                           _p(out, C.c_double), _p(stats, C.c_double))
    return out, rc, dict(iterations=int(stats[0]), relres=float(stats[1]), n=int(stats[2]), nnz=int(stats[3]))
