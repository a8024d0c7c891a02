"""ctypes binding of libremo3d_hip.so (include/remo3d_hip.h).

The library is the product: there is no Python / CPU fallback.  If the shared object is missing
the import of this module raises, and if no MI355X is visible ``Context()`` raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# REMO_LIB: another build of the same library (tools/: the -DREMO_PROBES build, compile-time variants); the product loads its own
LIB_PATH = os.environ.get("REMO_LIB") or os.path.join(_HERE, "libremo3d_hip.so")
REMO_MAX_RHS = 8


class RemoMesh(C.Structure):
    _fields_ = [("dim", C.c_int32), ("n_nodes", C.c_int64), ("coords", C.POINTER(C.c_double)),
                ("n_elems", C.c_int64), ("conn", C.POINTER(C.c_int32)), ("mat", C.POINTER(C.c_int32)),
                ("n_bfacets", C.c_int64), ("bconn", C.POINTER(C.c_int32)), ("bdirichlet", C.POINTER(C.c_uint8))]


class RemoOpts(C.Structure):
    _fields_ = [("preconditioner", C.c_int32), ("condense", C.c_int32), ("maxsteps", C.c_int32),
                ("check_every", C.c_int32), ("rtol", C.c_double), ("time_kernels", C.c_int32),
                ("coarse_degree", C.c_int32), ("coarse_ratio", C.c_int32), ("precision", C.c_int32), ("inner_digits", C.c_int32),
                ("serialize_solves", C.c_int32), ("op", C.c_int32), ("coarse", C.c_int32), ("assemble", C.c_int32), ("quadrature", C.c_int32)]


class RemoStats(C.Structure):
    _fields_ = [("n_dof", C.c_int64), ("n_free", C.c_int64), ("nnz", C.c_int64), ("n_edges", C.c_int64),
                ("n_faces", C.c_int64), ("n_rhs", C.c_int32), ("max_iterations", C.c_int32),
                ("iterations", C.c_int32 * REMO_MAX_RHS), ("relres", C.c_double * REMO_MAX_RHS),
                ("ms_symbolic", C.c_double), ("ms_h2d", C.c_double), ("ms_assemble", C.c_double),
                ("ms_solve", C.c_double), ("ms_eval", C.c_double), ("ms_total", C.c_double),
                ("spmv_ms", C.c_double), ("spmv_launches", C.c_int64), ("spmv_bytes", C.c_double),
                ("pcg_steps", C.c_int64), ("spmv_ms_raw", C.c_double), ("event_overhead_ms", C.c_double),
                ("refinement_cycles", C.c_int64), ("op_used", C.c_int32), ("coarse_used", C.c_int32),
                ("assembled", C.c_int32), ("reserved_", C.c_int32)]

    def as_dict(self):
        d = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            d[name] = list(v) if hasattr(v, "__len__") else v
        return d


# include/remo3d_hip.h: the drop-in boundary + inspection hooks of the parity tests
EXPORTS = ["remo_abi_version", "remo_opts_default", "remo_ctx_create", "remo_ctx_destroy", "remo_last_error",
           "remo_solve_batch", "remo_batch_create", "remo_batch_run", "remo_batch_fetch", "remo_batch_destroy",
           "remo_batch_eval", "remo_batch_get_system", "remo_batch_get_vectors", "remo_batch_apply_coarse", "remo_batch_spmv",
           "remo_host_element_matrix", "remo_host_factor_error", "remo_host_symbolic"]
# include/remo3d_hip_debug.h: probes and tuning knobs (tests, tools, bench.py's `box` record) - not part of the boundary
DEBUG_EXPORTS = ["remo_debug_stream", "remo_debug_clock", "remo_debug_device", "remo_debug_cache_gather", "remo_debug_xcc", "remo_debug_tune", "remo_debug_patch_phases", "remo_debug_patch_phases_p", "remo_debug_grid_barrier"]

_lib = None


def load():
    """Load libremo3d_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make -C remo3d_amd/csrc` "
                          "(or __graft_entry__.build()); remo3d_amd has no CPU fallback")
    L = C.CDLL(LIB_PATH)
    dp, ip, i64p, u8p = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_uint8)
    vp = C.c_void_p
    L.remo_abi_version.restype = C.c_int
    L.remo_opts_default.argtypes = [C.POINTER(RemoOpts)]
    L.remo_ctx_create.restype = vp
    L.remo_ctx_create.argtypes = [C.c_int]
    L.remo_ctx_destroy.argtypes = [vp]
    L.remo_last_error.restype = C.c_char_p
    L.remo_last_error.argtypes = [vp]
    batch_args = [vp, C.POINTER(RemoMesh), C.c_int32, dp, C.c_int32, ip, dp, dp, ip, dp]
    L.remo_solve_batch.restype = C.c_int
    L.remo_solve_batch.argtypes = batch_args + [dp, C.POINTER(RemoOpts), C.POINTER(RemoStats)]
    L.remo_batch_create.restype = C.c_int
    L.remo_batch_create.argtypes = batch_args + [C.POINTER(vp)]
    L.remo_batch_run.restype = C.c_int
    L.remo_batch_run.argtypes = [vp, vp, C.POINTER(RemoOpts), C.POINTER(RemoStats)]
    L.remo_batch_fetch.restype = C.c_int
    L.remo_batch_fetch.argtypes = [vp, vp, dp]
    L.remo_batch_destroy.argtypes = [vp, vp]
    L.remo_batch_eval.restype = C.c_int
    L.remo_batch_eval.argtypes = [vp, vp, C.c_int32, C.c_int32, dp, dp]
    L.remo_batch_get_system.restype = C.c_int
    L.remo_batch_get_system.argtypes = [vp, vp, ip, ip, dp, dp, ip]
    L.remo_batch_get_vectors.restype = C.c_int
    L.remo_batch_get_vectors.argtypes = [vp, vp, dp, dp, ip]
    L.remo_batch_apply_coarse.restype = C.c_int
    L.remo_batch_apply_coarse.argtypes = [vp, vp, C.c_int32, dp, dp, C.c_int32, i64p]
    L.remo_debug_patch_phases.restype = C.c_int
    L.remo_debug_patch_phases.argtypes = [vp, vp, C.c_int32, dp]
    L.remo_debug_patch_phases_p.restype = C.c_int
    L.remo_debug_patch_phases_p.argtypes = [vp, vp, C.c_int32, dp]
    L.remo_debug_grid_barrier.restype = C.c_int
    L.remo_debug_grid_barrier.argtypes = [vp, C.c_int32, C.c_int32, dp]
    L.remo_debug_xcc.restype = C.c_int
    L.remo_debug_xcc.argtypes = [vp, ip, C.c_int32]
    L.remo_debug_cache_gather.restype = C.c_int
    L.remo_debug_cache_gather.argtypes = [vp, C.c_int64, dp]
    L.remo_debug_device.restype = C.c_int
    L.remo_debug_device.argtypes = [vp, i64p]
    L.remo_debug_clock.restype = C.c_int
    L.remo_debug_clock.argtypes = [vp, dp]
    L.remo_debug_stream.restype = C.c_int
    L.remo_debug_stream.argtypes = [vp, C.c_int64, dp, dp]
    L.remo_batch_spmv.restype = C.c_int
    L.remo_batch_spmv.argtypes = [vp, vp, C.c_int32, dp, dp, C.c_int32, dp]
    L.remo_host_element_matrix.restype = C.c_int
    L.remo_host_element_matrix.argtypes = [C.c_int32, dp, C.c_double, dp]
    L.remo_host_factor_error.restype = C.c_double
    L.remo_host_symbolic.restype = C.c_int
    L.remo_host_symbolic.argtypes = [C.POINTER(RemoMesh), C.c_int32, i64p, ip, ip, ip]
    L.remo_debug_tune.restype = C.c_int
    L.remo_debug_tune.argtypes = [C.c_int32, C.c_int32]
    _lib = L
    return L


def ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def mesh_struct(mesh):
    """RemoMesh view of a meshgen.Mesh-like object; returns (struct, keepalive)."""
    coords = np.ascontiguousarray(mesh.coords, dtype=np.float64)
    conn = np.ascontiguousarray(mesh.conn, dtype=np.int32)
    mat = np.ascontiguousarray(mesh.mat, dtype=np.int32)
    bconn = np.ascontiguousarray(mesh.bconn, dtype=np.int32).reshape(-1, mesh.dim)
    bdir = np.ascontiguousarray(mesh.bdirichlet, dtype=np.uint8)
    m = RemoMesh(int(mesh.dim), coords.shape[0], ptr(coords, C.c_double), conn.shape[0], ptr(conn, C.c_int32),
                 ptr(mat, C.c_int32), bconn.shape[0], ptr(bconn, C.c_int32), ptr(bdir, C.c_uint8))
    return m, (coords, conn, mat, bconn, bdir)
