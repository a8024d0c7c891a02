// symbolic_gpu.h — device-side dof numbering / CSR pattern (symbolic_gpu.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <stdexcept>
#include <string>

#include "../../include/remo3d_hip.h"

namespace remo {

// Two-ended bump allocator over one device allocation: results grow from the bottom and stay,
// scratch grows from the top and is released phase by phase.
struct Arena {
    char *base = nullptr;
    size_t cap = 0, lo_off = 0, hi_off = 0;
    static size_t up(size_t x) { return (x + 255) / 256 * 256; }
    template <class T> T *lo(size_t count) {
        const size_t b = up(count * sizeof(T));
        if (lo_off + b + hi_off > cap) throw std::runtime_error("device arena exhausted");
        T *p = reinterpret_cast<T *>(base + lo_off);
        lo_off += b;
        return p;
    }
    template <class T> T *hi(size_t count) {
        const size_t b = up(count * sizeof(T));
        if (lo_off + b + hi_off > cap) throw std::runtime_error("device arena exhausted");
        hi_off += b;
        return reinterpret_cast<T *>(base + cap - hi_off);
    }
    size_t hi_mark() const { return hi_off; }
    void hi_release(size_t mark) { hi_off = mark; }
    void reset() { lo_off = hi_off = 0; }
};

struct DeviceSymbolic {
    int dim = 0, nld = 0, nld_full = 0;
    bool condense = false;
    int64_t nv = 0, nt = 0, ne = 0, nf = 0, ndof = 0, nfree = 0, nnz = 0, nadj = 0;
    int64_t nvfree = 0;  // free vertex dofs: rows/cols [0, nvfree) are the P1 block of the matrix
    int64_t nvefree = 0; // free vertex + edge dofs: rows [nvfree, nvefree) are edge dofs, two consecutive rows per edge
    int32_t *conn = nullptr;    // [nt][dim+1] ascending per element; elements sorted by their two smallest vertices
    int32_t *eperm = nullptr;   // [nt] input element of element t (nullptr: input order kept)
    int32_t *eldof = nullptr;   // [nt][nld_full] free row or -1
    int32_t *freeid = nullptr;  // [ndof]
    bool vertex_block_only = false;   // rowptr / col / nnz describe only the leading nvfree x nvfree (P1) block: the caller applies A matrix-free
    int32_t *rowptr = nullptr;  // [nfree+1] (vertex_block_only: [nvfree+1] valid)
    int32_t *col = nullptr;     // [nnz]
    int32_t *adjptr = nullptr;  // [nfree+1]
    uint32_t *adj = nullptr;    // element << 5 | local dof, ascending per row
};

size_t symbolic_gpu_arena_bytes(int dim, int64_t nv, int64_t nt, int64_t nbf);

// All device work is enqueued on `s`; the function synchronises three times to read sizes back.
int build_symbolic_gpu(Arena &ar, hipStream_t s, int dim, int64_t nv, int64_t nt, const int32_t *d_conn_in, int64_t nbf,
                       const int32_t *d_bconn, const uint8_t *d_bdir, bool condense, int32_t *d_err, DeviceSymbolic &out,
                       std::string &err, int64_t vertex_block_above = -1);   // >= 0: 3D meshes with more elements get the pattern of the P1 block only

// probe hook: 0 = build the CSR pattern by the global sort instead of row by row
void set_symbolic_tuning(int row_pattern);
void set_element_order(int on);   // probe hook: 0 = keep the caller's element order

}  // namespace remo
