// symbolic_gpu.hip — dof numbering, Dirichlet elimination, row->element adjacency and CSR
// pattern of one batch mesh, entirely on the GPU.  Device counterpart of symbolic.cpp (which now
// only serves the CPU-side test hook remo_host_symbolic) and of what H1(order=3, dirichlet=...)
// plus the sparsity part of BilinearForm.Assemble() do in the reference
// (ngsolve_functions.py:27, 47).  Same numbering as symbolic.cpp (tests compare the two).
//
// The work is integer bookkeeping: hand-written gfx950 kernels generate / search keys; the
// device-wide radix sorts, scans and the unique-compaction are rocPRIM primitives.
//   edges : 64-bit keys (a << nbits | b)            sort + unique  -> edge numbers by rank
//   faces : 64-bit keys (a,b,c packed, 3 x nbits)   sort + unique
//   dofs  : flags of Dirichlet facets -> exclusive scan -> free numbering
//   adj   : (row, element<<5|local) pairs, stable sort by row
//   CSR   : (row << cbits | col) keys of every element, sort + unique -> col, rowptr by search
#include "symbolic_gpu.h"

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>

#include "fem_p3.h"

namespace remo {

namespace {

#define HIP_OK(expr)                                                                                 \
    do {                                                                                             \
        hipError_t e__ = (expr);                                                                     \
        if (e__ != hipSuccess) {                                                                     \
            err = std::string(#expr) + ": " + hipGetErrorString(e__);                                \
            return REMO_ERR_DEVICE;                                                                  \
        }                                                                                            \
    } while (0)

__device__ __forceinline__ void sort4(int32_t *c, int n) {
    for (int i = 1; i < n; ++i) {
        const int32_t v = c[i];
        int j = i - 1;
        while (j >= 0 && c[j] > v) { c[j + 1] = c[j]; --j; }
        c[j + 1] = v;
    }
}

template <int DIM>
__global__ void __launch_bounds__(256) k_sort_conn(int64_t nt, int64_t nv, const int32_t *__restrict__ in, int32_t *__restrict__ out,
                                                   int32_t *errflag) {
    constexpr int NB = DIM + 1;
    const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    int32_t c[NB];
#pragma unroll
    for (int a = 0; a < NB; ++a) c[a] = in[t * NB + a];
    sort4(c, NB);
    bool bad = (c[0] < 0 || c[NB - 1] >= nv);
#pragma unroll
    for (int a = 1; a < NB; ++a) bad |= (c[a] == c[a - 1]);
    if (bad) {
        atomicOr(errflag, 4);
#pragma unroll
        for (int a = 0; a < NB; ++a) c[a] = a;  // keep later kernels in range
    }
#pragma unroll
    for (int a = 0; a < NB; ++a) out[t * NB + a] = c[a];
}

// Element order.  Vertices arrive in a locality-preserving (Morton) order from the meshers; elements arrive in whatever order
// the triangulation produced them.  Sorting the elements by their two smallest vertices makes neighbours in the list neighbours
// in the mesh: the element-wise kernels (assembly walk, metric terms, the patch operator of patch.hip, which cuts the list into
// runs of elements that share most of their dofs) then touch compact ranges of the vectors.  eperm[t] = input element of sorted
// element t (the material array stays in input order).
template <int DIM>
__global__ void __launch_bounds__(256) k_element_order_keys(int64_t nt, const int32_t *__restrict__ conn, int nbits, uint64_t *__restrict__ keys,
                                                            int32_t *__restrict__ ids) {
    const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    keys[t] = (uint64_t(uint32_t(conn[t * (DIM + 1)])) << nbits) | uint32_t(conn[t * (DIM + 1) + 1]);
    ids[t] = int32_t(t);
}
template <int DIM>
__global__ void __launch_bounds__(256) k_gather_conn(int64_t nt, const int32_t *__restrict__ in, const int32_t *__restrict__ perm,
                                                     int32_t *__restrict__ out) {
    const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    const int64_t src = perm[t];
#pragma unroll
    for (int a = 0; a <= DIM; ++a) out[t * (DIM + 1) + a] = in[src * (DIM + 1) + a];
}

__device__ __forceinline__ uint64_t edge_key(int32_t a, int32_t b, int nbits) {
    if (a > b) { const int32_t t = a; a = b; b = t; }
    return (uint64_t(uint32_t(a)) << nbits) | uint32_t(b);
}
__device__ __forceinline__ uint64_t face_key(int32_t a, int32_t b, int32_t c, int nbits) {
    int32_t t;
    if (a > b) { t = a; a = b; b = t; }
    if (b > c) { t = b; b = c; c = t; }
    if (a > b) { t = a; a = b; b = t; }
    return (uint64_t(uint32_t(a)) << (2 * nbits)) | (uint64_t(uint32_t(b)) << nbits) | uint32_t(c);
}

template <int DIM>
__global__ void __launch_bounds__(256) k_entity_keys(int64_t nt, const int32_t *__restrict__ conn, int nbits, uint64_t *__restrict__ ekeys,
                                                     uint64_t *__restrict__ fkeys) {
    constexpr int NB = DIM + 1, NE = P3<DIM>::NEDGE;
    const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    int32_t c[NB];
#pragma unroll
    for (int a = 0; a < NB; ++a) c[a] = conn[t * NB + a];
#pragma unroll
    for (int e = 0; e < NE; ++e) ekeys[t * NE + e] = edge_key(c[edge_a(DIM, e)], c[edge_b(DIM, e)], nbits);
    if (DIM == 3) {
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            int a, b, cc;
            face_abc(f, a, b, cc);
            fkeys[t * 4 + f] = face_key(c[a], c[b], c[cc], nbits);
        }
    }
}

__device__ __forceinline__ int64_t find_key(const uint64_t *__restrict__ a, int64_t n, uint64_t k) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (a[mid] < k) lo = mid + 1; else hi = mid;
    }
    return (lo < n && a[lo] == k) ? lo : -1;
}

// global dof numbers of every element (before elimination)
template <int DIM>
__global__ void __launch_bounds__(256) k_eldof_global(int64_t nt, int64_t nv, int64_t ne, int64_t nf, const int32_t *__restrict__ conn,
                                                      const int32_t *__restrict__ eperm, const uint64_t *__restrict__ eku,
                                                      const uint64_t *__restrict__ fku, int nbits, int32_t *__restrict__ eldof) {
    constexpr int NB = DIM + 1, NE = P3<DIM>::NEDGE, N = P3<DIM>::NLD;
    const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    int32_t c[NB];
#pragma unroll
    for (int a = 0; a < NB; ++a) c[a] = conn[t * NB + a];
    int32_t *ed = eldof + t * N;
    int k = 0;
#pragma unroll
    for (int a = 0; a < NB; ++a) ed[k++] = c[a];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int64_t id = find_key(eku, ne, edge_key(c[edge_a(DIM, e)], c[edge_b(DIM, e)], nbits));
        ed[k++] = int32_t(nv + 2 * id);
        ed[k++] = int32_t(nv + 2 * id + 1);
    }
    if (DIM == 3) {
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            int a, b, cc;
            face_abc(f, a, b, cc);
            ed[k++] = int32_t(nv + 2 * ne + find_key(fku, nf, face_key(c[a], c[b], c[cc], nbits)));
        }
    } else {
        ed[k++] = int32_t(nv + 2 * ne + (eperm ? int64_t(eperm[t]) : t));  // cell bubble, numbered by the caller's element order (dropped below when condensed)
    }
}

// isfree[] starts at 1; every dof of a flagged boundary facet is cleared
template <int DIM>
__global__ void __launch_bounds__(256) k_mark_dirichlet(int64_t nbf, int64_t nv, int64_t ne, int64_t nf, const int32_t *__restrict__ bconn,
                                                        const uint8_t *__restrict__ bdir, const uint64_t *__restrict__ eku,
                                                        const uint64_t *__restrict__ fku, int nbits, int32_t *isfree, int32_t *errflag) {
    const int64_t b = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (b >= nbf) return;
    int32_t c[DIM];
    bool bad = false;
#pragma unroll
    for (int i = 0; i < DIM; ++i) {
        c[i] = bconn[b * DIM + i];
        bad |= (c[i] < 0 || c[i] >= nv);
    }
    if (bad) { atomicOr(errflag, 8); return; }
    if (!bdir[b]) return;
#pragma unroll
    for (int i = 0; i < DIM; ++i) isfree[c[i]] = 0;
#pragma unroll
    for (int i = 0; i < DIM; ++i)
#pragma unroll
        for (int j = i + 1; j < DIM; ++j) {
            const int64_t id = find_key(eku, ne, edge_key(c[i], c[j], nbits));
            if (id < 0) { atomicOr(errflag, 16); continue; }
            isfree[nv + 2 * id] = 0;
            isfree[nv + 2 * id + 1] = 0;
        }
    if (DIM == 3) {
        const int64_t id = find_key(fku, nf, face_key(c[0], c[1], c[2], nbits));
        if (id < 0) atomicOr(errflag, 16);
        else isfree[nv + 2 * ne + id] = 0;
    }
}

__global__ void __launch_bounds__(256) k_fill_i32(int64_t n, int32_t v, int32_t *p) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// freeid = isfree ? scan : -1   (scan = exclusive prefix sum of isfree)
__global__ void __launch_bounds__(256) k_freeid(int64_t n, const int32_t *__restrict__ isfree, const int32_t *__restrict__ scan,
                                                int32_t *__restrict__ freeid, int32_t *__restrict__ nfree_out, int64_t nv, int64_t nve) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    freeid[i] = isfree[i] ? scan[i] : -1;
    if (i == n - 1) nfree_out[0] = scan[i] + isfree[i];
    if (i == nv - 1) nfree_out[1] = scan[i] + isfree[i];  // free vertex dofs = leading block of the matrix
    if (i == nve - 1) nfree_out[2] = scan[i] + isfree[i]; // free vertex + edge dofs: edge rows end here
}

// eldof: global dof -> free row (or -1); adjacency pairs; CSR keys
template <int DIM>
__global__ void __launch_bounds__(256) k_element_rows(int64_t nt, int64_t ndof, int nld, const int32_t *__restrict__ freeid,
                                                      int32_t *__restrict__ eldof, uint32_t *__restrict__ adj_keys,
                                                      uint32_t *__restrict__ adj_vals) {
    constexpr int N = P3<DIM>::NLD;
    const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= nt) return;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int32_t g = eldof[t * N + i];
        const int32_t r = (i < nld && g < ndof) ? freeid[g] : -1;
        eldof[t * N + i] = r;
        if (i < nld) {
            adj_keys[t * nld + i] = (r >= 0) ? uint32_t(r) : 0xFFFFFFFFu;
            adj_vals[t * nld + i] = (uint32_t(t) << 5) | uint32_t(i);
        }
    }
}

// CSR pattern row by row (the default; the sort of all T x nld^2 element pairs below is the fallback): one wave per
// row walks the row's incident elements (adjacency above), puts their free dofs into an LDS hash set, counts (PASS 0)
// or compacts the set and writes it out in ascending order by rank counting (PASS 1).  The candidates are the
// same T x nld^2 pairs, but they never leave the CU: ~0.3 ms against ~1.8 ms for 25 M sorted keys at 63 k tetrahedra.
// A row with more than kRowSlots - 256 distinct columns raises err bit 32 and the caller sorts instead.
constexpr int kRowSlots = 1024;
template <int PASS>
__global__ void __launch_bounds__(256) k_row_pattern(int64_t nfree, int nld, int N, const int32_t *__restrict__ adjptr,     // nld: candidate local dofs per element (the first nld)
                                                     const uint32_t *__restrict__ adj, const int32_t *__restrict__ eldof,
                                                     int32_t *__restrict__ cnt, const int32_t *__restrict__ rowptr,
                                                     int32_t *__restrict__ col, int32_t *errflag) {
    __shared__ int32_t tab[4][kRowSlots];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t row = int64_t(blockIdx.x) * 4 + wave;
    const bool active = row < nfree;
    int32_t *T = tab[wave];
    for (int sl = lane; sl < kRowSlots; sl += 64) T[sl] = INT_MAX;
    __syncthreads();
    bool overflow = false;
    if (active) {
        const int32_t a0 = adjptr[row];
        const int32_t total = (adjptr[row + 1] - a0) * nld;
        for (int32_t item = lane; item < total; item += 64) {
            const int32_t ai = item / nld, j = item - ai * nld;
            const int32_t c = eldof[int64_t(adj[a0 + ai] >> 5) * N + j];
            if (c < 0) continue;
            uint32_t h = (uint32_t(c) * 2654435761u) >> 22;
            bool placed = false;
            for (int probe = 0; probe < kRowSlots; ++probe) {
                const int32_t old = atomicCAS(&T[h], INT_MAX, c);
                if (old == INT_MAX || old == c) { placed = true; break; }
                h = (h + 1) & (kRowSlots - 1);
            }
            overflow |= !placed;
        }
    }
    __syncthreads();
    // compact the occupied slots to the front of a list (wave-wide prefix by ballot); the list reuses the table from the top
    int c = 0;
    int32_t mine[kRowSlots / 64];
#pragma unroll
    for (int b = 0; b < kRowSlots / 64; ++b) mine[b] = T[b * 64 + lane];
    __syncthreads();
#pragma unroll
    for (int b = 0; b < kRowSlots / 64; ++b) {
        const bool occ = mine[b] != INT_MAX;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(occ);
        if (PASS == 1 && occ) T[c + __popcll(m & ((1ull << lane) - 1ull))] = mine[b];
        c += __popcll(m);
    }
    if (__builtin_amdgcn_ballot_w64(overflow) != 0 || c > kRowSlots - 256) {
        if (lane == 0) atomicOr(errflag, 32);
        c = 0;
    }
    if (PASS == 0) {
        if (active && lane == 0) cnt[row] = c;
        return;
    }
    __syncthreads();
    if (!active) return;
    const int32_t off = rowptr[row];
    for (int t = lane; t < c; t += 64) {      // ascending order by rank: the keys are distinct
        const int32_t key = T[t];
        int rank = 0;
        for (int i = 0; i < c; ++i) rank += T[i] < key;   // LDS broadcast reads
        col[off + rank] = key;
    }
}

template <int DIM>
__global__ void __launch_bounds__(256) k_coo_keys(int64_t nt, int nld, int64_t nfree, int cbits, const int32_t *__restrict__ eldof,
                                                  uint64_t *__restrict__ keys) {
    constexpr int N = P3<DIM>::NLD;
    const int64_t idx = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;  // one thread per (element, local row)
    if (idx >= nt * nld) return;
    const int64_t t = idx / nld;
    const int i = int(idx - t * nld);
    const int32_t r = eldof[t * N + i];
    // key = row << cbits | col with cbits = bits of nfree: only 2 cbits key bits take part in the radix sort
    const uint64_t sentinel = uint64_t(nfree) << cbits;
    uint64_t *out = keys + idx * nld;
    for (int j = 0; j < nld; ++j) {
        const int32_t c = eldof[t * N + j];
        out[j] = (r >= 0 && c >= 0) ? ((uint64_t(uint32_t(r)) << cbits) | uint32_t(c)) : sentinel;
    }
}

// ptr[r] = first position p with keys[p] >= r  (r = 0..n), keys sorted ascending
__global__ void __launch_bounds__(256) k_row_starts_u32(int64_t n, const uint32_t *__restrict__ keys, int64_t nkeys, int32_t *__restrict__ ptr) {
    const int64_t r = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (r > n) return;
    int64_t lo = 0, hi = nkeys;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < uint32_t(r)) lo = mid + 1; else hi = mid;
    }
    ptr[r] = int32_t(lo);
}
__global__ void __launch_bounds__(256) k_row_starts_u64(int64_t n, int cbits, const uint64_t *__restrict__ keys, int64_t nkeys, int32_t *__restrict__ ptr) {
    const int64_t r = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (r > n) return;
    const uint64_t k = uint64_t(r) << cbits;
    int64_t lo = 0, hi = nkeys;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < k) lo = mid + 1; else hi = mid;
    }
    ptr[r] = int32_t(lo);
}
__global__ void __launch_bounds__(256) k_low32(int64_t n, int cbits, const uint64_t *__restrict__ keys, int32_t *__restrict__ col) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) col[i] = int32_t(uint32_t(keys[i] & ((uint64_t(1) << cbits) - 1)));
}

int g_row_pattern = 1;   // 0: always build the pattern by sorting (A/B probe, set_symbolic_tuning)
int g_element_order = 1; // 0: keep the caller's element order (A/B probe, set_element_order)
inline int grid_for(int64_t n) { return int((n + 255) / 256); }
inline int bits_for(uint64_t v) {  // bits needed to represent values 0..v
    int b = 1;
    while ((v >> b) != 0) ++b;
    return b;
}

}  // namespace

size_t symbolic_gpu_arena_bytes(int dim, int64_t nv, int64_t nt, int64_t nbf) {
    // generous upper bound: persistent outputs + scratch (two COO key buffers + sort temporaries)
    const int64_t N = (dim == 2) ? 10 : 20;
    const int64_t ndof_max = nv + (dim == 2 ? 7 : 16) * nt;
    size_t b = 0;
    b += size_t(nt) * (dim + 1) * 4 + size_t(nt) * N * 4 + size_t(ndof_max) * 4 * 3;   // conn, eldof, freeid/isfree/scan
    b += size_t(ndof_max + 1) * 4 * 2;                                                  // rowptr, adjptr
    b += size_t(nt) * N * 4 * 5;                                                        // adj pairs in/out + adj
    b += size_t(nt) * N * N * 4;                                                        // col (<= all pairs)
    b += size_t(nt) * N * N * 8 * 5;                                                    // COO keys in/sorted/unique + sort & select temporaries
    b += size_t(nt) * 10 * 8 * 4;                                                       // entity keys
    b += size_t(nt) * (4 + (dim + 1) * 4 + 4 + 16 + 16);                                // element order: permutation, scratch
    b += 64 << 20;
    return b;
}

int build_symbolic_gpu(Arena &ar, hipStream_t s, int dim, int64_t nv, int64_t nt, const int32_t *d_conn_in, int64_t nbf,
                       const int32_t *d_bconn, const uint8_t *d_bdir, bool condense_in, int32_t *d_err, DeviceSymbolic &out,
                       std::string &err, int64_t vertex_block_above) {
    out = DeviceSymbolic();
    if (dim != 2 && dim != 3) { err = "dim must be 2 or 3"; return REMO_ERR_ARG; }
    const int nb = dim + 1;
    const int N = (dim == 2) ? 10 : 20;
    const int NE = (dim == 2) ? 3 : 6;
    const bool condense = (dim == 2) && condense_in;
    const int nld = condense ? 9 : N;
    const int nbits = bits_for(uint64_t(nv));
    if (dim == 3 && 3 * nbits > 63) { err = "too many vertices for packed face keys"; return REMO_ERR_ARG; }
    if (nt >= (int64_t(1) << 27)) { err = "too many elements"; return REMO_ERR_ARG; }
    out.dim = dim; out.nld = nld; out.nld_full = N; out.condense = condense; out.nv = nv; out.nt = nt;

    // ---- persistent outputs that are sized by the mesh alone (bottom of the arena) ------------
    out.conn = ar.lo<int32_t>(nt * nb);
    out.eldof = ar.lo<int32_t>(nt * N);
    int32_t *d_cnt = ar.lo<int32_t>(8);           // [0]=nfree
    size_t *d_ucount = ar.lo<size_t>(4);          // unique counts
    HIP_OK(hipMemsetAsync(d_err, 0, sizeof(int32_t), s));
    if (g_element_order) {
        out.eperm = ar.lo<int32_t>(nt);
        const size_t mark = ar.hi_mark();
        int32_t *conn_tmp = ar.hi<int32_t>(nt * nb), *ids = ar.hi<int32_t>(nt);
        uint64_t *k_in = ar.hi<uint64_t>(nt), *k_out = ar.hi<uint64_t>(nt);
        if (dim == 2) {
            hipLaunchKernelGGL(k_sort_conn<2>, dim3(grid_for(nt)), dim3(256), 0, s, nt, nv, d_conn_in, conn_tmp, d_err);
            hipLaunchKernelGGL(k_element_order_keys<2>, dim3(grid_for(nt)), dim3(256), 0, s, nt, conn_tmp, nbits, k_in, ids);
        } else {
            hipLaunchKernelGGL(k_sort_conn<3>, dim3(grid_for(nt)), dim3(256), 0, s, nt, nv, d_conn_in, conn_tmp, d_err);
            hipLaunchKernelGGL(k_element_order_keys<3>, dim3(grid_for(nt)), dim3(256), 0, s, nt, conn_tmp, nbits, k_in, ids);
        }
        size_t tb = 0;
        HIP_OK(rocprim::radix_sort_pairs(nullptr, tb, k_in, k_out, ids, out.eperm, size_t(nt), 0u, unsigned(2 * nbits), s));
        void *tmp = ar.hi<char>(tb + 256);
        HIP_OK(rocprim::radix_sort_pairs(tmp, tb, k_in, k_out, ids, out.eperm, size_t(nt), 0u, unsigned(2 * nbits), s));
        if (dim == 2) hipLaunchKernelGGL(k_gather_conn<2>, dim3(grid_for(nt)), dim3(256), 0, s, nt, conn_tmp, out.eperm, out.conn);
        else hipLaunchKernelGGL(k_gather_conn<3>, dim3(grid_for(nt)), dim3(256), 0, s, nt, conn_tmp, out.eperm, out.conn);
        ar.hi_release(mark);     // the stream orders the kernels that reuse this scratch behind the gather
    } else {
        if (dim == 2) hipLaunchKernelGGL(k_sort_conn<2>, dim3(grid_for(nt)), dim3(256), 0, s, nt, nv, d_conn_in, out.conn, d_err);
        else hipLaunchKernelGGL(k_sort_conn<3>, dim3(grid_for(nt)), dim3(256), 0, s, nt, nv, d_conn_in, out.conn, d_err);
    }

    // ---- edges / faces: sort + unique ----------------------------------------------------------
    const size_t hi_mark0 = ar.hi_mark();
    const int64_t nek = nt * NE, nfk = (dim == 3) ? nt * 4 : 0;
    uint64_t *ek_in = ar.hi<uint64_t>(nek), *ek_sorted = ar.hi<uint64_t>(nek);
    uint64_t *fk_in = ar.hi<uint64_t>(nfk ? nfk : 1), *fk_sorted = ar.hi<uint64_t>(nfk ? nfk : 1);
    uint64_t *eku = ar.lo<uint64_t>(nek);               // unique edge keys stay until the Dirichlet pass
    uint64_t *fku = ar.lo<uint64_t>(nfk ? nfk : 1);
    if (dim == 2) hipLaunchKernelGGL(k_entity_keys<2>, dim3(grid_for(nt)), dim3(256), 0, s, nt, out.conn, nbits, ek_in, fk_in);
    else hipLaunchKernelGGL(k_entity_keys<3>, dim3(grid_for(nt)), dim3(256), 0, s, nt, out.conn, nbits, ek_in, fk_in);
    {
        size_t tb = 0;
        HIP_OK(rocprim::radix_sort_keys(nullptr, tb, ek_in, ek_sorted, size_t(nek), 0u, unsigned(2 * nbits), s));
        void *tmp = ar.hi<char>(tb + 256);
        HIP_OK(rocprim::radix_sort_keys(tmp, tb, ek_in, ek_sorted, size_t(nek), 0u, unsigned(2 * nbits), s));
        size_t ub = 0;
        HIP_OK(rocprim::unique(nullptr, ub, ek_sorted, eku, d_ucount, size_t(nek), rocprim::equal_to<uint64_t>(), s));
        void *tmp2 = ar.hi<char>(ub + 256);
        HIP_OK(rocprim::unique(tmp2, ub, ek_sorted, eku, d_ucount, size_t(nek), rocprim::equal_to<uint64_t>(), s));
        if (dim == 3) {
            size_t tb3 = 0;
            HIP_OK(rocprim::radix_sort_keys(nullptr, tb3, fk_in, fk_sorted, size_t(nfk), 0u, unsigned(3 * nbits), s));
            void *tmp3 = ar.hi<char>(tb3 + 256);
            HIP_OK(rocprim::radix_sort_keys(tmp3, tb3, fk_in, fk_sorted, size_t(nfk), 0u, unsigned(3 * nbits), s));
            size_t ub3 = 0;
            HIP_OK(rocprim::unique(nullptr, ub3, fk_sorted, fku, d_ucount + 1, size_t(nfk), rocprim::equal_to<uint64_t>(), s));
            void *tmp4 = ar.hi<char>(ub3 + 256);
            HIP_OK(rocprim::unique(tmp4, ub3, fk_sorted, fku, d_ucount + 1, size_t(nfk), rocprim::equal_to<uint64_t>(), s));
        }
    }
    size_t h_uc[2] = {0, 0};
    HIP_OK(hipMemcpyAsync(h_uc, d_ucount, sizeof(size_t) * 2, hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    ar.hi_release(hi_mark0);
    const int64_t ne = int64_t(h_uc[0]), nf = (dim == 3) ? int64_t(h_uc[1]) : 0;
    out.ne = ne; out.nf = nf;
    const int64_t ndof = nv + 2 * ne + nf + ((dim == 2 && !condense) ? nt : 0);
    if (ndof >= (int64_t(1) << 31)) { err = "too many dofs for 32-bit indices"; return REMO_ERR_ARG; }
    out.ndof = ndof;

    // ---- element dofs, Dirichlet flags, free numbering -----------------------------------------
    if (dim == 2) hipLaunchKernelGGL(k_eldof_global<2>, dim3(grid_for(nt)), dim3(256), 0, s, nt, nv, ne, nf, out.conn, (const int32_t *)out.eperm, eku, fku, nbits, out.eldof);
    else hipLaunchKernelGGL(k_eldof_global<3>, dim3(grid_for(nt)), dim3(256), 0, s, nt, nv, ne, nf, out.conn, (const int32_t *)out.eperm, eku, fku, nbits, out.eldof);
    out.freeid = ar.lo<int32_t>(ndof);
    const size_t hi_mark1 = ar.hi_mark();
    int32_t *isfree = ar.hi<int32_t>(ndof), *scan = ar.hi<int32_t>(ndof);
    hipLaunchKernelGGL(k_fill_i32, dim3(grid_for(ndof)), dim3(256), 0, s, ndof, 1, isfree);
    if (nbf > 0) {
        if (dim == 2) hipLaunchKernelGGL(k_mark_dirichlet<2>, dim3(grid_for(nbf)), dim3(256), 0, s, nbf, nv, ne, nf, d_bconn, d_bdir, eku, fku, nbits, isfree, d_err);
        else hipLaunchKernelGGL(k_mark_dirichlet<3>, dim3(grid_for(nbf)), dim3(256), 0, s, nbf, nv, ne, nf, d_bconn, d_bdir, eku, fku, nbits, isfree, d_err);
    }
    {
        size_t tb = 0;
        HIP_OK(rocprim::exclusive_scan(nullptr, tb, isfree, scan, int32_t(0), size_t(ndof), rocprim::plus<int32_t>(), s));
        void *tmp = ar.hi<char>(tb + 256);
        HIP_OK(rocprim::exclusive_scan(tmp, tb, isfree, scan, int32_t(0), size_t(ndof), rocprim::plus<int32_t>(), s));
    }
    hipLaunchKernelGGL(k_freeid, dim3(grid_for(ndof)), dim3(256), 0, s, ndof, isfree, scan, out.freeid, d_cnt, nv, nv + 2 * ne);
    int32_t h_cnt[3] = {0, 0, 0}, h_err = 0;
    HIP_OK(hipMemcpyAsync(h_cnt, d_cnt, 3 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_OK(hipMemcpyAsync(&h_err, d_err, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    ar.hi_release(hi_mark1);
    if (h_err & 4) { err = "element vertex index out of range or repeated"; return REMO_ERR_MESH; }
    if (h_err & 8) { err = "boundary facet vertex out of range"; return REMO_ERR_MESH; }
    if (h_err & 16) { err = "Dirichlet facet is not a facet of the mesh"; return REMO_ERR_MESH; }
    const int64_t nfree = h_cnt[0];
    if (nfree <= 0) { err = "no free dofs"; return REMO_ERR_MESH; }
    out.nfree = nfree;
    out.nvfree = h_cnt[1];
    out.nvefree = h_cnt[2];

    // ---- element rows, adjacency --------------------------------------------------------------
    const int64_t npairs = nt * nld;
    out.adjptr = ar.lo<int32_t>(nfree + 1);
    out.rowptr = ar.lo<int32_t>(nfree + 1);
    out.adj = ar.lo<uint32_t>(npairs);
    const size_t hi_mark2 = ar.hi_mark();
    uint32_t *ak_in = ar.hi<uint32_t>(npairs), *ak_out = ar.hi<uint32_t>(npairs), *av_in = ar.hi<uint32_t>(npairs);
    if (dim == 2) hipLaunchKernelGGL(k_element_rows<2>, dim3(grid_for(nt)), dim3(256), 0, s, nt, ndof, nld, out.freeid, out.eldof, ak_in, av_in);
    else hipLaunchKernelGGL(k_element_rows<3>, dim3(grid_for(nt)), dim3(256), 0, s, nt, ndof, nld, out.freeid, out.eldof, ak_in, av_in);
    {
        size_t tb = 0;
        HIP_OK(rocprim::radix_sort_pairs(nullptr, tb, ak_in, ak_out, av_in, out.adj, size_t(npairs), 0u, 32u, s));
        void *tmp = ar.hi<char>(tb + 256);
        HIP_OK(rocprim::radix_sort_pairs(tmp, tb, ak_in, ak_out, av_in, out.adj, size_t(npairs), 0u, 32u, s));
    }
    hipLaunchKernelGGL(k_row_starts_u32, dim3(grid_for(nfree + 1)), dim3(256), 0, s, nfree, ak_out, npairs, out.adjptr);

    // ---- CSR pattern, row by row through LDS ---------------------------------------------------------
    // vertex_block_only: the caller applies A without its stored entries (patch operator) and only the leading P1 block (vertex
    // rows x vertex columns: the preconditioner's coarse matrix) gets a pattern - rows [0, nvfree), candidates = the dim + 1
    // vertex dofs of every incident element.  The pattern of a 2 M-row matrix costs 2.4 ms per batch, its values 2.1 ms.
    out.vertex_block_only = dim == 3 && vertex_block_above >= 0 && nt > vertex_block_above && out.nvfree > 0 && nfree < (int64_t(1) << 23);   // (the patch operator addresses rows and slab slots with 24 bits)
    const int64_t prow_n = out.vertex_block_only ? out.nvfree : nfree;
    const int pcand = out.vertex_block_only ? nb : nld;
    if (g_row_pattern || out.vertex_block_only) {
        int32_t *cnt = ar.hi<int32_t>(prow_n + 1);
        HIP_OK(hipMemsetAsync(cnt + prow_n, 0, sizeof(int32_t), s));
        const int gp = int((prow_n + 3) / 4);
        hipLaunchKernelGGL((k_row_pattern<0>), dim3(gp), dim3(256), 0, s, prow_n, pcand, N, out.adjptr, out.adj, out.eldof, cnt, (const int32_t *)nullptr,
                           (int32_t *)nullptr, d_err);
        {
            size_t tb = 0;
            HIP_OK(rocprim::exclusive_scan(nullptr, tb, cnt, out.rowptr, int32_t(0), size_t(prow_n + 1), rocprim::plus<int32_t>(), s));
            void *tmp = ar.hi<char>(tb + 256);
            HIP_OK(rocprim::exclusive_scan(tmp, tb, cnt, out.rowptr, int32_t(0), size_t(prow_n + 1), rocprim::plus<int32_t>(), s));
        }
        int32_t h_nnz = 0, h_err2 = 0;
        HIP_OK(hipMemcpyAsync(&h_nnz, out.rowptr + prow_n, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        HIP_OK(hipMemcpyAsync(&h_err2, d_err, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        HIP_OK(hipStreamSynchronize(s));
        if ((h_err2 & 32) && out.vertex_block_only) { err = "a vertex with more than 768 neighbours"; return REMO_ERR_MESH; }
        if (!(h_err2 & 32)) {
            if (h_nnz <= 0) { err = "empty matrix pattern"; return REMO_ERR_MESH; }
            out.nnz = h_nnz;
            out.col = ar.lo<int32_t>(h_nnz);
            hipLaunchKernelGGL((k_row_pattern<1>), dim3(gp), dim3(256), 0, s, prow_n, pcand, N, out.adjptr, out.adj, out.eldof, (int32_t *)nullptr,
                               (const int32_t *)out.rowptr, out.col, d_err);
            HIP_OK(hipStreamSynchronize(s));  // scratch is released below
            ar.hi_release(hi_mark2);
            out.nadj = 0;
            return REMO_OK;
        }
        // a row with more distinct columns than the LDS table holds: clear the bit and sort instead
        int32_t cleared = h_err2 & ~32;
        HIP_OK(hipMemcpyAsync(d_err, &cleared, sizeof(int32_t), hipMemcpyHostToDevice, s));
        HIP_OK(hipStreamSynchronize(s));
    }

    // ---- CSR pattern (fallback): sort + unique of every element's (row, col) keys -----------------
    const int64_t ncoo = nt * int64_t(nld) * nld;
    uint64_t *ck_in = ar.hi<uint64_t>(ncoo), *ck_sorted = ar.hi<uint64_t>(ncoo), *ck_u = ar.hi<uint64_t>(ncoo);
    const int cbits = bits_for(uint64_t(nfree));
    if (dim == 2) hipLaunchKernelGGL(k_coo_keys<2>, dim3(grid_for(npairs)), dim3(256), 0, s, nt, nld, nfree, cbits, out.eldof, ck_in);
    else hipLaunchKernelGGL(k_coo_keys<3>, dim3(grid_for(npairs)), dim3(256), 0, s, nt, nld, nfree, cbits, out.eldof, ck_in);
    {
        const unsigned end_bit = unsigned(2 * cbits);
        size_t tb = 0;
        HIP_OK(rocprim::radix_sort_keys(nullptr, tb, ck_in, ck_sorted, size_t(ncoo), 0u, end_bit, s));
        void *tmp = ar.hi<char>(tb + 256);
        HIP_OK(rocprim::radix_sort_keys(tmp, tb, ck_in, ck_sorted, size_t(ncoo), 0u, end_bit, s));
        size_t ub = 0;
        HIP_OK(rocprim::unique(nullptr, ub, ck_sorted, ck_u, d_ucount + 2, size_t(ncoo), rocprim::equal_to<uint64_t>(), s));
        void *tmp2 = ar.hi<char>(ub + 256);
        HIP_OK(rocprim::unique(tmp2, ub, ck_sorted, ck_u, d_ucount + 2, size_t(ncoo), rocprim::equal_to<uint64_t>(), s));
    }
    size_t h_u = 0;
    HIP_OK(hipMemcpyAsync(&h_u, d_ucount + 2, sizeof(size_t), hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    uint64_t h_last = 0;
    if (h_u > 0) HIP_OK(hipMemcpy(&h_last, ck_u + (h_u - 1), sizeof(uint64_t), hipMemcpyDeviceToHost));
    int64_t nnz = int64_t(h_u);
    if (h_u > 0 && (h_last >> cbits) == uint64_t(nfree)) nnz -= 1;  // the sentinel of constrained pairs
    if (nnz <= 0 || nnz >= (int64_t(1) << 31)) { err = "nnz out of range for 32-bit row pointers"; return REMO_ERR_ARG; }
    out.nnz = nnz;
    out.col = ar.lo<int32_t>(nnz);
    hipLaunchKernelGGL(k_low32, dim3(grid_for(nnz)), dim3(256), 0, s, nnz, cbits, ck_u, out.col);
    hipLaunchKernelGGL(k_row_starts_u64, dim3(grid_for(nfree + 1)), dim3(256), 0, s, nfree, cbits, ck_u, nnz, out.rowptr);
    HIP_OK(hipStreamSynchronize(s));  // scratch is released below
    ar.hi_release(hi_mark2);
    out.nadj = 0;  // adjptr[nfree] on the device holds it
    return REMO_OK;
}

void set_symbolic_tuning(int row_pattern) { g_row_pattern = row_pattern; }
void set_element_order(int on) { g_element_order = on; }

}  // namespace remo
