// amg.hip — smoothed-aggregation multigrid on the P1 (vertex) block, gfx950.
//
// Why: in 2D (graded axisymmetric meshes, 80 k vertices at config 2) the Chebyshev polynomial on the vertex block needs degree
// 28 on [lmax / 2400, lmax] - 14 paired launches, two thirds of a PCG step - and still takes twice the steps of an exact
// vertex solve (offline, tools/amg_study_2d.py: exact 58, Chebyshev(28) 119, this cycle 71 steps at 3.7 sweeps of the block).
//
// Setup per batch (the matrix changes with every batch), all on the device, deterministic (no floating-point atomics):
//   aggregation   distance-2 maximal independent set of the matrix graph by hashed priorities (Bell / Dalton / Olson 2012):
//                 roots, then their neighbours, then the neighbours' neighbours by the strongest connection
//   prolongator   P = (I - 4 / (3 lmax) D^-1 A) P0, one generic row-wise sparse product (LDS hash set per wave for the
//                 pattern, sorted; values by sorted-row lookups in a fixed order),
//                 R = P^T by count / scan / scatter and an LDS sort of every row
//   coarse matrix A' = R (A P), two more products
//   coarsest      dense inverse (Gauss-Jordan in LDS, <= 64 rows)
// Cycle: V(1,1) with damped Jacobi (omega = 1.6 / lmax, the one-term Chebyshev polynomial of [lmax / 4, lmax]), symmetric, so
// the PCG sees a fixed symmetric positive definite preconditioner.  The host waits ONCE per level during the setup (number of
// aggregates); product sizes stay on the device (capacity-bound arrays, clamped row pointers, flags read at the next wait).
#include <hip/hip_runtime.h>
#include <limits.h>

#include <rocprim/device/device_scan.hpp>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "amg.h"
#include "kernels.h"
#include "wave_util.h"

namespace remo {
namespace {

#define HIP_OK(expr)                                                                                       \
    do {                                                                                                   \
        hipError_t e__ = (expr);                                                                           \
        if (e__ != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ bool amg_done(const double *scal, int step) {
    if (!scal) return false;
    const int d = reinterpret_cast<const int *>(scal + kDoneSlot)[0];
    return d != 0 && d <= step;
}

__device__ __forceinline__ int32_t lower_bound_i32(const int32_t *__restrict__ a, int32_t lo, int32_t hi, int32_t key) {
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// ---- leading block -> compact CSR ---------------------------------------------------------------------------------------
template <int PASS>
__global__ void __launch_bounds__(256) k_block_copy(int64_t nv, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                    const double *__restrict__ val, int32_t *__restrict__ cnt, const int32_t *__restrict__ orp,
                                                    int32_t *__restrict__ oc, double *__restrict__ ov) {
    const int64_t row = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (row >= nv) return;
    const int32_t rs = rowptr[row], re = rowptr[row + 1];
    if (PASS == 0) {
        cnt[row] = lower_bound_i32(col, rs, re, int32_t(nv)) - rs;
    } else {
        const int32_t at = orp[row], len = orp[row + 1] - at;
        for (int32_t e = 0; e < len; ++e) { oc[at + e] = col[rs + e]; ov[at + e] = val[rs + e]; }
    }
}

// Jacobi factors and the Gershgorin bound of D^-1 A; flag |= 1 when a diagonal entry is missing or not positive
__global__ void __launch_bounds__(256) k_level_diag(int64_t n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                    const double *__restrict__ val, double *__restrict__ dinv, unsigned long long *bound_bits,
                                                    int32_t *flag) {
    double m = 0.0;
    const int64_t row = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (row < n) {
        double s = 0.0, d = 0.0;
        for (int32_t p = rowptr[row]; p < rowptr[row + 1]; ++p) {
            s += fabs(val[p]);
            if (col[p] == row) d = val[p];
        }
        if (!(d > 0.0)) { atomicOr(flag, 1); d = 1.0; }
        dinv[row] = 1.0 / d;
        m = s / d;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(bound_bits, (unsigned long long)__double_as_longlong(m));   // positive doubles order like their bits
}

// ---- aggregation: distance-2 maximal independent set ---------------------------------------------------------------------
// tuple = state (2 bits: 2 root, 1 undecided, 0 removed) | hashed priority (30 bits) | index (32 bits): unique, so maxima are too
constexpr uint64_t kTupMask = (uint64_t(1) << 62) - 1;

__device__ __forceinline__ uint32_t hash_u32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__global__ void __launch_bounds__(256) k_mis_init(int64_t n, uint64_t *__restrict__ tup) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) tup[i] = (uint64_t(1) << 62) | (uint64_t(hash_u32(uint32_t(i)) >> 2) << 32) | uint64_t(uint32_t(i));
}

__global__ void __launch_bounds__(256) k_mis_max(int64_t n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                 const uint64_t *__restrict__ tup, uint64_t *__restrict__ m1) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t m = tup[i];
    for (int32_t p = rowptr[i]; p < rowptr[i + 1]; ++p) { const uint64_t t = tup[col[p]]; m = t > m ? t : m; }
    m1[i] = m;
}

// second hop + decision; tup is updated in place (the other threads read m1 only)
__global__ void __launch_bounds__(256) k_mis_update(int64_t n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                    uint64_t *__restrict__ tup, const uint64_t *__restrict__ m1) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t t = tup[i];
    if ((t >> 62) != 1) return;
    uint64_t m = m1[i];
    for (int32_t p = rowptr[i]; p < rowptr[i + 1]; ++p) { const uint64_t v = m1[col[p]]; m = v > m ? v : m; }
    if (m == t) tup[i] = (uint64_t(2) << 62) | (t & kTupMask);
    else if ((m >> 62) == 2) tup[i] = t & kTupMask;
}

// After the fixed number of iterations a few nodes (0.6 % at 80 k rows after six) are still undecided: they have no root
// within two hops.  One more sweep settles them without further iterations: root if no undecided neighbour has a higher tuple,
// removed otherwise (such a node sits next to a new root or next to a node that does, and joins through k_agg_near / k_agg_far;
// the odd node at the end of a longer chain stays without an aggregate: it is smoothed, not coarse-corrected).  out = final tuples.
__global__ void __launch_bounds__(256) k_mis_finish(int64_t n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                    const uint64_t *__restrict__ tup, uint64_t *__restrict__ out) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t t = tup[i];
    uint64_t r = t;
    if ((t >> 62) == 1) {
        bool top = true;
        for (int32_t p = rowptr[i]; p < rowptr[i + 1]; ++p) {
            const uint64_t v = tup[col[p]];
            if ((v >> 62) == 1 && v > t) top = false;
        }
        r = top ? ((uint64_t(2) << 62) | (t & kTupMask)) : (t & kTupMask);
    }
    out[i] = r;
}

__global__ void __launch_bounds__(256) k_mis_flags(int64_t n, const uint64_t *__restrict__ tup, int32_t *__restrict__ isroot) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) isroot[i] = (tup[i] >> 62) == 2 ? 1 : 0;
}

// roots take their number, the others the root neighbour of the highest priority (-1: none)
__global__ void __launch_bounds__(256) k_agg_near(int64_t n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                  const uint64_t *__restrict__ tup, const int32_t *__restrict__ id, int32_t *__restrict__ agg) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if ((tup[i] >> 62) == 2) { agg[i] = id[i]; return; }
    uint64_t best = 0;
    int32_t bj = -1;
    for (int32_t p = rowptr[i]; p < rowptr[i + 1]; ++p) {
        const int32_t j = col[p];
        const uint64_t t = tup[j];
        if ((t >> 62) == 2 && t > best) { best = t; bj = j; }
    }
    agg[i] = bj >= 0 ? id[bj] : -1;
}

// the rest: the aggregate of the neighbour with the strongest connection (ties: the lower index, columns ascend)
__global__ void __launch_bounds__(256) k_agg_far(int64_t n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                 const double *__restrict__ val, const int32_t *__restrict__ agg_in, int32_t *__restrict__ agg_out) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t a = agg_in[i];
    if (a < 0) {
        double w = -1.0;
        for (int32_t p = rowptr[i]; p < rowptr[i + 1]; ++p) {
            const int32_t j = col[p];
            if (j == i) continue;
            const int32_t aj = agg_in[j];
            const double v = fabs(val[p]);
            if (aj >= 0 && v > w) { w = v; a = aj; }
        }
    }
    agg_out[i] = a;
}

// S = I - wp D^-1 A on the pattern of A, and the tentative prolongator (one unit entry per row) as CSR
__global__ void __launch_bounds__(256) k_smoothing_factor(int64_t n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                          const double *__restrict__ val, const double *__restrict__ dinv, const unsigned long long *bound_bits,
                                                          double *__restrict__ sval, const int32_t *__restrict__ agg, int32_t *__restrict__ p0rp,
                                                          int32_t *__restrict__ p0c, double *__restrict__ p0v) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i > n) return;
    p0rp[i] = int32_t(i);
    if (i == n) return;
    const double f = 4.0 / (3.0 * __longlong_as_double((long long)bound_bits[0])) * dinv[i];   // prolongator smoothing 4 / (3 lmax)
    for (int32_t p = rowptr[i]; p < rowptr[i + 1]; ++p) sval[p] = (col[p] == i ? 1.0 : 0.0) - f * val[p];
    const int32_t a = agg[i];
    p0c[i] = a >= 0 ? a : 0;
    p0v[i] = a >= 0 ? 1.0 : 0.0;
}

// ---- generic sparse product C = X Y, one wave per row of X ----------------------------------------------------------------
// PASS 0 counts the distinct columns of every row; PASS 1 (row offsets known) writes them sorted with their values.
// Columns of Y rows ascend (binary search); X rows in any order.  flag |= 1: a row does not fit the hash table.
template <int PASS, int SLOTS>
__global__ void __launch_bounds__(256) k_spgemm(int64_t nx, const int32_t *__restrict__ xrp, const int32_t *__restrict__ xc, const double *__restrict__ xv,
                                                const int32_t *__restrict__ yrp, const int32_t *__restrict__ yc, const double *__restrict__ yv,
                                                int32_t *__restrict__ cnt, const int32_t *__restrict__ crp, int32_t *__restrict__ cc,
                                                double *__restrict__ cv, int32_t *flag) {   // PASS 1: crp has been clamped to the capacity of cc / cv (k_clamp_rowptr)
    __shared__ int32_t keys[4][SLOTS];
    __shared__ int32_t list[4][SLOTS];
    __shared__ double terms[4][PASS == 1 ? SLOTS / 2 : 1];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t row = int64_t(blockIdx.x) * 4 + wave;
    if (row >= nx) return;                      // no block-wide barriers below: a wave works on its own tables
    int32_t *K = keys[wave], *L = list[wave];
    for (int sl = lane; sl < SLOTS; sl += 64) K[sl] = INT_MAX;
    wave_sync();
    const int32_t xs = xrp[row], xe = xrp[row + 1];
    bool overflow = false;
    for (int32_t p = xs + lane; p < xe; p += 64) {
        const int32_t k = xc[p];
        const int32_t ye = yrp[k + 1];
        for (int32_t q = yrp[k]; q < ye; ++q) {
            const int32_t j = yc[q];
            uint32_t h = ((uint32_t(j) * 2654435761u) >> 16) & (SLOTS - 1);
            bool placed = false;
            for (int probe = 0; probe < SLOTS; ++probe) {
                const int32_t old = atomicCAS(&K[h], INT_MAX, j);
                if (old == INT_MAX || old == j) { placed = true; break; }
                h = (h + 1) & (SLOTS - 1);
            }
            overflow |= !placed;
        }
    }
    wave_sync();
    // compact the occupied slots into the list (any order), count them
    int c = 0;
    for (int base = 0; base < SLOTS; base += 64) {
        const int32_t v = K[base + lane];
        const bool occ = v != INT_MAX;
        const uint64_t bal = __builtin_amdgcn_ballot_w64(occ);
        if (occ) L[c + __builtin_popcountll(bal & ((uint64_t(1) << lane) - 1))] = v;
        c += __builtin_popcountll(bal);
    }
    if (__builtin_amdgcn_ballot_w64(overflow) != 0 || c > SLOTS / 2) {
        if (lane == 0) atomicOr(flag, 1);
        c = 0;
    }
    if (PASS == 0) {
        if (lane == 0) cnt[row] = c;
        return;
    }
    if (c == 0) return;
    int m = 64;                                  // bitonic sort of the list padded to a power of two
    while (m < c) m <<= 1;
    for (int t = c + lane; t < m; t += 64) L[t] = INT_MAX;
    wave_sync();
    for (int size = 2; size <= m; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = lane; t < m / 2; t += 64) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool up = (lo & size) == 0;
                const int32_t a = L[lo], b = L[hi];
                if ((a > b) == up) { L[lo] = b; L[hi] = a; }
            }
            wave_sync();
        }
    const int64_t off = crp[row];
    const int room = int(crp[row + 1] - crp[row]);   // < c only when the product outgrew its arrays (flagged by k_clamp_rowptr): stay inside them
    if (c > room) c = room;
    if (c <= 0) return;
    // values: the (output column, entry of X) pairs are spread over the lanes, one sorted-row lookup each, a chunk of X entries at
    // a time; the terms go through LDS and every output column adds its own in the order of X's row (not found = + 0.0):
    // the same sums as a serial walk, without its chain of dependent lookups
    double *V = terms[wave];
    constexpr int NACC = SLOTS / 128;            // output columns per lane: c <= SLOTS / 2
    double acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a) acc[a] = 0.0;
    const int pc = c >= 64 ? 1 : 64 / c;         // entries of X per round
    for (int32_t p0 = xs; p0 < xe; p0 += pc) {
        const int np = (xe - p0 < pc) ? int(xe - p0) : pc;
        for (int idx = lane; idx < c * np; idx += 64) {
            const int t = idx / np, pp = idx - t * np;
            const int32_t j = L[t];
            const int32_t p = p0 + pp;
            const int32_t k = xc[p];
            const int32_t ys = yrp[k], ye = yrp[k + 1];
            const int32_t q = lower_bound_i32(yc, ys, ye, j);
            V[idx] = (q < ye && yc[q] == j) ? xv[p] * yv[q] : 0.0;
        }
        wave_sync();
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            const int t = lane + 64 * a;
            if (t < c)
                for (int pp = 0; pp < np; ++pp) acc[a] += V[t * np + pp];
        }
        wave_sync();
    }
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
        const int t = lane + 64 * a;
        if (t < c) { cc[off + t] = L[t]; cv[off + t] = acc[a]; }
    }
}

// row pointers of a product whose arrays were allocated by an upper bound, before the size is known on the host: anything
// beyond the capacity is cut off (rows stay inside the arrays, consumers never index past them) and flagged
__global__ void __launch_bounds__(256) k_clamp_rowptr(int64_t n, int32_t *__restrict__ rowptr, int32_t capacity, int32_t *flag) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i > n) return;
    const int32_t v = rowptr[i];
    if (v > capacity) { rowptr[i] = capacity; if (i == n) atomicOr(flag, 2); }
}

// ---- transpose without a device-wide sort: count, scan, scatter in any order, then every row sorted by a wave in LDS --------
__global__ void __launch_bounds__(256) k_tr_count(int64_t n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, int32_t *__restrict__ cnt) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int32_t p = rowptr[i]; p < rowptr[i + 1]; ++p) atomicAdd(&cnt[col[p]], 1);
}
__global__ void __launch_bounds__(256) k_tr_fill(int64_t n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, const double *__restrict__ val,
                                                 const int32_t *__restrict__ rrp, int32_t *__restrict__ cursor, int32_t *__restrict__ rc, double *__restrict__ rv) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int32_t p = rowptr[i]; p < rowptr[i + 1]; ++p) {
        const int32_t c = col[p];
        const int32_t at = rrp[c] + atomicAdd(&cursor[c], 1);
        rc[at] = int32_t(i);
        rv[at] = val[p];
    }
}
// Rows of R longer than MINLEN and up to MAXLEN entries, WAVES rows per workgroup (a wave sorts its row in LDS).  Two launches cover a
// level: <0, 512, 4> takes nearly every row; <512, 4096, 1> the few long ones (at 184 k vertices of a graded lattice mesh a coarse
// aggregate's smoothed prolongator column holds more than 512 entries: the hierarchy could not be built) - beyond 4096 flag 4.
constexpr int kTrMax = 512, kTrMaxLong = 4096;
template <int MINLEN, int MAXLEN, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k_tr_sort(int64_t nr, const int32_t *__restrict__ rrp, int32_t *__restrict__ rc, double *__restrict__ rv, int32_t *flag) {
    __shared__ int32_t keys[WAVES][MAXLEN];
    __shared__ double vals[WAVES][MAXLEN];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t row = int64_t(blockIdx.x) * WAVES + wave;
    if (row >= nr) return;
    const int32_t rs = rrp[row], len = rrp[row + 1] - rs;
    if (len <= 1 || len <= MINLEN) return;
    if (len > MAXLEN) { if (lane == 0 && MAXLEN == kTrMaxLong) atomicOr(flag, 4); return; }
    int32_t *K = keys[wave];
    double *V = vals[wave];
    int m = 64;
    while (m < len) m <<= 1;
    for (int t = lane; t < m; t += 64) { K[t] = t < len ? rc[rs + t] : INT_MAX; V[t] = t < len ? rv[rs + t] : 0.0; }
    wave_sync();
    for (int size = 2; size <= m; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = lane; t < m / 2; t += 64) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool up = (lo & size) == 0;
                const int32_t a = K[lo], b = K[hi];
                if ((a > b) == up) { K[lo] = b; K[hi] = a; const double x = V[lo]; V[lo] = V[hi]; V[hi] = x; }
            }
            wave_sync();
        }
    for (int t = lane; t < len; t += 64) { rc[rs + t] = K[t]; rv[rs + t] = V[t]; }
}

// ---- dense inverse of the coarsest operator: in-place Gauss-Jordan in LDS, no pivoting (symmetric positive definite) --------
__global__ void __launch_bounds__(256) k_dense_inverse(int n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                       const double *__restrict__ val, double *__restrict__ inv, int32_t *flag) {
    __shared__ double a[kAmgDenseMax * kAmgDenseMax];
    const int tid = threadIdx.x;
    for (int e = tid; e < n * n; e += 256) a[e] = 0.0;
    __syncthreads();
    for (int i = tid; i < n; i += 256)
        for (int32_t p = rowptr[i]; p < rowptr[i + 1]; ++p) a[i * n + col[p]] = val[p];
    __syncthreads();
    for (int p = 0; p < n; ++p) {
        const double piv = a[p * n + p];
        if (!(piv > 0.0)) { if (tid == 0) atomicOr(flag, 2); return; }   // uniform: every thread reads the same entry
        const double ip = 1.0 / piv;
        __syncthreads();
        for (int j = tid; j < n; j += 256)
            if (j != p) a[p * n + j] *= ip;
        __syncthreads();
        for (int e = tid; e < n * n; e += 256) {
            const int i = e / n, j = e - i * n;
            if (i != p && j != p) a[e] -= a[i * n + p] * a[p * n + j];
        }
        __syncthreads();
        for (int i = tid; i < n; i += 256)
            a[i * n + p] = (i == p) ? ip : -a[i * n + p] * ip;
        __syncthreads();
    }
    for (int e = tid; e < n * n; e += 256) inv[e] = a[e];
}

template <class S, class D> __global__ void __launch_bounds__(256) k_convert(int64_t n, const S *__restrict__ src, D *__restrict__ dst) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = D(src[i]);
}

// ---- the cycle -------------------------------------------------------------------------------------------------------------
// Row operations: 8 lanes per row split its entries, every lane carries all K columns, the transposing reduction
// (wave_util.h) leaves column `mycol` of the row sum in lanes with `mine`.  The same device functions serve the one-launch-
// per-operation kernels of the large levels and the single-workgroup kernel that walks all small levels (k_amg_tail).
constexpr int kLpr = 8;

template <class T, int K, bool SCALE, int LPR = kLpr, class TX = T>
__device__ __forceinline__ void row_product(int32_t rs, int32_t re, int sub, const int32_t *__restrict__ col, const T *__restrict__ val,
                                            const T *__restrict__ dinv, const TX *__restrict__ x, T (&acc)[K]) {
#pragma unroll
    for (int c = 0; c < K; ++c) acc[c] = T(0);
    for (int32_t p0 = rs + sub; p0 < re; p0 += 2 * LPR) {   // two passes of loads in flight
        int32_t j[2];
        T a[2], xv[2][K];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int32_t p = p0 + u * LPR;
            j[u] = -1; a[u] = T(0);
            if (p < re) { j[u] = col[p]; a[u] = val[p]; }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int c = 0; c < K; ++c) xv[u][c] = T(0);
            if (j[u] >= 0) {
                if (SCALE) a[u] *= dinv[j[u]];
                const TX *xr = x + int64_t(j[u]) * K;
#pragma unroll
                for (int c = 0; c < K; ++c) xv[u][c] = T(xr[c]);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int c = 0; c < K; ++c) acc[c] += a[u] * xv[u][c];
    }
    TReduce<K, LPR>::run(acc, sub);
}

struct Lane {
    int sub, mycol;
    bool mine;
};
template <int K, int LPR = kLpr> __device__ __forceinline__ Lane lane_of(int tid) {
    int idx[K], own[K];
    Lane l;
    l.sub = tid % LPR;
    towner_init<K, LPR>(idx, own, l.sub);
    l.mycol = idx[0];
    l.mine = own[0] != 0;
    return l;
}

// first half on a level: z = w D^-1 r (from zero), t = r - A z
template <class T, int K, class TR = T> __device__ __forceinline__ void op_pre(const AmgLevelT<T> &a, const TR *__restrict__ r, int64_t row, const Lane &ln) {
    T acc[K];
    row_product<T, K, true, kLpr, TR>(a.rowptr[row], a.rowptr[row + 1], ln.sub, a.col, a.val, a.dinv, r, acc);
    if (ln.mine) {
        const int64_t at = row * K + ln.mycol;
        const T w = T(a.omega), ri = T(r[at]);
        a.z[at] = w * a.dinv[row] * ri;
        a.t[at] = ri - w * acc[0];
    }
}
// r_next = R t
template <class T, int K, int LPR = kLpr> __device__ __forceinline__ void op_restrict(const AmgLevelT<T> &a, T *__restrict__ rn, int64_t row, const Lane &ln) {
    T acc[K];
    row_product<T, K, false, LPR>(a.r_rowptr[row], a.r_rowptr[row + 1], ln.sub, a.r_col, a.r_val, (const T *)nullptr, (const T *)a.t, acc);
    if (ln.mine) rn[row * K + ln.mycol] = acc[0];
}
// z += P z_next
template <class T, int K> __device__ __forceinline__ void op_prolong(const AmgLevelT<T> &a, const T *__restrict__ zn, int64_t row, const Lane &ln) {
    T acc[K];
    row_product<T, K, false>(a.p_rowptr[row], a.p_rowptr[row + 1], ln.sub, a.p_col, a.p_val, (const T *)nullptr, zn, acc);
    if (ln.mine) a.z[row * K + ln.mycol] += acc[0];
}
// second half: zout = z + w D^-1 (r - A z); LAST (finest level): zout / dinv is stored, the return value is r * zout
template <class T, int K, bool LAST, class TR = T>
__device__ __forceinline__ double op_post(const AmgLevelT<T> &a, const TR *__restrict__ r, TR *__restrict__ zout, int64_t row, const Lane &ln) {
    T acc[K];
    row_product<T, K, false>(a.rowptr[row], a.rowptr[row + 1], ln.sub, a.col, a.val, (const T *)nullptr, (const T *)a.z, acc);
    double dot = 0.0;
    if (ln.mine) {
        const int64_t at = row * K + ln.mycol;
        const TR rr = r[at];
        const T ri = T(rr), di = a.dinv[row];
        const T zn = a.z[at] + T(a.omega) * di * (ri - acc[0]);
        zout[at] = TR(LAST ? zn / di : zn);
        dot = double(rr) * double(zn);
    }
    return dot;
}

constexpr int kRpb = 256 / kLpr;   // rows per workgroup of the one-operation kernels

template <class T, int K, class TR>
__global__ void __launch_bounds__(256) k_amg_pre(AmgLevelT<T> a, const TR *__restrict__ r, const double *scal, int step) {
    if (amg_done(scal, step)) return;
    const Lane ln = lane_of<K>(threadIdx.x);
    const int64_t row = int64_t(blockIdx.x) * kRpb + threadIdx.x / kLpr;
    if (row < a.n) op_pre<T, K, TR>(a, r, row, ln);
}
template <class T, int K>
__global__ void __launch_bounds__(256) k_amg_restrict(AmgLevelT<T> a, int64_t n_next, T *__restrict__ rn, const double *scal, int step) {
    if (amg_done(scal, step)) return;
    const Lane ln = lane_of<K>(threadIdx.x);
    const int64_t row = int64_t(blockIdx.x) * kRpb + threadIdx.x / kLpr;
    if (row < n_next) op_restrict<T, K>(a, rn, row, ln);
}
template <class T, int K>
__global__ void __launch_bounds__(256) k_amg_prolong(AmgLevelT<T> a, const T *__restrict__ zn, const double *scal, int step) {
    if (amg_done(scal, step)) return;
    const Lane ln = lane_of<K>(threadIdx.x);
    const int64_t row = int64_t(blockIdx.x) * kRpb + threadIdx.x / kLpr;
    if (row < a.n) op_prolong<T, K>(a, zn, row, ln);
}
// (prolongation folded into this sweep - P z_next formed on the fly for the row and its neighbours - was measured: 24.7 us
// against 5.3 + 12 at 80 k rows, the inner loops over P's rows are chains of dependent loads)
template <class T, int K, bool LAST, int THREADS, class TR>
__global__ void __launch_bounds__(THREADS) k_amg_post(AmgLevelT<T> a, const TR *__restrict__ r, TR *__restrict__ zout,
                                                     double *__restrict__ part, const double *scal, int step) {
    if (amg_done(scal, step)) return;
    const Lane ln = lane_of<K>(threadIdx.x);
    constexpr int RPB = THREADS / kLpr;
    double dot = 0.0;
    for (int64_t row = int64_t(blockIdx.x) * RPB + threadIdx.x / kLpr; row < a.n; row += int64_t(gridDim.x) * RPB)
        dot += op_post<T, K, LAST, TR>(a, r, zout, row, ln);
    if (LAST) {   // <r, z> partials of the workgroup, one per column
        constexpr int NW = THREADS / 64;
        __shared__ double smem[NW * K];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const double x = wave_sum((ln.mine && k == ln.mycol) ? dot : 0.0);
            if (lane == 0) smem[wave * K + k] = x;
        }
        __syncthreads();
        if (threadIdx.x < K) {
            double t = 0.0;
            for (int w = 0; w < NW; ++w) t += smem[w * K + threadIdx.x];
            part[blockIdx.x * K + threadIdx.x] = t;
        }
    }
}

template <class T, int K>
__device__ __forceinline__ void op_dense(int n, const T *__restrict__ inv, const T *__restrict__ r, T *__restrict__ z, int t) {
    const int row = t / K, c = t - row * K;
    T acc = T(0);
    for (int j = 0; j < n; ++j) acc += inv[j * n + row] * r[j * K + c];   // the inverse is symmetric: column = row, coalesced
    z[t] = acc;
}

// all small levels in ONE workgroup: pre / restrict down to the dense coarsest solve, prolong / post back up; the barriers of
// the workgroup stand in for the kernel boundaries (every vector goes through the L2, one CU, one L1)
constexpr int kTailLevels = 3;
constexpr int kTailRows = 128;     // levels of at most this many rows go into the tail kernel (the coarsest one always)
template <class T> struct AmgTailT {
    int levels;
    AmgLevelT<T> up;               // the level above the tail: its R and t feed the first restriction (16 lanes per row: R's rows are long)
    AmgLevelT<T> lev[kTailLevels];
    const T *inv;
};
template <class T, int K>
__global__ void __launch_bounds__(1024) k_amg_tail(AmgTailT<T> A, const double *scal, int step) {
    if (amg_done(scal, step)) return;
    const Lane ln = lane_of<K>(threadIdx.x);
    const int g = threadIdx.x / kLpr, ng = 1024 / kLpr;
    const int L = A.levels;
    {
        const Lane l16 = lane_of<K, 16>(threadIdx.x);
        for (int64_t row = threadIdx.x / 16; row < A.lev[0].n; row += 1024 / 16) op_restrict<T, K, 16>(A.up, A.lev[0].r, row, l16);
        __syncthreads();
    }
    for (int l = 0; l + 1 < L; ++l) {
        const AmgLevelT<T> &a = A.lev[l];
        for (int64_t row = g; row < a.n; row += ng) op_pre<T, K>(a, (const T *)a.r, row, ln);
        __syncthreads();
        const AmgLevelT<T> &b = A.lev[l + 1];
        for (int64_t row = g; row < b.n; row += ng) op_restrict<T, K>(a, b.r, row, ln);
        __syncthreads();
    }
    {
        const AmgLevelT<T> &c = A.lev[L - 1];
        for (int t = threadIdx.x; t < int(c.n) * K; t += 1024) op_dense<T, K>(int(c.n), A.inv, (const T *)c.r, c.z2, t);
        __syncthreads();
    }
    for (int l = L - 2; l >= 0; --l) {
        const AmgLevelT<T> &a = A.lev[l];
        const AmgLevelT<T> &b = A.lev[l + 1];
        for (int64_t row = g; row < a.n; row += ng) op_prolong<T, K>(a, (const T *)b.z2, row, ln);
        __syncthreads();
        for (int64_t row = g; row < a.n; row += ng) (void)op_post<T, K, false>(a, (const T *)a.r, a.z2, row, ln);
        __syncthreads();
    }
}

int grid_rows(int64_t n, int per_block) { return int((n + per_block - 1) / per_block); }

// out[0 .. n] = exclusive prefix sums of cnt[0 .. n) (out[n] = total; cnt[n] must be addressable, its value does not matter);
// the temporary storage comes from the scratch end of the arena and is released by the caller's mark
void scan_counts(Arena &ar, hipStream_t s, int64_t n, const int32_t *cnt, int32_t *out) {
    size_t tb = 0;
    HIP_OK(rocprim::exclusive_scan(nullptr, tb, cnt, out, int32_t(0), size_t(n + 1), rocprim::plus<int32_t>(), s));
    void *tmp = ar.hi<char>(tb + 256);
    HIP_OK(rocprim::exclusive_scan(tmp, tb, cnt, out, int32_t(0), size_t(n + 1), rocprim::plus<int32_t>(), s));
}

// one product C = X Y into arena memory: count, scan, (sync: size), fill.  Returns nnz(C); -1 when a row overflowed.
struct Csr {
    int64_t n = 0, nnz = 0;
    int32_t *rowptr = nullptr, *col = nullptr;
    double *val = nullptr;
};

// Products without a read-back.  The result arrays are allocated by an upper bound (`cap` entries), the row pointers come
// from the count pass and a scan on the device and are clamped to the capacity; overflowing hash tables (bit 1) or arrays
// (bit 2) raise the product's own flag slot, which the host reads at the NEXT point where it has to wait anyway.
// variant 1: a wave per row with 128 hash slots, 2: with 512; remembered per kind of matrix, level and product (process-wide;
// the meshes of a sweep look alike).  (Variant 0 was one LANE per row with a private sorted list of at most 32 columns for the
// short rows of S P0 and A P: 230 us per launch against 70 - private arrays with dynamic indices live in scratch memory; removed.)
template <int PASS>
void launch_product(int variant, hipStream_t s, const Csr &X, const Csr &Y, int32_t *cnt, const int32_t *crp, int32_t *cc, double *cv, int32_t *d_over) {
    if (variant <= 1)
        hipLaunchKernelGGL((k_spgemm<PASS, 128>), dim3(grid_rows(X.n, 4)), dim3(256), 0, s, X.n, X.rowptr, X.col, X.val, Y.rowptr, Y.col, Y.val, cnt, crp, cc, cv, d_over);
    else
        hipLaunchKernelGGL((k_spgemm<PASS, 512>), dim3(grid_rows(X.n, 4)), dim3(256), 0, s, X.n, X.rowptr, X.col, X.val, Y.rowptr, Y.col, Y.val, cnt, crp, cc, cv, d_over);
}

void spgemm_async(Arena &ar, bool permanent, hipStream_t s, const Csr &X, const Csr &Y, int64_t cap, int variant, int32_t *d_over, Csr &C) {
    if (cap > INT_MAX - 8) cap = INT_MAX - 8;
    C.n = X.n;
    C.nnz = cap;   // upper bound; the exact count is rowptr[n] on the device
    C.rowptr = permanent ? ar.lo<int32_t>(size_t(X.n) + 2) : ar.hi<int32_t>(size_t(X.n) + 2);
    int32_t *cnt = ar.hi<int32_t>(size_t(X.n) + 2);   // scratch of the level
    launch_product<0>(variant, s, X, Y, cnt, nullptr, nullptr, nullptr, d_over);
    scan_counts(ar, s, X.n, cnt, C.rowptr);
    hipLaunchKernelGGL(k_clamp_rowptr, dim3(grid_rows(X.n + 1, 256)), dim3(256), 0, s, X.n, C.rowptr, int32_t(cap), d_over);
    C.col = permanent ? ar.lo<int32_t>(size_t(cap) + 2) : ar.hi<int32_t>(size_t(cap) + 2);
    C.val = permanent ? ar.lo<double>(size_t(cap) + 2) : ar.hi<double>(size_t(cap) + 2);
    launch_product<1>(variant, s, X, Y, nullptr, C.rowptr, C.col, C.val, d_over);
}

std::atomic<int> g_product_hint[2][kAmgMaxLevels][3];   // [2D | 3D matrices][level][S P0, A P, R (A P)]; 0 / 1: 128 slots, 2: 512
std::atomic<int> g_ap_room[2];                          // log2 of the extra room of A P beyond 4 nnz(A) (0 .. 3)
constexpr int kFlagSlots = 64;                          // [0] setup flags, [8 + 3 level + product] product flags

// one attempt: 0 = built, 1 = a product outgrew its tables or arrays (the hints have been raised: try again), -1 = failed
int amg_build(Arena &ar, hipStream_t s, int hs, int64_t nv, const int32_t *rowptr, const int32_t *col, const double *val, int kmax, AmgT<double> &H,
              std::string &why) {
    H = AmgT<double>{};
    int32_t *d_flag = ar.lo<int32_t>(kFlagSlots);
    unsigned long long *d_bound = ar.lo<unsigned long long>(kAmgMaxLevels + 1);
    HIP_OK(hipMemsetAsync(d_flag, 0, sizeof(int32_t) * kFlagSlots, s));
    HIP_OK(hipMemsetAsync(d_bound, 0, sizeof(unsigned long long) * (kAmgMaxLevels + 1), s));
    int32_t h_flags[kFlagSlots] = {};
    int32_t h_nnz_a[kAmgMaxLevels] = {}, h_nnz_p[kAmgMaxLevels] = {};
    const int32_t *d_nnz_p[kAmgMaxLevels] = {};   // where the exact entry counts of the prolongators live on the device
    // looks at the flags read back so far; products of the levels below `upto`
    auto check = [&](int upto) -> int {
        int again = 0;   // the products first: a cut-off product leaves rows without a diagonal behind, which is then flagged as well
        for (int l = 0; l < upto; ++l)
            for (int p = 0; p < 3; ++p) {
                const int f = h_flags[8 + 3 * l + p];
                if (f == 0) continue;
                if (f & 1) {   // hash tables
                    if (g_product_hint[hs][l][p].load() >= 2) { why = "a product row with more than 256 distinct columns"; return -1; }
                    g_product_hint[hs][l][p].store(2);
                    again = 1;
                }
                if (f & 2) {   // arrays
                    if (p != 1 || g_ap_room[hs].load() >= 3) { why = "a product outgrew its arrays"; return -1; }
                    g_ap_room[hs].fetch_add(1);
                    again = 1;
                }
            }
        if (again) return 1;
        if (h_flags[0] != 0) {
            why = "setup raised flag " + std::to_string(h_flags[0]) + " (1 diagonal, 2 dense pivot, 4 a row of R beyond the LDS sort)";
            return -1;
        }
        return 0;
    };
    // level 0: compact copy of the leading block
    Csr A;
    A.n = nv;
    A.rowptr = ar.lo<int32_t>(size_t(nv) + 2);
    {
        const size_t mark = ar.hi_mark();
        int32_t *cnt = ar.hi<int32_t>(size_t(nv) + 2);
        hipLaunchKernelGGL((k_block_copy<0>), dim3(grid_rows(nv, 256)), dim3(256), 0, s, nv, rowptr, col, val, cnt, (const int32_t *)nullptr,
                           (int32_t *)nullptr, (double *)nullptr);
        scan_counts(ar, s, nv, cnt, A.rowptr);
        int32_t h = 0;
        HIP_OK(hipMemcpyAsync(&h, A.rowptr + nv, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        HIP_OK(hipStreamSynchronize(s));
        ar.hi_release(mark);
        if (h <= 0 || h == INT_MAX) { why = "empty vertex block"; return -1; }
        A.nnz = h;
        h_nnz_a[0] = h;
        A.col = ar.lo<int32_t>(size_t(h) + 2);
        A.val = ar.lo<double>(size_t(h) + 2);
        hipLaunchKernelGGL((k_block_copy<1>), dim3(grid_rows(nv, 256)), dim3(256), 0, s, nv, rowptr, col, val, (int32_t *)nullptr,
                           (const int32_t *)A.rowptr, A.col, A.val);
    }
    int L = 0;
    for (;;) {
        AmgLevelT<double> &lv = H.lev[L];
        const int64_t n = A.n;
        lv.n = n; lv.rowptr = A.rowptr; lv.col = A.col; lv.val = A.val;
        double *dinv = ar.lo<double>(size_t(n) + 2);
        lv.dinv = dinv;
        lv.z = ar.lo<double>(size_t(n) * kmax + 2);
        lv.z2 = ar.lo<double>(size_t(n) * kmax + 2);
        lv.t = ar.lo<double>(size_t(n) * kmax + 2);
        if (L > 0) lv.r = ar.lo<double>(size_t(n) * kmax + 2);
        const int g = grid_rows(n, 256);
        hipLaunchKernelGGL(k_level_diag, dim3(g), dim3(256), 0, s, n, A.rowptr, A.col, A.val, dinv, d_bound + L, d_flag);
        if (n <= kAmgDenseMax) {   // coarsest: dense inverse
            double *inv = ar.lo<double>(size_t(n) * n + 2);
            hipLaunchKernelGGL(k_dense_inverse, dim3(1), dim3(256), 0, s, int(n), A.rowptr, A.col, A.val, inv, d_flag);
            H.inv = inv;
            H.levels = L + 1;
            break;
        }
        if (L + 1 >= kAmgMaxLevels) { why = "too many levels"; return -1; }
        // ---- aggregation ----
        const size_t mark = ar.hi_mark();   // the scratch of a level is reused by the next one: the stream orders the kernels
        uint64_t *tup = ar.hi<uint64_t>(size_t(n) + 2), *m1 = ar.hi<uint64_t>(size_t(n) + 2);
        int32_t *isroot = ar.hi<int32_t>(size_t(n) + 2), *id = ar.hi<int32_t>(size_t(n) + 2);
        int32_t *agg1 = ar.hi<int32_t>(size_t(n) + 2), *agg = ar.hi<int32_t>(size_t(n) + 2);
        hipLaunchKernelGGL(k_mis_init, dim3(g), dim3(256), 0, s, n, tup);
        for (int it = 0; it < 10; ++it) {   // a fixed number of iterations, no read-back (460 of 80 k nodes are open after six, none after twelve); k_mis_finish settles stragglers
            hipLaunchKernelGGL(k_mis_max, dim3(g), dim3(256), 0, s, n, A.rowptr, A.col, tup, m1);
            hipLaunchKernelGGL(k_mis_update, dim3(g), dim3(256), 0, s, n, A.rowptr, A.col, tup, m1);
        }
        hipLaunchKernelGGL(k_mis_finish, dim3(g), dim3(256), 0, s, n, A.rowptr, A.col, tup, m1);
        hipLaunchKernelGGL(k_mis_flags, dim3(g), dim3(256), 0, s, n, m1, isroot);
        scan_counts(ar, s, n, isroot, id);
        hipLaunchKernelGGL(k_agg_near, dim3(g), dim3(256), 0, s, n, A.rowptr, A.col, m1, id, agg1);
        hipLaunchKernelGGL(k_agg_far, dim3(g), dim3(256), 0, s, n, A.rowptr, A.col, A.val, agg1, agg);
        // ---- the level's ONE wait: number of aggregates, spectrum bound, everything flagged so far, exact size of this level's matrix
        int32_t h_nc = 0;
        unsigned long long h_bound = 0;
        HIP_OK(hipMemcpyAsync(&h_nc, id + n, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        HIP_OK(hipMemcpyAsync(&h_bound, d_bound + L, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        HIP_OK(hipMemcpyAsync(h_flags, d_flag, sizeof(int32_t) * kFlagSlots, hipMemcpyDeviceToHost, s));
        if (L > 0) HIP_OK(hipMemcpyAsync(&h_nnz_a[L], A.rowptr + n, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        HIP_OK(hipStreamSynchronize(s));
        const int verdict = check(L);
        if (verdict != 0) return verdict;
        double lmax;
        std::memcpy(&lmax, &h_bound, sizeof lmax);
        const int64_t nnz_a = h_nnz_a[L];
        if (getenv("REMO_AMG_DEBUG")) fprintf(stderr, "amg level %d: n %lld nnz %lld -> %d aggregates, lmax %.3f\n", L, (long long)n, (long long)nnz_a, h_nc, lmax);
        if (!(lmax > 0.0) || !std::isfinite(lmax) || nnz_a <= 0) { why = "level operator without a spectrum bound"; return -1; }
        lv.omega = 1.6 / lmax;
        lv.nnz = nnz_a;
        const int64_t nc = h_nc;
        if (nc <= 0 || nc * 10 > n * 8) { why = "no coarsening"; return -1; }
        // ---- P = (I - 4/(3 lmax) D^-1 A) P0: at most as many entries as A ----
        Csr S = A, P0;
        S.val = ar.hi<double>(size_t(nnz_a) + 2);
        P0.n = n; P0.nnz = n;
        P0.rowptr = ar.hi<int32_t>(size_t(n) + 2); P0.col = ar.hi<int32_t>(size_t(n) + 2); P0.val = ar.hi<double>(size_t(n) + 2);
        hipLaunchKernelGGL(k_smoothing_factor, dim3(grid_rows(n + 1, 256)), dim3(256), 0, s, n, A.rowptr, A.col, A.val, dinv, d_bound + L, S.val,
                           agg, P0.rowptr, P0.col, P0.val);
        int32_t *pf = d_flag + 8 + 3 * L;
        Csr P;
        spgemm_async(ar, true, s, S, P0, nnz_a, g_product_hint[hs][L][0].load(), pf + 0, P);
        lv.p_rowptr = P.rowptr; lv.p_col = P.col; lv.p_val = P.val;
        d_nnz_p[L] = P.rowptr + n;
        // ---- R = P^T: count the entries of every column, scan, scatter (any order), sort every row of R by its column = row of P ----
        Csr R;
        R.n = nc; R.nnz = nnz_a;
        R.rowptr = ar.lo<int32_t>(size_t(nc) + 2); R.col = ar.lo<int32_t>(size_t(nnz_a) + 2); R.val = ar.lo<double>(size_t(nnz_a) + 2);
        {
            int32_t *tcnt = ar.hi<int32_t>(size_t(nc) + 2), *cursor = ar.hi<int32_t>(size_t(nc) + 2);
            HIP_OK(hipMemsetAsync(tcnt, 0, sizeof(int32_t) * (size_t(nc) + 2), s));
            HIP_OK(hipMemsetAsync(cursor, 0, sizeof(int32_t) * (size_t(nc) + 2), s));
            hipLaunchKernelGGL(k_tr_count, dim3(g), dim3(256), 0, s, n, P.rowptr, P.col, tcnt);
            scan_counts(ar, s, nc, tcnt, R.rowptr);
            hipLaunchKernelGGL(k_tr_fill, dim3(g), dim3(256), 0, s, n, P.rowptr, P.col, P.val, R.rowptr, cursor, R.col, R.val);
            hipLaunchKernelGGL((k_tr_sort<0, kTrMax, 4>), dim3(grid_rows(nc, 4)), dim3(256), 0, s, nc, R.rowptr, R.col, R.val, d_flag);
            hipLaunchKernelGGL((k_tr_sort<kTrMax, kTrMaxLong, 1>), dim3(grid_rows(nc, 1)), dim3(64), 0, s, nc, R.rowptr, R.col, R.val, d_flag);
        }
        lv.r_rowptr = R.rowptr; lv.r_col = R.col; lv.r_val = R.val;
        // ---- A' = R (A P): A P in scratch with room for (4 << room) nnz(A) entries, A' at most as many entries as A ----
        Csr AP, Ac;
        spgemm_async(ar, false, s, A, P, (nnz_a * 4) << g_ap_room[hs].load(), g_product_hint[hs][L][1].load(), pf + 1, AP);
        spgemm_async(ar, true, s, R, AP, nnz_a, g_product_hint[hs][L][2].load(), pf + 2, Ac);
        ar.hi_release(mark);
        A = Ac;
        A.n = nc;
        ++L;
    }
    // ---- the last wait: flags of the last level's products and of the dense inverse, exact sizes for the fp32 image ----
    HIP_OK(hipMemcpyAsync(h_flags, d_flag, sizeof(int32_t) * kFlagSlots, hipMemcpyDeviceToHost, s));
    HIP_OK(hipMemcpyAsync(&h_nnz_a[L], A.rowptr + A.n, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    for (int l = 0; l < L; ++l) HIP_OK(hipMemcpyAsync(&h_nnz_p[l], d_nnz_p[l], sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    const int verdict = check(L);
    if (verdict != 0) return verdict;
    H.lev[L].nnz = h_nnz_a[L];
    for (int l = 0; l < L; ++l) H.lev[l].nnz_p = h_nnz_p[l];
    H.launches = 4 * (H.levels - 1);   // pre, restrict, prolong, post per level; the tail kernel stands for the last restriction + dense solve
    H.kmax = kmax;
    return 0;
}

}  // namespace

bool amg_setup(Arena &ar, hipStream_t s, int dim, int64_t nv, const int32_t *rowptr, const int32_t *col, const double *val, int kmax, AmgT<double> &H,
               std::string &why) {
    H = AmgT<double>{};
    if (nv <= kAmgDenseMax) { why = "vertex block too small"; return false; }
    const int hs = dim == 3 ? 1 : 0;   // what earlier batches taught about the products: per kind of matrix
    const size_t hi0 = ar.hi_mark(), lo0 = ar.lo_off;
    try {
        for (int attempt = 0; attempt < 6; ++attempt) {
            ar.hi_release(hi0);
            ar.lo_off = lo0;               // a repeated attempt reuses the memory of the failed one (the stream orders the kernels)
            const int rc = amg_build(ar, s, hs, nv, rowptr, col, val, kmax, H, why);
            ar.hi_release(hi0);
            if (rc == 0) return true;
            if (rc < 0) break;
        }
        if (why.empty()) why = "the products did not fit after several attempts";
        H = AmgT<double>{};
        ar.lo_off = lo0;                   // the fallback (polynomial) sees the arena as it was before the hierarchy was attempted
        return false;
    } catch (const std::exception &ex) {
        ar.hi_release(hi0);
        ar.lo_off = lo0;
        H = AmgT<double>{};
        why = ex.what();
        return false;
    }
}

void amg_to_float(Arena &ar, hipStream_t s, const AmgT<double> &in, int kmax, AmgT<float> &out) {
    auto conv = [&](const double *src, size_t count) {
        float *dst = ar.lo<float>(count + 4);
        hipLaunchKernelGGL((k_convert<double, float>), dim3(grid_rows(int64_t(count), 256)), dim3(256), 0, s, int64_t(count), src, dst);
        return (const float *)dst;
    };
    out = AmgT<float>{};
    out.levels = in.levels;
    out.launches = in.launches;
    out.kmax = kmax;
    for (int l = 0; l < in.levels; ++l) {
        const AmgLevelT<double> &a = in.lev[l];
        AmgLevelT<float> &b = out.lev[l];
        b.n = a.n; b.nnz = a.nnz; b.rowptr = a.rowptr; b.col = a.col; b.omega = a.omega;
        b.val = conv(a.val, size_t(a.nnz));
        b.dinv = conv(a.dinv, size_t(a.n));
        b.z = ar.lo<float>(size_t(a.n) * kmax + 4); b.z2 = ar.lo<float>(size_t(a.n) * kmax + 4); b.t = ar.lo<float>(size_t(a.n) * kmax + 4);
        if (l > 0) b.r = ar.lo<float>(size_t(a.n) * kmax + 4);
        if (l + 1 < in.levels) {
            b.p_rowptr = a.p_rowptr; b.p_col = a.p_col; b.r_rowptr = a.r_rowptr; b.r_col = a.r_col;
            b.nnz_p = a.nnz_p;
            b.p_val = conv(a.p_val, size_t(a.nnz_p));
            b.r_val = conv(a.r_val, size_t(a.nnz_p));
        }
    }
    const int64_t nc = in.lev[in.levels - 1].n;
    out.inv = conv(in.inv, size_t(nc * nc));
}

template <class T, class TR> void launch_amg_cycle(const AmgT<T> &H, int k, int step, const TR *r, TR *cz, double *part, int nblocks, const double *scal, hipStream_t s) {
    const int L = H.levels;
#define AMG_K_SWITCH(CALL)                                  \
    switch (k) {                                            \
        case 1: { constexpr int KK = 1; CALL; } break;      \
        case 2: { constexpr int KK = 2; CALL; } break;      \
        case 3: { constexpr int KK = 3; CALL; } break;      \
        case 4: { constexpr int KK = 4; CALL; } break;      \
        case 5: { constexpr int KK = 5; CALL; } break;      \
        case 6: { constexpr int KK = 6; CALL; } break;      \
        case 7: { constexpr int KK = 7; CALL; } break;      \
        default: { constexpr int KK = 8; CALL; } break;     \
    }
    auto rows_grid = [&](int64_t n) { return int((n + kRpb - 1) / kRpb); };
    // the levels from `lt` on go into the single-workgroup kernel (always at least the coarsest one)
    int lt = L - 1;
    while (lt > 1 && H.lev[lt - 1].n <= kTailRows && L - (lt - 1) <= kTailLevels) --lt;
    for (int l = 0; l < lt; ++l) {   // down
        const AmgLevelT<T> &a = H.lev[l], &b = H.lev[l + 1];
        if (l == 0) {
            AMG_K_SWITCH(hipLaunchKernelGGL((k_amg_pre<T, KK, TR>), dim3(rows_grid(a.n)), dim3(256), 0, s, a, r, scal, step));
        } else {
            AMG_K_SWITCH(hipLaunchKernelGGL((k_amg_pre<T, KK, T>), dim3(rows_grid(a.n)), dim3(256), 0, s, a, (const T *)a.r, scal, step));
        }
        if (l + 1 < lt) AMG_K_SWITCH(hipLaunchKernelGGL((k_amg_restrict<T, KK>), dim3(rows_grid(b.n)), dim3(256), 0, s, a, b.n, b.r, scal, step));
    }
    {
        AmgTailT<T> tail;
        tail.levels = L - lt;
        tail.up = H.lev[lt - 1];
        for (int l = lt; l < L; ++l) tail.lev[l - lt] = H.lev[l];
        tail.inv = H.inv;
        AMG_K_SWITCH(hipLaunchKernelGGL((k_amg_tail<T, KK>), dim3(1), dim3(1024), 0, s, tail, scal, step));
    }
    for (int l = lt - 1; l >= 0; --l) {   // up
        const AmgLevelT<T> &a = H.lev[l], &b = H.lev[l + 1];
        AMG_K_SWITCH(hipLaunchKernelGGL((k_amg_prolong<T, KK>), dim3(rows_grid(a.n)), dim3(256), 0, s, a, (const T *)b.z2, scal, step));
        if (l == 0) {   // the launch that leaves the partial sums is held to `nblocks` workgroups: 128 rows each
            AMG_K_SWITCH(hipLaunchKernelGGL((k_amg_post<T, KK, true, 1024, TR>), dim3(nblocks), dim3(1024), 0, s, a, r, cz, part, scal, step));
        } else {
            AMG_K_SWITCH(hipLaunchKernelGGL((k_amg_post<T, KK, false, 256, T>), dim3(rows_grid(a.n)), dim3(256), 0, s, a, (const T *)a.r, a.z2, (double *)nullptr, scal, step));
        }
    }
#undef AMG_K_SWITCH
}

template void launch_amg_cycle<double, double>(const AmgT<double> &, int, int, const double *, double *, double *, int, const double *, hipStream_t);
template void launch_amg_cycle<float, float>(const AmgT<float> &, int, int, const float *, float *, double *, int, const double *, hipStream_t);
template void launch_amg_cycle<float, double>(const AmgT<float> &, int, int, const double *, double *, double *, int, const double *, hipStream_t);

}  // namespace remo
