// kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the ReMo3D hot path:
//   metric terms -> CSR value gather-assembly -> multi-RHS PCG (edge-pair SpMM, fused vector kernels,
//   two-level preconditioner: Chebyshev on the P1 vertex block + Jacobi; fp64 or fp32 storage with fp64
//   residual replacement) -> axis point location / RHS build / evaluation.
// All of it is HBM/L2-bound sparse work: no MFMA (a sparse row is not a dense contraction); the levers are
// coalesced CSR streams, DPP (not LDS) cross-lane reductions, XCD-aware row placement, LDS-staged reference
// tensors and few launches per PCG step.  Reference lines each kernel replaces are cited at the kernel.
#include "kernels.h"
#include "amg.h"

#include <limits.h>

#include "fem_p3.h"
#include "wave_util.h"
#include "kutil.h"

namespace remo {

// Progress records live in mapped, coherent host memory: the stores bypass the caches (sc0 sc1).  A system-scope RELEASE
// store would also write back the whole L2 of the XCD (buffer_wbl2) on every PCG step; the record only needs its data to
// land before its step number, so the data stores are relaxed, the wave waits for their acknowledgement, then stores the
// step number.
__device__ __forceinline__ void publish_progress(PcgProgress *pr, const double *rz, int k, int step) {
    for (int c = 0; c < k; ++c) __hip_atomic_store(&pr->rz[c], rz[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(&pr->step, step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Storage position of stored entry e of `row` (rows of an edge pair keep their values INTERLEAVED: the two rows have the
// same column pattern and the SpMM reads both values of an entry with one 16-byte load; every other row is plain CSR).
// rs = rowptr[row], len = row length.
__device__ __forceinline__ int64_t value_pos(int64_t row, int32_t rs, int32_t len, int32_t e, int64_t pair_begin, int64_t pair_end) {
    if (row < pair_begin || row >= pair_end) return int64_t(rs) + e;
    const bool second = ((row - pair_begin) & 1) != 0;
    return int64_t(second ? rs - len : rs) + 2 * int64_t(e) + (second ? 1 : 0);
}

// ------------------------------------------------------------------------------------------
// metric terms: one thread per element (ngsolve_functions.py:33-36: the coefficient part of the
// integrand; sigma per material as worker.py:101)

template <int DIM>
__global__ void __launch_bounds__(256) k_metric_terms(int64_t nt, const double *__restrict__ coords,
                                                      const int32_t *__restrict__ conn, const int32_t *__restrict__ mat,
                                                      const int32_t *__restrict__ eperm, const double *__restrict__ sigma, int nmat,
                                                      double *__restrict__ C, int32_t *errflag) {
    constexpr int NB = DIM + 1, NT = P3<DIM>::NTERM;
    const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    double X[NB * DIM];
#pragma unroll
    for (int a = 0; a < NB; ++a) {
        const int64_t v = conn[t * NB + a];
#pragma unroll
        for (int k = 0; k < DIM; ++k) X[a * DIM + k] = coords[v * DIM + k];
    }
    const int m = mat[eperm ? int64_t(eperm[t]) : t];   // materials stay in the caller's element order (symbolic_gpu.hip)
    double c[NT];
    bool ok = (m >= 0 && m < nmat);
    if (ok) ok = metric_terms<DIM>(X, sigma[m], c);
    if (!ok) {
        atomicOr(errflag, 1);
#pragma unroll
        for (int i = 0; i < NT; ++i) c[i] = 0.0;
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) C[t * NT + i] = c[i];
}

void launch_metric_terms(int dim, int64_t nt, const double *coords, const int32_t *conn, const int32_t *mat, const int32_t *eperm,
                         const double *sigma, int nmat, double *C, int32_t *errflag, hipStream_t s) {
    const int grid = int((nt + 255) / 256);
    if (dim == 2)
        hipLaunchKernelGGL(k_metric_terms<2>, dim3(grid), dim3(256), 0, s, nt, coords, conn, mat, eperm, sigma, nmat, C, errflag);
    else
        hipLaunchKernelGGL(k_metric_terms<3>, dim3(grid), dim3(256), 0, s, nt, coords, conn, mat, eperm, sigma, nmat, C, errflag);
}

// ------------------------------------------------------------------------------------------
// CSR value assembly, gather formulation (a.Assemble(), ngsolve_functions.py:47).
// One wave owns one row; lane p owns stored entry p of the row and walks the row's incident
// elements in ascending order, adding K_e[li][lj] where the element's local dof lj is its
// column.  No atomics, every value written exactly once (coalesced), bit-reproducible.
// Reference tensors are staged in LDS (19.2 KB in 3D, 7.2 KB in 2D).

constexpr int kAsmRow = 448;   // stored entries of a row handled through LDS in k_assemble (longer rows: lane-per-entry walk)
template <int DIM, bool CONDENSE>
__global__ void __launch_bounds__(256) k_assemble(int64_t nfree, int64_t pair_begin, int64_t pair_end, const int32_t *__restrict__ rowptr,
                                                  const int32_t *__restrict__ col, const int32_t *__restrict__ adjptr,
                                                  const uint32_t *__restrict__ adj, const int32_t *__restrict__ eldof,
                                                  const double *__restrict__ C, const double *__restrict__ Mg,
                                                  double *__restrict__ val, double *__restrict__ dinv) {
    constexpr int N = P3<DIM>::NLD, NT = P3<DIM>::NTERM;
    constexpr int NK = CONDENSE ? 9 : N;  // local dofs that are unknowns
    __shared__ double M[NT * N * N];
    for (int i = threadIdx.x; i < NT * N * N; i += blockDim.x) M[i] = Mg[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Rows of up to kAsmRow stored entries (all but pathological vertex rows): the row's columns and two accumulators
    // per stored entry live in LDS.  For every incident element, lane q < NK takes the element's local dof q: it finds
    // the position of that column in the row by binary search in LDS (the 20-compare search per stored entry AND element
    // of the first version was the cost of the kernel: 508 us per batch at 63 k tetrahedra, 5 % of the HBM roofline),
    // forms K_e[li][q] and adds it at that position.  The positions of one element are distinct and the elements are
    // walked in ascending order by the whole wave, so every stored entry still sums its contributions in the same
    // fixed order: bit-identical to the lane-per-entry walk, which remains for longer rows.
    __shared__ int32_t colL[4][kAsmRow];
    __shared__ double accL[4][kAsmRow];      // single rows: entry p at [p]; edge-row pairs: entry p of the two rows at [2p], [2p + 1]
    // workgroups are persistent over rows: the reference tensors are staged in LDS once per workgroup, not once per 4 rows
    for (int64_t row = int64_t(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6); row < nfree; row += int64_t(gridDim.x) * (blockDim.x >> 6)) {
    // the two dofs of an edge (rows r, r + 1 of the pair range) meet the same elements with local numbers li, li + 1 and
    // have the same columns: the wave of the first row computes both (one column search), the wave of the second rests
    const bool in_pairs = row >= pair_begin && row < pair_end;
    if (in_pairs && ((row - pair_begin) & 1)) continue;
    const int32_t rs = rowptr[row], re = rowptr[row + 1];
    const int32_t as = adjptr[row], ae = adjptr[row + 1];
    if (re - rs <= (in_pairs ? kAsmRow / 2 : kAsmRow)) {
        const int32_t len = re - rs;
        int32_t *cl = colL[wave];
        double *aa = accL[wave];
        const int sh = in_pairs ? 1 : 0;
        for (int32_t p = lane; p < len; p += 64) cl[p] = col[rs + p];
        for (int32_t p = lane; p < (len << sh); p += 64) aa[p] = 0.0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // NG incident elements per trip, one group of NK lanes each: the dependent chain of a trip (element's dofs -> position
        // in the row by binary search -> element matrix entries) is latency, so the groups run it side by side; only the adds
        // into the accumulators are taken group by group, in element order (two elements of a trip can hit the same entry).
        constexpr int NG = 64 / NK;
        const int grp = lane / NK, q = lane - grp * NK;
        for (int32_t a = as; a < ae; a += NG) {
            const bool have = grp < NG && a + grp < ae;
            int32_t pos = -1;
            double k = 0.0, k2 = 0.0;
            if (have) {
                const uint32_t code = adj[a + grp];
                const int64_t t = code >> 5;
                const int li = int(code & 31u);
                const int32_t j = eldof[t * N + q];
                if (j >= 0) {
                    int32_t lo = 0, hi = len;
                    while (lo < hi) {                  // columns ascend; j is one of them
                        const int32_t mid = (lo + hi) >> 1;
                        if (cl[mid] < j) lo = mid + 1; else hi = mid;
                    }
                    // (vertex-block-only assembly: the row holds vertex columns only, every other local dof of the element ends
                    // behind the row's last column or between two of them - a hit counts only if the column is really there;
                    // a row of exactly kAsmRow entries would otherwise add into the next wave's accumulators)
                    pos = (lo < len && cl[lo] == j) ? lo : -1;
                    const double *c = C + t * NT;
                    k = kentry<DIM>(c, M, li, q);
                    if (in_pairs) k2 = kentry<DIM>(c, M, li + 1, q);
                    if (CONDENSE) {  // Schur complement of the cell bubble (condense=True, ngsolve_functions.py:31)
                        const double kbj = kentry<DIM>(c, M, 9, q), kbb = kentry<DIM>(c, M, 9, 9);
                        k -= kentry<DIM>(c, M, li, 9) * kbj / kbb;
                        if (in_pairs) k2 -= kentry<DIM>(c, M, li + 1, 9) * kbj / kbb;
                    }
                }
            }
#pragma unroll
            for (int gg = 0; gg < NG; ++gg) {
                if (grp == gg && pos >= 0) {
                    aa[pos << sh] += k;
                    if (in_pairs) aa[2 * pos + 1] += k2;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        for (int32_t p = lane; p < len; p += 64) {
            const int32_t j = cl[p];
            const double acc = aa[p << sh], acc2 = in_pairs ? aa[2 * p + 1] : 0.0;
            if (in_pairs) {   // interleaved values of the pair (value_pos)
                val[int64_t(rs) + 2 * p] = acc;
                val[int64_t(rs) + 2 * p + 1] = acc2;
                if (j == row) dinv[row] = 1.0 / acc;
                if (j == row + 1) dinv[row + 1] = 1.0 / acc2;
            } else {
                val[rs + p] = acc;
                if (j == row) dinv[row] = 1.0 / acc;  // Jacobi = Preconditioner(a, "local"), ngsolve_functions.py:46
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        continue;
    }
    for (int32_t base = rs; base < re; base += 64) {
        const int32_t p = base + lane;
        const int32_t j = (p < re) ? col[p] : -2;
        double acc = 0.0, acc2 = 0.0;
        for (int32_t a = as; a < ae; ++a) {
            const uint32_t code = adj[a];  // wave-uniform
            const int64_t t = code >> 5;
            const int li = int(code & 31u);
            const int32_t *ed = eldof + t * N;
            const double *c = C + t * NT;
            int lj = -1;
#pragma unroll
            for (int q = 0; q < NK; ++q)
                if (ed[q] == j) lj = q;
            if (lj >= 0) {
                double k = kentry<DIM>(c, M, li, lj);
                double k2 = in_pairs ? kentry<DIM>(c, M, li + 1, lj) : 0.0;
                if (CONDENSE) {  // Schur complement of the cell bubble (condense=True, ngsolve_functions.py:31)
                    const double kbj = kentry<DIM>(c, M, 9, lj), kbb = kentry<DIM>(c, M, 9, 9);
                    k -= kentry<DIM>(c, M, li, 9) * kbj / kbb;
                    if (in_pairs) k2 -= kentry<DIM>(c, M, li + 1, 9) * kbj / kbb;
                }
                acc += k;
                acc2 += k2;
            }
        }
        if (p < re) {
            if (in_pairs) {   // interleaved values of the pair (value_pos): entry e of rows r, r + 1 at rs + 2e, rs + 2e + 1
                val[int64_t(rs) + 2 * (p - rs)] = acc;
                val[int64_t(rs) + 2 * (p - rs) + 1] = acc2;
                if (j == row) dinv[row] = 1.0 / acc;
                if (j == row + 1) dinv[row + 1] = 1.0 / acc2;
            } else {
                val[p] = acc;
                if (j == row) dinv[row] = 1.0 / acc;  // Jacobi = Preconditioner(a, "local"), ngsolve_functions.py:46
            }
        }
    }
    }
}

// Jacobi factors of rows [row0, nfree) WITHOUT the assembled matrix (the patch operator's batches assemble only the P1 block):
// diagonal entry = sum over the row's incident elements of K_e[li][li].  One thread per row, same element order as the assembly.
template <int DIM>
__global__ void __launch_bounds__(256) k_diag_rows(int64_t row0, int64_t nfree, const int32_t *__restrict__ adjptr, const uint32_t *__restrict__ adj,
                                                   const double *__restrict__ C, const double *__restrict__ Mg, double *__restrict__ dinv) {
    constexpr int N = P3<DIM>::NLD, NT = P3<DIM>::NTERM;
    __shared__ double Md[NT * N];
    for (int i = threadIdx.x; i < NT * N; i += blockDim.x) Md[i] = Mg[((i / N) * N + (i % N)) * N + (i % N)];
    __syncthreads();
    const int64_t row = row0 + int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (row >= nfree) return;
    double s = 0.0;
    for (int32_t a = adjptr[row]; a < adjptr[row + 1]; ++a) {
        const uint32_t code = adj[a];
        const double *c = C + int64_t(code >> 5) * NT;
        const int li = int(code & 31u);
        double e = 0.0;
#pragma unroll
        for (int t = 0; t < NT; ++t) e += c[t] * Md[t * N + li];
        s += e;
    }
    dinv[row] = 1.0 / s;
}
void launch_diag_rows(int dim, int64_t row0, int64_t nfree, const int32_t *adjptr, const uint32_t *adj, const double *C, const double *M, double *dinv, hipStream_t s) {
    if (nfree <= row0) return;
    const int grid = int((nfree - row0 + 255) / 256);
    if (dim == 3) hipLaunchKernelGGL(k_diag_rows<3>, dim3(grid), dim3(256), 0, s, row0, nfree, adjptr, adj, C, M, dinv);
    else hipLaunchKernelGGL(k_diag_rows<2>, dim3(grid), dim3(256), 0, s, row0, nfree, adjptr, adj, C, M, dinv);
}

void launch_assemble(int dim, bool condense, int64_t nfree, int64_t pair_begin, int64_t pair_end, const int32_t *rowptr, const int32_t *col,
                     const int32_t *adjptr, const uint32_t *adj, const int32_t *eldof, const double *C,
                     const double *M, double *val, double *dinv, hipStream_t s) {
    int64_t g64 = (nfree + 3) / 4;
    if (g64 > 4096) g64 = 4096;
    const int grid = int(g64 < 1 ? 1 : g64);
    if (dim == 3)
        hipLaunchKernelGGL((k_assemble<3, false>), dim3(grid), dim3(256), 0, s, nfree, pair_begin, pair_end, rowptr, col, adjptr, adj, eldof, C, M, val, dinv);
    else if (condense)
        hipLaunchKernelGGL((k_assemble<2, true>), dim3(grid), dim3(256), 0, s, nfree, pair_begin, pair_end, rowptr, col, adjptr, adj, eldof, C, M, val, dinv);
    else
        hipLaunchKernelGGL((k_assemble<2, false>), dim3(grid), dim3(256), 0, s, nfree, pair_begin, pair_end, rowptr, col, adjptr, adj, eldof, C, M, val, dinv);
}

// ------------------------------------------------------------------------------------------
// CSR SpMM  y = A x  for K interleaved right-hand sides (x[n][K] row-major), the kernel the
// CG hot loop spends its time in (CGSolver, ngsolve_functions.py:50-51; cusparseSpMV in the
// reference's CUDA attempt, ngsolve_functions_gpu.py:41-47).
// LPR lanes cooperate on a row: values/columns are read as contiguous runs (rows are contiguous
// in CSR, so a wave streams one contiguous span), x rows are gathered K doubles at a time, the
// LPR partial sums are combined with wave shuffles.  Optionally leaves per-block partial sums of
// <x, y> (the CG's <p, Ap>) so the dot product costs no extra pass.

// Variant A ("lane per stored entry"): the fallback for matrices without edge-row pairs and the
// baseline of tools/probe_spmm.py.  T = double (the product path) or float (inner solver of the
// mixed-precision mode); dot-product partials are always accumulated in double.
template <class T, int K, int LPR, bool DOT>
__global__ void __launch_bounds__(512) k_spmm(int64_t n, int64_t pair_begin, int64_t pair_end, const int32_t *__restrict__ rowptr,
                                              const int32_t *__restrict__ col, const T *__restrict__ val,
                                              const T *__restrict__ x, T *__restrict__ y, double *__restrict__ part, const double *__restrict__ scal, int step) {
    if (scal && solve_done(scal, step)) return;
    const int rpb = blockDim.x / LPR;
    const int sub = threadIdx.x % LPR;
    const int grp = threadIdx.x / LPR;
    double dot[K];
#pragma unroll
    for (int c = 0; c < K; ++c) dot[c] = 0.0;
    for (int64_t row = int64_t(blockIdx.x) * rpb + grp; row < n; row += int64_t(gridDim.x) * rpb) {
        const int32_t rs = rowptr[row], re = rowptr[row + 1];
        T acc[K];
#pragma unroll
        for (int c = 0; c < K; ++c) acc[c] = T(0);
        for (int32_t p = rs + sub; p < re; p += LPR) {
            const T v = val[value_pos(row, rs, re - rs, p - rs, pair_begin, pair_end)];
            const T *xr = x + int64_t(col[p]) * K;
#pragma unroll
            for (int c = 0; c < K; ++c) acc[c] += v * xr[c];
        }
#pragma unroll
        for (int c = 0; c < K; ++c) acc[c] = group_sum<LPR>(acc[c]);
        if (sub == 0) {
#pragma unroll
            for (int c = 0; c < K; ++c) y[row * K + c] = acc[c];
            if (DOT) {
                const T *xr = x + row * K;
#pragma unroll
                for (int c = 0; c < K; ++c) dot[c] += double(acc[c]) * double(xr[c]);
            }
        }
    }
    if (DOT) {
        __shared__ double smem[16 * K];
        block_sum<K>(dot, smem);
        if (threadIdx.x < K) part[blockIdx.x * K + threadIdx.x] = pick<K>(dot, threadIdx.x);
    }
}

// Variant C ("edge row pairs", the default): the two dofs of an edge are consecutive rows with the
// SAME column pattern, and edge rows hold ~3/4 of the stored entries.  A lane group takes both
// rows at once: one column index and ONE gather of the x row serve two stored entries.  Vertex and
// face rows are walked singly.
// The kernel is latency-bound, not bandwidth-bound (a stored entry costs two dependent memory
// round trips: column index, then the x row), so each lane issues the index/value loads of U
// passes of its row up front and then all U gathers, before any arithmetic: a row of <= U * LPR
// entries pays the two latencies once instead of once per pass.  Lanes past the row end read a
// safe address with a zero value (no branches between the loads).
template <class T, int K, int LPR, bool DOT, int MODE = 0>   // MODE != 0: ablations for tools/probe_ablate.py (wrong results on purpose)
__global__ void __launch_bounds__(512) k_spmm_pair(int64_t n, int64_t pair_begin, int64_t pair_end, int xcd_windows, const int32_t *__restrict__ rowptr,
                                                   const int32_t *__restrict__ col, const T *__restrict__ val,
                                                   const T *__restrict__ x, T *__restrict__ y, double *__restrict__ part, const double *__restrict__ scal, int step) {
    if (scal && solve_done(scal, step)) return;
    constexpr int U = 2;
    constexpr int MFP = treduce_out(2 * K, LPR), MFS = treduce_out(K, LPR);   // sums per lane after the reduction
    const int rpb = blockDim.x / LPR;
    const int sub = threadIdx.x % LPR;
    const int grp = threadIdx.x / LPR;
    const int64_t npair = (pair_end - pair_begin) >> 1;
    const int64_t ngroups = n - npair;
    // which of the 2K (pair) / K (single row) sums this lane ends up with: entry t of a pair is y[row*K + t]
    int idx_p[2 * K], idx_s[K], own_p[2 * K], own_s[K];
    towner_init<2 * K, LPR>(idx_p, own_p, sub);
    towner_init<K, LPR>(idx_s, own_s, sub);
    double dot_p[MFP], dot_s[MFS];
#pragma unroll
    for (int f = 0; f < MFP; ++f) dot_p[f] = 0.0;
#pragma unroll
    for (int f = 0; f < MFS; ++f) dot_s[f] = 0.0;
    // row of lane-group index g (edge rows are taken two at a time)
    auto row_of = [&](int64_t g, bool &pair) -> int64_t {
        pair = false;
        if (g < pair_begin) return g;
        if (g < pair_begin + npair) { pair = true; return pair_begin + 2 * (g - pair_begin); }
        return g + npair;
    };
    // Row schedule.  Workgroups b, b + 8, ... share an XCD (and its L2).
    //  xcd_windows 1: every sweep step of the chip is cut into 8 windows of neighbouring row groups, one per XCD
    //    (an XCD gathers from one window of x per step, but over the launch it walks through all of x, three
    //    times: vertex rows, edge rows, face rows);
    //  xcd_windows >= 16: REGIONS.  The dofs of each class are in Morton order of the mesh, so the same fraction
    //    of the vertex, edge and face rows covers the same part of space.  XCD j takes chunk j*nc .. j*nc + nc - 1
    //    (nc = xcd_windows / 16) of all three classes, one chunk after the other, vertex, edge and face rows
    //    of a chunk back to back: its L2 serves the three passes over a chunk's part of x from one fetch.
    const int per = int(gridDim.x >> 3), slot = int(blockIdx.x >> 3), xcd = int(blockIdx.x & 7);
    const int nchunk = xcd_windows >> 4;
    const int64_t nface = n - pair_end;
    // class sizes of one chunk, whole waves (4 lane groups of 16): waves stay uniform in the row class
    const int64_t tc = int64_t(8) * (nchunk > 0 ? nchunk : 1);
    const int64_t gran = 64 / LPR > 0 ? 64 / LPR : 1;
    const int64_t sv = ((pair_begin + tc - 1) / tc + gran - 1) / gran * gran;
    const int64_t se = ((npair + tc - 1) / tc + gran - 1) / gran * gran;
    const int64_t sf = ((nface + tc - 1) / tc + gran - 1) / gran * gran;
    const uint32_t S = uint32_t(sv + se + sf);
    int64_t g, gstep, glimit;
    if (nchunk > 0) {
        g = int64_t(slot) * rpb + grp; gstep = int64_t(per) * rpb; glimit = int64_t(nchunk) * S;
    } else {
        int vblock = int(blockIdx.x);
        if (xcd_windows) {
            const int inner = (xcd_windows == 2 && (per & 31) == 0) ? (slot & 31) * (per >> 5) + (slot >> 5) : slot;   // 2: also CU-adjacent (probe)
            vblock = xcd * per + inner;
        }
        g = int64_t(vblock) * rpb + grp; gstep = int64_t(gridDim.x) * rpb; glimit = ngroups;
    }
    // schedule position -> (row, pair); false: an empty slot of the last chunks
    auto locate = [&](int64_t pos, int64_t &row, bool &pair) -> bool {
        if (nchunk == 0) { row = row_of(pos, pair); return true; }
        const uint32_t c = uint32_t(pos) / S, o = uint32_t(pos) - c * S;
        const int64_t gc = int64_t(xcd) * nchunk + c;
        pair = false;
        if (o < uint32_t(sv)) { row = gc * sv + o; return row < pair_begin; }
        if (o < uint32_t(sv + se)) { const int64_t e = gc * se + (o - uint32_t(sv)); pair = true; row = pair_begin + 2 * e; return e < npair; }
        const int64_t f = gc * sf + (o - uint32_t(sv + se));
        row = pair_end + f;
        return f < nface;
    };
    bool pair_n = false, valid_n = false;
    int64_t row_n = 0;
    int32_t rs_n = 0, re_n = 0;
    if (g < glimit) {
        valid_n = locate(g, row_n, pair_n);
        if (valid_n) { rs_n = rowptr[row_n]; re_n = rowptr[row_n + 1]; }
    }
    for (; g < glimit; g += gstep) {
        const int64_t row = row_n;
        const bool pair = pair_n, valid = valid_n;
        const int32_t rs = rs_n, re = re_n;
        // the row pointers of the NEXT row are requested now: one of the three dependent round trips
        // of a row (pointers -> indices -> x) leaves the critical path
        if (g + gstep < glimit) {
            valid_n = locate(g + gstep, row_n, pair_n);
            if (valid_n) { rs_n = rowptr[row_n]; re_n = rowptr[row_n + 1]; }
        }
        if (!valid) continue;
        T acc[2 * K];                             // [0, K): row, [K, 2K): row + 1
#pragma unroll
        for (int c = 0; c < 2 * K; ++c) acc[c] = T(0);
        for (int32_t p0 = rs + sub; p0 < re; p0 += U * LPR) {
            int32_t j[U];
            T v0[U], v1[U], xv[U][K];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int32_t p = p0 + u * LPR;
                j[u] = -1; v0[u] = T(0); v1[u] = T(0);
                if (p < re) {                        // lanes past the row end issue nothing (exec-masked loads)
                    j[u] = col[p];
                    // one 16-byte (fp32: 8-byte) load either way: the interleaved values of the two rows of a pair, or a
                    // single row's value plus its unused right neighbour (val is allocated with one element of slack, CsrViewT)
                    typedef T pair_t __attribute__((ext_vector_type(2), aligned(sizeof(T))));
                    const pair_t vv = *reinterpret_cast<const pair_t *>(val + rs + (pair ? 2 * (p - rs) : (p - rs)));
                    v0[u] = vv.x; v1[u] = vv.y;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int c = 0; c < K; ++c) xv[u][c] = T(0);
                if (j[u] >= 0) {
                    if (MODE == 1) {          // no gather at all
#pragma unroll
                        for (int c = 0; c < K; ++c) xv[u][c] = T(j[u]);
                    } else {
                        const T *xr = x + int64_t(MODE == 2 ? (j[u] & 255) : j[u]) * K;   // 2: L1-resident gather
#pragma unroll
                        for (int c = 0; c < K; ++c) xv[u][c] = xr[c];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int c = 0; c < K; ++c) {
                    acc[c] += v0[u] * xv[u][c];
                    acc[K + c] += v1[u] * xv[u][c];
                }
        }
        if (MODE == 3) {   // no reduction, (practically) no store
            T t = T(0);
#pragma unroll
            for (int c = 0; c < 2 * K; ++c) t += acc[c];
            if (t == T(1.2345e30)) y[row * K] = t;
            continue;
        }
        // wave-uniform choice (row classes are contiguous, so all but a handful of waves are uniform)
        if (__builtin_amdgcn_ballot_w64(pair) != 0) {
            TReduce<2 * K, LPR>::run(acc, sub);
#pragma unroll
            for (int f = 0; f < MFP; ++f)
                if (own_p[f] && (pair || idx_p[f] < K)) {
                    const int64_t at = row * K + idx_p[f];
                    y[at] = acc[f];
                    if (DOT) dot_p[f] += double(acc[f]) * double(x[at]);
                }
        } else {
            TReduce<K, LPR>::run(acc, sub);
#pragma unroll
            for (int f = 0; f < MFS; ++f)
                if (own_s[f]) {
                    const int64_t at = row * K + idx_s[f];
                    y[at] = acc[f];
                    if (DOT) dot_s[f] += double(acc[f]) * double(x[at]);
                }
        }
    }
    if (DOT) {
        __shared__ double smem[16 * K];
        double dot[K];   // back to one slot per column for the block sum
#pragma unroll
        for (int c = 0; c < K; ++c) {
            double d = 0.0;
#pragma unroll
            for (int f = 0; f < MFP; ++f) d += (idx_p[f] % K == c) ? dot_p[f] : 0.0;
#pragma unroll
            for (int f = 0; f < MFS; ++f) d += (idx_s[f] == c) ? dot_s[f] : 0.0;
            dot[c] = d;
        }
        block_sum<K>(dot, smem);
        if (threadIdx.x < K) part[blockIdx.x * K + threadIdx.x] = pick<K>(dot, threadIdx.x);
    }
}

// tuning knobs (remo_debug_tune): 0 = heuristic default
struct SpmmTuning {
    int mode = 0;     // ablation mode of the pair kernel (K = 5, 16 lanes per row only)
    int variant = 0;  // 1 = lane per stored entry, 3 = edge row pairs (default)
    int lpr = 0;
    int threads = 0;
    int mapping = -1;
    int grid = 0;
};
static SpmmTuning g_tune;
static int g_fold_first = 1;   // remo_debug_tune key 9: FIRST Chebyshev step inside the update launch (0 = own launch)
void set_fold_first(int v) { g_fold_first = v; }
void set_spmm_tuning(int key, int value) {
    switch (key) {
        case 0: g_tune.variant = value; break;
        case 1: g_tune.lpr = value; break;
        case 2: g_tune.threads = value; break;
        case 3: g_tune.mapping = value; break;
        case 4: g_tune.grid = value; break;
        case 5: g_tune.mode = value; break;
        default: break;
    }
}

int choose_lanes_per_row(int64_t n, int64_t nnz) {
    if (g_tune.lpr) return g_tune.lpr;
    const double avg = double(nnz) / double(n > 0 ? n : 1);
    if (avg > 40) return 16;
    if (avg > 20) return 8;
    return 4;
}
static int spmm_threads() { return g_tune.threads ? g_tune.threads : 256; }

int spmv_grid(int64_t n, int lpr) {
    if (g_tune.grid) return g_tune.grid;
    const int64_t rpb = spmm_threads() / lpr;
    int64_t g = (n + rpb - 1) / rpb;
    g = (g + 7) / 8 * 8;  // whole residue classes mod 8 (one per XCD)
    // whole multiples of the 256 CUs finish together; measured at k = 5, 329 k rows, interleaved pair values:
    // 1024 -> 55.4 us, 896 -> 61.0, 768 -> 57.5, 640 -> 69.3 (before the interleaving 768 was ahead: 59.3 vs 59.7)
    constexpr int kSpmmBlocks = 1024;
    static_assert(kSpmmBlocks <= kMaxPartialBlocks, "partials buffer");
    if (g > kSpmmBlocks) g = kSpmmBlocks;
    if (g < 8) g = 8;
    return int(g);
}

template <class T, int K> static void spmm_dispatch(const CsrViewT<T> &A, const T *x, T *y, double *part, const double *scal, int step, int nb, hipStream_t s) {
    int lpr = choose_lanes_per_row(A.n, A.nnz);
    const int threads = spmm_threads();
    int variant = g_tune.variant ? g_tune.variant : 3;
    if (variant == 3 && !(A.pair_end > A.pair_begin)) variant = 1;
    // default row schedule of the pair kernel: XCD windows (measured 69 -> 60 us at 334k rows, k = 5); once the matrix no
    // longer stays in the 256 MB of MALL between launches (3D, more than ~20 M stored entries), XCD regions (4 chunks
    // each in these measurements): inside the solver loop (bench.py --tune 3=1 against 3=64 on one box) 152.7 -> 149.8 us at 716k rows,
    // 490.8 -> 480.2 us at 2.17 M rows, but 58.6 -> 60.0 us at 289k rows (the stand-alone probe, tools/probe_regions.py,
    // shows 2 ... 16 chunks within 1 % of each other)
    // chunks per XCD: about 0.4 MB of x per chunk, 4 ... 64 (bench at 5.4 M rows: 4 chunks 1257 us, 16: 1227, 64: 1216)
    int64_t nc = (A.n * int64_t(K) * int64_t(sizeof(T)) / 8 + 210000) / 420000;
    nc = nc < 4 ? 4 : (nc > 64 ? 64 : nc);
    const int mapping = (g_tune.mapping >= 0) ? g_tune.mapping : ((lpr == 16 && A.nnz > 20000000) ? int(16 * nc) : 1);
#define REMO_SPMM(L)                                                                                                        \
    if (part)                                                                                                               \
        hipLaunchKernelGGL((k_spmm<T, K, L, true>), dim3(nb), dim3(threads), 0, s, A.n, A.pair_begin, A.pair_end, A.rowptr, A.col, A.val, x, y, part, scal, step); \
    else                                                                                                                    \
        hipLaunchKernelGGL((k_spmm<T, K, L, false>), dim3(nb), dim3(threads), 0, s, A.n, A.pair_begin, A.pair_end, A.rowptr, A.col, A.val, x, y, part, scal, step)
#define REMO_SPMM_PAIR(L)                                                                                                               \
    if (part)                                                                                                                           \
        hipLaunchKernelGGL((k_spmm_pair<T, K, L, true>), dim3(nb), dim3(threads), 0, s, A.n, A.pair_begin, A.pair_end, mapping, A.rowptr, A.col, A.val, x, y, part, scal, step); \
    else                                                                                                                                \
        hipLaunchKernelGGL((k_spmm_pair<T, K, L, false>), dim3(nb), dim3(threads), 0, s, A.n, A.pair_begin, A.pair_end, mapping, A.rowptr, A.col, A.val, x, y, part, scal, step)
    if constexpr (K == 5 && sizeof(T) == 8) {   // ablation modes of tools/probe_ablate.py
        if (variant == 3 && lpr == 16 && g_tune.mode >= 1 && g_tune.mode <= 3 && !part) {
            if (g_tune.mode == 1) hipLaunchKernelGGL((k_spmm_pair<T, 5, 16, false, 1>), dim3(nb), dim3(threads), 0, s, A.n, A.pair_begin, A.pair_end, mapping, A.rowptr, A.col, A.val, x, y, part, scal, step);
            if (g_tune.mode == 2) hipLaunchKernelGGL((k_spmm_pair<T, 5, 16, false, 2>), dim3(nb), dim3(threads), 0, s, A.n, A.pair_begin, A.pair_end, mapping, A.rowptr, A.col, A.val, x, y, part, scal, step);
            if (g_tune.mode == 3) hipLaunchKernelGGL((k_spmm_pair<T, 5, 16, false, 3>), dim3(nb), dim3(threads), 0, s, A.n, A.pair_begin, A.pair_end, mapping, A.rowptr, A.col, A.val, x, y, part, scal, step);
            return;
        }
    }
    if (variant == 3) {
        if (lpr >= 32) { REMO_SPMM_PAIR(32); }
        else if (lpr == 16) { REMO_SPMM_PAIR(16); }
        else if (lpr == 8) { REMO_SPMM_PAIR(8); }
        else { REMO_SPMM_PAIR(4); }
    } else {
        if (lpr >= 32) { REMO_SPMM(32); }
        else if (lpr == 16) { REMO_SPMM(16); }
        else if (lpr == 8) { REMO_SPMM(8); }
        else { REMO_SPMM(4); }
    }
#undef REMO_SPMM
#undef REMO_SPMM_PAIR
}

template <class T> bool patch_applies(const CsrViewT<T> &A, int k) {
    return A.patch && k * A.patch->t.E <= A.patch->t.block && patch_lds_bytes(A.patch->lds_rows, k, A.patch->t.block, A.patch->t.all_slab != 0) <= kPatchLdsLimit;
}
template bool patch_applies<double>(const CsrViewT<double> &, int);
template bool patch_applies<float>(const CsrViewT<float> &, int);

template <class T> void launch_spmm(const CsrViewT<T> &A, int k, const T *x, T *y, double *part, const double *scal, int nb, hipStream_t s, int step, bool defer) {
    // patch operator (patch.hip): its tables are laid out for the batch's own column count - a product with more columns than
    // that (inspection hooks only) goes through the stored matrix
    if (patch_applies(A, k)) {
        launch_patch_spmm(A, k, x, y, part, scal, nb, s, step, defer);
        return;
    }
    switch (k) {
        case 1: spmm_dispatch<T, 1>(A, x, y, part, scal, step, nb, s); break;
        case 2: spmm_dispatch<T, 2>(A, x, y, part, scal, step, nb, s); break;
        case 3: spmm_dispatch<T, 3>(A, x, y, part, scal, step, nb, s); break;
        case 4: spmm_dispatch<T, 4>(A, x, y, part, scal, step, nb, s); break;
        case 5: spmm_dispatch<T, 5>(A, x, y, part, scal, step, nb, s); break;
        case 6: spmm_dispatch<T, 6>(A, x, y, part, scal, step, nb, s); break;
        case 7: spmm_dispatch<T, 7>(A, x, y, part, scal, step, nb, s); break;
        default: spmm_dispatch<T, 8>(A, x, y, part, scal, step, nb, s); break;
    }
}
template void launch_spmm<double>(const CsrViewT<double> &, int, const double *, double *, double *, const double *, int, hipStream_t, int, bool);
template void launch_spmm<float>(const CsrViewT<float> &, int, const float *, float *, double *, const double *, int, hipStream_t, int, bool);

// ------------------------------------------------------------------------------------------
// Jacobi-PCG vector kernels (CGSolver(a.mat, c.mat), ngsolve_functions.py:50-51), K columns at
// once with per-column step lengths.  Three launches per step:
//   spmm      q = A p, partials of <p,q>
//   update    alpha = <Cr,r>/<p,q>;  x += alpha p;  r -= alpha q;  partials of <C r, r>
//   direction beta = <Cr,r>_new/<Cr,r>_old;  p = C r + beta p
// Scalars never visit the host: every block re-reduces the (<= 1024 x K) per-block partials of
// the previous launch in a fixed order, so results are bit-reproducible and there is no atomic.
// A column whose <Cr,r> has dropped below tol^2 <Cr0,r0> (or that broke down) is frozen
// (alpha = beta = 0), which makes post-convergence steps harmless.

// Two-level preconditioner ("multigrid" of the reference, ngsolve_functions.py:46: a lowest-order
// coarse space plus a smoother on the high-order dofs).  In the hierarchical basis the vertex
// functions ARE the P1 space and vertex dofs are numbered first, so the coarse problem is the leading
// nv x nv block of the assembled matrix, read in place (columns are sorted: the block's entries
// lead every row).  C = blockdiag( q_d(A_vv), D_hh^-1 ): a fixed Chebyshev polynomial of degree d in
// the Jacobi-scaled vertex block (spectrum bounds [lmax/ratio, lmax], lmax = Gershgorin bound, so
// q_d is positive definite on the whole spectrum and plain PCG stays valid) and Jacobi on edge/face
// dofs.  nv = 0 gives plain Jacobi ("local").
template <class T> struct ChebArgsT {
    int64_t nv;       // free vertex dofs (0: Jacobi)
    double inv_theta; // 1 / theta, theta = (lmax + lmin) / 2
    T *z, *res;       // [nv][K] polynomial value so far / residual of the vertex block system
    T *d0;            // [nv][K] first Chebyshev direction
};
// FIRST Chebyshev step folded into the update launch (k_pcg_update): nb_flat = 0 switches it off
// q = A p of the patch operator with the rows shared by several patches still in the boundary slab (PcgBuffersT::defer_q)
template <class T> struct QViewT {
    const int32_t *bptr = nullptr, *bslot = nullptr;
    const T *Yb = nullptr;
    int ahead = 1;      // 0: the slots of a shared row one by one (remo_debug_tune key 27)
    int skip_x = 0;     // 1: x += alpha p is left to the direction launch of the step (PcgBuffersT::x_in_direction)
    int tile = 0;              // 1: the tile form of the update launch (k_pcg_update; remo_debug_tune key 31)
    const int32_t *row4 = nullptr;   // PatchTables::row4 (tile form)
    uint64_t slab_bytes = 0;   // != 0 (slab below 4 GB): the slots a row does not have are not fetched at all (buffer loads, offset out of range)
};
template <class T> struct FoldArgsT {
    int nb_flat = 0;             // workgroups [0, nb_flat) do the flat update of the rows >= nv, the rest the vertex rows
    const int32_t *rowptr = nullptr, *col = nullptr;
    const T *val = nullptr;
    T *d_new = nullptr, *stage = nullptr;
    double c1 = 0.0, c2 = 0.0;
};
// All PCG kernels are templates on the storage type T of matrix values and vectors: double = the
// product path, float = the inner solver of the mixed-precision mode (BASELINE config 5).  Scalars,
// partial sums and the convergence test are double in both.
// scal[kFloorSlot + c]: absolute floor of <Cr,r> below which column c is frozen as well (0 in the plain
// fp64 solve; the outer target of the refinement in the mixed mode)
constexpr int kFloorSlot = 5 * 8;

template <class T, int K>
__global__ void __launch_bounds__(256) k_pcg_init(int64_t n, ChebArgsT<T> ch, const T *__restrict__ f, const T *__restrict__ dinv,
                                                  T *__restrict__ x, T *__restrict__ r, T *__restrict__ p,
                                                  double *__restrict__ part_rz) {
    __shared__ double smem[16 * K];
    double rz[K];
#pragma unroll
    for (int c = 0; c < K; ++c) rz[c] = 0.0;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
        const T d = dinv[i];
        const bool coarse = i < ch.nv;
#pragma unroll
        for (int c = 0; c < K; ++c) {
            const T ri = f[i * K + c];
            const T zi = d * ri;
            x[i * K + c] = T(0);
            r[i * K + c] = ri;
            p[i * K + c] = coarse ? T(0) : zi;
            rz[c] += coarse ? 0.0 : double(ri) * double(zi);   // the vertex block's share comes from the Chebyshev kernels
        }
    }
    block_sum<K>(rz, smem);
    if (threadIdx.x < K) part_rz[blockIdx.x * K + threadIdx.x] = rz[threadIdx.x];
}

template <class T, int K>
__global__ void __launch_bounds__(256) k_pcg_update(int64_t n, int step, double tol2, int x_only, int nb_spmv, int nb_rz, ChebArgsT<T> ch,
                                                    const double *__restrict__ part_pq, const double *__restrict__ part_rz_cur,
                                                    double *__restrict__ part_rz_next, double *__restrict__ rz0,
                                                    PcgProgress *progress, int progress_len, const T *__restrict__ p,
                                                    const T *__restrict__ q, T *__restrict__ x, T *__restrict__ r,
                                                    const T *__restrict__ dinv, FoldArgsT<T> fold, QViewT<T> qv, double *__restrict__ clear_bins = nullptr) {
    // scal = rz0[8] | pq[8] | rz of even steps[8] | rz of odd steps[8]: totals forwarded between launches
    // by workgroup 0, so every launch re-reduces only the ONE partial array that is new to it
    __shared__ double smem[16 * 3 * K];
    double *scal = rz0;
    if (solve_done(scal, step)) return;
    double pq[K], rz[K], unused[K], alpha[K], acc[K];
    if (step == 0) {
        reduce_partials3<K>(part_pq, nb_spmv, part_rz_cur, nb_rz, nullptr, 0, pq, rz, unused, smem);
    } else {
        reduce_partials<K>(part_pq, nb_spmv, pq, smem);
#pragma unroll
        for (int c = 0; c < K; ++c) rz[c] = scal[16 + 8 * (step & 1) + c];
    }
#pragma unroll
    for (int c = 0; c < K; ++c) {
        const double r0 = (step == 0) ? rz[c] : rz0[c];
        const bool live = (rz[c] > tol2 * r0) && (rz[c] > scal[kFloorSlot + c]) && (pq[c] > 0.0);
        alpha[c] = live ? rz[c] / pq[c] : 0.0;
        acc[c] = 0.0;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        bool any_live = false;
#pragma unroll
        for (int c = 0; c < K; ++c) any_live |= (alpha[c] != 0.0);
        if (!any_live) {   // every column frozen: later launches of this solve are no-ops; tell the host where it ended
            reinterpret_cast<int *>(scal + kDoneSlot)[0] = step + 1;   // acts on the launches of steps > step only (solve_done)
            publish_progress(progress + (progress_len - 1), rz, K, step);
        }
#pragma unroll
        for (int c = 0; c < K; ++c) scal[8 + c] = pq[c];
        if (step == 0)
#pragma unroll
            for (int c = 0; c < K; ++c) { rz0[c] = rz[c]; scal[16 + c] = rz[c]; }
        // progress record in mapped host memory: data first, then the step number (system scope)
        publish_progress(progress + (step % (progress_len - 1)), rz, K, step);   // the last slot is the "done" record
    }
    if (clear_bins) {     // the patch operator's <p, A p> bins of the NEXT step (PcgBuffersT::pq_bins): this launch is the last reader of that set
        for (int i = int(blockIdx.x) * int(blockDim.x) + int(threadIdx.x); i < kPqBins * K; i += int(gridDim.x) * int(blockDim.x)) clear_bins[i] = 0.0;
    }
    if (x_only) {   // residual replacement step (mixed precision): r and the <Cr,r> partials come from k_mixed_replace
        for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
#pragma unroll
            for (int c = 0; c < K; ++c) x[i * K + c] += T(alpha[c]) * p[i * K + c];
        }
        return;
    }
    // fold.nb_flat > 0: the workgroups behind the first nb_flat take the vertex rows, 8 lanes per row, and run the FIRST
    // Chebyshev step on them in the same pass (its operand D^-1 (r - alpha q) / theta is formed per gathered entry from
    // the OLD r and q, both complete at this point).  The new vertex residual goes to a staging vector (fold.stage =
    // the free one of the two direction buffers): r itself is still being gathered by the neighbours' rows; the next
    // Chebyshev launch commits it.  One launch less per PCG step.
    const int nb_flat = fold.nb_flat > 0 ? fold.nb_flat : int(gridDim.x);
    if (int(blockIdx.x) >= nb_flat) {
        constexpr int LPR = 8, RPB = 256 / LPR;
        static_assert(K <= LPR, "one column per lane after the transposing reduction");
        const int sub = threadIdx.x % LPR, grp = threadIdx.x / LPR;
        int idx[K], own[K];
        towner_init<K, LPR>(idx, own, sub);
        const int mycol = idx[0];
        const bool mine = own[0] != 0;
        const T inv_theta = T(ch.inv_theta), c1 = T(fold.c1), c2 = T(fold.c2);
        T al[K], a_mine = T(0);
#pragma unroll
        for (int c = 0; c < K; ++c) { al[c] = T(alpha[c]); a_mine = (c == mycol) ? al[c] : a_mine; }
        const int64_t nv = ch.nv;
        for (int64_t row = int64_t(int(blockIdx.x) - nb_flat) * RPB + grp; row < nv; row += int64_t(int(gridDim.x) - nb_flat) * RPB) {
            const int32_t rs = fold.rowptr[row], re = fold.rowptr[row + 1];
            const int64_t at = row * K + mycol;
            T di = T(0), rn = T(0);
            if (mine) {
                di = dinv[row];
                if (!qv.skip_x) x[at] += a_mine * p[at];
                rn = r[at] - a_mine * q[at];
            }
            T t[K];
#pragma unroll
            for (int c = 0; c < K; ++c) t[c] = T(0);
            for (int32_t pp = rs + sub; pp < re; pp += LPR) {
                const int32_t j = fold.col[pp];
                if (j >= nv) break;  // columns ascend: the vertex block leads the row
                const T v = fold.val[pp] * dinv[j] * inv_theta;
                const T *rj = r + int64_t(j) * K, *qj = q + int64_t(j) * K;
#pragma unroll
                for (int c = 0; c < K; ++c) t[c] += v * (rj[c] - al[c] * qj[c]);
            }
            TReduce<K, LPR>::run(t, sub);
            if (mine) {
                const T dold = di * rn * inv_theta;
                const T ri = rn - t[0];
                ch.z[at] = dold;
                ch.res[at] = ri;
                fold.d_new[at] = c1 * dold + c2 * di * ri;
                fold.stage[at] = rn;
            }
        }
        return;
    }
    if (qv.tile) {
        // TILE form (launcher: patch operator with every row in the slab and its row4 table, x left to the direction launch, no folded
        // Chebyshev step, n K sizeof(T) and the slab below 4 GB).  A wave takes 64 rows = 64 K values at a time.  Lane l fetches the
        // four slots of row l of the tile (ONE 16-byte load per lane: 1 KB per wave); then, pass by pass, lane l handles value
        // 64 t + l of the tile: r as 512 consecutive bytes per wave and instruction, the slots of the value's row from the lane that
        // holds them (ds_bpermute), and ITS column of each slab slot - the K lanes of a row read a slot's 40 bytes side by side.
        // All 6 K loads of a lane are in flight together.  Same sums, same order as the row form (rows of five or more patches, the
        // first vertex rows: their further slots are summed by the row's lane and handed over through LDS - another association).
        constexpr uint32_t S = sizeof(T);
        __shared__ double ex_lds[4 * 64 * K];
        __syncthreads();          // (the reductions' last reads of smem)
        if (threadIdx.x == 0) {
#pragma unroll
            for (int c = 0; c < K; ++c) smem[c] = alpha[c];
        }
        __syncthreads();
        const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const uint32_t N = uint32_t(n) * K, NV = uint32_t(ch.nv) * K;
        const rsrc_t rr = make_rsrc(r, uint64_t(N) * S), rd = make_rsrc(dinv, uint64_t(n) * S), rs = make_rsrc(qv.Yb, qv.slab_bytes),
                     r4 = make_rsrc(qv.row4, uint64_t(n) * 16);
        uint32_t rowof[K], colof[K];      // pass t: this lane's value is (row rowof[t] of the tile, column colof[t]) - the same for every tile
        T al[K];
        double accv[K];
#pragma unroll
        for (int t = 0; t < K; ++t) {
            const uint32_t v = 64u * t + lane;
            rowof[t] = v / uint32_t(K); colof[t] = v - rowof[t] * uint32_t(K);
            al[t] = T(smem[colof[t]]); accv[t] = 0.0;
        }
        double *exw = ex_lds + wave * (64 * K);
        const uint32_t nwaves = uint32_t(gridDim.x) * 4u;
        // (the slots of the NEXT tile are requested before this tile's values: one memory round trip per tile on a wave's critical path, not two)
        int32_t w4n[4];
        {
            const uint32_t row0 = (uint32_t(blockIdx.x) * 4u + wave) * 64u + lane;
            buf_load<int32_t, 4>(r4, row0 < uint32_t(n) ? row0 * 16u : kOutOfRange, w4n);
        }
        for (uint32_t R0 = (uint32_t(blockIdx.x) * 4u + wave) * 64u; R0 < uint32_t(n); R0 += nwaves * 64u) {
            const uint32_t myrow = R0 + lane;
            int32_t w4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) w4[j] = myrow < uint32_t(n) ? w4n[j] : -1;
            {
                const uint32_t nrow = myrow + nwaves * 64u;
                buf_load<int32_t, 4>(r4, nrow < uint32_t(n) ? nrow * 16u : kOutOfRange, w4n);
            }
            const uint32_t e0 = R0 * uint32_t(K);
            T rv[K], dv[K], part[K][4];
#pragma unroll
            for (int t = 0; t < K; ++t) {
                const uint32_t e = e0 + 64u * t + lane;
                T one[1];
                buf_load<T, 1>(rr, e < N ? e * S : kOutOfRange, one);
                rv[t] = one[0];
                buf_load<T, 1>(rd, e < N ? (R0 + rowof[t]) * S : kOutOfRange, one);
                dv[t] = one[0];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int32_t sj = __builtin_amdgcn_ds_bpermute(int(rowof[t] << 2), w4[j]);
                    buf_load<T, 1>(rs, sj >= 0 ? (uint32_t(sj) * uint32_t(K) + colof[t]) * S : kOutOfRange, one);
                    part[t][j] = one[0];
                }
            }
            const bool more = w4[3] == -2;
            const bool any_more = __builtin_amdgcn_ballot_w64(more) != 0;
            if (any_more) {       // (wave-uniform) rows of five or more patches in this tile: their lanes sum the slots from the fourth on
#pragma unroll
                for (int c = 0; c < K; ++c) exw[lane * K + c] = 0.0;
                if (more) {
                    const int32_t c0 = qv.bptr[myrow], c1 = qv.bptr[myrow + 1];
                    for (int32_t sl = c0 + 3; sl < c1; ++sl) {
                        const int64_t a2 = qv.bslot[sl];
#pragma unroll
                        for (int c = 0; c < K; ++c) exw[lane * K + c] += double(qv.Yb[a2 * K + c]);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's LDS writes are done before its lanes read each other's
            }
#pragma unroll
            for (int t = 0; t < K; ++t) {
                const uint32_t e = e0 + 64u * t + lane;
                T qi = part[t][0];
#pragma unroll
                for (int j = 1; j < 4; ++j) qi += part[t][j];
                if (any_more) qi += T(exw[64 * t + lane]);
                const T ri = rv[t] - al[t] * qi;
                T one[1] = {ri};
                buf_store<T, 1>(rr, e < N ? e * S : kOutOfRange, one);
                accv[t] += (e < NV || e >= N) ? 0.0 : double(ri) * double(ri) * double(dv[t]);
            }
            if (any_more) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads done before the next tile's zeros
        }
#pragma unroll
        for (int c = 0; c < K; ++c) {
            double tsum = 0.0;
#pragma unroll
            for (int t = 0; t < K; ++t) tsum += (colof[t] == uint32_t(c)) ? accv[t] : 0.0;
            acc[c] = tsum;
        }
        __syncthreads();
        const double mine = block_sum_column<K>(acc, smem);
        if (threadIdx.x < K) part_rz_next[blockIdx.x * K + threadIdx.x] = mine;
        return;
    }
    const int64_t i0 = fold.nb_flat > 0 ? ch.nv : 0;
#ifndef REMO_UPD_UNROLL
#define REMO_UPD_UNROLL 2
#endif
#pragma unroll REMO_UPD_UNROLL
    for (int64_t i = i0 + int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(nb_flat) * blockDim.x) {
        const T d = dinv[i];
        const bool coarse = i < ch.nv;
        T qi[K];
        int32_t b0 = 0, b1 = 0;
        if (qv.bptr) { b0 = qv.bptr[i]; b1 = qv.bptr[i + 1]; }
        // x, p and r of the row are requested HERE, with the row's slab pointers, not behind the branch that gathers q: three more
        // vectors in flight while the slab slots make their two round trips
        T xv[K], pv[K], rv[K];
        if (qv.skip_x) {      // x += alpha p rides on the direction launch, which reads p anyway: neither x nor p is touched here
#pragma unroll
            for (int c = 0; c < K; ++c) { xv[c] = T(0); pv[c] = T(0); rv[c] = r[i * K + c]; }
        } else {
#pragma unroll
            for (int c = 0; c < K; ++c) { xv[c] = x[i * K + c]; pv[c] = p[i * K + c]; rv[c] = r[i * K + c]; }
        }
        if (b1 > b0) {      // a row shared by several patches: its q is still spread over the slab, one slot per patch, ascending
            // the first kSlabAhead slots without a branch and with all their loads in flight together (slot numbers, then slab
            // rows: two round trips; the plain loop made two per slot, and a wave waits for its row with the most slots) - a
            // missing slot reads slot 0 and counts for nothing; further slots (rare) one by one
            constexpr int kSlabAhead = 4;
            if (qv.ahead) {
            int32_t at[kSlabAhead];
#pragma unroll
            for (int j = 0; j < kSlabAhead; ++j) at[j] = qv.bslot[b0 + j < b1 ? b0 + j : b0];
            T part[kSlabAhead][K];
            if (qv.slab_bytes) {
                // a row has 1.84 slots on average (44 % of the rows exactly one, now that every row goes through the slab): a lane asks
                // only for the slots its row has - the others get an offset beyond the buffer, which sends no request and returns 0
                const rsrc_t rs = make_rsrc(qv.Yb, qv.slab_bytes);
#pragma unroll
                for (int j = 0; j < kSlabAhead; ++j)
                    buf_load<T, K>(rs, b0 + j < b1 ? uint32_t(at[j]) * uint32_t(K * sizeof(T)) : kOutOfRange, part[j]);
#pragma unroll
                for (int c = 0; c < K; ++c) qi[c] = part[0][c];
#pragma unroll
                for (int j = 1; j < kSlabAhead; ++j)
#pragma unroll
                    for (int c = 0; c < K; ++c) qi[c] += part[j][c];
            } else {
#pragma unroll
            for (int j = 0; j < kSlabAhead; ++j)
#pragma unroll
                for (int c = 0; c < K; ++c) part[j][c] = qv.Yb[int64_t(at[j]) * K + c];
#pragma unroll
            for (int c = 0; c < K; ++c) qi[c] = part[0][c];
#pragma unroll
            for (int j = 1; j < kSlabAhead; ++j) {
                const T w = b0 + j < b1 ? T(1) : T(0);
#pragma unroll
                for (int c = 0; c < K; ++c) qi[c] += w * part[j][c];
            }
            }
            } else {
#pragma unroll
                for (int c = 0; c < K; ++c) qi[c] = T(0);
            }
            for (int32_t sl = b0 + (qv.ahead ? kSlabAhead : 0); sl < b1; ++sl) {
                const int64_t a2 = qv.bslot[sl];
#pragma unroll
                for (int c = 0; c < K; ++c) qi[c] += qv.Yb[a2 * K + c];
            }
        } else {
#pragma unroll
            for (int c = 0; c < K; ++c) qi[c] = q[i * K + c];
        }
#pragma unroll
        for (int c = 0; c < K; ++c) {
            const T a = T(alpha[c]);
            const T xi = xv[c] + a * pv[c];
            const T ri = rv[c] - a * qi[c];
            if (!qv.skip_x) x[i * K + c] = xi;
            r[i * K + c] = ri;
            acc[c] += coarse ? 0.0 : double(ri) * double(ri) * double(d);   // the vertex block's share comes from the Chebyshev kernels
        }
    }
    __syncthreads();
    block_sum<K>(acc, smem);
    if (threadIdx.x < K) part_rz_next[blockIdx.x * K + threadIdx.x] = acc[threadIdx.x];
}

// One Chebyshev step on the vertex block, 8 lanes per row:
//   z += d;  res -= A_vv d;  d' = c1 d + c2 D^-1 res
// FIRST: z = 0, res = r, d = D^-1 r / theta are formed on the fly from the PCG residual (no set-up
// pass).  LAST = 1 (degree 1 only): z = d, nothing else.  LAST = 2: the polynomial's last term needs no
// product of its own (z_final = z + d + d'), so the launch that forms d' also finishes z and leaves the
// <r, z> partial sums for the PCG scalars: a polynomial of `degree` terms costs degree - 1 launches.
// TC = storage type of the chain (block values, Jacobi factors, directions, residual, running z): T, or float inside an fp64
// solve (large vertex blocks: the launches are HBM streams there, and a preconditioner may be applied inexactly - the
// recurrences of x and r never see it).  r (read) and the finished z (written for the direction kernel) stay in T.
// ELL: the first kEllWidth entries of a row come from the fixed-width image (ecol, eval; k_vblock_ell) and `rowptr` holds the
// begin / end PAIRS of the entries beyond them in (col, val).
template <class T, class TC, int K, bool FIRST, int LAST, bool ELL = false>
__global__ void __launch_bounds__(256) k_cheb_step(int64_t nv, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                   const TC *__restrict__ val, const TC *__restrict__ dinv,
                                                   const TC *__restrict__ d_old, TC *__restrict__ d_new,
                                                   TC *__restrict__ zc, TC *__restrict__ res, T *__restrict__ z, double c1_, double c2_, double inv_theta_,
                                                   T *__restrict__ r, double *__restrict__ part, const double *__restrict__ scal, int commit, int step,
                                                   const int32_t *__restrict__ ecol = nullptr, const TC *__restrict__ eval = nullptr) {
    const TC c1 = TC(c1_), c2 = TC(c2_), inv_theta = TC(inv_theta_);
    constexpr int LPR = 8, RPB = 256 / LPR;
    constexpr bool SAME = sizeof(T) == sizeof(TC);
    static_assert(K <= LPR, "one column per lane after the transposing reduction");
    if (solve_done(scal, step)) return;
    const int sub = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    // the kernel is a chain of dependent round trips on a tiny block (launch-latency class): after the
    // transposing reduction lane `sub` owns column mycol of its row, so the vectors of the update are
    // requested per lane BEFORE the row is walked and need no round trip of their own
    int idx[K], own[K];
    towner_init<K, LPR>(idx, own, sub);
    const int mycol = idx[0];
    const bool mine = own[0] != 0;
    double dot = 0.0;
    for (int64_t row = int64_t(blockIdx.x) * RPB + grp; row < nv; row += int64_t(gridDim.x) * RPB) {
        constexpr int NE = ELL ? kEllWidth / LPR : 1;
        int32_t ej[NE];
        TC ev[NE];
        int32_t rs, re;
        if constexpr (ELL) {
            const int64_t eb = row * kEllWidth + sub;
#pragma unroll
            for (int u = 0; u < NE; ++u) { ej[u] = ecol[eb + u * LPR]; ev[u] = eval[eb + u * LPR]; }
            rs = rowptr[2 * row]; re = rowptr[2 * row + 1];
        } else {
            rs = rowptr[row]; re = rowptr[row + 1];
        }
        const int64_t at = row * K + mycol;
        TC di = TC(0), dold_in = TC(0), z_in = TC(0), res_in = TC(0);
        T rr = T(0);
        if (mine) {
            di = dinv[row];
            if (SAME && commit) {   // the update launch ran the FIRST step and left the new vertex residual in d_new (free until this
                rr = T(d_new[at]);  // launch writes it): r could not take it while the neighbours' rows were still gathering r
                r[at] = rr;
            } else {
                rr = r[at];
            }
            if (!FIRST) { dold_in = d_old[at]; z_in = zc[at]; res_in = res[at]; }
        }
        TC t[K];
#pragma unroll
        for (int c = 0; c < K; ++c) t[c] = TC(0);
        if constexpr (ELL) {
#pragma unroll
            for (int u = 0; u < NE; ++u) {
                const int32_t j = ej[u];
                const TC v = FIRST ? ev[u] * dinv[j] * inv_theta : ev[u];
                if (FIRST) {
                    const T *dj = r + int64_t(j) * K;
#pragma unroll
                    for (int c = 0; c < K; ++c) t[c] += v * TC(dj[c]);
                } else {
                    const TC *dj = d_old + int64_t(j) * K;
#pragma unroll
                    for (int c = 0; c < K; ++c) t[c] += v * dj[c];
                }
            }
        }
        for (int32_t p = rs + sub; p < re; p += LPR) {
            const int32_t j = col[p];
            if (j >= nv) break;  // columns ascend: the vertex block leads the row
            const TC v = FIRST ? val[p] * dinv[j] * inv_theta : val[p];
            if (FIRST) {
                const T *dj = r + int64_t(j) * K;
#pragma unroll
                for (int c = 0; c < K; ++c) t[c] += v * TC(dj[c]);
            } else {
                const TC *dj = d_old + int64_t(j) * K;
#pragma unroll
                for (int c = 0; c < K; ++c) t[c] += v * dj[c];
            }
        }
        TReduce<K, LPR>::run(t, sub);
        if (mine) {
            const TC dold = FIRST ? di * TC(rr) * inv_theta : dold_in;
            TC zi = FIRST ? dold : z_in + dold;
            const TC ri = (FIRST ? TC(rr) : res_in) - t[0];
            const TC dn = c1 * dold + c2 * di * ri;
            if (LAST == 2) zi += dn;
            if (LAST) z[at] = T(zi / di);   // LAST: stored pre-divided by dinv so the direction kernel treats it like r
            else {
                zc[at] = zi;
                res[at] = ri;
                d_new[at] = dn;
            }
            if (LAST) dot += double(rr) * double(zi);
        }
    }
    if (LAST) {
        __shared__ double smem[16 * K];
        double dcol[K];
#pragma unroll
        for (int c = 0; c < K; ++c) dcol[c] = (mine && c == mycol) ? dot : 0.0;
        block_sum<K>(dcol, smem);
#pragma unroll
        for (int c = 0; c < K; ++c)
            if (threadIdx.x == c) part[blockIdx.x * K + c] = dcol[c];
    }
}

template <class T, int K>
__global__ void __launch_bounds__(256) k_pcg_direction(int64_t n, int first, int step, double tol2, int nb_rz, ChebArgsT<T> ch,
                                                       const double *__restrict__ part_rz_new, double *__restrict__ scal,
                                                       const T *__restrict__ r, T *__restrict__ p,
                                                       const T *__restrict__ dinv, T *__restrict__ x = nullptr, int flat = 0) {
    __shared__ double smem[16 * K];
    if (solve_done(scal, step)) return;
    double beta[K], alpha[K];
#pragma unroll
    for (int c = 0; c < K; ++c) alpha[c] = 0.0;
    if (first) {  // p0 = C r0
#pragma unroll
        for (int c = 0; c < K; ++c) beta[c] = 0.0;
    } else {
        double rzn[K];
        reduce_partials<K>(part_rz_new, nb_rz, rzn, smem);
#pragma unroll
        for (int c = 0; c < K; ++c) {
            const double pq = scal[8 + c], rzo = scal[16 + 8 * (step & 1) + c];   // forwarded by the update launch
            const bool live = (rzo > tol2 * scal[c]) && (rzo > scal[kFloorSlot + c]) && (pq > 0.0);
            beta[c] = live ? rzn[c] / rzo : 0.0;
            alpha[c] = live ? rzo / pq : 0.0;      // the step's alpha, from the same operands as in its update launch: the same bits
        }
        if (blockIdx.x == 0 && threadIdx.x == 0)
#pragma unroll
            for (int c = 0; c < K; ++c) scal[16 + 8 * ((step + 1) & 1) + c] = rzn[c];   // read by the next update launch
    }
    if (flat) {
        // FLAT form (launcher: n K sizeof(T) below 4 GB): the vectors as arrays of n K values, 16 bytes per lane and access, consecutive
        // lanes on consecutive bytes - every load and store instruction of a wave covers 1 KB of whole lines (the row form's cover 1 KB
        // out of a 2.5 KB span, three instructions per line).  The column of a value is its index mod K: beta and alpha come from LDS
        // (an index into registers would put them into scratch memory); the Jacobi factor of its row is an 8-byte load (an L1 hit for
        // four of five).  Buffer accesses: the range check drops what lies behind the last value, dword by dword.
        constexpr int VEC = 16 / int(sizeof(T));
        constexpr uint32_t S = sizeof(T);
        constexpr int UF = 4;
        __syncthreads();          // (block_sum's last reads of smem)
        if (threadIdx.x == 0) {
#pragma unroll
            for (int c = 0; c < K; ++c) { smem[c] = beta[c]; smem[K + c] = alpha[c]; }
        }
        __syncthreads();
        const uint32_t N = uint32_t(n) * K, NV = uint32_t(ch.nv) * K;
        const rsrc_t rr = make_rsrc(r, uint64_t(N) * S), rz = make_rsrc(ch.z, uint64_t(NV) * S), rp = make_rsrc(p, uint64_t(N) * S),
                     rd = make_rsrc(dinv, uint64_t(n) * S), rx = make_rsrc(x ? x : p, uint64_t(N) * S);
        const bool with_x = x != nullptr && !first;
        const uint32_t span = uint32_t(gridDim.x) * blockDim.x * VEC;
        for (uint32_t g0 = (uint32_t(blockIdx.x) * blockDim.x + threadIdx.x) * VEC; g0 < N; g0 += UF * span) {
            T zv[UF][VEC], pv[UF][VEC], xv[UF][VEC], dv[UF][VEC];
#pragma unroll
            for (int u = 0; u < UF; ++u) {
                const uint32_t e0 = g0 + u * span;
                const uint32_t off = e0 < N ? e0 * S : kOutOfRange;
                buf_load<T, VEC>(rr, off, zv[u]);
                {                   // vertex rows: the vertex-block solver's result (stored pre-divided by the Jacobi factor) instead of r
                    T zz[VEC];      // (no branch: the other lanes hand over an offset out of range and send no request)
                    buf_load<T, VEC>(rz, e0 < NV ? e0 * S : kOutOfRange, zz);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) zv[u][v] = (e0 + v < NV) ? zz[v] : zv[u][v];
                }
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    T one[1];
                    buf_load<T, 1>(rd, e0 < N ? ((e0 + v) / uint32_t(K)) * S : kOutOfRange, one);
                    dv[u][v] = one[0];
                }
                if (!first) buf_load<T, VEC>(rp, off, pv[u]);
                else {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) pv[u][v] = T(0);
                }
                if (with_x) buf_load<T, VEC>(rx, off, xv[u]);
                else {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) xv[u][v] = T(0);
                }
            }
#pragma unroll
            for (int u = 0; u < UF; ++u) {
                const uint32_t e0 = g0 + u * span;
                const uint32_t off = e0 < N ? e0 * S : kOutOfRange;
                T pn[VEC], xn[VEC];
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const uint32_t c = (e0 + v) % uint32_t(K);
                    const T zi = dv[u][v] * zv[u][v];
                    pn[v] = first ? zi : zi + T(smem[c]) * pv[u][v];
                    xn[v] = xv[u][v] + T(smem[K + c]) * pv[u][v];
                }
                buf_store<T, VEC>(rp, off, pn);
                if (with_x) buf_store<T, VEC>(rx, off, xn);
            }
        }
        return;
    }
    // U rows per thread are loaded before any of them is stored: p is read and written through the same pointer, and a store
    // of one row otherwise holds back the loads of the next (one row in flight per thread: 2.7 TB/s at 5.4 M rows in fp32)
#ifndef REMO_DIR_U
#define REMO_DIR_U 4
#endif
    constexpr int U = REMO_DIR_U;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t i0 = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i0 < n; i0 += U * stride) {
        T d[U], zv[U][K], pv[U][K], xv[U][K];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * stride;
            d[u] = T(0);
#pragma unroll
            for (int c = 0; c < K; ++c) { zv[u][c] = T(0); pv[u][c] = T(0); xv[u][c] = T(0); }
            if (i < n) {
                d[u] = dinv[i];
                const T *src = (i < ch.nv) ? ch.z : r;   // C r: Chebyshev result on the vertex block (stored as z / dinv), Jacobi elsewhere
#pragma unroll
                for (int c = 0; c < K; ++c) zv[u][c] = src[i * K + c];
                if (!first)
#pragma unroll
                    for (int c = 0; c < K; ++c) pv[u][c] = p[i * K + c];
                if (x && !first)
#pragma unroll
                    for (int c = 0; c < K; ++c) xv[u][c] = x[i * K + c];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * stride;
            if (i < n)
#pragma unroll
                for (int c = 0; c < K; ++c) {
                    const T zi = d[u] * zv[u][c];
                    p[i * K + c] = first ? zi : zi + T(beta[c]) * pv[u][c];
                    // x += alpha p of THIS step, with the old direction that is in registers anyway (PcgBuffersT::x_in_direction):
                    // the update launch then neither reads p nor touches x - one pass over p less per step
                    if (x && !first) x[i * K + c] = xv[u][c] + T(alpha[c]) * pv[u][c];
                }
        }
    }
}

template <int K>
__global__ void __launch_bounds__(256) k_pcg_final(int step, int nb_rz, const double *__restrict__ part_rz, const double *__restrict__ scal,
                                                   PcgProgress *progress, int progress_len) {
    __shared__ double smem[16 * K];
    if (solve_done(scal, step)) return;   // the "done" record already holds the final <Cr,r>
    double rz[K];
    reduce_partials<K>(part_rz, nb_rz, rz, smem);
    if (threadIdx.x == 0) publish_progress(progress + (step % (progress_len - 1)), rz, K, step);
}

// Gershgorin bound of the Jacobi-scaled vertex block: max_i dinv_i * sum_j |a_ij|, j < nv
__global__ void __launch_bounds__(256) k_vblock_bound(int64_t nv, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                      const double *__restrict__ val, const double *__restrict__ dinv,
                                                      unsigned long long *out_bits) {
    double m = 0.0;
    for (int64_t row = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; row < nv; row += int64_t(gridDim.x) * blockDim.x) {
        double s = 0.0;
        for (int32_t p = rowptr[row]; p < rowptr[row + 1]; ++p) {
            if (col[p] >= nv) break;
            s += fabs(val[p]);
        }
        m = fmax(m, s * dinv[row]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out_bits, (unsigned long long)__double_as_longlong(m));  // positive doubles order like their bits
}

void launch_vblock_bound(int64_t nv, const CsrView &A, const double *dinv, unsigned long long *out_bits, hipStream_t s) {
    int64_t g = (nv + 255) / 256;
    if (g > 512) g = 512;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(k_vblock_bound, dim3(int(g)), dim3(256), 0, s, nv, A.rowptr, A.col, A.val, dinv, out_bits);
}

// ------------------------------------------------------------------------------------------
// Squared vertex block.  A Chebyshev step is a launch of ~5 us on a block of ~1e4 rows: pure launch latency, and
// there are 6 (3D) / 8 (2D) of them per PCG step.  Two consecutive Richardson factors of the same polynomial,
//   res'' = (I - b M)(I - a M) res = res - (a + b) M res + a b M^2 res,   M = A_vv D^-1,
// need M^2, i.e. B = A_vv D^-1 A_vv, once per matrix (a 2-hop pattern, ~65 entries per row in 3D): then ONE launch
// applies two factors, and the chain is half as long.  The polynomial is the same (product over the Chebyshev
// roots instead of the three-term recurrence), so the PCG iteration is unchanged up to rounding.
// One wave per row: distinct 2-hop columns through an LDS hash set, sorted (bitonic, so that the result does
// not depend on the insertion order), values by sorted-row lookups in a fixed order: bit-reproducible.

__device__ __forceinline__ int32_t lower_bound_col(const int32_t *__restrict__ col, int32_t lo, int32_t hi, int32_t key) {
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if (col[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

template <int PASS>   // 0: count the distinct columns of every row; 1: fill (row offsets known)
__global__ void __launch_bounds__(256) k_vblock_square(int64_t nv, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                       const double *__restrict__ val, const double *__restrict__ dinv,
                                                       int32_t *__restrict__ cnt, const int32_t *__restrict__ sq_rowptr,
                                                       int32_t *__restrict__ sq_col, double *__restrict__ sq_a, double *__restrict__ sq_b,
                                                       int64_t capacity, int32_t *flag) {
    __shared__ int32_t keys[4][kSquareSlots];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t row = int64_t(blockIdx.x) * 4 + wave;
    const bool active = row < nv;
    int32_t *K = keys[wave];
    for (int sl = lane; sl < kSquareSlots; sl += 64) K[sl] = INT_MAX;
    __syncthreads();
    int32_t rs = 0, re = 0;
    bool overflow = false;
    if (active) {
        rs = rowptr[row]; re = rowptr[row + 1];
        for (int32_t p = rs; p < re; ++p) {
            const int32_t k = col[p];
            if (k >= nv) break;                       // wave-uniform: the vertex block leads the row
            const int32_t ke = rowptr[k + 1];
            for (int32_t q = rowptr[k] + lane; q < ke; q += 64) {
                const int32_t j = col[q];
                if (j >= nv) break;
                uint32_t h = (uint32_t(j) * 2654435761u) >> 23;
                bool placed = false;
                for (int probe = 0; probe < kSquareSlots; ++probe) {
                    const int32_t old = atomicCAS(&K[h], INT_MAX, j);
                    if (old == INT_MAX || old == j) { placed = true; break; }
                    h = (h + 1) & (kSquareSlots - 1);
                }
                overflow |= !placed;
            }
        }
    }
    __syncthreads();
    int c = 0;
    for (int sl = lane; sl < kSquareSlots; sl += 64) c += K[sl] != INT_MAX;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
    if (__builtin_amdgcn_ballot_w64(overflow) != 0 || c > kSquareSlots - 64) {   // keep the table sparse enough to probe quickly
        if (lane == 0) atomicOr(flag, 1);
        c = 0;
    }
    if (PASS == 0) {
        if (active && lane == 0) cnt[row] = c;
        return;
    }
    // bitonic sort of the table (empty slots = INT_MAX end up last); every wave sorts its own table, the
    // barriers keep the four waves of the block in step
    for (int size = 2; size <= kSquareSlots; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = lane; t < kSquareSlots / 2; t += 64) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool up = (lo & size) == 0;
                const int32_t a = K[lo], b = K[hi];
                if ((a > b) == up) { K[lo] = b; K[hi] = a; }
            }
        }
    __syncthreads();
    if (!active || c == 0) return;
    const int64_t off = sq_rowptr[row];
    if (off + c > capacity) {
        if (lane == 0) atomicOr(flag, 2);
        return;
    }
    for (int t = lane; t < c; t += 64) {
        const int32_t j = K[t];
        double a = 0.0, b = 0.0;
        for (int32_t p = rs; p < re; ++p) {           // fixed order over the row's vertex entries
            const int32_t k = col[p];
            if (k >= nv) break;
            if (k == j) a = val[p];
            const int32_t ks = rowptr[k], ke = rowptr[k + 1];
            const int32_t q = lower_bound_col(col, ks, ke, j);
            if (q < ke && col[q] == j) b += val[p] * dinv[k] * val[q];
        }
        sq_col[off + t] = j;
        sq_a[off + t] = a;
        sq_b[off + t] = b;
    }
}

// exclusive scan of cnt[0 .. n) into out[0 .. n]; one workgroup (n is the vertex count)
__global__ void __launch_bounds__(1024) k_scan_counts(int64_t n, const int32_t *__restrict__ cnt, int32_t *__restrict__ out) {
    __shared__ int64_t part[1024];
    const int64_t chunk = (n + 1023) / 1024;
    const int64_t b = int64_t(threadIdx.x) * chunk, e = (b + chunk < n) ? b + chunk : n;
    int64_t sum = 0;
    for (int64_t i = b; i < e; ++i) sum += cnt[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t run = 0;
        for (int t = 0; t < 1024; ++t) { const int64_t v = part[t]; part[t] = run; run += v; }
        out[n] = int32_t(run < INT_MAX ? run : INT_MAX);
    }
    __syncthreads();
    int64_t run = part[threadIdx.x];
    for (int64_t i = b; i < e; ++i) { out[i] = int32_t(run < INT_MAX ? run : INT_MAX); run += cnt[i]; }
}

// compact copy of the vertex block: PASS 0 counts the leading entries (column < nv) of every vertex row, PASS 1 copies them
template <int PASS>
__global__ void __launch_bounds__(256) k_vblock_compact(int64_t nv, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                        const double *__restrict__ val, int32_t *__restrict__ cnt,
                                                        const int32_t *__restrict__ vb_rowptr, int32_t *__restrict__ vb_col,
                                                        double *__restrict__ vb_val, int64_t capacity, int32_t *flag) {
    const int64_t row = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (row >= nv) return;
    const int32_t rs = rowptr[row], re = rowptr[row + 1];
    if (PASS == 0) {
        cnt[row] = lower_bound_col(col, rs, re, int32_t(nv)) - rs;
    } else {
        const int32_t at = vb_rowptr[row], len = vb_rowptr[row + 1] - at;
        if (int64_t(at) + len > capacity) { if (len > 0) atomicOr(flag, 1); return; }
        for (int32_t e = 0; e < len; ++e) { vb_col[at + e] = col[rs + e]; vb_val[at + e] = val[rs + e]; }   // vertex rows are plain CSR
    }
}

void launch_vblock_compact(int64_t nv, const CsrView &A, int32_t *vb_rowptr, int32_t *vb_col, double *vb_val, int64_t capacity, int32_t *flag,
                           hipStream_t s) {
    if (nv <= 0) return;
    const int g = int((nv + 255) / 256);
    int32_t *cnt = vb_col;   // the counts live in the (not yet used) column array: capacity >= nv is required by the caller
    hipLaunchKernelGGL((k_vblock_compact<0>), dim3(g), dim3(256), 0, s, nv, A.rowptr, A.col, A.val, cnt, (const int32_t *)nullptr, (int32_t *)nullptr,
                       (double *)nullptr, capacity, flag);
    hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, s, nv, cnt, vb_rowptr);
    hipLaunchKernelGGL((k_vblock_compact<1>), dim3(g), dim3(256), 0, s, nv, A.rowptr, A.col, A.val, (int32_t *)nullptr, vb_rowptr, vb_col, vb_val,
                       capacity, flag);
}

// ---- grid barrier (probe and, if it pays, the Chebyshev chain as one launch) ---------------------------------------------------
// Every workgroup of a launch whose workgroups are all resident (grid <= what the chip holds at once: the caller's duty) arrives
// at a counter that only grows - arrival e of nblocks workgroups waits for e * nblocks - and leaves when all have.  Data that
// crosses the barrier is written and read with agent-scope accesses (the eight L2s of the chip are not coherent with each other
// for plain ones).  A wait gives up after kBarrierSpins polls and raises *fail: every wave reaches the end of the kernel.
constexpr long long kBarrierSpins = 1ll << 22;
__device__ __forceinline__ bool grid_barrier(unsigned *counter, unsigned nblocks, unsigned &epoch, int *fail) {
    __shared__ int ok;
    __syncthreads();
    if (threadIdx.x == 0) {
        ++epoch;
        const unsigned target = epoch * nblocks;
        __atomic_thread_fence(__ATOMIC_RELEASE);   // (agent scope is HIP's default for the builtin)
        (void)__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long long spins = 0;
        int good = 1;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > kBarrierSpins || __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { good = 0; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (!good) __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        ok = good;
    }
    __syncthreads();
    return ok != 0;
}

__global__ void __launch_bounds__(256) k_barrier_probe(unsigned nblocks, int nbar, unsigned *counter, int *fail, float *buf, int *mismatch) {
    unsigned epoch = 0;
    const unsigned other = (blockIdx.x + 37u) % nblocks;
    int bad = 0;
    for (int it = 0; it < nbar; ++it) {
        __hip_atomic_store(buf + size_t(blockIdx.x) * 256 + threadIdx.x, float(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!grid_barrier(counter, nblocks, epoch, fail)) return;
        const float v = __hip_atomic_load(buf + size_t(other) * 256 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bad += (v != float(it + 1)) ? 1 : 0;
        if (!grid_barrier(counter, nblocks, epoch, fail)) return;     // nobody overwrites what a neighbour still reads
    }
    if (bad) atomicAdd(mismatch, bad);
}

void launch_barrier_probe(int nblocks, int nbar, unsigned *counter, int *fail, float *buf, int *mismatch, hipStream_t s) {
    hipLaunchKernelGGL(k_barrier_probe, dim3(nblocks), dim3(256), 0, s, unsigned(nblocks), nbar, counter, fail, buf, mismatch);
}

// Fixed-width image of the vertex block for the Chebyshev launches.  A launch on the CSR form is a chain of dependent round
// trips per row - row bounds, then columns and values (once per eight entries), then the gathered vector rows - on a block that
// lives in the caches: latency, not bytes.  Here the first kEllWidth entries of row i sit at [i][0 .. kEllWidth) (padded with
// (i, 0)), so a lane asks for all its columns and values at once, then for all the vector rows: two trips.  tail[i] = the
// begin / end of the row's remaining entries in the arrays the image was made from (equal: none - the usual case).
__global__ void __launch_bounds__(256) k_vblock_ell(int64_t nv, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                    const double *__restrict__ val, int32_t *__restrict__ ecol, int32_t *__restrict__ tail,
                                                    double *__restrict__ eval64, float *__restrict__ eval32) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= nv * kEllWidth) return;
    const int64_t row = i / kEllWidth;
    const int e = int(i - row * kEllWidth);
    const int32_t rs = rowptr[row], re = rowptr[row + 1];
    int32_t c = int32_t(row);
    double v = 0.0;
    if (rs + e < re) {
        const int32_t j = col[rs + e];
        if (j < nv) { c = j; v = val[rs + e]; }     // columns ascend: the vertex block leads a row of the whole matrix
    }
    ecol[i] = c;
    if (eval64) eval64[i] = v;
    if (eval32) eval32[i] = float(v);
    if (e == 0) {
        const int32_t end = lower_bound_col(col, rs, re, int32_t(nv));
        const bool more = end - rs > kEllWidth;
        tail[2 * row] = more ? rs + kEllWidth : 0;
        tail[2 * row + 1] = more ? end : 0;
    }
}

void launch_vblock_ell(int64_t nv, const int32_t *rowptr, const int32_t *col, const double *val, int32_t *ecol, int32_t *tail, double *eval64,
                       float *eval32, hipStream_t s) {
    if (nv <= 0) return;
    hipLaunchKernelGGL(k_vblock_ell, dim3(int((nv * kEllWidth + 255) / 256)), dim3(256), 0, s, nv, rowptr, col, val, ecol, tail, eval64, eval32);
}

void launch_vblock_square(int64_t nv, const CsrView &A, const double *dinv, int32_t *sq_rowptr, int32_t *sq_col, double *sq_a, double *sq_b,
                          int64_t capacity, int32_t *flag, hipStream_t s) {
    if (nv <= 0) return;
    const int g = int((nv + 3) / 4);
    int32_t *cnt = sq_col;   // the counts live in the (not yet used) column array: capacity >= nv is required by the caller
    hipLaunchKernelGGL((k_vblock_square<0>), dim3(g), dim3(256), 0, s, nv, A.rowptr, A.col, A.val, dinv, cnt, (const int32_t *)nullptr, (int32_t *)nullptr,
                       (double *)nullptr, (double *)nullptr, capacity, flag);
    hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, s, nv, cnt, sq_rowptr);
    hipLaunchKernelGGL((k_vblock_square<1>), dim3(g), dim3(256), 0, s, nv, A.rowptr, A.col, A.val, dinv, (int32_t *)nullptr, sq_rowptr, sq_col, sq_a, sq_b,
                       capacity, flag);
}

// Two Chebyshev (Richardson) factors per launch, 16 lanes per row of B's pattern.  State: z and w = D^-1 res.
//   t1 = A w, t2 = B w;   z += (a + b) w - a b D^-1 t1;   w' = w - (a + b) D^-1 t1 + a b D^-1 t2
// FIRST: z = 0, w = D^-1 r formed on the fly; LAST: z is stored divided by dinv (the direction kernel treats it like
// r) and the <r, z> partial sums are left for the PCG scalars.
template <class T, int K, int LPR, bool FIRST, bool LAST>
__global__ void __launch_bounds__(256) k_cheb_pair(int64_t nv, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                   const T *__restrict__ va, const T *__restrict__ vb, const T *__restrict__ dinv,
                                                   const T *__restrict__ w_old, T *__restrict__ w_new, T *__restrict__ z, double ab_sum_, double ab_prod_,
                                                   const T *__restrict__ r, double *__restrict__ part, const double *__restrict__ scal, int step) {
    constexpr int RPB = 256 / LPR, U = 2;
    static_assert(K <= LPR, "one column per lane after the transposing reduction");
    if (solve_done(scal, step)) return;
    const T ab_sum = T(ab_sum_), ab_prod = T(ab_prod_);
    const int sub = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    int idx[K], own[K];
    towner_init<K, LPR>(idx, own, sub);
    const int mycol = idx[0];
    const bool mine = own[0] != 0;
    double dot = 0.0;
    for (int64_t row = int64_t(blockIdx.x) * RPB + grp; row < nv; row += int64_t(gridDim.x) * RPB) {
        const int32_t rs = rowptr[row], re = rowptr[row + 1];
        const int64_t at = row * K + mycol;
        T di = T(0), wi = T(0), zi = T(0), rr = T(0);
        if (mine) {
            di = dinv[row];
            if (FIRST || LAST) rr = r[at];
            wi = FIRST ? di * rr : w_old[at];
            if (!FIRST) zi = z[at];
        }
        T t1[K], t2[K];
#pragma unroll
        for (int c = 0; c < K; ++c) { t1[c] = T(0); t2[c] = T(0); }
        for (int32_t p0 = rs + sub; p0 < re; p0 += U * LPR) {   // U passes of loads in flight, as in the SpMM
            int32_t j[U];
            T a[U], b[U], w[U][K];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int32_t p = p0 + u * LPR;
                j[u] = -1; a[u] = T(0); b[u] = T(0);
                if (p < re) { j[u] = col[p]; a[u] = va[p]; b[u] = vb[p]; }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int c = 0; c < K; ++c) w[u][c] = T(0);
                if (j[u] >= 0) {
                    const T *wj = (FIRST ? r : w_old) + int64_t(j[u]) * K;
                    const T dj = FIRST ? dinv[j[u]] : T(1);
#pragma unroll
                    for (int c = 0; c < K; ++c) w[u][c] = FIRST ? dj * wj[c] : wj[c];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int c = 0; c < K; ++c) { t1[c] += a[u] * w[u][c]; t2[c] += b[u] * w[u][c]; }
        }
        TReduce<K, LPR>::run(t1, sub);
        TReduce<K, LPR>::run(t2, sub);
        if (mine) {
            const T zn = zi + ab_sum * wi - ab_prod * di * t1[0];
            z[at] = LAST ? zn / di : zn;
            if (!LAST) w_new[at] = wi - ab_sum * di * t1[0] + ab_prod * di * t2[0];
            if (LAST) dot += double(rr) * double(zn);
        }
    }
    if (LAST) {
        __shared__ double smem[16 * K];
        double dcol[K];
#pragma unroll
        for (int c = 0; c < K; ++c) dcol[c] = (mine && c == mycol) ? dot : 0.0;
        block_sum<K>(dcol, smem);
#pragma unroll
        for (int c = 0; c < K; ++c)
            if (threadIdx.x == c) part[blockIdx.x * K + c] = dcol[c];
    }
}

int g_slab_ahead = 1;      // remo_debug_tune key 27: 0 = the update launch walks the slab slots of a shared row one by one
int g_slab_masked = 1;   // remo_debug_tune key 29: 0 = every row fetches four slab slots and weights the ones it does not have by zero (the form before)
int g_flat_direction = 1;   // remo_debug_tune key 30: 1 (default) = the direction launch walks its vectors as flat arrays, 16 bytes per lane; 0 = a k-wide row per lane
int g_tile_update = 1;      // remo_debug_tune key 31: 1 (default, fp64 storage) = the update launch takes 64 rows per wave, a value per lane and pass (k_pcg_update, tile form)
void set_tile_update(int v) { g_tile_update = v ? 1 : 0; }
void set_flat_direction(int v) { g_flat_direction = v ? 1 : 0; }
void set_slab_masked(int v) { g_slab_masked = v ? 1 : 0; }
void set_slab_ahead(int v) { g_slab_ahead = v ? 1 : 0; }
int vec_grid(int64_t n) {
    int64_t g = (n + 255) / 256;
    if (g > kMaxPartialBlocks / 2) g = kMaxPartialBlocks / 2;
    if (g < 1) g = 1;
    return int(g);
}

int cheb_grid(int64_t nv) {
    if (nv <= 0) return 0;
    int64_t g = (nv + 31) / 32;
    if (g > kMaxPartialBlocks / 2) g = kMaxPartialBlocks / 2;
    return int(g);
}
#define REMO_K_SWITCH(k, CALL) \
    switch (k) {               \
        case 1: { constexpr int KK = 1; CALL; } break; \
        case 2: { constexpr int KK = 2; CALL; } break; \
        case 3: { constexpr int KK = 3; CALL; } break; \
        case 4: { constexpr int KK = 4; CALL; } break; \
        case 5: { constexpr int KK = 5; CALL; } break; \
        case 6: { constexpr int KK = 6; CALL; } break; \
        case 7: { constexpr int KK = 7; CALL; } break; \
        default: { constexpr int KK = 8; CALL; } break; \
    }

template <class T> static ChebArgsT<T> cheb_args(const PcgBuffersT<T> &b) {
    ChebArgsT<T> c;
    c.nv = b.cheb_degree > 0 ? b.nv_coarse : 0;
    c.inv_theta = b.cheb_degree > 0 ? 1.0 / (0.5 * (b.cheb_lmax + b.cheb_lmin)) : 0.0;
    c.z = b.cz; c.res = b.cres; c.d0 = b.cd[0];
    return c;
}

// C r for the vertex block: `degree` Chebyshev steps; the last one leaves the <r_v, z_v> partials
// behind the nb_vec partials of the high-order part (slot = even / odd step buffer)
// the update launch can take the FIRST step along when the polynomial has launches of its own left to commit the
// vertex residual (degree >= 3) and is not applied through the squared block (2D)
template <class T> static bool cheb_first_folds(const PcgBuffersT<T> &b);
template <class T> bool pcg_update_folds(const PcgBuffersT<T> &b) { return cheb_first_folds(b); }
template bool pcg_update_folds<double>(const PcgBuffersT<double> &);
template bool pcg_update_folds<float>(const PcgBuffersT<float> &);
template <class T> static bool cheb_first_folds(const PcgBuffersT<T> &b) {
    // measured in the bench, fold on vs off on one box: -2.3 % solve time at 12.8 k vertices, -0.9 % at 27 k, +0.4 % at 83 k
    // (there the step is real work, not launch latency): small vertex blocks only
    return g_fold_first && !b.amg && b.cheb_degree >= 3 && b.nv_coarse > 0 && b.nv_coarse <= 32768 && !(b.sq_rowptr && (b.cheb_degree & 1) == 0);
}

template <class T> static void launch_cheb(const CsrViewT<T> &A, int k, int step, const PcgBuffersT<T> &b, double *part_slot, hipStream_t s, bool first_done = false) {
    if (b.cheb_degree <= 0 || b.nv_coarse <= 0) return;
    if (b.amg) {   // multigrid cycle instead of the polynomial (amg.hip)
        if constexpr (sizeof(T) == 8) {
            if (b.amg32) {
                launch_amg_cycle<float, double>(*b.amg32, k, step, (const double *)b.r, b.cz, part_slot + int64_t(b.nb_vec) * k, cheb_grid(b.nv_coarse), (const double *)b.rz0, s);
                return;
            }
        }
        launch_amg_cycle<T, T>(*b.amg, k, step, (const T *)b.r, b.cz, part_slot + int64_t(b.nb_vec) * k, cheb_grid(b.nv_coarse), (const double *)b.rz0, s);
        return;
    }
    if (b.sq_rowptr && (b.cheb_degree & 1) == 0) {   // two Richardson factors of the Chebyshev polynomial per launch
        const double theta = 0.5 * (b.cheb_lmax + b.cheb_lmin), delta = 0.5 * (b.cheb_lmax - b.cheb_lmin);
        const int m = b.cheb_degree, np = m / 2;
        // 3D rows of B hold ~65 entries (32 lanes per row), 2D rows ~19 (8 lanes)
        const int g_last = cheb_grid(b.nv_coarse);
        double *part = part_slot + int64_t(b.nb_vec) * k;
        for (int j = 0; j < np; ++j) {
            // roots of the shifted Chebyshev polynomial, paired from the two ends of the interval inwards
            const double r1 = theta - delta * cos(M_PI * (2.0 * (j + 1) - 1.0) / (2.0 * m));
            const double r2 = theta - delta * cos(M_PI * (2.0 * (m - j) - 1.0) / (2.0 * m));
            const double a = 1.0 / r1, bb = 1.0 / r2;
            const T *wold = b.cd[j & 1];
            T *wnew = b.cd[(j + 1) & 1];
            const bool first = (j == 0), last = (j + 1 == np);
            // only the LAST launch leaves partial sums, so only it is tied to the cheb_grid slots
            auto grid_for = [&](int lpr) { int64_t gg = (b.nv_coarse + 256 / lpr - 1) / (256 / lpr); if (last && gg > g_last) gg = g_last; if (gg > 4096) gg = 4096; return int(gg); };
#define REMO_CHEB2(LPRV, F, L)                                                                                                                        \
    REMO_K_SWITCH(k, hipLaunchKernelGGL((k_cheb_pair<T, KK, LPRV, F, L>), dim3(grid_for(LPRV)), dim3(256), 0, s, b.nv_coarse, b.sq_rowptr, b.sq_col, b.sq_a, \
                                        b.sq_b, b.dinv, wold, wnew, b.cz, a + bb, a * bb, b.r, part, b.rz0, step))
#define REMO_CHEB2_FL(LPRV)                                 \
    if (first && last) { REMO_CHEB2(LPRV, true, true); }    \
    else if (first) { REMO_CHEB2(LPRV, true, false); }      \
    else if (last) { REMO_CHEB2(LPRV, false, true); }       \
    else { REMO_CHEB2(LPRV, false, false); }
            if (b.sq_lanes >= 32) { REMO_CHEB2_FL(32) }
            else if (b.sq_lanes >= 16) { REMO_CHEB2_FL(16) }
            else { REMO_CHEB2_FL(8) }
#undef REMO_CHEB2_FL
#undef REMO_CHEB2
        }
        return;
    }
    const double theta = 0.5 * (b.cheb_lmax + b.cheb_lmin), delta = 0.5 * (b.cheb_lmax - b.cheb_lmin);
    const double sig = theta / delta, inv_theta = 1.0 / theta;
    double rho = 1.0 / sig;
    const int g = cheb_grid(b.nv_coarse);
    double *part = part_slot + int64_t(b.nb_vec) * k;
    const int launches = b.cheb_degree > 1 ? b.cheb_degree - 1 : 1;   // the last term rides on the launch before it
    const int32_t *vrow = b.vb_rowptr ? b.vb_rowptr : A.rowptr, *vcol = b.vb_rowptr ? b.vb_col : A.col;   // compact vertex block if there is one
    const T *vval = b.vb_rowptr ? b.vb_val : A.val;
    // fp32 chain inside an fp64 solve: needs the compact block (its float copy), never together with the folded first step
    const bool chain32 = sizeof(T) == 8 && b.c32_val != nullptr && b.vb_rowptr != nullptr && !first_done;
    // the fixed-width image of the block (k_vblock_ell) in the storage type of the chain, if the host made one
    const bool ell = b.ell_col != nullptr && b.ell_tail != nullptr && (chain32 ? b.c32_ell_val != nullptr : b.ell_val != nullptr);
    for (int j = 0; j < launches; ++j) {
        const double rho_new = 1.0 / (2.0 * sig - rho);
        const double c1 = rho_new * rho, c2 = 2.0 * rho_new / delta;
        rho = rho_new;
        const T *dold = b.cd[j & 1];
        T *dnew = b.cd[(j + 1) & 1];
        const bool first = (j == 0), last = (j + 1 == launches);
        if (first && first_done) continue;        // k_pcg_update did it (with the c1, c2 of cheb_first_coefficients)
        const int commit = (first_done && j == 1) ? 1 : 0;
        // only the launch that leaves partial sums is tied to the cheb_grid slots; the others take one row group per
        // workgroup slot (at 83 k vertices 512 workgroups walk five row groups each, a chain of five dependent round trips:
        // 17.9 us per launch under rocprofv3, profiles/r01_f_kernel_stats_sizeL_10depths_before_grid_fix.csv; after: 14.2 us, …_after_grid_fix.csv)
        const int64_t g_rows = (b.nv_coarse + 31) / 32;
        const int gl = last ? g : int(g_rows < 8192 ? g_rows : 8192);
#define REMO_CHEB_E(F, L, E)                                                                                                                        \
    if (chain32) {                                                                                                                                  \
        REMO_K_SWITCH(k, hipLaunchKernelGGL((k_cheb_step<T, float, KK, F, L, E>), dim3(gl), dim3(256), 0, s, b.nv_coarse, E ? b.ell_tail : vrow, vcol, b.c32_val, b.c32_dinv, \
                                            (const float *)b.c32_d[j & 1], b.c32_d[(j + 1) & 1], b.c32_z, b.c32_res, b.cz, c1, c2, inv_theta, b.r, part, b.rz0, 0, step, \
                                            b.ell_col, b.c32_ell_val));                                                                             \
    } else {                                                                                                                                        \
        REMO_K_SWITCH(k, hipLaunchKernelGGL((k_cheb_step<T, T, KK, F, L, E>), dim3(gl), dim3(256), 0, s, b.nv_coarse, E ? b.ell_tail : vrow, vcol, vval, b.dinv, dold, dnew, \
                                            b.cz, b.cres, b.cz, c1, c2, inv_theta, b.r, part, b.rz0, commit, step, b.ell_col, b.ell_val));               \
    }
#define REMO_CHEB(F, L)                \
    if (ell) { REMO_CHEB_E(F, L, true) } \
    else { REMO_CHEB_E(F, L, false) }
        if (first && last) { if (b.cheb_degree == 1) { REMO_CHEB(true, 1); } else { REMO_CHEB(true, 2); } }
        else if (first) { REMO_CHEB(true, 0); }
        else if (last) { REMO_CHEB(false, 2); }
        else { REMO_CHEB(false, 0); }
#undef REMO_CHEB
#undef REMO_CHEB_E
    }
}

template <class T> static int nb_rz(const PcgBuffersT<T> &b) { return b.nb_vec + ((b.cheb_degree > 0 && b.nv_coarse > 0) ? cheb_grid(b.nv_coarse) : 0); }

template <class T> void launch_pcg_init(const CsrViewT<T> &A, int k, const T *f, const PcgBuffersT<T> &b, hipStream_t s) {
    const int64_t n = A.n;
    const int g = b.nb_vec;
    const ChebArgsT<T> ch = cheb_args(b);
    REMO_K_SWITCH(k, hipLaunchKernelGGL((k_pcg_init<T, KK>), dim3(g), dim3(256), 0, s, n, ch, f, b.dinv, b.x, b.r, b.p, b.part_rz));
    launch_cheb(A, k, 0, b, b.part_rz, s);
    if (ch.nv > 0)
        REMO_K_SWITCH(k, hipLaunchKernelGGL((k_pcg_direction<T, KK>), dim3(g), dim3(256), 0, s, n, 1, 0, 0.0, nb_rz(b), ch, b.part_rz, b.rz0, b.r, b.p, b.dinv));
}

template <class T> void launch_pcg_update(const CsrViewT<T> &A, int k, int step, double tol2, const PcgBuffersT<T> &b, hipStream_t s) {
    const int64_t n = A.n;
    const int g = b.nb_vec;
    double *cur = b.part_rz + (step & 1) * (kMaxPartialBlocks * 8);
    double *nxt = b.part_rz + ((step + 1) & 1) * (kMaxPartialBlocks * 8);
    const ChebArgsT<T> ch = cheb_args(b);
    FoldArgsT<T> fold;
    int grid = g;
    const bool folded = cheb_first_folds(b);
    if (folded) {
        const double theta = 0.5 * (b.cheb_lmax + b.cheb_lmin), delta = 0.5 * (b.cheb_lmax - b.cheb_lmin);
        const double sig = theta / delta, rho = 1.0 / sig, rho_new = 1.0 / (2.0 * sig - rho);
        fold.nb_flat = g;
        fold.rowptr = b.vb_rowptr ? b.vb_rowptr : A.rowptr; fold.col = b.vb_rowptr ? b.vb_col : A.col; fold.val = b.vb_rowptr ? b.vb_val : A.val;
        fold.d_new = b.cd[1]; fold.stage = b.cd[0];
        fold.c1 = rho_new * rho; fold.c2 = 2.0 * rho_new / delta;      // the j = 0 coefficients of launch_cheb
        grid = g + int((b.nv_coarse + 31) / 32);   // the vertex workgroups leave no partial sums: one row group each
    }
    QViewT<T> qv;
    if (b.defer_q && A.patch && !folded) { qv.bptr = A.patch->t.bptr; qv.bslot = A.patch->t.bslot; qv.Yb = A.patch->Yb; qv.ahead = g_slab_ahead; }
    qv.skip_x = b.x_in_direction ? 1 : 0;
    if (qv.bptr && g_slab_masked) {
        const uint64_t bytes = uint64_t(A.patch->t.nslot_cap) * uint64_t(k) * sizeof(T);
        qv.slab_bytes = bytes < 0xFFFFF000ull ? bytes : 0;
        qv.row4 = A.patch->t.row4;
        qv.tile = (g_tile_update && sizeof(T) == 8 && qv.slab_bytes && qv.row4 && qv.skip_x &&     // (fp32 storage: 256 bytes per wave and access - the row form is ahead there, 74.7 against 76.3 ms)
                   uint64_t(n) * uint64_t(k) * sizeof(T) < 0xFFFFF000ull &&
                   uint64_t(n) * 16 < 0xFFFFF000ull) ? 1 : 0;
    }
    const bool bins = b.pq_bins && b.defer_q && A.patch && !folded;
    const double *pq_rows = bins ? b.part_pq + (step & 1) * (kPqBins * 8) : b.part_pq;
    double *pq_clear = bins ? b.part_pq + ((step + 1) & 1) * (kPqBins * 8) : nullptr;
    REMO_K_SWITCH(k, hipLaunchKernelGGL((k_pcg_update<T, KK>), dim3(grid), dim3(256), 0, s, n, step, tol2, 0, bins ? kPqBins : b.nb_spmv, nb_rz(b), ch, pq_rows, cur, nxt,
                                        b.rz0, b.progress, b.progress_len, b.p, b.q, b.x, b.r, b.dinv, fold, qv, pq_clear));
    launch_cheb(A, k, step, b, nxt, s, folded);
}

// Residual replacement of the mixed mode, in place of launch_pcg_update at the chosen steps:
//   x32 += alpha p;  x64 += x32, x32 = 0;  r32 = float(f - A64 x64);  C r;  <Cr,r> partials
// The search direction and the scalars carry on, so the Krylov process is not restarted; what is
// removed is the drift of the fp32 recurrence residual from the true one.
template <int K>
__global__ void __launch_bounds__(256) k_mixed_replace(int64_t n, int64_t nv, const double *__restrict__ f, const double *__restrict__ q64,
                                                       float *__restrict__ r, const float *__restrict__ dinv, double *__restrict__ part_rz_next,
                                                       const double *__restrict__ scal, int step) {
    __shared__ double smem[16 * K];
    if (solve_done(scal, step)) return;
    double acc[K];
#pragma unroll
    for (int c = 0; c < K; ++c) acc[c] = 0.0;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
        const float d = dinv[i];
        const bool coarse = i < nv;
#pragma unroll
        for (int c = 0; c < K; ++c) {
            const float ri = float(f[i * K + c] - q64[i * K + c]);
            r[i * K + c] = ri;
            acc[c] += coarse ? 0.0 : double(ri) * double(ri) * double(d);
        }
    }
    block_sum<K>(acc, smem);
    if (threadIdx.x < K) part_rz_next[blockIdx.x * K + threadIdx.x] = acc[threadIdx.x];
}

void launch_pcg_replace(const CsrViewT<float> &A, const CsrViewT<double> &A64, int k, int step, double tol2, const PcgBuffersT<float> &b,
                        const double *f64, double *x64, double *q64, hipStream_t s) {
    const int64_t n = A.n;
    const int g = b.nb_vec;
    double *cur = b.part_rz + (step & 1) * (kMaxPartialBlocks * 8);
    double *nxt = b.part_rz + ((step + 1) & 1) * (kMaxPartialBlocks * 8);
    const ChebArgsT<float> ch = cheb_args(b);
    const bool bins = b.pq_bins && b.defer_q && A.patch && !pcg_update_folds(b);
    const double *pq_rows = bins ? b.part_pq + (step & 1) * (kPqBins * 8) : b.part_pq;
    double *pq_clear = bins ? b.part_pq + ((step + 1) & 1) * (kPqBins * 8) : nullptr;
    REMO_K_SWITCH(k, hipLaunchKernelGGL((k_pcg_update<float, KK>), dim3(g), dim3(256), 0, s, n, step, tol2, 1, bins ? kPqBins : b.nb_spmv, nb_rz(b), ch, pq_rows, cur, nxt,
                                        b.rz0, b.progress, b.progress_len, b.p, b.q, b.x, b.r, b.dinv, FoldArgsT<float>(), QViewT<float>(), pq_clear));
    launch_mixed_accumulate(n * k, x64, b.x, 1, s);
    launch_spmm(A64, k, (const double *)x64, q64, (double *)nullptr, (const double *)nullptr, b.nb_spmv, s, 0);
    REMO_K_SWITCH(k, hipLaunchKernelGGL((k_mixed_replace<KK>), dim3(g), dim3(256), 0, s, n, ch.nv, f64, q64, b.r, b.dinv, nxt, b.rz0, step));
    launch_cheb(A, k, step, b, nxt, s);
}

// add_x: this launch also forms x += alpha p of the step (b.x_in_direction and the step's update launch left x alone; a residual
// replacement of the mixed mode updates x itself)
template <class T> void launch_pcg_direction(const CsrViewT<T> &A, int k, int step, double tol2, const PcgBuffersT<T> &b, hipStream_t s, bool add_x) {
    const int64_t n = A.n;
    const int g = b.nb_vec;
    const double *nw = b.part_rz + ((step + 1) & 1) * (kMaxPartialBlocks * 8);
    const ChebArgsT<T> ch = cheb_args(b);
    T *xp = (add_x && b.x_in_direction) ? b.x : nullptr;
    const int flat = (g_flat_direction && uint64_t(n) * uint64_t(k) * sizeof(T) < 0xFFFFF000ull) ? 1 : 0;
    REMO_K_SWITCH(k, hipLaunchKernelGGL((k_pcg_direction<T, KK>), dim3(g), dim3(256), 0, s, n, 0, step, tol2, nb_rz(b), ch, nw, b.rz0, b.r, b.p, b.dinv, xp, flat));
}

template <class T> void launch_pcg_final(int k, int step, const PcgBuffersT<T> &b, hipStream_t s) {
    const double *cur = b.part_rz + (step & 1) * (kMaxPartialBlocks * 8);
    REMO_K_SWITCH(k, hipLaunchKernelGGL(k_pcg_final<KK>, dim3(1), dim3(256), 0, s, step, nb_rz(b), cur, b.rz0, b.progress, b.progress_len));
}

#define REMO_INSTANTIATE_PCG(T)                                                                                          \
    template void launch_pcg_init<T>(const CsrViewT<T> &, int, const T *, const PcgBuffersT<T> &, hipStream_t);          \
    template void launch_pcg_update<T>(const CsrViewT<T> &, int, int, double, const PcgBuffersT<T> &, hipStream_t);      \
    template void launch_pcg_direction<T>(const CsrViewT<T> &, int, int, double, const PcgBuffersT<T> &, hipStream_t, bool);   \
    template void launch_pcg_final<T>(int, int, const PcgBuffersT<T> &, hipStream_t);
REMO_INSTANTIATE_PCG(double)
REMO_INSTANTIATE_PCG(float)
#undef REMO_INSTANTIATE_PCG

// ------------------------------------------------------------------------------------------
// mixed precision (BASELINE config 5): fp32 inner PCG, fp64 residual refinement.  The outer loop is
//   r = f - A x (fp64 SpMM)  ->  r32 = (float) r  ->  inner PCG on A32 e = r32  ->  x += e

__global__ void __launch_bounds__(256) k_to_float(int64_t n, const double *__restrict__ src, float *__restrict__ dst) {
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) dst[i] = float(src[i]);
}
// r32 = float(f - q)   (q = nullptr: x = 0, r = f)
__global__ void __launch_bounds__(256) k_mixed_residual(int64_t n, const double *__restrict__ f, const double *__restrict__ q, float *__restrict__ r32) {
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x)
        r32[i] = float(q ? f[i] - q[i] : f[i]);
}
// x += e (and e = 0 when the inner solve carries on)
__global__ void __launch_bounds__(256) k_mixed_accumulate(int64_t n, double *__restrict__ x, float *__restrict__ e, int zero) {
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
        x[i] += double(e[i]);
        if (zero) e[i] = 0.f;
    }
}
static int stream_grid(int64_t n) {
    int64_t g = (n + 255) / 256;
    if (g > 2048) g = 2048;
    return int(g < 1 ? 1 : g);
}
void launch_to_float(int64_t n, const double *src, float *dst, hipStream_t s) {
    hipLaunchKernelGGL(k_to_float, dim3(stream_grid(n)), dim3(256), 0, s, n, src, dst);
}
void launch_mixed_residual(int64_t n, const double *f, const double *q, float *r32, hipStream_t s) {
    hipLaunchKernelGGL(k_mixed_residual, dim3(stream_grid(n)), dim3(256), 0, s, n, f, q, r32);
}
void launch_mixed_accumulate(int64_t n, double *x, float *e, int zero, hipStream_t s) {
    hipLaunchKernelGGL(k_mixed_accumulate, dim3(stream_grid(n)), dim3(256), 0, s, n, x, e, zero);
}

// ------------------------------------------------------------------------------------------
// axis points: mesh(0, z) / mesh(0, 0, z) point location, AddPointSource and gfu(point)
// (ngsolve_functions.py:10-21, worker.py:124-131).  All points lie on the borehole axis.

template <int DIM>
__global__ void __launch_bounds__(256) k_locate(int64_t nt, const double *__restrict__ coords, const int32_t *__restrict__ conn,
                                                int npts, const double *__restrict__ pz, int32_t *found) {
    constexpr int NB = DIM + 1;
    const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    double X[NB * DIM];
    double lo[DIM], hi[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) { lo[k] = 1e300; hi[k] = -1e300; }
#pragma unroll
    for (int a = 0; a < NB; ++a) {
        const int64_t v = conn[t * NB + a];
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
            const double c = coords[v * DIM + k];
            X[a * DIM + k] = c;
            lo[k] = fmin(lo[k], c);
            hi[k] = fmax(hi[k], c);
        }
    }
    // the axis is x = 0 (2D) / x = y = 0 (3D): reject elements that do not touch it
    bool touch = true;
#pragma unroll
    for (int k = 0; k < DIM - 1; ++k) {
        const double ext = 1e-9 * (1.0 + hi[k] - lo[k]);
        if (lo[k] > ext || hi[k] < -ext) touch = false;
    }
    if (!touch) return;
    const double extz = 1e-9 * (1.0 + hi[DIM - 1] - lo[DIM - 1]);
    for (int q = 0; q < npts; ++q) {
        const double z = pz[q];
        if (z < lo[DIM - 1] - extz || z > hi[DIM - 1] + extz) continue;
        double P[DIM], l[NB];
#pragma unroll
        for (int k = 0; k < DIM; ++k) P[k] = 0.0;
        P[DIM - 1] = z;
        if (!barycentrics<DIM>(X, P, l)) continue;
        bool in = true;
#pragma unroll
        for (int a = 0; a < NB; ++a)
            if (l[a] < -1e-10) in = false;
        if (in) atomicMin(&found[q], int32_t(t));  // lowest element number: deterministic choice
    }
}

void launch_locate(int dim, int64_t nt, const double *coords, const int32_t *conn, int npts, const double *pz, int32_t *found,
                   hipStream_t s) {
    const int grid = int((nt + 255) / 256);
    if (dim == 2)
        hipLaunchKernelGGL(k_locate<2>, dim3(grid), dim3(256), 0, s, nt, coords, conn, npts, pz, found);
    else
        hipLaunchKernelGGL(k_locate<3>, dim3(grid), dim3(256), 0, s, nt, coords, conn, npts, pz, found);
}

template <int DIM>
__global__ void k_point_shapes(int npts, const double *__restrict__ pz, const int32_t *__restrict__ found,
                               const double *__restrict__ coords, const int32_t *__restrict__ conn, double *__restrict__ phi,
                               int32_t *errflag) {
    constexpr int NB = DIM + 1, N = P3<DIM>::NLD;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= npts) return;
    const int32_t t = found[q];
    if (t == INT_MAX) {
        atomicOr(errflag, 2);
        for (int i = 0; i < N; ++i) phi[q * N + i] = 0.0;
        return;
    }
    double X[NB * DIM], P[DIM], l[NB], ph[N];
    for (int a = 0; a < NB; ++a) {
        const int64_t v = conn[int64_t(t) * NB + a];
        for (int k = 0; k < DIM; ++k) X[a * DIM + k] = coords[v * DIM + k];
    }
    for (int k = 0; k < DIM; ++k) P[k] = 0.0;
    P[DIM - 1] = pz[q];
    barycentrics<DIM>(X, P, l);
    shape<DIM>(l, ph);
    for (int i = 0; i < N; ++i) phi[q * N + i] = ph[i];
}

void launch_point_shapes(int dim, int npts, const double *pz, const int32_t *found, const double *coords, const int32_t *conn,
                         double *phi, int32_t *errflag, hipStream_t s) {
    const int grid = (npts + 63) / 64;
    if (dim == 2)
        hipLaunchKernelGGL(k_point_shapes<2>, dim3(grid), dim3(64), 0, s, npts, pz, found, coords, conn, phi, errflag);
    else
        hipLaunchKernelGGL(k_point_shapes<3>, dim3(grid), dim3(64), 0, s, npts, pz, found, coords, conn, phi, errflag);
}

// One thread per source: f[dof][rhs] += I phi (a handful of points: atomics are irrelevant here).
template <int DIM, bool CONDENSE>
__global__ void k_build_rhs(int npts, const int32_t *__restrict__ pt_rhs, const double *__restrict__ pt_I,
                            const int32_t *__restrict__ found, const double *__restrict__ phi, const int32_t *__restrict__ eldof,
                            const double *__restrict__ C, const double *__restrict__ M, int k, double *f, double *fint) {
    constexpr int N = P3<DIM>::NLD, NT = P3<DIM>::NTERM, NK = CONDENSE ? 9 : N;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= npts) return;
    fint[q] = 0.0;
    const double I = pt_I[q];
    const int32_t t = found[q];
    if (I == 0.0 || t == INT_MAX) return;  // zero strengths are skipped (ngsolve_functions.py:43)
    const int c = pt_rhs[q];
    const int32_t *ed = eldof + int64_t(t) * N;
    for (int i = 0; i < NK; ++i) {
        const int32_t row = ed[i];
        if (row >= 0) atomicAdd(&f[int64_t(row) * k + c], I * phi[q * N + i]);
    }
    if (CONDENSE) {  // fold the bubble load: f_b -= K_bi K_ii^-1 f_i
        const double fi = I * phi[q * N + 9];
        fint[q] = fi;
        if (fi != 0.0) {
            const double *ce = C + int64_t(t) * NT;
            const double kbb = kentry<DIM>(ce, M, 9, 9);
            for (int j = 0; j < 9; ++j) {
                const int32_t row = ed[j];
                if (row >= 0) atomicAdd(&f[int64_t(row) * k + c], -kentry<DIM>(ce, M, 9, j) * fi / kbb);
            }
        }
    }
}

void launch_build_rhs(int dim, bool condense, int npts, const int32_t *pt_rhs, const double *pt_I, const int32_t *found,
                      const double *phi, const int32_t *eldof, const double *C, const double *M, int k, double *f, double *fint,
                      hipStream_t s) {
    const int grid = (npts + 63) / 64;
    if (dim == 3)
        hipLaunchKernelGGL((k_build_rhs<3, false>), dim3(grid), dim3(64), 0, s, npts, pt_rhs, pt_I, found, phi, eldof, C, M, k, f, fint);
    else if (condense)
        hipLaunchKernelGGL((k_build_rhs<2, true>), dim3(grid), dim3(64), 0, s, npts, pt_rhs, pt_I, found, phi, eldof, C, M, k, f, fint);
    else
        hipLaunchKernelGGL((k_build_rhs<2, false>), dim3(grid), dim3(64), 0, s, npts, pt_rhs, pt_I, found, phi, eldof, C, M, k, f, fint);
}

// One thread per point: u_h(point) = sum phi_i x[dof_i] (+ recovered bubble when condensed:
// u_i = K_ii^-1 (f_i - K_ib u_b), ngsolve_functions.py:53-56).  Points with I != 0 are sources:
// they get NaN-free zeros in `out` slots they do not own (out is indexed by point).
template <int DIM, bool CONDENSE>
__global__ void k_eval(int npts, const int32_t *__restrict__ pt_rhs, const double *__restrict__ pt_I,
                       const int32_t *__restrict__ found, const double *__restrict__ phi, const int32_t *__restrict__ eldof,
                       const double *__restrict__ C, const double *__restrict__ M, int k, const double *__restrict__ x,
                       const double *__restrict__ fint, double *__restrict__ out) {
    constexpr int N = P3<DIM>::NLD, NT = P3<DIM>::NTERM, NK = CONDENSE ? 9 : N;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= npts) return;
    const int32_t t = found[q];
    if (t == INT_MAX) { out[q] = nan(""); return; }
    const int c = pt_rhs[q];
    const int32_t *ed = eldof + int64_t(t) * N;
    double s = 0.0;
    for (int i = 0; i < NK; ++i) {
        const int32_t row = ed[i];
        if (row >= 0) s += phi[q * N + i] * x[int64_t(row) * k + c];
    }
    if (CONDENSE) {
        const double pb = phi[q * N + 9];
        if (pb != 0.0) {
            const double *ce = C + int64_t(t) * NT;
            double acc = 0.0;
            for (int w = 0; w < npts; ++w)
                if (pt_I[w] != 0.0 && found[w] == t && pt_rhs[w] == c) acc += fint[w];
            for (int j = 0; j < 9; ++j) {
                const int32_t row = ed[j];
                if (row >= 0) acc -= kentry<DIM>(ce, M, 9, j) * x[int64_t(row) * k + c];
            }
            s += pb * acc / kentry<DIM>(ce, M, 9, 9);
        }
    }
    out[q] = s;
}

void launch_eval(int dim, bool condense, int npts, const int32_t *pt_rhs, const double *pt_I, const int32_t *found, const double *phi,
                 const int32_t *eldof, const double *C, const double *M, int k, const double *x, const double *fint, double *out,
                 hipStream_t s) {
    const int grid = (npts + 63) / 64;
    if (dim == 3)
        hipLaunchKernelGGL((k_eval<3, false>), dim3(grid), dim3(64), 0, s, npts, pt_rhs, pt_I, found, phi, eldof, C, M, k, x, fint, out);
    else if (condense)
        hipLaunchKernelGGL((k_eval<2, true>), dim3(grid), dim3(64), 0, s, npts, pt_rhs, pt_I, found, phi, eldof, C, M, k, x, fint, out);
    else
        hipLaunchKernelGGL((k_eval<2, false>), dim3(grid), dim3(64), 0, s, npts, pt_rhs, pt_I, found, phi, eldof, C, M, k, x, fint, out);
}

}  // namespace remo
