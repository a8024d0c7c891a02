// kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the ReMo3D hot path:
//   metric terms -> CSR value gather-assembly -> Jacobi-PCG (multi-RHS SpMM + fused vector
//   kernels) -> axis point location / RHS build / evaluation.
// All of it is HBM/L2-bound sparse fp64 work: no MFMA (a sparse row is not a dense contraction);
// the levers are coalesced CSR streams, wave-shuffle reductions, LDS-staged reference tensors and
// few launches per PCG step.  Reference lines each kernel replaces are cited at the kernel.
#include "kernels.h"

#include <limits.h>

#include "fem_p3.h"

namespace remo {

// ------------------------------------------------------------------------------------------
// wave / block reductions (wave = 64 lanes)

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int W> __device__ __forceinline__ double group_sum(double v) {
#pragma unroll
    for (int off = W / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Sum K per-thread values over the block (blockDim.x multiple of 64, <= 1024).  Result valid in
// every thread.  Deterministic: fixed tree.
template <int K> __device__ __forceinline__ void block_sum(double (&v)[K], double *smem /* [16*K] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int c = 0; c < K; ++c) v[c] = wave_sum(v[c]);
    __syncthreads();
    if (lane == 0)
#pragma unroll
        for (int c = 0; c < K; ++c) smem[wave * K + c] = v[c];
    __syncthreads();
#pragma unroll
    for (int c = 0; c < K; ++c) {
        double s = 0.0;
        for (int w = 0; w < nw; ++w) s += smem[w * K + c];
        v[c] = s;
    }
}

// Sum of per-block partials part[nb][K] in a fixed order; result in every thread.
template <int K> __device__ __forceinline__ void reduce_partials(const double *part, int nb, double (&out)[K], double *smem) {
    double v[K];
#pragma unroll
    for (int c = 0; c < K; ++c) v[c] = 0.0;
    for (int b = threadIdx.x; b < nb; b += blockDim.x)
#pragma unroll
        for (int c = 0; c < K; ++c) v[c] += part[b * K + c];
    block_sum<K>(v, smem);
#pragma unroll
    for (int c = 0; c < K; ++c) out[c] = v[c];
}

// ------------------------------------------------------------------------------------------
// metric terms: one thread per element (ngsolve_functions.py:33-36: the coefficient part of the
// integrand; sigma per material as worker.py:101)

template <int DIM>
__global__ void __launch_bounds__(256) k_metric_terms(int64_t nt, const double *__restrict__ coords,
                                                      const int32_t *__restrict__ conn, const int32_t *__restrict__ mat,
                                                      const double *__restrict__ sigma, int nmat, double *__restrict__ C,
                                                      int32_t *errflag) {
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
    const int m = mat[t];
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

void launch_metric_terms(int dim, int64_t nt, const double *coords, const int32_t *conn, const int32_t *mat,
                         const double *sigma, int nmat, double *C, int32_t *errflag, hipStream_t s) {
    const int grid = int((nt + 255) / 256);
    if (dim == 2)
        hipLaunchKernelGGL(k_metric_terms<2>, dim3(grid), dim3(256), 0, s, nt, coords, conn, mat, sigma, nmat, C, errflag);
    else
        hipLaunchKernelGGL(k_metric_terms<3>, dim3(grid), dim3(256), 0, s, nt, coords, conn, mat, sigma, nmat, C, errflag);
}

// ------------------------------------------------------------------------------------------
// CSR value assembly, gather formulation (a.Assemble(), ngsolve_functions.py:47).
// One wave owns one row; lane p owns stored entry p of the row and walks the row's incident
// elements in ascending order, adding K_e[li][lj] where the element's local dof lj is its
// column.  No atomics, every value written exactly once (coalesced), bit-reproducible.
// Reference tensors are staged in LDS (19.2 KB in 3D, 7.2 KB in 2D).

template <int DIM, bool CONDENSE>
__global__ void __launch_bounds__(256) k_assemble(int64_t nfree, const int32_t *__restrict__ rowptr,
                                                  const int32_t *__restrict__ col, const int32_t *__restrict__ adjptr,
                                                  const uint32_t *__restrict__ adj, const int32_t *__restrict__ eldof,
                                                  const double *__restrict__ C, const double *__restrict__ Mg,
                                                  double *__restrict__ val, double *__restrict__ dinv) {
    constexpr int N = P3<DIM>::NLD, NT = P3<DIM>::NTERM;
    constexpr int NK = CONDENSE ? 9 : N;  // local dofs that are unknowns
    __shared__ double M[NT * N * N];
    for (int i = threadIdx.x; i < NT * N * N; i += blockDim.x) M[i] = Mg[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t row = int64_t(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= nfree) return;
    const int32_t rs = rowptr[row], re = rowptr[row + 1];
    const int32_t as = adjptr[row], ae = adjptr[row + 1];
    for (int32_t base = rs; base < re; base += 64) {
        const int32_t p = base + lane;
        const int32_t j = (p < re) ? col[p] : -2;
        double acc = 0.0;
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
                if (CONDENSE) {  // Schur complement of the cell bubble (condense=True, ngsolve_functions.py:31)
                    const double kib = kentry<DIM>(c, M, li, 9), kbj = kentry<DIM>(c, M, 9, lj), kbb = kentry<DIM>(c, M, 9, 9);
                    k -= kib * kbj / kbb;
                }
                acc += k;
            }
        }
        if (p < re) {
            val[p] = acc;
            if (j == row) dinv[row] = 1.0 / acc;  // Jacobi = Preconditioner(a, "local"), ngsolve_functions.py:46
        }
    }
}

void launch_assemble(int dim, bool condense, int64_t nfree, const int32_t *rowptr, const int32_t *col,
                     const int32_t *adjptr, const uint32_t *adj, const int32_t *eldof, const double *C,
                     const double *M, double *val, double *dinv, hipStream_t s) {
    const int grid = int((nfree + 3) / 4);
    if (dim == 3)
        hipLaunchKernelGGL((k_assemble<3, false>), dim3(grid), dim3(256), 0, s, nfree, rowptr, col, adjptr, adj, eldof, C, M, val, dinv);
    else if (condense)
        hipLaunchKernelGGL((k_assemble<2, true>), dim3(grid), dim3(256), 0, s, nfree, rowptr, col, adjptr, adj, eldof, C, M, val, dinv);
    else
        hipLaunchKernelGGL((k_assemble<2, false>), dim3(grid), dim3(256), 0, s, nfree, rowptr, col, adjptr, adj, eldof, C, M, val, dinv);
}

// ------------------------------------------------------------------------------------------
// CSR SpMM  y = A x  for K interleaved right-hand sides (x[n][K] row-major), the kernel the
// CG hot loop spends its time in (CGSolver, ngsolve_functions.py:50-51; cusparseSpMV in the
// reference's CUDA attempt, ngsolve_functions_gpu.py:41-47).
// LPR lanes cooperate on a row: values/columns are read as contiguous runs (rows are contiguous
// in CSR, so a wave streams one contiguous span), x rows are gathered K doubles at a time, the
// LPR partial sums are combined with wave shuffles.  Optionally leaves per-block partial sums of
// <x, y> (the CG's <p, Ap>) so the dot product costs no extra pass.

template <int K, int LPR, bool DOT>
__global__ void __launch_bounds__(256) k_spmm(int64_t n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                              const double *__restrict__ val, const double *__restrict__ x,
                                              double *__restrict__ y, double *__restrict__ part) {
    constexpr int RPB = 256 / LPR;  // rows per block pass
    const int sub = threadIdx.x % LPR;
    const int grp = threadIdx.x / LPR;
    double dot[K];
#pragma unroll
    for (int c = 0; c < K; ++c) dot[c] = 0.0;
    for (int64_t row = int64_t(blockIdx.x) * RPB + grp; row < n; row += int64_t(gridDim.x) * RPB) {
        const int32_t rs = rowptr[row], re = rowptr[row + 1];
        double acc[K];
#pragma unroll
        for (int c = 0; c < K; ++c) acc[c] = 0.0;
        for (int32_t p = rs + sub; p < re; p += LPR) {
            const double v = val[p];
            const double *xr = x + int64_t(col[p]) * K;
#pragma unroll
            for (int c = 0; c < K; ++c) acc[c] += v * xr[c];
        }
#pragma unroll
        for (int c = 0; c < K; ++c) acc[c] = group_sum<LPR>(acc[c]);
        if (sub == 0) {
#pragma unroll
            for (int c = 0; c < K; ++c) y[row * K + c] = acc[c];
            if (DOT) {
                const double *xr = x + row * K;
#pragma unroll
                for (int c = 0; c < K; ++c) dot[c] += acc[c] * xr[c];
            }
        }
    }
    if (DOT) {
        __shared__ double smem[16 * K];
        block_sum<K>(dot, smem);
        if (threadIdx.x < K) part[blockIdx.x * K + threadIdx.x] = dot[threadIdx.x];
    }
}

int choose_lanes_per_row(int64_t n, int64_t nnz) {
    const double avg = double(nnz) / double(n > 0 ? n : 1);
    if (avg > 40) return 16;
    if (avg > 20) return 8;
    return 4;
}

int spmv_grid(int64_t n, int lpr) {
    const int64_t rpb = 256 / lpr;
    int64_t g = (n + rpb - 1) / rpb;
    if (g > kMaxPartialBlocks) g = kMaxPartialBlocks;
    if (g < 1) g = 1;
    return int(g);
}

template <int K> static void spmm_dispatch(const CsrView &A, const double *x, double *y, double *part, int nb, hipStream_t s) {
    const int lpr = choose_lanes_per_row(A.n, A.nnz);
#define REMO_SPMM(L)                                                                                                  \
    if (part)                                                                                                         \
        hipLaunchKernelGGL((k_spmm<K, L, true>), dim3(nb), dim3(256), 0, s, A.n, A.rowptr, A.col, A.val, x, y, part); \
    else                                                                                                              \
        hipLaunchKernelGGL((k_spmm<K, L, false>), dim3(nb), dim3(256), 0, s, A.n, A.rowptr, A.col, A.val, x, y, part)
    if (lpr == 16) { REMO_SPMM(16); }
    else if (lpr == 8) { REMO_SPMM(8); }
    else { REMO_SPMM(4); }
#undef REMO_SPMM
}

void launch_spmm(const CsrView &A, int k, const double *x, double *y, double *part, int nb, hipStream_t s) {
    switch (k) {
        case 1: spmm_dispatch<1>(A, x, y, part, nb, s); break;
        case 2: spmm_dispatch<2>(A, x, y, part, nb, s); break;
        case 3: spmm_dispatch<3>(A, x, y, part, nb, s); break;
        case 4: spmm_dispatch<4>(A, x, y, part, nb, s); break;
        case 5: spmm_dispatch<5>(A, x, y, part, nb, s); break;
        case 6: spmm_dispatch<6>(A, x, y, part, nb, s); break;
        case 7: spmm_dispatch<7>(A, x, y, part, nb, s); break;
        default: spmm_dispatch<8>(A, x, y, part, nb, s); break;
    }
}

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

template <int K>
__global__ void __launch_bounds__(256) k_pcg_init(int64_t n, const double *__restrict__ f, const double *__restrict__ dinv,
                                                  double *__restrict__ x, double *__restrict__ r, double *__restrict__ p,
                                                  double *__restrict__ part_rz) {
    __shared__ double smem[16 * K];
    double rz[K];
#pragma unroll
    for (int c = 0; c < K; ++c) rz[c] = 0.0;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
        const double d = dinv[i];
#pragma unroll
        for (int c = 0; c < K; ++c) {
            const double ri = f[i * K + c];
            const double zi = d * ri;
            x[i * K + c] = 0.0;
            r[i * K + c] = ri;
            p[i * K + c] = zi;
            rz[c] += ri * zi;
        }
    }
    block_sum<K>(rz, smem);
    if (threadIdx.x < K) part_rz[blockIdx.x * K + threadIdx.x] = rz[threadIdx.x];
}

template <int K>
__global__ void __launch_bounds__(256) k_pcg_update(int64_t n, int step, double tol2, int nb_spmv, int nb_vec,
                                                    const double *__restrict__ part_pq, const double *__restrict__ part_rz_cur,
                                                    double *__restrict__ part_rz_next, double *__restrict__ rz0,
                                                    PcgProgress *progress, int progress_len, const double *__restrict__ p,
                                                    const double *__restrict__ q, double *__restrict__ x, double *__restrict__ r,
                                                    const double *__restrict__ dinv) {
    __shared__ double smem[16 * K];
    double pq[K], rz[K], alpha[K], acc[K];
    reduce_partials<K>(part_pq, nb_spmv, pq, smem);
    __syncthreads();
    reduce_partials<K>(part_rz_cur, nb_vec, rz, smem);
#pragma unroll
    for (int c = 0; c < K; ++c) {
        const double r0 = (step == 0) ? rz[c] : rz0[c];
        const bool live = (rz[c] > tol2 * r0) && (pq[c] > 0.0);
        alpha[c] = live ? rz[c] / pq[c] : 0.0;
        acc[c] = 0.0;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (step == 0)
#pragma unroll
            for (int c = 0; c < K; ++c) rz0[c] = rz[c];
        // progress record in mapped host memory: data first, then the step number (system scope)
        PcgProgress *pr = progress + (step % progress_len);
#pragma unroll
        for (int c = 0; c < K; ++c) __hip_atomic_store(&pr->rz[c], rz[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&pr->step, step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
        const double d = dinv[i];
#pragma unroll
        for (int c = 0; c < K; ++c) {
            const double a = alpha[c];
            const double xi = x[i * K + c] + a * p[i * K + c];
            const double ri = r[i * K + c] - a * q[i * K + c];
            x[i * K + c] = xi;
            r[i * K + c] = ri;
            acc[c] += ri * ri * d;
        }
    }
    __syncthreads();
    block_sum<K>(acc, smem);
    if (threadIdx.x < K) part_rz_next[blockIdx.x * K + threadIdx.x] = acc[threadIdx.x];
}

template <int K>
__global__ void __launch_bounds__(256) k_pcg_direction(int64_t n, double tol2, int nb_spmv, int nb_vec,
                                                       const double *__restrict__ part_pq, const double *__restrict__ part_rz_old,
                                                       const double *__restrict__ part_rz_new, const double *__restrict__ rz0,
                                                       const double *__restrict__ r, double *__restrict__ p,
                                                       const double *__restrict__ dinv) {
    __shared__ double smem[16 * K];
    double pq[K], rzo[K], rzn[K], beta[K];
    reduce_partials<K>(part_pq, nb_spmv, pq, smem);
    __syncthreads();
    reduce_partials<K>(part_rz_old, nb_vec, rzo, smem);
    __syncthreads();
    reduce_partials<K>(part_rz_new, nb_vec, rzn, smem);
#pragma unroll
    for (int c = 0; c < K; ++c) {
        const bool live = (rzo[c] > tol2 * rz0[c]) && (pq[c] > 0.0);
        beta[c] = live ? rzn[c] / rzo[c] : 0.0;
    }
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
        const double d = dinv[i];
#pragma unroll
        for (int c = 0; c < K; ++c) p[i * K + c] = d * r[i * K + c] + beta[c] * p[i * K + c];
    }
}

template <int K>
__global__ void __launch_bounds__(256) k_pcg_final(int step, int nb_vec, const double *__restrict__ part_rz, PcgProgress *progress,
                                                   int progress_len) {
    __shared__ double smem[16 * K];
    double rz[K];
    reduce_partials<K>(part_rz, nb_vec, rz, smem);
    if (threadIdx.x == 0) {
        PcgProgress *pr = progress + (step % progress_len);
#pragma unroll
        for (int c = 0; c < K; ++c) __hip_atomic_store(&pr->rz[c], rz[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&pr->step, step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

static int vec_grid(int64_t n) {
    int64_t g = (n + 255) / 256;
    if (g > kMaxPartialBlocks / 2) g = kMaxPartialBlocks / 2;
    if (g < 1) g = 1;
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

void launch_pcg_init(int64_t n, int k, const double *f, const PcgBuffers &b, hipStream_t s) {
    const int g = vec_grid(n);
    REMO_K_SWITCH(k, hipLaunchKernelGGL(k_pcg_init<KK>, dim3(g), dim3(256), 0, s, n, f, b.dinv, b.x, b.r, b.p, b.part_rz));
}

void launch_pcg_update(int64_t n, int k, int step, double tol2, const PcgBuffers &b, hipStream_t s) {
    const int g = vec_grid(n);
    double *cur = b.part_rz + (step & 1) * (kMaxPartialBlocks * 8);
    double *nxt = b.part_rz + ((step + 1) & 1) * (kMaxPartialBlocks * 8);
    REMO_K_SWITCH(k, hipLaunchKernelGGL(k_pcg_update<KK>, dim3(g), dim3(256), 0, s, n, step, tol2, b.nb_spmv, g, b.part_pq, cur, nxt,
                                        b.rz0, b.progress, b.progress_len, b.p, b.q, b.x, b.r, b.dinv));
}

void launch_pcg_direction(int64_t n, int k, int step, double tol2, const PcgBuffers &b, hipStream_t s) {
    const int g = vec_grid(n);
    const double *old = b.part_rz + (step & 1) * (kMaxPartialBlocks * 8);
    const double *nw = b.part_rz + ((step + 1) & 1) * (kMaxPartialBlocks * 8);
    REMO_K_SWITCH(k, hipLaunchKernelGGL(k_pcg_direction<KK>, dim3(g), dim3(256), 0, s, n, tol2, b.nb_spmv, g, b.part_pq, old, nw, b.rz0,
                                        b.r, b.p, b.dinv));
}

void launch_pcg_final(int k, int step, const PcgBuffers &b, hipStream_t s) {
    const double *cur = b.part_rz + (step & 1) * (kMaxPartialBlocks * 8);
    REMO_K_SWITCH(k, hipLaunchKernelGGL(k_pcg_final<KK>, dim3(1), dim3(256), 0, s, step, b.nb_vec, cur, b.progress, b.progress_len));
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
