// amg.h — smoothed-aggregation multigrid cycle on the P1 (vertex) block (amg.hip): the coarse solver of the two-level
// preconditioner where the Chebyshev polynomial needs a high degree (2D: graded axisymmetric meshes).  Replaces the
// "multigrid" choice of ngsolve_functions.py:46 on the vertex block; the edge / face dofs keep their Jacobi factors.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "symbolic_gpu.h"

namespace remo {

constexpr int kAmgMaxLevels = 8;
constexpr int kAmgDenseMax = 64;   // the coarsest operator is inverted densely in LDS

// One level: operator A (CSR, columns ascending), its Jacobi factors, the prolongator P to the next level (n x n_next)
// and its transpose R, level vectors [n][k].  T = storage type of values and vectors.
template <class T> struct AmgLevelT {
    int64_t n = 0, nnz = 0, nnz_p = 0;   // rows and entries of A, entries of P (= of R)
    const int32_t *rowptr = nullptr, *col = nullptr;
    const T *val = nullptr, *dinv = nullptr;
    const int32_t *p_rowptr = nullptr, *p_col = nullptr;
    const T *p_val = nullptr;
    const int32_t *r_rowptr = nullptr, *r_col = nullptr;
    const T *r_val = nullptr;
    double omega = 0.0;            // damped Jacobi smoother z += omega D^-1 (r - A z)
    T *r = nullptr, *z = nullptr, *z2 = nullptr, *t = nullptr;
};

template <class T> struct AmgT {
    int levels = 0;
    AmgLevelT<T> lev[kAmgMaxLevels];
    const T *inv = nullptr;        // dense inverse of the coarsest operator [nc][nc]
    int launches = 0;              // kernel launches of one cycle
    int kmax = 0;                  // columns the level vectors were allocated for
};

// Build the hierarchy (fp64) of the leading nv x nv block of A (rows' vertex entries lead: columns ascend).  Everything is
// enqueued on s; the function synchronises a few times per level to read sizes back.  Returns false (with the reason) when
// the hierarchy cannot be built (a row too long for the LDS tables, no coarsening, arena exhausted): the caller keeps the
// Chebyshev polynomial.
bool amg_setup(Arena &ar, hipStream_t s, int dim, int64_t nv, const int32_t *rowptr, const int32_t *col, const double *val, int kmax,
               AmgT<double> &out, std::string &why);
// fp32 image of a hierarchy (values converted, patterns shared, own vectors) for the inner solver of the mixed mode
void amg_to_float(Arena &ar, hipStream_t s, const AmgT<double> &in, int kmax, AmgT<float> &out);

// z = V(1,1)-cycle(r) on the vertex rows; the last launch stores cz = z / dinv (the direction kernel treats it like r) and the
// <r, z> partial sums of `nblocks` workgroups [nblocks][k] in part.  scal / step: early exit of finished solves (kernels.hip)
// T = storage type of the hierarchy, TR = type of r and cz: <float, double> is the cycle in fp32 inside an fp64 solve (a
// preconditioner may be applied inexactly; the recurrences of x and r never see it)
template <class T, class TR> void launch_amg_cycle(const AmgT<T> &H, int k, int step, const TR *r, TR *cz, double *part, int nblocks, const double *scal,
                                                   hipStream_t s);

}  // namespace remo
