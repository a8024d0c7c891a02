// fem_p3.h — order-3 H1 element on straight-sided simplices, shared by host and gfx950 code.
//
// Replaces what NGSolve's H1(order=3) + SymbolicBFI do for ReMo3D's bilinear form
// (ngsolve_functions.py:27-36).  The element matrix is never integrated per element: with the
// vertices of every element sorted by global number, it is the contraction
//
//     K_e[i][j] = sum_t  C_e[t] * M[t][i][j]
//
// of per-element metric terms C_e (6 in 3D, 9 in 2D) with reference tensors M that are
// integrated exactly once on the host (ref_tables.cpp).
//   3D: t = (a,b), 1<=a<=b<=3:   C = sigma |T| grad(l_a).grad(l_b)
//   2D: t = 3k + (a,b), k=0..2, 1<=a<=b<=2:  C = 2 pi sigma |T| r_k grad(l_a).grad(l_b)
//       (axisymmetric weight 2 pi r, r = first coordinate, ngsolve_functions.py:34)
//
// Basis (hierarchical, barycentric; any basis of P3 gives the same Galerkin solution):
//   vertex i: l_i | edge (a,b), a<b: l_a l_b and l_a l_b (l_b - l_a) | face/cell: l_a l_b l_c
// Local order: vertices, edges (01,02,03,12,13,23 | 01,02,12) two dofs each, faces
// (012,013,023,123) | the cell bubble.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define REMO_HD __host__ __device__ __forceinline__

namespace remo {

template <int DIM> struct P3 {
    static constexpr int NB = DIM + 1;
    static constexpr int NLD = (DIM == 2) ? 10 : 20;  // local dofs incl. the 2D bubble
    static constexpr int NTERM = (DIM == 2) ? 9 : 6;
    static constexpr int NEDGE = (DIM == 2) ? 3 : 6;
    static constexpr int NFACE = (DIM == 2) ? 0 : 4;
};

REMO_HD int edge_a(int dim, int e) {
    // 3D: 01 02 03 12 13 23 ; 2D: 01 02 12
    if (dim == 3) return (e < 3) ? 0 : ((e < 5) ? 1 : 2);
    return (e < 2) ? 0 : 1;
}
REMO_HD int edge_b(int dim, int e) {
    if (dim == 3) return (e < 3) ? e + 1 : ((e < 5) ? e - 1 : 3);
    return (e < 2) ? e + 1 : 2;
}
REMO_HD void face_abc(int f, int &a, int &b, int &c) {
    // 012 013 023 123
    a = (f == 3) ? 1 : 0;
    b = (f < 2) ? 1 : 2;
    c = (f == 0) ? 2 : 3;
}

// shape functions at barycentrics l[DIM+1]
template <int DIM> REMO_HD void shape(const double *l, double *phi) {
    int k = 0;
    for (int i = 0; i <= DIM; ++i) phi[k++] = l[i];
    for (int e = 0; e < P3<DIM>::NEDGE; ++e) {
        const double la = l[edge_a(DIM, e)], lb = l[edge_b(DIM, e)];
        phi[k++] = la * lb;
        phi[k++] = la * lb * (lb - la);
    }
    if (DIM == 2) {
        phi[k++] = l[0] * l[1] * l[2];
    } else {
        for (int f = 0; f < 4; ++f) {
            int a, b, c;
            face_abc(f, a, b, c);
            phi[k++] = l[a] * l[b] * l[c];
        }
    }
}

// Gradients of l_1..l_DIM (rows of the inverse Jacobian) and the measure |T|.
// X = coordinates of the (sorted) vertices, [NB][DIM].  Returns |T|; 0 for a degenerate element.
template <int DIM> REMO_HD double bary_gradients(const double *X, double g[DIM][DIM]) {
    if (DIM == 2) {
        const double a11 = X[2] - X[0], a21 = X[3] - X[1];
        const double a12 = X[4] - X[0], a22 = X[5] - X[1];
        const double det = a11 * a22 - a12 * a21;
        if (det == 0.0) return 0.0;
        const double id = 1.0 / det;
        g[0][0] = a22 * id;  g[0][1] = -a12 * id;
        g[1][0] = -a21 * id; g[1][1] = a11 * id;
        return (det < 0 ? -det : det) * 0.5;
    } else {
        double A[3][3];
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i) A[i][j] = X[3 * (j + 1) + i] - X[i];
        const double c00 = A[1][1] * A[2][2] - A[1][2] * A[2][1];
        const double c01 = A[1][2] * A[2][0] - A[1][0] * A[2][2];
        const double c02 = A[1][0] * A[2][1] - A[1][1] * A[2][0];
        const double det = A[0][0] * c00 + A[0][1] * c01 + A[0][2] * c02;
        if (det == 0.0) return 0.0;
        const double id = 1.0 / det;
        g[0][0] = c00 * id;
        g[0][1] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) * id;
        g[0][2] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) * id;
        g[1][0] = c01 * id;
        g[1][1] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) * id;
        g[1][2] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) * id;
        g[2][0] = c02 * id;
        g[2][1] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) * id;
        g[2][2] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) * id;
        return (det < 0 ? -det : det) / 6.0;
    }
}

// Metric terms C[NTERM] of one element.  Returns false for a degenerate element.
template <int DIM> REMO_HD bool metric_terms(const double *X, double sigma, double *C) {
    double g[DIM][DIM];
    const double vol = bary_gradients<DIM>(X, g);
    if (!(vol > 0.0)) return false;
    if (DIM == 3) {
        const double s = sigma * vol;
        int t = 0;
        for (int a = 0; a < 3; ++a)
            for (int b = a; b < 3; ++b)
                C[t++] = s * (g[a][0] * g[b][0] + g[a][1] * g[b][1] + g[a][2] * g[b][2]);
    } else {
        const double s = 6.283185307179586476925286766559 * sigma * vol;
        const double g11 = g[0][0] * g[0][0] + g[0][1] * g[0][1];
        const double g12 = g[0][0] * g[1][0] + g[0][1] * g[1][1];
        const double g22 = g[1][0] * g[1][0] + g[1][1] * g[1][1];
        for (int k = 0; k < 3; ++k) {
            const double r = X[2 * k];
            C[3 * k + 0] = s * r * g11;
            C[3 * k + 1] = s * r * g12;
            C[3 * k + 2] = s * r * g22;
        }
    }
    return true;
}

// K_e[i][j] from metric terms and reference tensors M[NTERM][NLD][NLD]
template <int DIM> REMO_HD double kentry(const double *C, const double *M, int i, int j) {
    constexpr int N = P3<DIM>::NLD;
    double s = 0.0;
#pragma unroll
    for (int t = 0; t < P3<DIM>::NTERM; ++t) s += C[t] * M[(t * N + i) * N + j];
    return s;
}

// Barycentrics of point P (DIM coords) in element with vertices X.  Returns false if degenerate.
template <int DIM> REMO_HD bool barycentrics(const double *X, const double *P, double *l) {
    double g[DIM][DIM];
    const double vol = bary_gradients<DIM>(X, g);
    if (!(vol > 0.0)) return false;
    double s = 0.0;
    for (int a = 0; a < DIM; ++a) {
        double v = 0.0;
        for (int k = 0; k < DIM; ++k) v += g[a][k] * (P[k] - X[k]);
        l[a + 1] = v;
        s += v;
    }
    l[0] = 1.0 - s;
    return true;
}

// host: exact reference tensors (ref_tables.cpp)
const double *ref_tables(int dim);  // [NTERM][NLD][NLD], exact integrals
const double *ref_tables2_rule4();   // 2D tensors by the 6-point degree-4 rule instead (remo_opts_t.quadrature = 1)
const double *ref_factors3();        // [3][10][20]: B[a][m][i], mean_T(D_a phi_i D_b phi_j) = sum_m B[a][m][i] B[b][m][j]
double ref_factors3_error();         // max deviation of the factorised form from ref_tables(3)

}  // namespace remo
