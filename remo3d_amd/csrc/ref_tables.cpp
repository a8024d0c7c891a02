// ref_tables.cpp — exact reference tensors of the order-3 H1 stiffness form (host, once).
//
// M3[(a,b)][i][j]   = mean_T( D_a phi_i D_b phi_j  [+ D_b phi_i D_a phi_j if a != b] )
// M2[k][(a,b)][i][j] = mean_T( l_k * (same) )           (axisymmetric weight, r = sum r_k l_k)
// with D_a = d/dl_a - d/dl_0 (reference coordinates xi_a = l_a, l_0 = 1 - sum xi).
// Integration is exact: every integrand is a polynomial in the barycentrics and
//   mean_T( l^alpha ) = d! prod(alpha_k!) / (d + |alpha|)!
// The integrand of the reference's 2D form is of degree 5 (r * grad * grad); NGSolve's rule
// order is part of the un-pinned third-party arithmetic — the exact
// integral is used here and the generator is the single place a different rule would plug in.
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "fem_p3.h"

namespace remo {
namespace {

struct Mono {
    int e[4];
    double c;
};
using Poly = std::vector<Mono>;

Poly mono(double c, int a = -1, int pa = 0, int b = -1, int pb = 0, int c3 = -1, int pc = 0) {
    Mono m{{0, 0, 0, 0}, c};
    if (a >= 0) m.e[a] += pa;
    if (b >= 0) m.e[b] += pb;
    if (c3 >= 0) m.e[c3] += pc;
    return Poly{m};
}
Poly add(const Poly &p, const Poly &q, double s = 1.0) {
    Poly r = p;
    for (auto m : q) {
        m.c *= s;
        r.push_back(m);
    }
    return r;
}
Poly mul(const Poly &p, const Poly &q) {
    Poly r;
    for (const auto &a : p)
        for (const auto &b : q) {
            Mono m{{a.e[0] + b.e[0], a.e[1] + b.e[1], a.e[2] + b.e[2], a.e[3] + b.e[3]}, a.c * b.c};
            r.push_back(m);
        }
    return r;
}
Poly diff(const Poly &p, int a) {
    Poly r;
    for (const auto &m : p) {
        if (m.e[a] == 0) continue;
        Mono d = m;
        d.c *= m.e[a];
        d.e[a] -= 1;
        r.push_back(d);
    }
    return r;
}
double fact(int n) {
    double f = 1;
    for (int i = 2; i <= n; ++i) f *= i;
    return f;
}
double mean(const Poly &p, int dim) {
    long double s = 0;
    for (const auto &m : p) {
        int tot = m.e[0] + m.e[1] + m.e[2] + m.e[3];
        long double v = fact(dim) * fact(m.e[0]) * fact(m.e[1]) * fact(m.e[2]) * fact(m.e[3]) / fact(dim + tot);
        s += (long double)m.c * v;
    }
    return (double)s;
}

// mean_T of a polynomial by a QUADRATURE RULE instead of the exact formula (2D only): the 6-point rule of Strang and Fix /
// Dunavant, exact to degree 4 - one order below the degree-5 integrand of the axisymmetric form.  Which rule NGSolve's
// SymbolicBFI applies to `2 pi x sigma grad(u) grad(v)` (ngsolve_functions.py:34) is part of the un-pinned third-party
// arithmetic; this is the alternative SURVEY.md section 7.3-2 asks to have pluggable (remo_opts_t.quadrature = 1).
double mean_rule4(const Poly &p) {
    static const double a1 = 0.445948490915965, b1 = 0.108103018168070, w1 = 0.223381589678011;
    static const double a2 = 0.091576213509771, b2 = 0.816847572980459, w2 = 0.109951743655322;
    const double pts[6][3] = {{b1, a1, a1}, {a1, b1, a1}, {a1, a1, b1}, {b2, a2, a2}, {a2, b2, a2}, {a2, a2, b2}};
    const double wts[6] = {w1, w1, w1, w2, w2, w2};
    long double s = 0;
    for (int q = 0; q < 6; ++q) {
        long double v = 0;
        for (const auto &m : p) v += (long double)m.c * std::pow(pts[q][0], m.e[0]) * std::pow(pts[q][1], m.e[1]) * std::pow(pts[q][2], m.e[2]);
        s += wts[q] * v;
    }
    return (double)s;
}

std::vector<Poly> basis(int dim) {
    std::vector<Poly> phi;
    for (int i = 0; i <= dim; ++i) phi.push_back(mono(1.0, i, 1));
    const int ne = (dim == 2) ? 3 : 6;
    for (int e = 0; e < ne; ++e) {
        const int a = edge_a(dim, e), b = edge_b(dim, e);
        phi.push_back(mono(1.0, a, 1, b, 1));
        phi.push_back(add(mono(1.0, a, 1, b, 2), mono(-1.0, a, 2, b, 1)));
    }
    if (dim == 2) {
        phi.push_back(mono(1.0, 0, 1, 1, 1, 2, 1));
    } else {
        for (int f = 0; f < 4; ++f) {
            int a, b, c;
            face_abc(f, a, b, c);
            phi.push_back(mono(1.0, a, 1, b, 1, c, 1));
        }
    }
    return phi;
}

std::vector<double> build(int dim, int rule = 0) {
    const int n = (dim == 2) ? 10 : 20;
    const auto phi = basis(dim);
    // D[a][i] = d phi_i / d xi_a, a = 1..dim
    std::vector<std::vector<Poly>> D(dim + 1, std::vector<Poly>(n));
    for (int a = 1; a <= dim; ++a)
        for (int i = 0; i < n; ++i) D[a][i] = add(diff(phi[i], a), diff(phi[i], 0), -1.0);
    std::vector<double> M;
    auto pair_tensor = [&](int a, int b, int k /* -1 or weight index */) {
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                Poly p = mul(D[a][i], D[b][j]);
                if (a != b) p = add(p, mul(D[b][i], D[a][j]));
                if (k >= 0) p = mul(p, mono(1.0, k, 1));
                M.push_back((rule == 1 && dim == 2) ? mean_rule4(p) : mean(p, dim));
            }
    };
    if (dim == 3) {
        for (int a = 1; a <= 3; ++a)
            for (int b = a; b <= 3; ++b) pair_tensor(a, b, -1);
    } else {
        for (int k = 0; k < 3; ++k)
            for (int a = 1; a <= 2; ++a)
                for (int b = a; b <= 2; ++b) pair_tensor(a, b, k);
    }
    return M;
}

// ---- factorised form of the 3D tensors (element-wise operator, kernels.hip k_elem_apply) -----------------------------
// The reference gradient D_a phi_i of a P3 function is a P2 polynomial.  With psi_1..psi_10 an orthonormal basis of
// P2(T) in the mean_T inner product and B[a][m][i] = mean_T(psi_m D_a phi_i):
//     mean_T(D_a phi_i D_b phi_j) = sum_m B[a][m][i] B[b][m][j]          (exactly: both factors lie in span psi)
// so K_e X = sum_a B_a^T ( sum_b c~_ab (B_b X) ),  c~ the symmetric 3 x 3 matrix of the metric terms: 6450 multiply-adds
// for 5 right-hand sides against 12000 through the six 20 x 20 tensors.
std::vector<double> build_factors() {
    const int dim = 3, n = 20;
    const auto phi = basis(dim);
    std::vector<std::vector<Poly>> D(dim + 1, std::vector<Poly>(n));
    for (int a = 1; a <= dim; ++a)
        for (int i = 0; i < n; ++i) D[a][i] = add(diff(phi[i], a), diff(phi[i], 0), -1.0);
    // monomials of degree <= 2 in xi_1..xi_3 (l_0 eliminated), Gram-Schmidt in the mean_T inner product
    std::vector<Poly> mon;
    mon.push_back(mono(1.0));
    for (int a = 1; a <= 3; ++a) mon.push_back(mono(1.0, a, 1));
    for (int a = 1; a <= 3; ++a)
        for (int b = a; b <= 3; ++b) mon.push_back(a == b ? mono(1.0, a, 2) : mono(1.0, a, 1, b, 1));
    std::vector<Poly> psi;
    for (const auto &m : mon) {
        Poly v = m;
        for (int pass = 0; pass < 2; ++pass)           // twice: classical Gram-Schmidt loses digits once
            for (const auto &q : psi) v = add(v, q, -mean(mul(v, q), dim));
        const double nrm = std::sqrt(mean(mul(v, v), dim));
        Poly u;
        for (auto t : v) { t.c /= nrm; u.push_back(t); }
        psi.push_back(u);
    }
    std::vector<double> B(3 * 10 * n);
    for (int a = 1; a <= 3; ++a)
        for (int m = 0; m < 10; ++m)
            for (int i = 0; i < n; ++i) {
                const double v = mean(mul(psi[m], D[a][i]), dim);
                B[((a - 1) * 10 + m) * n + i] = std::fabs(v) < 1e-14 ? 0.0 : v;
            }
    return B;
}

}  // namespace

const double *ref_factors3() {
    static std::once_flag f;
    static std::vector<double> b;
    std::call_once(f, [] { b = build_factors(); });
    return b.data();
}

// max |sum_m B_a B_b (symmetrised like M3) - M3| over all entries: the factorisation reproduces the tensors
double ref_factors3_error() {
    const double *B = ref_factors3(), *M = ref_tables(3);
    double worst = 0.0;
    int t = 0;
    for (int a = 0; a < 3; ++a)
        for (int b = a; b < 3; ++b, ++t)
            for (int i = 0; i < 20; ++i)
                for (int j = 0; j < 20; ++j) {
                    double s = 0.0;
                    for (int m = 0; m < 10; ++m) {
                        s += B[(a * 10 + m) * 20 + i] * B[(b * 10 + m) * 20 + j];
                        if (a != b) s += B[(b * 10 + m) * 20 + i] * B[(a * 10 + m) * 20 + j];
                    }
                    const double e = std::fabs(s - M[(t * 20 + i) * 20 + j]);
                    if (e > worst) worst = e;
                }
    return worst;
}

const double *ref_tables2_rule4() {
    static std::once_flag f;
    static std::vector<double> t;
    std::call_once(f, [] { t = build(2, 1); });
    return t.data();
}

const double *ref_tables(int dim) {
    static std::once_flag f2, f3;
    static std::vector<double> t2, t3;
    if (dim == 2) {
        std::call_once(f2, [] { t2 = build(2); });
        return t2.data();
    }
    std::call_once(f3, [] { t3 = build(3); });
    return t3.data();
}

}  // namespace remo
