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

std::vector<double> build(int dim) {
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
                M.push_back(mean(p, dim));
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

}  // namespace

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
