/*
 * oracle/fem_oracle.c — CPU restatement of ReMo3D's per-measurement-point FEM hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (remo3d_amd/) may import, link or call
 * this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and
 * only as the checker / the CPU baseline.
 *
 * PARITY UNPINNED (against NGSolve): the arithmetic of the reference path lives in NGSolve,
 * a third-party dependency that is not vendored under /root/reference and has no pinned
 * version (setup.py:11 does not even list it).  The reference has no tests and no golden
 * vectors at the SolveBVP boundary.  This oracle is therefore pinned by (a) closed-form
 * physics (homogeneous grounded sphere, Ra == R identity of remo3d.py:285-306), and (b) the
 * reference's committed example logs through a different mesh (tolerance ~1e-3).  See
 * DESIGN.md "Oracle".
 *
 * What is restated, with the reference lines each function follows:
 *   orc_create      : H1(order=3, dirichlet=...) dof numbering + BilinearForm assembly
 *                     ngsolve_functions.py:27-36,47 ; sigma per material worker.py:100-101
 *   orc_rhs         : LinearForm + AddPointSource                ngsolve_functions.py:10-21,39-44
 *   orc_pcg         : Preconditioner("local") + CGSolver          ngsolve_functions.py:46,50-51
 *   condensation    : condense=True Schur complement + back-substitution  ngsolve_functions.py:31,53-56
 *   orc_eval        : gfu(mesh(0,z)) / gfu(mesh(0,0,z))           worker.py:124-131
 *   orc_solve_batch : the inner hot loop over the RHS of one batch worker.py:104-131
 *
 * Discretisation: straight-sided simplices, order-3 H1.  The Galerkin solution does not depend
 * on the basis, so a hierarchical basis in barycentric coordinates is used (vertices of every
 * element sorted by global number, which orients edges/faces consistently):
 *     vertex i        : l_i
 *     edge (a,b), a<b : l_a l_b          and   l_a l_b (l_b - l_a)
 *     face (a,b,c)    : l_a l_b l_c      (2D: the cell bubble)
 * Global numbering: vertices, then 2 per edge (edges sorted lexicographically by (min,max)
 * vertex), then 1 per face (3D; sorted lexicographically) or 1 per cell (2D, only when
 * condense == 0).  Constrained (Dirichlet) dofs are eliminated; free dofs are numbered
 * compactly in ascending global order.
 *
 * Element matrices are integrated HERE by numerical quadrature (collapsed Gauss-Legendre,
 * 4 points per direction, exact to total degree 5) on every element — deliberately a different
 * route from the product, which contracts per-element metric terms with pre-integrated
 * reference tensors.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAXLD 20
#define NGL 4

typedef struct {
    int dim, nvl, nel, nfl, nld_full, nld; /* local counts; nld = dofs kept in global system */
    int condense;
    long nv, nt, ne, nf, ndof, nfree, nnz;
    double *xyz;
    int *conn;  /* [nt*(dim+1)] sorted ascending per element */
    int *mat;
    uint64_t *ekeys; /* sorted unique */
    uint64_t *fkeys;
    int *eldof;  /* [nt*nld_full] global dof ids (full numbering incl. bubbles) */
    int *freeid; /* [ndof_full] -> free index or -1 */
    long ndof_full;
    long *rowptr;
    int *col;
    double *val;
    double *diag;
    /* 2D condensation: per element K_ib (9) and K_ii */
    double *kib, *kii;
    double *sigma;
    int nmat;
    /* quadrature */
    int nq;
    double *ql; /* [nq*(dim+1)] barycentrics */
    double *qw; /* [nq] weights, sum 1 */
    char err[256];
} orc_t;

static const int E3[6][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3}};
static const int F3[4][3] = {{0, 1, 2}, {0, 1, 3}, {0, 2, 3}, {1, 2, 3}};
static const int E2[3][2] = {{0, 1}, {0, 2}, {1, 2}};

static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return (x > y) - (x < y);
}
static int cmp_int(const void *a, const void *b) {
    int x = *(const int *)a, y = *(const int *)b;
    return (x > y) - (x < y);
}
static long uniq_u64(uint64_t *a, long n) {
    if (n == 0) return 0;
    long m = 1;
    for (long i = 1; i < n; i++)
        if (a[i] != a[m - 1]) a[m++] = a[i];
    return m;
}
static long find_u64(const uint64_t *a, long n, uint64_t k) {
    long lo = 0, hi = n - 1;
    while (lo <= hi) {
        long mid = (lo + hi) >> 1;
        if (a[mid] == k) return mid;
        if (a[mid] < k) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}
static uint64_t ekey(int a, int b) {
    if (a > b) { int t = a; a = b; b = t; }
    return ((uint64_t)(uint32_t)a << 32) | (uint32_t)b;
}
static uint64_t fkey(int a, int b, int c) {
    int t;
    if (a > b) { t = a; a = b; b = t; }
    if (b > c) { t = b; b = c; c = t; }
    if (a > b) { t = a; a = b; b = t; }
    return ((uint64_t)a << 42) | ((uint64_t)b << 21) | (uint64_t)c;
}

/* Gauss-Legendre on [0,1], n points, Newton on Legendre recurrence */
static void gauss_legendre01(int n, double *x, double *w) {
    for (int i = 0; i < n; i++) {
        double t = cos(M_PI * (i + 0.75) / (n + 0.5));
        for (int it = 0; it < 100; it++) {
            double p0 = 1.0, p1 = t;
            for (int k = 2; k <= n; k++) {
                double p2 = ((2.0 * k - 1.0) * t * p1 - (k - 1.0) * p0) / k;
                p0 = p1; p1 = p2;
            }
            double dp = n * (t * p1 - p0) / (t * t - 1.0);
            double dt = p1 / dp;
            t -= dt;
            if (fabs(dt) < 1e-16) break;
        }
        double p0 = 1.0, p1 = t;
        for (int k = 2; k <= n; k++) {
            double p2 = ((2.0 * k - 1.0) * t * p1 - (k - 1.0) * p0) / k;
            p0 = p1; p1 = p2;
        }
        double dp = n * (t * p1 - p0) / (t * t - 1.0);
        x[i] = 0.5 * (1.0 - t);
        w[i] = 1.0 / ((1.0 - t * t) * dp * dp); /* = (2/((1-t^2)dp^2))/2 */
    }
}

/* Collapsed (Duffy) rule on the unit simplex; weights normalised to sum 1. */
static void build_quadrature(orc_t *o) {
    double x[NGL], w[NGL];
    gauss_legendre01(NGL, x, w);
    int d = o->dim;
    o->nq = (d == 2) ? NGL * NGL : NGL * NGL * NGL;
    o->ql = (double *)malloc(sizeof(double) * o->nq * (d + 1));
    o->qw = (double *)malloc(sizeof(double) * o->nq);
    int q = 0;
    double sw = 0;
    if (d == 2) {
        for (int i = 0; i < NGL; i++)
            for (int j = 0; j < NGL; j++) {
                double a = x[i], b = x[j];
                double l1 = a, l2 = (1 - a) * b, l0 = 1 - l1 - l2;
                o->ql[3 * q] = l0; o->ql[3 * q + 1] = l1; o->ql[3 * q + 2] = l2;
                o->qw[q] = w[i] * w[j] * (1 - a);
                sw += o->qw[q]; q++;
            }
    } else {
        for (int i = 0; i < NGL; i++)
            for (int j = 0; j < NGL; j++)
                for (int k = 0; k < NGL; k++) {
                    double a = x[i], b = x[j], c = x[k];
                    double l1 = a, l2 = (1 - a) * b, l3 = (1 - a) * (1 - b) * c;
                    double l0 = 1 - l1 - l2 - l3;
                    o->ql[4 * q] = l0; o->ql[4 * q + 1] = l1; o->ql[4 * q + 2] = l2; o->ql[4 * q + 3] = l3;
                    o->qw[q] = w[i] * w[j] * w[k] * (1 - a) * (1 - a) * (1 - b);
                    sw += o->qw[q]; q++;
                }
    }
    for (int i = 0; i < o->nq; i++) o->qw[i] /= sw;
}

/* shape values and d(phi)/d(lambda_a) for barycentrics l[dim+1] */
static void shape(int dim, const double *l, double *phi, double *dphi /* [nld_full][dim+1] */) {
    int nb = dim + 1;
    int nld = (dim == 2) ? 10 : 20;
    memset(dphi, 0, sizeof(double) * nld * nb);
    int k = 0;
    for (int i = 0; i < nb; i++) { phi[k] = l[i]; dphi[k * nb + i] = 1.0; k++; }
    int ne = (dim == 2) ? 3 : 6;
    for (int e = 0; e < ne; e++) {
        int a = (dim == 2) ? E2[e][0] : E3[e][0];
        int b = (dim == 2) ? E2[e][1] : E3[e][1];
        double la = l[a], lb = l[b];
        phi[k] = la * lb; dphi[k * nb + a] = lb; dphi[k * nb + b] = la; k++;
        phi[k] = la * lb * (lb - la);
        dphi[k * nb + a] = lb * lb - 2 * la * lb;
        dphi[k * nb + b] = 2 * la * lb - la * la; k++;
    }
    if (dim == 2) {
        phi[k] = l[0] * l[1] * l[2];
        dphi[k * nb + 0] = l[1] * l[2]; dphi[k * nb + 1] = l[0] * l[2]; dphi[k * nb + 2] = l[0] * l[1]; k++;
    } else {
        for (int f = 0; f < 4; f++) {
            int a = F3[f][0], b = F3[f][1], c = F3[f][2];
            phi[k] = l[a] * l[b] * l[c];
            dphi[k * nb + a] = l[b] * l[c]; dphi[k * nb + b] = l[a] * l[c]; dphi[k * nb + c] = l[a] * l[b]; k++;
        }
    }
}

/* gradients of barycentrics and measure; returns |T| (area / volume) */
static double geom(const orc_t *o, long t, double gl[4][3]) {
    int d = o->dim;
    const int *c = o->conn + t * (d + 1);
    const double *X = o->xyz;
    if (d == 2) {
        double x0 = X[2 * c[0]], y0 = X[2 * c[0] + 1];
        double a11 = X[2 * c[1]] - x0, a21 = X[2 * c[1] + 1] - y0;
        double a12 = X[2 * c[2]] - x0, a22 = X[2 * c[2] + 1] - y0;
        double det = a11 * a22 - a12 * a21;
        /* rows of inverse: grad l1, grad l2 */
        gl[1][0] = a22 / det; gl[1][1] = -a12 / det; gl[1][2] = 0;
        gl[2][0] = -a21 / det; gl[2][1] = a11 / det; gl[2][2] = 0;
        gl[0][0] = -gl[1][0] - gl[2][0]; gl[0][1] = -gl[1][1] - gl[2][1]; gl[0][2] = 0;
        return fabs(det) * 0.5;
    }
    double p0[3] = {X[3 * c[0]], X[3 * c[0] + 1], X[3 * c[0] + 2]};
    double A[3][3];
    for (int j = 0; j < 3; j++)
        for (int i = 0; i < 3; i++) A[i][j] = X[3 * c[j + 1] + i] - p0[i];
    double c00 = A[1][1] * A[2][2] - A[1][2] * A[2][1];
    double c01 = A[1][2] * A[2][0] - A[1][0] * A[2][2];
    double c02 = A[1][0] * A[2][1] - A[1][1] * A[2][0];
    double det = A[0][0] * c00 + A[0][1] * c01 + A[0][2] * c02;
    double inv[3][3];
    inv[0][0] = c00 / det;
    inv[0][1] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) / det;
    inv[0][2] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) / det;
    inv[1][0] = c01 / det;
    inv[1][1] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) / det;
    inv[1][2] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) / det;
    inv[2][0] = c02 / det;
    inv[2][1] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) / det;
    inv[2][2] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) / det;
    for (int a = 0; a < 3; a++)
        for (int i = 0; i < 3; i++) gl[a + 1][i] = inv[a][i];
    for (int i = 0; i < 3; i++) gl[0][i] = -gl[1][i] - gl[2][i] - gl[3][i];
    return fabs(det) / 6.0;
}

/* full element matrix (nld_full x nld_full) by quadrature; ngsolve_functions.py:33-36 */
static void element_matrix(const orc_t *o, long t, double *K) {
    int d = o->dim, nb = d + 1, n = o->nld_full;
    double gl[4][3];
    double vol = geom(o, t, gl);
    double sig = o->sigma[o->mat[t]];
    const int *c = o->conn + t * nb;
    memset(K, 0, sizeof(double) * n * n);
    double phi[MAXLD], dphi[MAXLD * 4], g[MAXLD][3];
    for (int q = 0; q < o->nq; q++) {
        const double *l = o->ql + q * nb;
        shape(d, l, phi, dphi);
        double wgt = o->qw[q] * vol * sig;
        if (d == 2) { /* 2*pi*x weight, x = first coordinate (ngsolve_functions.py:34) */
            double r = 0;
            for (int a = 0; a < nb; a++) r += l[a] * o->xyz[2 * c[a]];
            wgt *= 2.0 * M_PI * r;
        }
        for (int i = 0; i < n; i++)
            for (int k = 0; k < 3; k++) {
                double s = 0;
                for (int a = 0; a < nb; a++) s += dphi[i * nb + a] * gl[a][k];
                g[i][k] = s;
            }
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++)
                K[i * n + j] += wgt * (g[i][0] * g[j][0] + g[i][1] * g[j][1] + g[i][2] * g[j][2]);
    }
}

void orc_destroy(orc_t *o) {
    if (!o) return;
    free(o->xyz); free(o->conn); free(o->mat); free(o->ekeys); free(o->fkeys); free(o->eldof);
    free(o->freeid); free(o->rowptr); free(o->col); free(o->val); free(o->diag); free(o->kib);
    free(o->kii); free(o->sigma); free(o->ql); free(o->qw);
    free(o);
}

static void sort_small(int *a, int n) {
    for (int i = 1; i < n; i++) {
        int v = a[i], j = i - 1;
        while (j >= 0 && a[j] > v) { a[j + 1] = a[j]; j--; }
        a[j + 1] = v;
    }
}

orc_t *orc_create(int dim, long nv, const double *xyz, long nt, const int *conn, const int *mat,
                  long nbf, const int *bconn, const unsigned char *bdir, int nmat,
                  const double *sigma, int condense) {
    orc_t *o = (orc_t *)calloc(1, sizeof(orc_t));
    o->dim = dim; o->nv = nv; o->nt = nt; o->nmat = nmat;
    int nb = dim + 1;
    o->nel = (dim == 2) ? 3 : 6;
    o->nfl = (dim == 2) ? 0 : 4;
    o->nld_full = (dim == 2) ? 10 : 20;
    o->condense = (dim == 2) ? (condense != 0) : 0; /* P3 tets have no interior dofs */
    o->nld = (dim == 2 && o->condense) ? 9 : o->nld_full;
    if (nv >= (1L << 21) && dim == 3) { orc_destroy(o); return NULL; }
    o->xyz = (double *)malloc(sizeof(double) * nv * dim);
    memcpy(o->xyz, xyz, sizeof(double) * nv * dim);
    o->conn = (int *)malloc(sizeof(int) * nt * nb);
    memcpy(o->conn, conn, sizeof(int) * nt * nb);
    o->mat = (int *)malloc(sizeof(int) * nt);
    memcpy(o->mat, mat, sizeof(int) * nt);
    o->sigma = (double *)malloc(sizeof(double) * nmat);
    memcpy(o->sigma, sigma, sizeof(double) * nmat);
    for (long t = 0; t < nt; t++) {
        sort_small(o->conn + t * nb, nb);
        if (o->mat[t] < 0 || o->mat[t] >= nmat) { orc_destroy(o); return NULL; }
    }
    build_quadrature(o);

    /* edges */
    long nek = nt * o->nel;
    o->ekeys = (uint64_t *)malloc(sizeof(uint64_t) * (nek ? nek : 1));
    for (long t = 0; t < nt; t++)
        for (int e = 0; e < o->nel; e++) {
            int a = (dim == 2) ? E2[e][0] : E3[e][0], b = (dim == 2) ? E2[e][1] : E3[e][1];
            o->ekeys[t * o->nel + e] = ekey(o->conn[t * nb + a], o->conn[t * nb + b]);
        }
    qsort(o->ekeys, nek, sizeof(uint64_t), cmp_u64);
    o->ne = uniq_u64(o->ekeys, nek);
    /* faces (3D) */
    if (dim == 3) {
        long nfk = nt * 4;
        o->fkeys = (uint64_t *)malloc(sizeof(uint64_t) * (nfk ? nfk : 1));
        for (long t = 0; t < nt; t++)
            for (int f = 0; f < 4; f++)
                o->fkeys[t * 4 + f] = fkey(o->conn[t * 4 + F3[f][0]], o->conn[t * 4 + F3[f][1]], o->conn[t * 4 + F3[f][2]]);
        qsort(o->fkeys, nfk, sizeof(uint64_t), cmp_u64);
        o->nf = uniq_u64(o->fkeys, nfk);
    }
    long ncell = (dim == 2) ? nt : 0;
    o->ndof_full = nv + 2 * o->ne + o->nf + ncell;
    o->ndof = nv + 2 * o->ne + o->nf + ((dim == 2 && !o->condense) ? nt : 0);

    /* element dof table */
    int n = o->nld_full;
    o->eldof = (int *)malloc(sizeof(int) * nt * n);
    for (long t = 0; t < nt; t++) {
        int *ed = o->eldof + t * n;
        const int *c = o->conn + t * nb;
        int k = 0;
        for (int i = 0; i < nb; i++) ed[k++] = c[i];
        for (int e = 0; e < o->nel; e++) {
            int a = (dim == 2) ? E2[e][0] : E3[e][0], b = (dim == 2) ? E2[e][1] : E3[e][1];
            long id = find_u64(o->ekeys, o->ne, ekey(c[a], c[b]));
            ed[k++] = (int)(nv + 2 * id);
            ed[k++] = (int)(nv + 2 * id + 1);
        }
        if (dim == 3)
            for (int f = 0; f < 4; f++) {
                long id = find_u64(o->fkeys, o->nf, fkey(c[F3[f][0]], c[F3[f][1]], c[F3[f][2]]));
                ed[k++] = (int)(nv + 2 * o->ne + id);
            }
        else
            ed[k++] = (int)(nv + 2 * o->ne + t);
    }

    /* Dirichlet: every dof on a flagged boundary facet (ngsolve_functions.py:27 dirichlet=...) */
    unsigned char *cons = (unsigned char *)calloc(o->ndof_full, 1);
    for (long b = 0; b < nbf; b++) {
        if (!bdir[b]) continue;
        const int *bc = bconn + b * dim;
        for (int i = 0; i < dim; i++) cons[bc[i]] = 1;
        for (int i = 0; i < dim; i++)
            for (int j = i + 1; j < dim; j++) {
                long id = find_u64(o->ekeys, o->ne, ekey(bc[i], bc[j]));
                if (id < 0) { free(cons); orc_destroy(o); return NULL; }
                cons[nv + 2 * id] = 1; cons[nv + 2 * id + 1] = 1;
            }
        if (dim == 3) {
            long id = find_u64(o->fkeys, o->nf, fkey(bc[0], bc[1], bc[2]));
            if (id < 0) { free(cons); orc_destroy(o); return NULL; }
            cons[nv + 2 * o->ne + id] = 1;
        }
    }
    o->freeid = (int *)malloc(sizeof(int) * o->ndof_full);
    long nf_ = 0;
    for (long i = 0; i < o->ndof_full; i++) {
        if (i >= o->ndof) { o->freeid[i] = -1; continue; } /* condensed bubbles */
        o->freeid[i] = cons[i] ? -1 : (int)nf_++;
    }
    o->nfree = nf_;
    free(cons);

    /* CSR pattern: dof -> element adjacency, per-row union */
    int nk = o->nld;
    long *cnt = (long *)calloc(o->nfree + 1, sizeof(long));
    for (long t = 0; t < nt; t++)
        for (int i = 0; i < nk; i++) {
            int r = o->freeid[o->eldof[t * n + i]];
            if (r >= 0) cnt[r + 1]++;
        }
    for (long i = 0; i < o->nfree; i++) cnt[i + 1] += cnt[i];
    int *adj = (int *)malloc(sizeof(int) * (cnt[o->nfree] ? cnt[o->nfree] : 1));
    long *fill = (long *)malloc(sizeof(long) * (o->nfree + 1));
    memcpy(fill, cnt, sizeof(long) * (o->nfree + 1));
    for (long t = 0; t < nt; t++)
        for (int i = 0; i < nk; i++) {
            int r = o->freeid[o->eldof[t * n + i]];
            if (r >= 0) adj[fill[r]++] = (int)t;
        }
    o->rowptr = (long *)calloc(o->nfree + 1, sizeof(long));
    long cap = 1 << 20, ncol = 0;
    int *cols = (int *)malloc(sizeof(int) * cap);
    int tmpcap = 4096;
    int *tmp = (int *)malloc(sizeof(int) * tmpcap);
    for (long r = 0; r < o->nfree; r++) {
        int m = 0;
        long need = (cnt[r + 1] - cnt[r]) * nk;
        if (need > tmpcap) { tmpcap = (int)need * 2; tmp = (int *)realloc(tmp, sizeof(int) * tmpcap); }
        for (long p = cnt[r]; p < cnt[r + 1]; p++) {
            long t = adj[p];
            for (int j = 0; j < nk; j++) {
                int cidx = o->freeid[o->eldof[t * n + j]];
                if (cidx >= 0) tmp[m++] = cidx;
            }
        }
        qsort(tmp, m, sizeof(int), cmp_int);
        int u = 0;
        for (int i = 0; i < m; i++)
            if (i == 0 || tmp[i] != tmp[i - 1]) tmp[u++] = tmp[i];
        if (ncol + u > cap) { while (ncol + u > cap) cap *= 2; cols = (int *)realloc(cols, sizeof(int) * cap); }
        memcpy(cols + ncol, tmp, sizeof(int) * u);
        ncol += u;
        o->rowptr[r + 1] = ncol;
    }
    free(tmp); free(adj); free(fill); free(cnt);
    o->nnz = ncol;
    o->col = cols;
    o->val = (double *)calloc(ncol ? ncol : 1, sizeof(double));
    o->diag = (double *)calloc(o->nfree ? o->nfree : 1, sizeof(double));
    if (o->condense) {
        o->kib = (double *)malloc(sizeof(double) * nt * 9);
        o->kii = (double *)malloc(sizeof(double) * nt);
    }

    /* numeric assembly (a.Assemble(), ngsolve_functions.py:47) */
    double K[MAXLD * MAXLD];
    for (long t = 0; t < nt; t++) {
        element_matrix(o, t, K);
        if (o->condense) { /* Schur complement of the cell bubble (local dof 9) */
            double kii = K[9 * n + 9];
            o->kii[t] = kii;
            for (int j = 0; j < 9; j++) o->kib[t * 9 + j] = K[9 * n + j];
            for (int i = 0; i < 9; i++)
                for (int j = 0; j < 9; j++) K[i * n + j] -= K[i * n + 9] * K[9 * n + j] / kii;
        }
        for (int i = 0; i < nk; i++) {
            int r = o->freeid[o->eldof[t * n + i]];
            if (r < 0) continue;
            for (int j = 0; j < nk; j++) {
                int cidx = o->freeid[o->eldof[t * n + j]];
                if (cidx < 0) continue;
                long lo = o->rowptr[r], hi = o->rowptr[r + 1] - 1;
                while (lo <= hi) {
                    long mid = (lo + hi) >> 1;
                    if (o->col[mid] == cidx) { o->val[mid] += K[i * n + j]; break; }
                    if (o->col[mid] < cidx) lo = mid + 1; else hi = mid - 1;
                }
            }
        }
    }
    for (long r = 0; r < o->nfree; r++)
        for (long p = o->rowptr[r]; p < o->rowptr[r + 1]; p++)
            if (o->col[p] == r) o->diag[r] = o->val[p];
    return o;
}

void orc_sizes(const orc_t *o, long *out /* [8] */) {
    out[0] = o->nv; out[1] = o->ne; out[2] = o->nf; out[3] = o->ndof; out[4] = o->nfree;
    out[5] = o->nnz; out[6] = o->nt; out[7] = o->nld;
}
void orc_get_csr(const orc_t *o, long *rowptr, int *col, double *val) {
    memcpy(rowptr, o->rowptr, sizeof(long) * (o->nfree + 1));
    memcpy(col, o->col, sizeof(int) * o->nnz);
    memcpy(val, o->val, sizeof(double) * o->nnz);
}
void orc_get_freeid(const orc_t *o, int *freeid) { memcpy(freeid, o->freeid, sizeof(int) * o->ndof); }
void orc_get_eldof(const orc_t *o, int *eldof) { memcpy(eldof, o->eldof, sizeof(int) * o->nt * o->nld_full); }

void orc_spmv(const orc_t *o, const double *x, double *y) {
    for (long r = 0; r < o->nfree; r++) {
        double s = 0;
        for (long p = o->rowptr[r]; p < o->rowptr[r + 1]; p++) s += o->val[p] * x[o->col[p]];
        y[r] = s;
    }
}

/* locate the element containing axis point z; lowest element index among candidates.
 * returns element or -1; l = barycentrics */
static long locate(const orc_t *o, double z, double *l) {
    int d = o->dim, nb = d + 1;
    double P[3] = {0, 0, 0};
    if (d == 2) P[1] = z; else P[2] = z;
    const double tol = 1e-10;
    for (long t = 0; t < o->nt; t++) {
        const int *c = o->conn + t * nb;
        /* bounding box reject */
        int out = 0;
        for (int k = 0; k < d && !out; k++) {
            double mn = 1e300, mx = -1e300;
            for (int a = 0; a < nb; a++) {
                double v = o->xyz[d * c[a] + k];
                if (v < mn) mn = v;
                if (v > mx) mx = v;
            }
            double ext = 1e-9 * (1.0 + mx - mn);
            if (P[k] < mn - ext || P[k] > mx + ext) out = 1;
        }
        if (out) continue;
        double gl[4][3];
        geom(o, t, gl);
        const double *x0 = o->xyz + d * c[0];
        double s = 0;
        int ok = 1;
        for (int a = 1; a < nb; a++) {
            double v = 0;
            for (int k = 0; k < d; k++) v += gl[a][k] * (P[k] - x0[k]);
            l[a] = v; s += v;
            if (v < -tol) ok = 0;
        }
        l[0] = 1 - s;
        if (l[0] < -tol) ok = 0;
        if (ok) return t;
    }
    return -1;
}

/* f = sum_s I_s delta(x_s) on free dofs (AddPointSource, ngsolve_functions.py:10-21);
 * with condensation the bubble load is folded: f_b -= K_bi K_ii^-1 f_i.
 * src_elem/src_fi (length nsrc) receive element and interior load for back-substitution. */
int orc_rhs(const orc_t *o, int nsrc, const double *z, const double *I, double *f, long *src_elem,
            double *src_fi) {
    int n = o->nld_full;
    memset(f, 0, sizeof(double) * o->nfree);
    for (int s = 0; s < nsrc; s++) {
        if (src_elem) { src_elem[s] = -1; src_fi[s] = 0; }
        if (I[s] == 0.0) continue; /* ngsolve_functions.py:43 */
        double l[4], phi[MAXLD], dphi[MAXLD * 4];
        long t = locate(o, z[s], l);
        if (t < 0) return -1;
        shape(o->dim, l, phi, dphi);
        for (int i = 0; i < o->nld; i++) {
            int r = o->freeid[o->eldof[t * n + i]];
            if (r >= 0) f[r] += I[s] * phi[i];
        }
        if (o->condense) {
            double fi = I[s] * phi[9];
            for (int j = 0; j < 9; j++) {
                int r = o->freeid[o->eldof[t * n + j]];
                if (r >= 0) f[r] -= o->kib[t * 9 + j] * fi / o->kii[t];
            }
            if (src_elem) { src_elem[s] = t; src_fi[s] = fi; }
        }
    }
    return 0;
}

/* Jacobi-PCG from u0 = 0, stop when sqrt(<Cr,r>) <= rtol*sqrt(<Cr0,r0>)
 * (CGSolver semantics, ngsolve_functions.py:50-51; Preconditioner "local", :46) */
int orc_pcg(const orc_t *o, const double *f, double *u, double rtol, int maxit, int *iters, double *relres) {
    long n = o->nfree;
    double *r = (double *)malloc(sizeof(double) * n), *zv = (double *)malloc(sizeof(double) * n);
    double *p = (double *)malloc(sizeof(double) * n), *q = (double *)malloc(sizeof(double) * n);
    double rz = 0;
    for (long i = 0; i < n; i++) { u[i] = 0; r[i] = f[i]; zv[i] = r[i] / o->diag[i]; p[i] = zv[i]; rz += r[i] * zv[i]; }
    double rz0 = rz;
    int it = 0;
    if (rz0 > 0)
        for (it = 0; it < maxit; it++) {
            if (sqrt(rz) <= rtol * sqrt(rz0)) break;
            orc_spmv(o, p, q);
            double pq = 0;
            for (long i = 0; i < n; i++) pq += p[i] * q[i];
            double alpha = rz / pq, rzn = 0;
            for (long i = 0; i < n; i++) {
                u[i] += alpha * p[i];
                r[i] -= alpha * q[i];
                zv[i] = r[i] / o->diag[i];
                rzn += r[i] * zv[i];
            }
            double beta = rzn / rz;
            rz = rzn;
            for (long i = 0; i < n; i++) p[i] = zv[i] + beta * p[i];
        }
    *iters = it;
    *relres = (rz0 > 0) ? sqrt(rz / rz0) : 0.0;
    free(r); free(zv); free(p); free(q);
    return (rz0 > 0 && sqrt(rz) > rtol * sqrt(rz0)) ? 1 : 0;
}

/* u_h at axis points (worker.py:124-131); bubble recovered per element when condensed
 * (ngsolve_functions.py:53-56): u_i = K_ii^-1 (f_i - K_ib u_b) */
int orc_eval(const orc_t *o, const double *u, int npts, const double *z, double *out, int nsrc,
             const long *src_elem, const double *src_fi) {
    int n = o->nld_full;
    for (int k = 0; k < npts; k++) {
        double l[4], phi[MAXLD], dphi[MAXLD * 4];
        long t = locate(o, z[k], l);
        if (t < 0) { out[k] = NAN; return -1; }
        shape(o->dim, l, phi, dphi);
        double s = 0;
        for (int i = 0; i < o->nld; i++) {
            int r = o->freeid[o->eldof[t * n + i]];
            if (r >= 0) s += u[r] * phi[i];
        }
        if (o->condense) {
            double fi = 0;
            for (int q = 0; q < nsrc; q++)
                if (src_elem && src_elem[q] == t) fi += src_fi[q];
            double acc = fi;
            for (int j = 0; j < 9; j++) {
                int r = o->freeid[o->eldof[t * n + j]];
                if (r >= 0) acc -= o->kib[t * 9 + j] * u[r];
            }
            s += (acc / o->kii[t]) * phi[9];
        }
        out[k] = s;
    }
    return 0;
}

/* one batch: all RHS on the same mesh/sigma (worker.py:104-131).  Same argument meaning as
 * remo_solve_batch in include/remo3d_hip.h.  stats: [0]=total iterations [1]=max relres
 * [2]=nfree [3]=nnz */
int orc_solve_batch(int dim, long nv, const double *xyz, long nt, const int *conn, const int *mat,
                    long nbf, const int *bconn, const unsigned char *bdir, int nmat,
                    const double *sigma, int condense, int nrhs, const int *src_ptr,
                    const double *src_z, const double *src_I, const int *eval_ptr,
                    const double *eval_z, double rtol, int maxit, double *u_out, double *stats) {
    int ntot = eval_ptr[nrhs];
    for (int i = 0; i < ntot; i++) u_out[i] = NAN;
    orc_t *o = orc_create(dim, nv, xyz, nt, conn, mat, nbf, bconn, bdir, nmat, sigma, condense);
    if (!o) return -1;
    double *f = (double *)malloc(sizeof(double) * o->nfree), *u = (double *)malloc(sizeof(double) * o->nfree);
    int rc = 0;
    double its = 0, mres = 0;
    for (int k = 0; k < nrhs && rc >= 0; k++) {
        int ns = src_ptr[k + 1] - src_ptr[k];
        long se[16];
        double sf[16];
        if (ns > 16) { rc = -2; break; }
        if (orc_rhs(o, ns, src_z + src_ptr[k], src_I + src_ptr[k], f, se, sf) < 0) { rc = -3; break; }
        int it;
        double rr;
        int c = orc_pcg(o, f, u, rtol, maxit, &it, &rr);
        if (c > 0) rc = 1;
        its += it;
        if (rr > mres) mres = rr;
        if (orc_eval(o, u, eval_ptr[k + 1] - eval_ptr[k], eval_z + eval_ptr[k], u_out + eval_ptr[k], ns, se, sf) < 0) { rc = -4; break; }
    }
    if (stats) { stats[0] = its; stats[1] = mres; stats[2] = (double)o->nfree; stats[3] = (double)o->nnz; }
    if (rc < 0)
        for (int i = 0; i < ntot; i++) u_out[i] = NAN;
    free(f); free(u);
    orc_destroy(o);
    return rc;
}
