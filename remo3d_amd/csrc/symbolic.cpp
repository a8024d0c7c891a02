// symbolic.cpp — see symbolic.h.  Integer bookkeeping on the host cores of the GPU box; the
// numbering it defines (documented in DESIGN.md) is:
//   vertices 0..nv-1 | two dofs per edge, edges ordered by (min vertex, max vertex) |
//   one dof per face (3D, ordered by sorted vertex triple) or per cell (2D, only if !condense)
//   then constrained dofs are dropped and the rest renumbered in ascending order.
#include "symbolic.h"

#include <algorithm>
#include <cstring>

#include "fem_p3.h"

namespace remo {
namespace {

inline uint64_t ekey(int32_t a, int32_t b) {
    if (a > b) std::swap(a, b);
    return (uint64_t(uint32_t(a)) << 32) | uint32_t(b);
}
struct FKey {
    int32_t a, b, c;
    bool operator<(const FKey &o) const { return a != o.a ? a < o.a : (b != o.b ? b < o.b : c < o.c); }
    bool operator==(const FKey &o) const { return a == o.a && b == o.b && c == o.c; }
};
inline FKey fkey(int32_t a, int32_t b, int32_t c) {
    if (a > b) std::swap(a, b);
    if (b > c) std::swap(b, c);
    if (a > b) std::swap(a, b);
    return FKey{a, b, c};
}

}  // namespace

int build_symbolic(const remo_mesh_t &m, bool condense_in, bool want_pattern, Symbolic &s, std::string &err) {
    const int dim = m.dim;
    if (dim != 2 && dim != 3) { err = "dim must be 2 or 3"; return REMO_ERR_ARG; }
    if (m.n_nodes <= 0 || m.n_elems <= 0 || !m.coords || !m.conn || !m.mat) { err = "empty mesh"; return REMO_ERR_ARG; }
    if (m.n_bfacets > 0 && (!m.bconn || !m.bdirichlet)) { err = "boundary arrays missing"; return REMO_ERR_ARG; }
    const int nb = dim + 1;
    const int64_t nt = m.n_elems, nv = m.n_nodes;
    if (nt >= (int64_t(1) << 27)) { err = "too many elements"; return REMO_ERR_ARG; }
    s = Symbolic();
    s.dim = dim; s.nt = nt; s.nv = nv;
    s.nld_full = (dim == 2) ? 10 : 20;
    s.condense = (dim == 2) && condense_in;  // P3 tets have no interior dofs: no-op in 3D
    s.nld = s.condense ? 9 : s.nld_full;
    const int nel = (dim == 2) ? 3 : 6;

    s.conn.assign(m.conn, m.conn + nt * nb);
    for (int64_t t = 0; t < nt; ++t) {
        int32_t *c = &s.conn[t * nb];
        std::sort(c, c + nb);
        if (c[0] < 0 || c[nb - 1] >= nv) { err = "element vertex index out of range"; return REMO_ERR_MESH; }
        for (int i = 1; i < nb; ++i)
            if (c[i] == c[i - 1]) { err = "element with repeated vertex"; return REMO_ERR_MESH; }
    }

    // edges
    std::vector<uint64_t> ek(size_t(nt) * nel);
    for (int64_t t = 0; t < nt; ++t)
        for (int e = 0; e < nel; ++e)
            ek[t * nel + e] = ekey(s.conn[t * nb + edge_a(dim, e)], s.conn[t * nb + edge_b(dim, e)]);
    std::vector<uint64_t> eu = ek;
    std::sort(eu.begin(), eu.end());
    eu.erase(std::unique(eu.begin(), eu.end()), eu.end());
    s.ne = int64_t(eu.size());
    auto edge_id = [&](uint64_t k) -> int64_t {
        auto it = std::lower_bound(eu.begin(), eu.end(), k);
        return (it != eu.end() && *it == k) ? int64_t(it - eu.begin()) : -1;
    };
    // faces
    std::vector<FKey> fu;
    if (dim == 3) {
        fu.resize(size_t(nt) * 4);
        for (int64_t t = 0; t < nt; ++t)
            for (int f = 0; f < 4; ++f) {
                int a, b, c;
                face_abc(f, a, b, c);
                fu[t * 4 + f] = fkey(s.conn[t * 4 + a], s.conn[t * 4 + b], s.conn[t * 4 + c]);
            }
        std::sort(fu.begin(), fu.end());
        fu.erase(std::unique(fu.begin(), fu.end()), fu.end());
        s.nf = int64_t(fu.size());
    }
    auto face_id = [&](const FKey &k) -> int64_t {
        auto it = std::lower_bound(fu.begin(), fu.end(), k);
        return (it != fu.end() && *it == k) ? int64_t(it - fu.begin()) : -1;
    };

    const int64_t ncell = (dim == 2 && !s.condense) ? nt : 0;
    s.ndof = nv + 2 * s.ne + s.nf + ncell;
    if (s.ndof >= (int64_t(1) << 31)) { err = "too many dofs for 32-bit indices"; return REMO_ERR_ARG; }

    // Dirichlet flags: every dof of a flagged boundary facet
    std::vector<uint8_t> cons(size_t(s.ndof), 0);
    for (int64_t b = 0; b < m.n_bfacets; ++b) {
        const int32_t *bc = m.bconn + b * dim;
        for (int i = 0; i < dim; ++i)
            if (bc[i] < 0 || bc[i] >= nv) { err = "boundary facet vertex out of range"; return REMO_ERR_MESH; }
        if (!m.bdirichlet[b]) continue;
        for (int i = 0; i < dim; ++i) cons[bc[i]] = 1;
        for (int i = 0; i < dim; ++i)
            for (int j = i + 1; j < dim; ++j) {
                const int64_t id = edge_id(ekey(bc[i], bc[j]));
                if (id < 0) { err = "Dirichlet facet edge is not a mesh edge"; return REMO_ERR_MESH; }
                cons[nv + 2 * id] = cons[nv + 2 * id + 1] = 1;
            }
        if (dim == 3) {
            const int64_t id = face_id(fkey(bc[0], bc[1], bc[2]));
            if (id < 0) { err = "Dirichlet facet is not a mesh face"; return REMO_ERR_MESH; }
            cons[nv + 2 * s.ne + id] = 1;
        }
    }
    s.freeid.resize(size_t(s.ndof));
    int64_t nfree = 0;
    for (int64_t i = 0; i < s.ndof; ++i) s.freeid[i] = cons[i] ? -1 : int32_t(nfree++);
    s.nfree = nfree;
    if (nfree == 0) { err = "no free dofs"; return REMO_ERR_MESH; }

    // element -> free rows
    const int n = s.nld_full;
    s.eldof.resize(size_t(nt) * n);
    for (int64_t t = 0; t < nt; ++t) {
        int32_t *ed = &s.eldof[t * n];
        const int32_t *c = &s.conn[t * nb];
        int k = 0;
        for (int i = 0; i < nb; ++i) ed[k++] = s.freeid[c[i]];
        for (int e = 0; e < nel; ++e) {
            const int64_t id = edge_id(ek[t * nel + e]);
            ed[k++] = s.freeid[nv + 2 * id];
            ed[k++] = s.freeid[nv + 2 * id + 1];
        }
        if (dim == 3) {
            for (int f = 0; f < 4; ++f) {
                int a, b, cc;
                face_abc(f, a, b, cc);
                ed[k++] = s.freeid[nv + 2 * s.ne + face_id(fkey(c[a], c[b], c[cc]))];
            }
        } else {
            ed[k++] = s.condense ? -1 : s.freeid[nv + 2 * s.ne + t];
        }
    }
    if (!want_pattern) return REMO_OK;

    // row -> (element, local dof) adjacency by counting sort (ascending element within a row)
    s.adjptr.assign(size_t(nfree) + 1, 0);
    for (int64_t t = 0; t < nt; ++t)
        for (int i = 0; i < s.nld; ++i) {
            const int32_t r = s.eldof[t * n + i];
            if (r >= 0) s.adjptr[r + 1]++;
        }
    for (int64_t r = 0; r < nfree; ++r) s.adjptr[r + 1] += s.adjptr[r];
    s.adj.resize(size_t(s.adjptr[nfree]));
    {
        std::vector<int32_t> fill(s.adjptr.begin(), s.adjptr.end() - 1);
        for (int64_t t = 0; t < nt; ++t)
            for (int i = 0; i < s.nld; ++i) {
                const int32_t r = s.eldof[t * n + i];
                if (r >= 0) s.adj[fill[r]++] = (uint32_t(t) << 5) | uint32_t(i);
            }
    }
    // per-row union of the dofs of the incident elements
    s.rowptr.assign(size_t(nfree) + 1, 0);
    s.col.clear();
    s.col.reserve(size_t(nfree) * (dim == 2 ? 17 : 50));
    std::vector<int32_t> tmp;
    for (int64_t r = 0; r < nfree; ++r) {
        tmp.clear();
        for (int32_t p = s.adjptr[r]; p < s.adjptr[r + 1]; ++p) {
            const int64_t t = s.adj[p] >> 5;
            for (int j = 0; j < s.nld; ++j) {
                const int32_t c = s.eldof[t * n + j];
                if (c >= 0) tmp.push_back(c);
            }
        }
        std::sort(tmp.begin(), tmp.end());
        tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
        if (int64_t(s.col.size()) + int64_t(tmp.size()) >= (int64_t(1) << 31)) { err = "nnz exceeds 32-bit row pointers"; return REMO_ERR_ARG; }
        s.col.insert(s.col.end(), tmp.begin(), tmp.end());
        s.rowptr[r + 1] = int32_t(s.col.size());
    }
    s.nnz = int64_t(s.col.size());
    return REMO_OK;
}

}  // namespace remo
