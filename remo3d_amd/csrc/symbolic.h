// symbolic.h — dof numbering, Dirichlet elimination and CSR pattern of one batch mesh.
// Replaces what H1(mesh, order=3, dirichlet=...) and the sparsity-pattern part of
// BilinearForm.Assemble() do in the reference (ngsolve_functions.py:27, 47).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/remo3d_hip.h"

namespace remo {

struct Symbolic {
    int dim = 0;
    int nld = 0;       // local dofs that are global unknowns (9 when the 2D bubble is condensed)
    int nld_full = 0;  // 10 / 20
    bool condense = false;
    int64_t nv = 0, nt = 0, ne = 0, nf = 0;
    int64_t ndof = 0;   // before Dirichlet elimination
    int64_t nfree = 0;  // rows
    int64_t nnz = 0;
    std::vector<int32_t> conn;    // [nt][dim+1] ascending per element
    std::vector<int32_t> eldof;   // [nt][nld_full] free row of each local dof, -1 if constrained / condensed
    std::vector<int32_t> freeid;  // [ndof] free row or -1
    std::vector<int32_t> rowptr;  // [nfree+1]
    std::vector<int32_t> col;     // [nnz] ascending per row
    std::vector<int32_t> adjptr;  // [nfree+1]
    std::vector<uint32_t> adj;    // row -> (element << 5 | local dof), ascending element
};

// Returns 0 or a REMO_ERR_* code; err receives a message.
int build_symbolic(const remo_mesh_t &mesh, bool condense, bool want_pattern, Symbolic &out, std::string &err);

}  // namespace remo
