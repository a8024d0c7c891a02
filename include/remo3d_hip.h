/*
 * remo3d_hip.h — C ABI of libremo3d_hip.so: the MI355X (gfx950) replacement for ReMo3D's
 * per-measurement-point FEM hot path.
 *
 * Reference interface replaced (eMWu94/ReMo3D @ 2025-04-04):
 *   remo3d/ngsolve_functions.py:23-58      SolveBVP(mesh, sigma, tool_geometry, source_terms,
 *                                                   dirichlet_boundary, preconditioner, condense)
 *   remo3d/ngsolve_functions_gpu.py:15-54  the same entry, CUDA attempt (the plugin slot)
 *   remo3d/workers/worker.py:100-134       sigma wrapping, loop over the RHS of one batch,
 *                                          evaluation of gfu at the measuring electrodes
 * Because `mesh` and `gfu` are NGSolve objects in the reference, the native boundary sits one
 * step outside SolveBVP: it takes the arrays that define one batch (mesh + sigma + the point
 * sources of every right-hand side + the axis points where the potential is read) and returns
 * the potentials.  See INTEGRATION.md for the ctypes binding a ReMo3D maintainer would add.
 *
 * Conventions: plain pointers and sizes only; the caller owns every buffer; the library never
 * keeps a host pointer past return.  Return codes: 0 ok, >0 warning (REMO_NOT_CONVERGED — the
 * reference is silent about that, ngsolve_functions.py:50), <0 error (outputs NaN-filled, which
 * reproduces the reference's NaN-per-batch convention, worker.py:135-138).  No C++ exception
 * crosses the ABI.  One calling thread per context; contexts are independent (own stream and device arena),
 * so one per GPU per process is the normal case and several may share a GPU (each with its own thread).
 */
#ifndef REMO3D_HIP_H
#define REMO3D_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define REMO_ABI_VERSION 7
#define REMO_MAX_RHS 8 /* right-hand sides solved as one block; longer batches are chunked */

#define REMO_OK 0
#define REMO_NOT_CONVERGED 1
#define REMO_ERR_ARG (-1)
#define REMO_ERR_DEVICE (-2)
#define REMO_ERR_MESH (-3)   /* inconsistent mesh (bad index, boundary facet not in mesh, ...) */
#define REMO_ERR_POINT (-4)  /* source / evaluation point outside the mesh */
#define REMO_ERR_NUMERIC (-5) /* breakdown (non-finite residual) */

/* One batch mesh = what worker.py:100 wraps with ngs.Mesh(): straight-sided simplices. */
typedef struct {
    int32_t dim;               /* 2 = axisymmetric (r, z) half disc; 3 = (x, y, z) half ball      */
    int64_t n_nodes;
    const double *coords;      /* [n_nodes * dim] row-major, metres, batch-centred frame          */
    int64_t n_elems;
    const int32_t *conn;       /* [n_elems * (dim+1)] 0-based vertex numbers, any orientation     */
    const int32_t *mat;        /* [n_elems] 0-based index into sigma (gmsh_functions.py:332-361)  */
    int64_t n_bfacets;
    const int32_t *bconn;      /* [n_bfacets * dim] boundary facets                               */
    const uint8_t *bdirichlet; /* [n_bfacets] 1 => u = 0 there (H1(..., dirichlet=...), :27)       */
} remo_mesh_t;

/* Solver options; zero-initialise then call remo_opts_default(). */
typedef struct {
    int32_t preconditioner; /* 0 = "local" (Jacobi, ngsolve_functions.py:46); 1 = "multigrid": two-level, Chebyshev
                               solve of the P1 (vertex) block + Jacobi on edge/face dofs                  */
    int32_t condense;       /* static condensation of the 2D cell bubble (ngsolve_functions.py:31) */
    int32_t maxsteps;       /* CG step limit, reference 1000 (ngsolve_functions.py:50)             */
    int32_t check_every;    /* host looks at the residual history every this many steps            */
    double rtol;            /* stop when sqrt(<Cr,r>) <= rtol * sqrt(<Cr0,r0>); NGSolve default 1e-8 */
    int32_t time_kernels;   /* k > 0: bracket every k-th SpMV launch of a solve with HIP events (bench roofline); 0 = off */
    int32_t coarse_degree;  /* "multigrid": Chebyshev degree on the vertex block (0 => default: 16 in 2D; in 3D 5 at 1e4 vertices,
                               growing like sqrt(vertices) up to 16) */
    int32_t coarse_ratio;   /* "multigrid": lmax / lmin of the Chebyshev interval (0 => default: 600 in 2D; in 3D 90 at 1e4 vertices, growing like
                               vertices^(2/3)) */
    int32_t precision;      /* 0 = fp64 throughout; 1 = mixed (BASELINE config 5): PCG in fp32 storage inside an fp64
                               residual-refinement loop, stopping test on the true fp64 residual           */
    int32_t inner_digits;   /* mixed: the fp32 residual is replaced by the true fp64 one every time <Cr,r> has gained this
                               many decimal digits (0 => 3; capped at 5, what an fp32 recurrence can hold)   */
    int32_t serialize_solves; /* 1: the solve phase of remo_batch_run (PCG + evaluation) takes a process-wide lock, so that with
                               several contexts driven by several host threads only ONE batch is in its PCG at any time while the
                               others number / assemble theirs beside it (software pipelining across batches); 0: no lock      */
    int32_t op;             /* how the CG applies A (CGSolver's a.mat, ngsolve_functions.py:50-51):
                               3 = patch operator (3D; a 2D batch runs on the CSR product whatever this says): matrix-free - the element
                                   list is cut into patches of 256 / k tetrahedra, a workgroup stages the x rows of its patch in LDS, applies
                                   every K_e through the factorised reference tensors and writes each row once (rows shared by patches
                                   through a compact slab); same operator to rounding; results reproducible to rounding, not bit for bit
                                   (LDS atomics) - the one kernel of the path for which that holds;
                               2 = CSR SpMM on the assembled matrix (bit-reproducible);
                               0 = default: 3 in 3D (at the reference's resolution an application takes ~100 us against 370-450 us
                                   of the CSR product and moves 0.4 GB instead of 1.25 GB), 2 in 2D.
                               (1 was the round-2 element-wise operator with a slab of element results: removed with ABI 7, REMO_ERR_ARG.)
                               What is assembled: remo_opts_t.assemble */
    int32_t coarse;         /* "multigrid": the solver of the P1 (vertex) block.
                               1 = Chebyshev polynomial (coarse_degree, coarse_ratio);
                               2 = one V(1,1) cycle of a smoothed-aggregation multigrid hierarchy built per batch on the device
                                   (distance-2 independent-set aggregates, Galerkin products, dense coarsest solve);
                               0 = by dimension (default; a coarse_degree > 0 selects the polynomial): the cycle in 2D (graded axisymmetric meshes: 40 % fewer PCG steps than
                                   the polynomial of degree 28 at a quarter of its launches), the polynomial in 3D (there it is
                                   within 15 % of an exact vertex solve at degree 5-13 and the cycle gains nothing).
                               3 = the cycle if its hierarchy can be built, else the polynomial (coarse_degree / coarse_ratio): what `Model` asks
                                   for on the graded, sheared interface-conforming 3D meshes it builds, where the cycle takes 17 % less
                                   time than the best polynomial (187 against 198 steps at two thirds of the launches).
                               If the hierarchy cannot be built (a vertex of extreme valence) 0 and 3 fall back to the polynomial, 2 fails */
    int32_t assemble;       /* what a 3D batch that runs on the patch operator assembles (CGSolver's a.mat is never read there):
                               0 = by size (default): the whole matrix up to 200 k tetrahedra (inspection hooks, small cost), above that only
                                   the Jacobi diagonal and the P1 (vertex) block the preconditioner solves - pattern and values of a
                                   2 M-row matrix are 4.5 ms of an 80 ms batch and 1.2 GB;
                               1 = always the whole matrix; 2 = diagonal + P1 block whenever the patch operator runs.
                               Without the matrix remo_batch_get_system (rowptr / col / val) and products with more columns than the
                               batch has right-hand sides fail with REMO_ERR_ARG (remo_stats_t.assembled says which it was); op = 2 always assembles */
    int32_t quadrature;     /* 2D: how the reference tensors of `2 pi x sigma grad(u) grad(v)` (ngsolve_functions.py:34; a degree-5 integrand) are
                               integrated: 0 = exactly (default); 1 = by the 6-point rule that is exact to degree 4 - the alternative NGSolve
                               may be using (its rule order is not pinned by the reference).  3D integrands have degree 4: always exact */
} remo_opts_t;

typedef struct {
    int64_t n_dof;   /* dofs before Dirichlet elimination (condensed bubbles not counted)          */
    int64_t n_free;  /* rows of the linear system                                                 */
    int64_t nnz;     /* stored entries of the CSR matrix                                          */
    int64_t n_edges, n_faces;
    int32_t n_rhs;
    int32_t max_iterations;              /* over the RHS                                           */
    int32_t iterations[REMO_MAX_RHS];    /* of the last chunk                                      */
    double relres[REMO_MAX_RHS];         /* sqrt(<Cr,r>/<Cr0,r0>) of the last chunk                */
    double ms_symbolic;  /* dof numbering + CSR pattern                                            */
    double ms_h2d;       /* uploads                                                                */
    double ms_assemble;  /* geometry terms + CSR values (device, HIP events)                       */
    double ms_solve;     /* PCG, all RHS (device, HIP events)                                      */
    double ms_eval;      /* point location + RHS build + evaluation                                */
    double ms_total;     /* wall clock of the call                                                 */
    double spmv_ms;      /* sum of the event-timed SpMV launches (time_kernels > 0), minus the bracket
                            overhead below per launch                                             */
    int64_t spmv_launches;
    double spmv_bytes;   /* algorithmic bytes of ONE operator application: CSR product 12 nnz + 4 n + 16 k n;
                            patch operator 16 k n + 88 T (x read and y written once, 40 B of local indices + 48 B of
                            metric terms per element); fp32 storage (mixed): 8 nnz + 4 n + 8 k n / 8 k n + 88 T        */
    int64_t pcg_steps;   /* total PCG steps executed on the device (incl. post-convergence slack)  */
    double spmv_ms_raw;  /* the same sum before the correction                                     */
    double event_overhead_ms; /* elapsed time of an EMPTY hipEvent pair on the stream (min of 16),
                            i.e. what a bracket measures beyond the kernel it encloses             */
    int64_t refinement_cycles; /* mixed precision: fp32 inner solves that contributed a correction (all chunks) */
    int32_t op_used;     /* 0 = the CG applied A as a CSR SpMM, 3 = patch operator (remo_opts_t.op) */
    int32_t coarse_used; /* vertex-block solver of the run: 0 = none ("local"), 1 = Chebyshev polynomial, 2 = multigrid cycle */
    int32_t assembled;   /* 1 = the whole matrix, 2 = only the Jacobi diagonal and the P1 (vertex) block (remo_opts_t.assemble):
                            then remo_batch_get_system hands out dinv / freeid only and nnz reads 0                  */
    int32_t reserved_;
} remo_stats_t;

typedef struct remo_ctx remo_ctx_t;
typedef struct remo_batch remo_batch_t;

int remo_abi_version(void);
void remo_opts_default(remo_opts_t *opts);

/* One context per GPU (hipSetDevice(device_id), one stream, a grow-only device arena). */
remo_ctx_t *remo_ctx_create(int device_id);
void remo_ctx_destroy(remo_ctx_t *ctx);
/* Text of the last error on this context (or of the failed remo_ctx_create when ctx == NULL). */
const char *remo_last_error(remo_ctx_t *ctx);

/*
 * Whole batch in one call: the inner hot loop of worker.py:104-131.
 * RHS k has point sources src_z[src_ptr[k] .. src_ptr[k+1]) on the borehole axis with strengths
 * src_I (zero strengths are skipped, ngsolve_functions.py:43) and is read at axis positions
 * eval_z[eval_ptr[k] .. eval_ptr[k+1]).  u_out[eval_ptr[n_rhs]] receives u_h at those points
 * (the raw FE potential; the 1/2 of the 3D half-space model and the geometric factor are applied
 * by the caller as in worker.py:124-131).
 */
int remo_solve_batch(remo_ctx_t *ctx, const remo_mesh_t *mesh, int32_t n_mat, const double *sigma,
                     int32_t n_rhs, const int32_t *src_ptr, const double *src_z, const double *src_I,
                     const int32_t *eval_ptr, const double *eval_z, double *u_out,
                     const remo_opts_t *opts, remo_stats_t *stats);

/*
 * Staged form of the same work, for callers that keep a batch resident (bench.py: inputs are in
 * HBM before the timed region).  create = validate + upload; run = numbering, pattern, assembly,
 * PCG, evaluation, all RHS; fetch = potentials to the host.
 */
int remo_batch_create(remo_ctx_t *ctx, const remo_mesh_t *mesh, int32_t n_mat, const double *sigma,
                      int32_t n_rhs, const int32_t *src_ptr, const double *src_z, const double *src_I,
                      const int32_t *eval_ptr, const double *eval_z, remo_batch_t **out);
int remo_batch_run(remo_ctx_t *ctx, remo_batch_t *batch, const remo_opts_t *opts, remo_stats_t *stats);
int remo_batch_fetch(remo_ctx_t *ctx, remo_batch_t *batch, double *u_out);
void remo_batch_destroy(remo_ctx_t *ctx, remo_batch_t *batch);

/*
 * u_h of right-hand side `rhs` of the last remo_batch_run at further axis points - the
 * `gfu(mesh(0.0, z))` / `gfu(mesh(0.0, 0.0, z))` of worker.py:124-131 for callers that keep the
 * solution object around (remo3d_amd/ngsolve_functions_hip.py).  Valid until the next run on the
 * context, for batches of at most REMO_MAX_RHS right-hand sides.  Points outside the mesh give
 * NaN and REMO_ERR_POINT.
 */
int remo_batch_eval(remo_ctx_t *ctx, remo_batch_t *batch, int32_t rhs, int32_t n_points, const double *z, double *u_out);

/*
 * Inspection hooks used by the parity tests (tests/): the assembled system of the last
 * remo_batch_run on this batch.  Pass NULL to skip an array.  rowptr[n_free+1], col[nnz],
 * val[nnz], dinv[n_free] (Jacobi), freeid[n_dof] (free row of each dof or -1).
 */
int remo_batch_get_system(remo_ctx_t *ctx, remo_batch_t *batch, int32_t *rowptr, int32_t *col,
                          double *val, double *dinv, int32_t *freeid);
/* Solution and load vectors of the LAST chunk of right-hand sides of the last remo_batch_run (chunks hold at most
 * REMO_MAX_RHS columns): x[n_free * k], f[n_free * k], row-major with k = *k_out columns.  Either pointer may be NULL.
 * With remo_batch_spmv this lets a test measure the TRUE residual f - A x of what CGSolver's silent stopping rule
 * (ngsolve_functions.py:50-51) left behind. */
int remo_batch_get_vectors(remo_ctx_t *ctx, remo_batch_t *batch, double *x, double *f, int32_t *k_out);
/* Inspection hook (tests): z = one application of the multigrid cycle the last run built on the P1 block (remo_opts_t.coarse;
 * replaces the "multigrid" preconditioner object of ngsolve_functions.py:46 on that block) to k column-interleaved vectors
 * r[nv][k]; fp32 != 0: through the fp32 image of the hierarchy (what fp64 solves use by default).  nv_out receives the block size
 * (r and z may be NULL to ask for it).  Fails when the last run used the polynomial. */
int remo_batch_apply_coarse(remo_ctx_t *ctx, remo_batch_t *batch, int32_t k, const double *r, double *z, int32_t fp32, int64_t *nv_out);

/* y = A x on the device with the batch's matrix, k interleaved columns (x[n_free*k] row-major);
 * reps >= 1 launches are timed with HIP events, average ms returned in *ms_avg. */
int remo_batch_spmv(remo_ctx_t *ctx, remo_batch_t *batch, int32_t k, const double *x, double *y,
                    int32_t reps, double *ms_avg);

/*
 * Host-side pieces exposed for CPU-only tests (no GPU needed):
 *  - the pre-integrated reference tensors contracted with one element's metric terms, i.e. the
 *    element matrix the assembly kernel gathers from (nld x nld, nld = 10 or 20);
 *  - dof numbering + CSR pattern of a mesh.
 */
int remo_host_element_matrix(int32_t dim, const double *vertex_coords /*[(dim+1)*dim], sorted vertices*/,
                             double sigma, double *K_out);
/* max |sum_m B_a[m][i] B_b[m][j] - M_ab[i][j]|: how well the factorised reference tensors of the patch operator
 * (remo_opts_t.op = 3) reproduce the tensors the CSR assembly contracts (both exact polynomial integrals). */
double remo_host_factor_error(void);
int remo_host_symbolic(const remo_mesh_t *mesh, int32_t condense, int64_t *sizes /*[6]: n_dof,n_free,nnz,n_edges,n_faces,nld*/,
                       int32_t *rowptr /*[n_free+1] or NULL*/, int32_t *col /*[nnz] or NULL*/,
                       int32_t *freeid /*[n_dof] or NULL*/);

#ifdef __cplusplus
}
#endif
#endif /* REMO3D_HIP_H */
