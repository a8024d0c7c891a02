// kernels.h — launchers of the gfx950 kernels (kernels.hip).  All launches are asynchronous on
// the given stream; no launcher allocates or synchronises.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace remo {

constexpr int kMaxPartialBlocks = 1024;   // grid cap of every kernel that leaves per-block partial sums
constexpr int kMaxPoints = 256;          // sources + evaluation points of one RHS chunk

// per-iteration record the PCG kernels write straight into mapped host memory
struct PcgProgress {
    double rz[8];        // <Cr,r> per column at the START of step `step`
    int32_t step;        // written last
    int32_t pad;
};

// T = storage type of matrix values and vectors: double (product path) or float (inner solver of the
// mixed-precision mode); scalars and partial sums are double in both
template <class T> struct AmgT;   // amg.h

template <class T> struct PcgBuffersT {
    T *x, *r, *p, *q;       // [n*k]
    const T *dinv;          // [n]
    double *part_pq;        // [kMaxPartialBlocks*8] per-workgroup partial sums of <p, Ap>
    double *part_rz;        // [2][kMaxPartialBlocks*8] per-workgroup partial sums of <Cr, r> (even / odd step)
    double *rz0;            // [48] totals forwarded between launches: <Cr0,r0>[8] | <p,Ap>[8] | <Cr,r> of even / odd steps [8+8] | done flag [8] | floor of <Cr,r> [8]
    // two-level preconditioner (vertex-block Chebyshev); cheb_degree = 0 -> Jacobi
    int64_t nv_coarse;      // free vertex dofs = size of the leading P1 block
    int cheb_degree;
    double cheb_lmax, cheb_lmin;
    T *cz, *cres;           // [nv_coarse*k]
    T *cd[2];               // [nv_coarse*k] ping-pong Chebyshev directions
    // paired Chebyshev steps (launch_vblock_square): the vertex block A_vv and B = A_vv D^-1 A_vv on B's pattern;
    // sq_rowptr = nullptr -> one launch per Chebyshev step on A_vv alone
    const int32_t *sq_rowptr = nullptr, *sq_col = nullptr;
    const T *sq_a = nullptr, *sq_b = nullptr;
    int sq_lanes = 16;      // lanes per row of the paired kernel (8 / 16 / 32 by the average row length of B)
    // compact copy of the vertex block (launch_vblock_compact): inside the full matrix a vertex row's ~15 leading entries sit
    // in a row of ~110, so a Chebyshev launch touches three partly used cache lines per row; nullptr -> read A in place
    const int32_t *vb_rowptr = nullptr, *vb_col = nullptr;
    const T *vb_val = nullptr;
    // fp32 Chebyshev chain inside an fp64 solve (vertex blocks above 32 k rows: the launches are HBM streams there, and a
    // preconditioner may be inexact); nullptr -> the chain runs in T.  Float copies of the compact block's values and of the
    // vertex rows' Jacobi factors, chain vectors [nv_coarse * k]
    const float *c32_val = nullptr, *c32_dinv = nullptr;
    // fixed-width image of the vertex block for the Chebyshev launches (launch_vblock_ell): [nv][kEllWidth] columns and values,
    // [nv][2] begin / end of a row's further entries in the arrays the chain reads otherwise; nullptr -> rows are walked as CSR
    const int32_t *ell_col = nullptr, *ell_tail = nullptr;
    const T *ell_val = nullptr;
    const float *c32_ell_val = nullptr;
    float *c32_z = nullptr, *c32_res = nullptr, *c32_d[2] = {nullptr, nullptr};
    // multigrid cycle on the vertex block instead of the polynomial (amg.h; 2D by default); nullptr -> Chebyshev
    const AmgT<T> *amg = nullptr;
    const AmgT<float> *amg32 = nullptr;   // fp64 solves: the cycle in fp32 storage (remo_debug_tune key 17), nullptr -> in T
    // patch operator only: q = A p is left WITHOUT the sums of the rows shared by several patches - the update launch, which reads
    // q once anyway, forms them from the boundary slab itself (one launch and one pass over the slab less per step).  Set by the
    // host when the operator is the patch operator and the update launch does not gather q (no folded Chebyshev step).
    bool defer_q = false;
    // x += alpha p is formed by the DIRECTION launch of the step (it reads the old p anyway; alpha recomputed there from the same operands:
    // bit-identical x) instead of the update launch, which then reads neither p nor x: one pass over p less per step
    bool x_in_direction = false;
    // patch operator, with defer_q: the patches add their <p, A p> into kPqBins rows of part_pq themselves (atomic adds; two sets of
    // rows taken in turn by step parity, the update launch of a step clears the set of the next) - no launch that folds them
    bool pq_bins = false;
    PcgProgress *progress;  // mapped host records [progress_len]: one per step, the last one is the "all columns frozen" record
    int progress_len;
    int nb_spmv, nb_vec;    // grid sizes actually used (partials valid for these many blocks)
};
using PcgBuffers = PcgBuffersT<double>;
constexpr int kPqBins = 256;       // rows of a set of <p, A p> bins (PcgBuffersT::pq_bins); set s starts at part_pq + s * kPqBins * 8
constexpr int kScalarSlots = 48;   // doubles behind PcgBuffersT::rz0
constexpr int kDoneSlot = 4 * 8;   // rz0[kDoneSlot] (as int): step + 1 of the update launch that froze every column (kernels.hip solve_done)

// Patch operator (3D, remo_opts_t.op = 3; patch.hip): the element list cut into runs of E elements, one workgroup each
struct PatchTables {
    int64_t nt = 0, n = 0, npatch = 0;
    int E = 0;                         // elements per patch = block / right-hand sides of the batch (one lane per element and column)
    int rows_cap = 0;                  // rows of prow / pout per patch
    int block = 256;                   // threads per workgroup the tables were laid out for (256 or 512)
    int trim = 1;                      // 1: a workgroup's staging / output phases stop at its own patch's row count
    int spread = 4;                    // the lanes of a wave take their elements from this many runs of the patch's list (k_patch_apply; <= 1: one run)
    const uint16_t *lidx = nullptr;    // [nt][20] local row of every element dof inside its patch, 0xFFFF = constrained
    const int32_t *pcount = nullptr;   // [npatch] distinct free rows of the patch
    const int32_t *prow = nullptr;     // [npatch][rows_cap] matrix row of local row m, ascending
    const int32_t *pout = nullptr;     // [npatch][rows_cap] -1: the row belongs to this patch alone (result goes to y); else its slot in the boundary slab
    const int32_t *pboff = nullptr;    // [npatch + 1] first slab slot of every patch (its rows' slots are consecutive, in ascending row order)
    int all_slab = 1;                  // 1 (default): EVERY row of a patch goes to the patch's block of the slab, also the rows no other patch touches
                                       //   (pout[p][m] = pboff[p] + m: the kernel stores one contiguous block and y is written by whoever sums the slab);
                                       //   0 (remo_debug_tune key 37, probe builds): only the rows shared by several patches
    const int32_t *row4 = nullptr;     // [n][4] all_slab: the first four slab slots of every row (-1 none, [3] = -2: five or more, the rest in bslot)
    const int32_t *bptr = nullptr;     // [n + 1] entries [bptr[r], bptr[r + 1]) of bslot belong to row r, one per patch that touches it (none: not shared)
    const int32_t *bslot = nullptr;    // slab slot of each (row, patch) pair; the slab itself is patch-major (a patch's shared rows are one block)
    const double *C = nullptr;         // [nt][6] metric terms (launch_metric_terms)
    int64_t nslot_cap = 0;             // upper bound of bptr[n]: rows of the slab
    int stagger = 0;                   // probe builds (remo_debug_tune key 38)
};
template <class T> struct PatchOpT {
    PatchTables t;
    T *Yb = nullptr;                   // [nslot_cap][k] boundary slab
    double *ppart = nullptr;           // [npatch][8] <x, y> of every patch's own rows
    int lds_rows = 0;                  // largest pcount: sizes the kernel's LDS
    bool dot_bins = false;             // PcgBuffersT::pq_bins of the solve that applies it
};
#ifndef REMO_PATCH_PASSES
#define REMO_PATCH_PASSES 12
#endif
constexpr int kPatchPasses = REMO_PATCH_PASSES;   // staging passes a lane's registers hold (k_patch_apply)
// dynamic LDS of k_patch_apply: staged k-wide rows (later the fp64 accumulators; + the zero row and one of slack) and the two
// row tables padded to whole staging passes; a workgroup may ask for 64 KB less the kernel's static 128 k bytes
// (all_slab, the product's form: no row tables in LDS - the row numbers go straight into registers - so five workgroups of a 700-row batch
// share a CU's 160 KB instead of four)
inline size_t patch_lds_bytes(int lds_rows, int k, int block, bool all_slab) {
    const int pass = kPatchPasses * (block / k);
    return size_t(lds_rows + 2) * size_t(k) * 8 + (all_slab ? size_t(0) : size_t((lds_rows + pass - 1) / pass) * pass * 8);
}
constexpr size_t kPatchLdsLimit = 63 * 1024;


template <class T> struct CsrViewT {
    int64_t n;
    int64_t nnz;
    const int32_t *rowptr;
    const int32_t *col;
    const T *val;          // [nnz + 1] allocated: the pair SpMM loads two values per stored entry, so the last entry of a single row reads one element past nnz
    // rows [pair_begin, pair_end) are the two dofs of each free edge, consecutive and with identical
    // column patterns; their VALUES are stored interleaved (launch_assemble).  0, 0 = no pairs, plain CSR
    int64_t pair_begin = 0, pair_end = 0;
    const PatchOpT<T> *patch = nullptr; // != nullptr: launch_spmm applies the patch operator (patch.hip)
    bool vertex_block_only = false;     // rowptr / col / val hold only the leading P1 block (rows and columns < the free vertex count): products need `patch`
};
using CsrView = CsrViewT<double>;

void launch_metric_terms(int dim, int64_t nt, const double *coords, const int32_t *conn, const int32_t *mat, const int32_t *eperm /* or null */,
                         const double *sigma, int nmat, double *C, int32_t *errflag, hipStream_t s);
// rows [pair_begin, pair_end) (the two dofs of every free edge) get their values interleaved: entry e of the first row
// at rowptr[row] + 2e, of the second at rowptr[row] + 2e + 1 (CsrViewT below); all other rows plain CSR
void launch_diag_rows(int dim, int64_t row0, int64_t nfree, const int32_t *adjptr, const uint32_t *adj, const double *C, const double *M, double *dinv, hipStream_t s);
void launch_assemble(int dim, bool condense, int64_t nfree, int64_t pair_begin, int64_t pair_end, const int32_t *rowptr, const int32_t *col,
                     const int32_t *adjptr, const uint32_t *adj, const int32_t *eldof, const double *C,
                     const double *M, double *val, double *dinv, hipStream_t s);

int spmv_grid(int64_t n, int lanes_per_row);
int vec_grid(int64_t n);
void set_fold_first(int v);                 // 1 (default): FIRST Chebyshev step inside the update launch
void set_spmm_tuning(int key, int value);  // 0 variant, 1 lanes per row, 2 threads, 3 mapping, 4 grid (0 = default)
int choose_lanes_per_row(int64_t n, int64_t nnz);
// y = A x for k interleaved columns; if part != nullptr also leaves per-block partial sums of <x_c, y_c>
// scal != nullptr: the launch belongs to PCG step `step` and returns at once when an earlier step froze every column
template <class T> bool patch_applies(const CsrViewT<T> &A, int k);   // launch_spmm will take the patch operator for k columns
// defer = true (patch operator inside the PCG): rows shared by several patches are NOT summed into y (PcgBuffersT::defer_q)
template <class T> void launch_spmm(const CsrViewT<T> &A, int k, const T *x, T *y, double *part, const double *scal, int nblocks, hipStream_t s, int step = 0, bool defer = false);

template <class T> void launch_patch_spmm(const CsrViewT<T> &A, int k, const T *x, T *y, double *part, const double *scal, int nblocks, hipStream_t s, int step, bool defer);   // patch.hip
template <class T> bool pcg_update_folds(const PcgBuffersT<T> &b);   // the update launch takes the first Chebyshev step along (and gathers q)
void set_patch_mode(int mode);
void set_patch_stagger(int units);   // key 38 (probe builds)
void set_patch_all_slab(int on);   // remo_debug_tune key 37 (probe builds): 0 = only the shared rows go through the slab (the form of rounds 3-4)
void set_patch_persist(int on);   // remo_debug_tune key 34 (patch.hip k_patch_apply_p)
void set_patch_wgs_per_xcd(int n);   // key 35
void set_patch_block(int threads);   // 256 (default) or 512
void set_slab_masked(int v);         // key 29
void set_flat_direction(int v);      // key 30
void set_tile_update(int v);         // key 31
void set_slab_ahead(int v);          // 0: slab slots of a shared row one by one in the update launch (default 1: four in flight)
void set_patch_trim(int v);
void set_patch_spread(int v);
void set_patch_lean(int v);          // register-lean arithmetic phase of the patch kernel: -1 fp32 only (default), 0 never, 1 always
void set_patch_slab_rows(int v);     // 1: boundary slab row-major (0: patch-major)
void set_patch_stamps(long long *device_buffer);   // mode 4: [workgroups][8] phase time stamps
int patch_elements_per_group(int kmax);

template <class T> void launch_pcg_init(const CsrViewT<T> &A, int k, const T *f, const PcgBuffersT<T> &b, hipStream_t s);       // + C r0, p0
template <class T> void launch_pcg_update(const CsrViewT<T> &A, int k, int step, double tol2, const PcgBuffersT<T> &b, hipStream_t s);  // + C r (Chebyshev steps)
template <class T> void launch_pcg_direction(const CsrViewT<T> &A, int k, int step, double tol2, const PcgBuffersT<T> &b, hipStream_t s, bool add_x = true);
template <class T> void launch_pcg_final(int k, int step, const PcgBuffersT<T> &b, hipStream_t s);
void launch_vblock_bound(int64_t nv, const CsrView &A, const double *dinv, unsigned long long *out_bits, hipStream_t s);
// B = A_vv D^-1 A_vv (and A_vv itself) on the pattern of B, rows sorted; cnt / rowptr [nv + 1], col / a / b [capacity];
// *flag (device int, pre-zeroed) is raised when a row has more than kSquareSlots distinct columns or the capacity
// is too small: the caller then keeps the one-step path
constexpr int kSquareSlots = 512;
void launch_vblock_square(int64_t nv, const CsrView &A, const double *dinv, int32_t *sq_rowptr, int32_t *sq_col, double *sq_a, double *sq_b,
                          int64_t capacity, int32_t *flag, hipStream_t s);
int cheb_grid(int64_t nv);

// probe of a grid-wide barrier (remo_debug_grid_barrier): nblocks must not exceed what the chip holds at once; counter / fail / mismatch zeroed by the caller
void launch_barrier_probe(int nblocks, int nbar, unsigned *counter, int *fail, float *buf, int *mismatch, hipStream_t s);
constexpr int kEllWidth = 24;   // entries per row of the fixed-width image: three per lane at eight lanes per row (3D P1 rows hold ~15)
// eval64 / eval32: either may be nullptr
void launch_vblock_ell(int64_t nv, const int32_t *rowptr, const int32_t *col, const double *val, int32_t *ecol, int32_t *tail, double *eval64,
                       float *eval32, hipStream_t s);


// mixed precision: conversions around the fp32 inner solve
void launch_to_float(int64_t n, const double *src, float *dst, hipStream_t s);
// compact CSR copy of the leading nv x nv block of A (columns ascend, so the block's entries lead every row); *flag is raised
// when it does not fit `capacity` entries; vb_rowptr[nv] = entries
void launch_vblock_compact(int64_t nv, const CsrView &A, int32_t *vb_rowptr, int32_t *vb_col, double *vb_val, int64_t capacity, int32_t *flag,
                           hipStream_t s);
void launch_mixed_residual(int64_t n, const double *f, const double *q /* may be null */, float *r32, hipStream_t s);
void launch_mixed_accumulate(int64_t n, double *x, float *e, int zero_e, hipStream_t s);
// one PCG step's update with residual replacement (x32 += alpha p; x64 += x32; r32 = f - A64 x64; C r)
void launch_pcg_replace(const CsrViewT<float> &A, const CsrViewT<double> &A64, int k, int step, double tol2, const PcgBuffersT<float> &b,
                        const double *f64, double *x64, double *q64, hipStream_t s);

// point location on the borehole axis + shape values; found[] must be pre-set to INT_MAX
void launch_locate(int dim, int64_t nt, const double *coords, const int32_t *conn, int npts, const double *pz,
                   int32_t *found, hipStream_t s);
void launch_point_shapes(int dim, int npts, const double *pz, const int32_t *found, const double *coords,
                         const int32_t *conn, double *phi, int32_t *errflag, hipStream_t s);
// f[n*k] += I * phi at sources (with the condensed-bubble fold in 2D); bubble loads kept in fint[npts]
void launch_build_rhs(int dim, bool condense, int npts, const int32_t *pt_rhs, const double *pt_I,
                      const int32_t *found, const double *phi, const int32_t *eldof, const double *C,
                      const double *M, int k, double *f, double *fint, hipStream_t s);
void launch_eval(int dim, bool condense, int npts, const int32_t *pt_rhs, const double *pt_I, const int32_t *found,
                 const double *phi, const int32_t *eldof, const double *C, const double *M, int k,
                 const double *x, const double *fint, double *out, hipStream_t s);

}  // namespace remo
