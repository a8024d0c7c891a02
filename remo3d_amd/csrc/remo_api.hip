// remo_api.hip — the C ABI of include/remo3d_hip.h: context, resident batches, and the host
// orchestration of one batch (numbering -> upload -> assembly -> multi-RHS PCG -> evaluation).
// The orchestration mirrors the inner loop of remo3d/workers/worker.py:100-134 with one
// difference the reference leaves on the table: the matrix of a batch is
// assembled once and all its right-hand sides are solved together.
#include <hip/hip_runtime.h>
#include <limits.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/remo3d_hip.h"
#include "../../include/remo3d_hip_debug.h"
#include "fem_p3.h"
#include "kernels.h"
#include "amg.h"
#include "patch.h"
#include "symbolic.h"
#include "symbolic_gpu.h"

using namespace remo;

namespace {

thread_local std::string g_create_error;

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e__ = (expr);                                                                        \
        if (e__ != hipSuccess) {                                                                        \
            char buf__[512];                                                                            \
            snprintf(buf__, sizeof buf__, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            throw std::runtime_error(buf__);                                                            \
        }                                                                                               \
    } while (0)

size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

}  // namespace

struct remo_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    Arena ar;
    double *d_M2 = nullptr, *d_M3 = nullptr, *d_M2q = nullptr;   // reference tensors: exact 2D / 3D, 2D by the degree-4 rule
    PcgProgress *progress = nullptr;  // mapped, coherent host memory
    PcgProgress *progress_dev = nullptr;
    int progress_len = 0;
    int32_t *d_err = nullptr;
    hipEvent_t ev[8] = {};
    std::vector<hipEvent_t> spmv_ev;
    uint64_t run_id = 0;  // the arena holds the system / solution of the batch that ran last
    // input pool of the one-shot entry (remo_solve_batch): the mesh arrays of the batch in hand, grow-only - a sweep of thousands of
    // batches then makes no hipMalloc / hipFree per batch (hipFree synchronises the whole device, i.e. the other contexts' streams)
    char *in_pool = nullptr;
    size_t in_cap = 0;
    double floor_stage[REMO_MAX_RHS] = {};   // host staging of the mixed mode's <Cr,r> floors (outlives the async copy)

    template <class T> T *take(size_t count) { return ar.lo<T>(count); }
    void reserve(size_t bytes) {
        ar.reset();
        if (bytes <= ar.cap) return;
        HIP_TRY(hipStreamSynchronize(stream));
        if (ar.base) HIP_TRY(hipFree(ar.base));
        ar.base = nullptr;
        ar.cap = 0;
        const size_t want = align_up(bytes + bytes / 4, 4096);  // the top-down end must stay aligned too
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&ar.base), want));
        ar.cap = want;
    }
    void ensure_progress(int len) {
        if (len <= progress_len) return;
        if (progress) HIP_TRY(hipHostFree(progress));
        progress = nullptr;
        progress_len = 0;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&progress), sizeof(PcgProgress) * size_t(len),
                              hipHostMallocMapped | hipHostMallocCoherent));
        HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&progress_dev), progress, 0));
        progress_len = len;
    }
};

struct remo_batch {
    int dim = 0;
    int64_t nv = 0, nt = 0, nbf = 0;
    int n_mat = 0;
    // points: per chunk [sources..., evals...]
    int n_rhs = 0;
    std::vector<int32_t> src_ptr, eval_ptr;
    std::vector<double> src_z, src_I, eval_z;
    // resident device inputs (everything the path reads is in HBM before remo_batch_run)
    double *d_coords = nullptr, *d_sigma = nullptr;
    int32_t *d_mat = nullptr, *d_conn = nullptr, *d_bconn = nullptr;
    uint8_t *d_bdir = nullptr;
    bool pooled = false;     // the six arrays live in the context's input pool (remo_solve_batch): not freed with the batch
    // last system (pointers into the context arena; valid until the next run on the context)
    bool has_system = false;
    DeviceSymbolic sym;
    CsrView A{};
    double *d_val = nullptr, *d_dinv = nullptr;
    double *d_x = nullptr, *d_C = nullptr;  // solution block [n][k_last] and metric terms of the last run
    double *d_f = nullptr;                  // load vectors [n][k_last] of the last chunk
    PatchOpT<double> patch64{};             // patch operator of the last run (remo_opts_t.op = 3), pointers into the arena
    PatchOpT<float> patch32{};
    AmgT<double> amg64{};                   // multigrid hierarchy of the vertex block of the last run (arena)
    AmgT<float> amg32{};
    int k_last = 0;
    const double *d_M_last = nullptr;       // reference tensors of the last run (remo_opts_t.quadrature)
    uint64_t run_id = 0;
    std::vector<double> u_out;
};

namespace {

int fail(remo_ctx *ctx, int code, const std::string &msg) {
    if (ctx) ctx->err = msg;
    return code;
}

struct ChunkResult {
    int steps = 0;
    bool converged = false;
    bool finite = true;
    int iters[REMO_MAX_RHS];
    double relres[REMO_MAX_RHS];
};

// One PCG solve in storage type T.  tol2: relative target on <Cr,r> (w.r.t. this solve's own start);
// floor: optional absolute per-column floor of <Cr,r> (mixed mode: the outer target).  rz_first /
// rz_last return <Cr,r> at the start and at the end.
std::mutex g_solve_mutex;   // remo_opts_t.serialize_solves
int g_square = 1;   // remo_debug_tune key 6: 0 = one launch per Chebyshev step, 1 = paired steps in 2D, 2 = paired steps always
int g_sq_lanes = 0;  // key 7: lanes per row of the paired kernel (0 = by row length)
int g_amg32 = 1;     // key 17: 1 = fp64 solves run the multigrid cycle in fp32 storage (default), 0 = in fp64
int g_amg = 0;       // key 16: 0 = remo_opts_t.coarse decides, 1 = never the multigrid cycle, 2 = always (any dimension)
int g_chain32 = 1;   // key 15: 1 = fp32 Chebyshev chain inside fp64 solves above 32 k vertex rows (default), 0 = chain in fp64
int g_ell = 1;        // key 24: 1 = the Chebyshev launches of 3D read the fixed-width image of the vertex block (default), 0 = its CSR form
int g_dot_bins = 1;   // key 28: 1 = the patches add their <p, A p> straight into the update launch's rows (default), 0 = a row per patch + k_patch_dot
int g_x_in_direction = 1;   // key 25: 1 = x += alpha p formed by the direction launch of the step (default), 0 = by the update launch
int g_defer_q = 1;    // key 22: 1 = the PCG's update launch sums the patch operator's shared rows itself (default), 0 = k_patch_reduce does
#ifdef REMO_PROBES
int g_extra_apply = 0; // key 36 (probe builds): extra operator applications (apply + shared-row sums, results discarded) per PCG step: what a step with more applications would cost
#endif
int g_compact = 1;   // key 13: 1 = Chebyshev launches read a compact copy of the vertex block, 0 = the leading entries of A's rows in place
constexpr int64_t kCompactPerRow = 48;   // capacity of the compact copy per vertex (3D P1 rows hold ~15 entries; a copy that does not fit is not used)

// fp64 side of a mixed-precision inner solve: where the residual replacements read and write
struct RefineHooks {
    const CsrView *A64 = nullptr;
    const double *f64 = nullptr;
    double *x64 = nullptr, *q64 = nullptr;
    double factor2 = 1e-6;   // replace once <Cr,r> of some column has dropped by this factor since the last replacement
    int replacements = 0;
};

template <class T>
ChunkResult run_pcg_t(remo_ctx *ctx, const CsrViewT<T> &A, int k, const T *d_f, PcgBuffersT<T> &buf, double tol2, const double *floor,
                      int maxit, int check, int time_kernels, remo_stats_t *st, size_t &ev_used, double *rz_first, double *rz_last,
                      RefineHooks *hooks = nullptr) {
    ChunkResult res;
    bool replace_next = false, have_ref = false;
    double rz_ref[REMO_MAX_RHS] = {0};
    hipStream_t s = ctx->stream;
    if (check <= 0) check = 10;
    for (int i = 0; i < ctx->progress_len; ++i) ctx->progress[i].step = -1;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    HIP_TRY(hipMemsetAsync(buf.rz0, 0, kScalarSlots * sizeof(double), s));   // forwarded totals + done flag + floor
    if (buf.defer_q) HIP_TRY(hipMemsetAsync(buf.part_pq, 0, sizeof(double) * kMaxPartialBlocks * 8, s));   // the patch operator's dot launch fills only its first rows
    if (floor) {
        std::memcpy(ctx->floor_stage, floor, sizeof(double) * REMO_MAX_RHS);
        HIP_TRY(hipMemcpyAsync(buf.rz0 + 5 * 8, ctx->floor_stage, sizeof(double) * REMO_MAX_RHS, hipMemcpyHostToDevice, s));
    }
    launch_pcg_init(A, k, d_f, buf, s);
    volatile int32_t *done_step = &ctx->progress[ctx->progress_len - 1].step;
    int step = 0;
    bool done = false;
    for (; step < maxit && !done;) {
        // time_kernels = k: every k-th SpMM launch is bracketed with events (a bracket costs the stream ~1.5 us)
        if (time_kernels > 0 && (step % time_kernels) == (time_kernels / 2) && ev_used + 2 <= ctx->spmv_ev.size()) {
            HIP_TRY(hipEventRecord(ctx->spmv_ev[ev_used], s));
            launch_spmm(A, k, (const T *)buf.p, buf.q, buf.part_pq, (const double *)buf.rz0, buf.nb_spmv, s, step, buf.defer_q);
            HIP_TRY(hipEventRecord(ctx->spmv_ev[ev_used + 1], s));
            ev_used += 2;
        } else {
            launch_spmm(A, k, (const T *)buf.p, buf.q, buf.part_pq, (const double *)buf.rz0, buf.nb_spmv, s, step, buf.defer_q);
        }
#ifdef REMO_PROBES
        for (int extra = 0; extra < g_extra_apply; ++extra)      // (idempotent: the same q and slab again, no dot products)
            launch_spmm(A, k, (const T *)buf.p, buf.q, (double *)nullptr, (const double *)buf.rz0, buf.nb_spmv, s, step, false);
#endif
        bool replaced = false;
        if constexpr (std::is_same<T, float>::value) {
            if (hooks && replace_next) {
                launch_pcg_replace(A, *hooks->A64, k, step, tol2, buf, hooks->f64, hooks->x64, hooks->q64, s);
                hooks->replacements += 1;
                replace_next = false;
                replaced = true;
            }
        }
        if (!replaced) launch_pcg_update(A, k, step, tol2, buf, s);
        launch_pcg_direction(A, k, step, tol2, buf, s, !replaced);
        ++step;
        if (*done_step >= 0) { done = true; break; }   // the device froze every column: the queued launches are no-ops
        if (step % check == 0) {
            // stay one check interval ahead of the device (a step is ~10 launches, ~30 us of host time
            // against ~150 us on the device); the wait also ends when the "done" record appears
            const int target = step - check;
            if (target >= 0) {
                volatile int32_t *flag = &ctx->progress[target % (ctx->progress_len - 1)].step;
                const double t0 = now_ms();
                int spins = 0;
                while (*flag != target && *done_step < 0) {
                    if (++spins > 64) {
                        std::this_thread::yield();
                        if (now_ms() - t0 > 2000.0) {
                            HIP_TRY(hipStreamSynchronize(s));
                            if (*flag != target && *done_step < 0) throw std::runtime_error("PCG progress record not visible to the host");
                        }
                    }
                }
                std::atomic_thread_fence(std::memory_order_acquire);
                if (*done_step >= 0) { done = true; break; }
                const PcgProgress &pr = ctx->progress[target % (ctx->progress_len - 1)];
                for (int c = 0; c < k; ++c)
                    if (!std::isfinite(pr.rz[c])) { res.finite = false; done = true; }
                if (hooks) {   // schedule a residual replacement when the (lagged) history has dropped far enough
                    if (!have_ref) {
                        for (int c = 0; c < k; ++c) rz_ref[c] = pr.rz[c];
                        have_ref = true;
                    } else {
                        bool hit = false;
                        for (int c = 0; c < k; ++c)
                            if (rz_ref[c] > 0.0 && pr.rz[c] > 0.0 && pr.rz[c] <= hooks->factor2 * rz_ref[c]) hit = true;
                        if (hit) {
                            replace_next = true;
                            for (int c = 0; c < k; ++c) rz_ref[c] = pr.rz[c];
                        }
                    }
                }
            }
        }
    }
    launch_pcg_final(k, step, buf, s);
    HIP_TRY(hipStreamSynchronize(s));
    std::atomic_thread_fence(std::memory_order_acquire);
    const PcgProgress &dn = ctx->progress[ctx->progress_len - 1];
    const int last = (dn.step >= 0) ? dn.step : step;   // index of the record that holds the final <Cr,r>
    const PcgProgress &fin = (dn.step >= 0) ? dn : ctx->progress[step % (ctx->progress_len - 1)];
    res.steps = (dn.step >= 0) ? dn.step : step;
    const PcgProgress &p0 = (last == 0) ? fin : ctx->progress[0];
    res.converged = true;
    for (int c = 0; c < k; ++c) {
        res.iters[c] = last;
        const double r0 = p0.rz[c];
        const double thr = std::max(tol2 * r0, floor ? floor[c] : 0.0);
        for (int i = 0; i <= last; ++i) {
            const PcgProgress &pr = (i == last) ? fin : ctx->progress[i % (ctx->progress_len - 1)];
            if (i != last && pr.step != i) continue;
            if (!std::isfinite(pr.rz[c])) res.finite = false;
            if (!(pr.rz[c] > thr)) { res.iters[c] = i; break; }
        }
        const double rl = fin.rz[c];
        res.relres[c] = (r0 > 0.0) ? std::sqrt(rl / r0) : 0.0;
        if (rl > thr) res.converged = false;
        if (!std::isfinite(rl)) res.finite = false;
        if (rz_first) rz_first[c] = r0;
        if (rz_last) rz_last[c] = rl;
    }
    if (st) st->pcg_steps += res.steps;
    return res;
}

ChunkResult run_pcg(remo_ctx *ctx, const CsrView &A, int k, const double *d_f, PcgBuffers &buf, const remo_opts_t &o,
                    remo_stats_t *st, size_t &ev_used) {
    return run_pcg_t<double>(ctx, A, k, d_f, buf, o.rtol * o.rtol, nullptr, o.maxsteps, o.check_every, o.time_kernels, st, ev_used, nullptr,
                             nullptr);
}

// Mixed precision (BASELINE config 5): PCG runs in fp32 storage (matrix values, vectors,
// preconditioner; scalars fp64) and its residual is refreshed from fp64 as it goes.  Every time <Cr,r>
// of a column has dropped by `inner_digits` decimal digits, the step's update is replaced by
//   x64 += x32, x32 = 0, r32 = float(f - A64 x64)            (launch_pcg_replace)
// while the search direction and the scalars carry on (residual replacement: the Krylov process is NOT
// restarted, which restart-style refinement pays for with 30-60 % more steps on these matrices).
// When the recurrence says converged, an outer cycle re-measures the TRUE residual in fp64; the solve
// ends with a cycle whose START already meets the target (normally the second one, at the cost of one
// fp64 SpMM and one inner step).  <Cr,r> is measured with the fp32 preconditioner.
struct MixedBuffers {
    CsrViewT<float> A32{};
    PcgBuffersT<float> b32{};
    float *f32 = nullptr;
};

ChunkResult run_pcg_mixed(remo_ctx *ctx, const CsrView &A, int k, const double *d_f, PcgBuffers &buf, MixedBuffers &mx, const remo_opts_t &o,
                          remo_stats_t *st, size_t &ev_used) {
    hipStream_t s = ctx->stream;
    const int64_t nk = A.n * k;
    const double tol2 = o.rtol * o.rtol;
    const int digits = o.inner_digits > 0 ? std::min(o.inner_digits, 5) : 3;   // fp32 recurrences do not hold more than ~5 digits
    const double tol2_in = std::pow(10.0, -2.0 * digits);
    ChunkResult out;
    out.converged = false;
    for (int c = 0; c < REMO_MAX_RHS; ++c) { out.iters[c] = 0; out.relres[c] = 0.0; }
    double rz0g[REMO_MAX_RHS] = {0}, floor[REMO_MAX_RHS] = {0}, first[REMO_MAX_RHS], last[REMO_MAX_RHS];
    HIP_TRY(hipMemsetAsync(buf.x, 0, sizeof(double) * nk, s));
    int total = 0;
    const int max_cycles = 40;
    for (int cycle = 0; cycle < max_cycles; ++cycle) {
        if (cycle == 0) {
            launch_mixed_residual(nk, d_f, nullptr, mx.f32, s);
        } else {
            launch_spmm(A, k, (const double *)buf.x, buf.q, (double *)nullptr, (const double *)nullptr, buf.nb_spmv, s);
            launch_mixed_residual(nk, d_f, buf.q, mx.f32, s);
        }
        const int budget = std::max(1, o.maxsteps - total);
        RefineHooks hooks;
        hooks.A64 = &A; hooks.f64 = d_f; hooks.x64 = buf.x; hooks.q64 = buf.q; hooks.factor2 = tol2_in;
        ChunkResult in = run_pcg_t<float>(ctx, mx.A32, k, mx.f32, mx.b32, 0.5 * tol2, cycle ? floor : nullptr, budget, o.check_every,
                                          o.time_kernels, st, ev_used, first, last, &hooks);
        if (st) st->refinement_cycles += hooks.replacements;
        if (cycle == 0)
            for (int c = 0; c < k; ++c) { rz0g[c] = first[c]; floor[c] = 0.5 * tol2 * rz0g[c]; }   // inner target: 0.7 of the outer one in norm
        out.finite = out.finite && in.finite;
        if (!in.finite) break;
        bool met = true;
        for (int c = 0; c < k; ++c) {
            out.relres[c] = rz0g[c] > 0.0 ? std::sqrt(first[c] / rz0g[c]) : 0.0;   // TRUE residual at the start of this cycle
            if (first[c] > tol2 * rz0g[c]) met = false;
        }
        if (met) { out.converged = true; break; }   // nothing to add: every column was frozen at step 0
        launch_mixed_accumulate(nk, buf.x, mx.b32.x, 0, s);
        total += in.steps;
        for (int c = 0; c < k; ++c) out.iters[c] += in.iters[c];
        if (st) st->refinement_cycles += 1;
        if (total >= o.maxsteps) break;
    }
    out.steps = total;
    HIP_TRY(hipStreamSynchronize(s));
    return out;
}

}  // namespace

extern "C" {

int remo_abi_version(void) { return REMO_ABI_VERSION; }

void remo_opts_default(remo_opts_t *o) {
    if (!o) return;
    std::memset(o, 0, sizeof *o);
    o->preconditioner = 1;  // remo3d.py:82 default "multigrid" => best available
    o->condense = 1;        // remo3d.py:83
    o->maxsteps = 1000;     // ngsolve_functions.py:50
    o->check_every = 5;
    o->rtol = 1e-8;         // NGSolve CGSolver default precision
    o->time_kernels = 0;
    o->coarse_degree = 0;   // 0 = by dimension and size (remo_batch_run): e.g. Chebyshev(5) on [lmax/90, lmax] at 1e4 vertices in 3D
    o->coarse_ratio = 0;
    o->coarse = 0;          // by dimension: multigrid cycle on the vertex block in 2D, Chebyshev polynomial in 3D
}

remo_ctx_t *remo_ctx_create(int device_id) {
    remo_ctx *ctx = nullptr;
    try {
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev <= 0) {
            g_create_error = std::string("no HIP device available: ") + hipGetErrorString(e);
            return nullptr;
        }
        if (device_id < 0 || device_id >= ndev) {
            g_create_error = "device_id out of range";
            return nullptr;
        }
        ctx = new remo_ctx();
        ctx->device = device_id;
        HIP_TRY(hipSetDevice(device_id));
        HIP_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        for (auto &ev : ctx->ev) HIP_TRY(hipEventCreate(&ev));
        const double *m2 = ref_tables(2), *m3 = ref_tables(3);
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->d_M2), sizeof(double) * 9 * 100));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->d_M3), sizeof(double) * 6 * 400));
        HIP_TRY(hipMemcpy(ctx->d_M2, m2, sizeof(double) * 9 * 100, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ctx->d_M3, m3, sizeof(double) * 6 * 400, hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->d_M2q), sizeof(double) * 9 * 100));
        HIP_TRY(hipMemcpy(ctx->d_M2q, ref_tables2_rule4(), sizeof(double) * 9 * 100, hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->d_err), sizeof(int32_t)));
        ctx->ensure_progress(1024 + 2);
        return ctx;
    } catch (const std::exception &ex) {
        g_create_error = ex.what();
        delete ctx;
        return nullptr;
    }
}

void remo_ctx_destroy(remo_ctx_t *ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    for (auto &ev : ctx->spmv_ev) hipEventDestroy(ev);
    for (auto &ev : ctx->ev)
        if (ev) hipEventDestroy(ev);
    if (ctx->ar.base) hipFree(ctx->ar.base);
    if (ctx->in_pool) hipFree(ctx->in_pool);
    if (ctx->d_M2) hipFree(ctx->d_M2);
    if (ctx->d_M3) hipFree(ctx->d_M3);
    if (ctx->d_M2q) hipFree(ctx->d_M2q);
    if (ctx->d_err) hipFree(ctx->d_err);
    if (ctx->progress) hipHostFree(ctx->progress);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *remo_last_error(remo_ctx_t *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

static int batch_create(remo_ctx_t *ctx, const remo_mesh_t *mesh, int32_t n_mat, const double *sigma, int32_t n_rhs,
                        const int32_t *src_ptr, const double *src_z, const double *src_I, const int32_t *eval_ptr,
                        const double *eval_z, remo_batch_t **out, bool pooled) {
    if (!ctx) return REMO_ERR_ARG;
    if (!mesh || !sigma || !out || n_mat <= 0 || n_rhs <= 0 || !src_ptr || !eval_ptr)
        return fail(ctx, REMO_ERR_ARG, "null or empty argument");
    if (mesh->dim != 2 && mesh->dim != 3) return fail(ctx, REMO_ERR_ARG, "dim must be 2 or 3");
    if (mesh->n_nodes <= 0 || mesh->n_elems <= 0 || !mesh->coords || !mesh->conn || !mesh->mat)
        return fail(ctx, REMO_ERR_ARG, "empty mesh");
    if (mesh->n_bfacets < 0 || (mesh->n_bfacets > 0 && (!mesh->bconn || !mesh->bdirichlet)))
        return fail(ctx, REMO_ERR_ARG, "boundary arrays missing");
    if (mesh->n_nodes >= (int64_t(1) << 31) || mesh->n_elems >= (int64_t(1) << 27)) return fail(ctx, REMO_ERR_ARG, "mesh too large");
    if (src_ptr[0] != 0 || eval_ptr[0] != 0) return fail(ctx, REMO_ERR_ARG, "src_ptr / eval_ptr must start at 0");
    for (int k = 0; k < n_rhs; ++k)
        if (src_ptr[k + 1] < src_ptr[k] || eval_ptr[k + 1] < eval_ptr[k]) return fail(ctx, REMO_ERR_ARG, "src_ptr / eval_ptr not monotone");
    if ((src_ptr[n_rhs] > 0 && (!src_z || !src_I)) || (eval_ptr[n_rhs] > 0 && !eval_z)) return fail(ctx, REMO_ERR_ARG, "point arrays missing");
    const int dim = mesh->dim, nb = dim + 1;
    for (int64_t i = 0; i < mesh->n_nodes * dim; ++i)
        if (!std::isfinite(mesh->coords[i])) return fail(ctx, REMO_ERR_MESH, "non-finite coordinate");
    for (int i = 0; i < n_mat; ++i)
        if (!(sigma[i] > 0.0) || !std::isfinite(sigma[i])) return fail(ctx, REMO_ERR_ARG, "sigma must be positive and finite");
    remo_batch *b = nullptr;
    try {
        HIP_TRY(hipSetDevice(ctx->device));
        b = new remo_batch();
        b->dim = dim; b->nv = mesh->n_nodes; b->nt = mesh->n_elems; b->nbf = mesh->n_bfacets; b->n_mat = n_mat;
        b->n_rhs = n_rhs;
        b->src_ptr.assign(src_ptr, src_ptr + n_rhs + 1);
        b->eval_ptr.assign(eval_ptr, eval_ptr + n_rhs + 1);
        b->src_z.assign(src_z, src_z + src_ptr[n_rhs]);
        b->src_I.assign(src_I, src_I + src_ptr[n_rhs]);
        b->eval_z.assign(eval_z, eval_z + eval_ptr[n_rhs]);
        hipStream_t s = ctx->stream;
        b->pooled = pooled;
        size_t pool_at = 0;
        if (pooled) {     // one grow-only device buffer per context holds the inputs of the batch in hand
            const size_t need = align_up(sizeof(double) * size_t(b->nv) * dim) + align_up(sizeof(int32_t) * size_t(b->nt) * nb) + align_up(sizeof(int32_t) * size_t(b->nt)) +
                                align_up(sizeof(double) * size_t(n_mat)) + align_up(sizeof(int32_t) * size_t(b->nbf) * dim + 4) + align_up(size_t(b->nbf) + 4) + 4096;
            if (need > ctx->in_cap) {
                HIP_TRY(hipStreamSynchronize(s));
                if (ctx->in_pool) HIP_TRY(hipFree(ctx->in_pool));
                ctx->in_pool = nullptr; ctx->in_cap = 0;
                const size_t want = align_up(need + need / 4, 4096);
                HIP_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->in_pool), want));
                ctx->in_cap = want;
            }
        }
        auto up = [&](auto **dst, const auto *src, size_t count) {
            using T = std::remove_const_t<std::remove_pointer_t<decltype(src)>>;
            if (pooled) {
                *dst = reinterpret_cast<T *>(ctx->in_pool + pool_at);
                pool_at += align_up(sizeof(T) * (count ? count : 1));
            } else {
                HIP_TRY(hipMalloc(reinterpret_cast<void **>(dst), sizeof(T) * (count ? count : 1)));
            }
            if (count) HIP_TRY(hipMemcpyAsync(*dst, src, sizeof(T) * count, hipMemcpyHostToDevice, s));
        };
        up(&b->d_coords, mesh->coords, size_t(b->nv) * dim);
        up(&b->d_conn, mesh->conn, size_t(b->nt) * nb);
        up(&b->d_mat, mesh->mat, size_t(b->nt));
        up(&b->d_sigma, sigma, size_t(n_mat));
        up(&b->d_bconn, mesh->bconn, size_t(b->nbf) * dim);
        up(&b->d_bdir, mesh->bdirichlet, size_t(b->nbf));
        HIP_TRY(hipStreamSynchronize(s));
        b->u_out.assign(size_t(eval_ptr[n_rhs]), std::nan(""));
        *out = b;
        return REMO_OK;
    } catch (const std::exception &ex) {
        remo_batch_destroy(ctx, b);
        return fail(ctx, REMO_ERR_DEVICE, ex.what());
    }
}

int remo_batch_create(remo_ctx_t *ctx, const remo_mesh_t *mesh, int32_t n_mat, const double *sigma, int32_t n_rhs,
                      const int32_t *src_ptr, const double *src_z, const double *src_I, const int32_t *eval_ptr,
                      const double *eval_z, remo_batch_t **out) {
    return batch_create(ctx, mesh, n_mat, sigma, n_rhs, src_ptr, src_z, src_I, eval_ptr, eval_z, out, false);
}

void remo_batch_destroy(remo_ctx_t *ctx, remo_batch_t *b) {
    if (!b) return;
    if (ctx) (void)hipSetDevice(ctx->device);
    if (!b->pooled)
    for (void *p : {(void *)b->d_coords, (void *)b->d_mat, (void *)b->d_sigma, (void *)b->d_conn, (void *)b->d_bconn, (void *)b->d_bdir})
        if (p) (void)hipFree(p);
    delete b;
}

int remo_batch_run(remo_ctx_t *ctx, remo_batch_t *b, const remo_opts_t *opts_in, remo_stats_t *st) {
    if (!ctx) return REMO_ERR_ARG;
    if (!b) return fail(ctx, REMO_ERR_ARG, "null batch");
    remo_opts_t o;
    if (opts_in) o = *opts_in; else remo_opts_default(&o);
    if (o.maxsteps <= 0 || !(o.rtol > 0.0)) return fail(ctx, REMO_ERR_ARG, "maxsteps and rtol must be positive");
    if (o.coarse < 0 || o.coarse > 3) return fail(ctx, REMO_ERR_ARG, "remo_opts_t.coarse must be 0 (by dimension), 1 (polynomial), 2 (multigrid cycle) or 3 (cycle, else polynomial)");
    if (o.op != 0 && o.op != 2 && o.op != 3)
        return fail(ctx, REMO_ERR_ARG, "remo_opts_t.op must be 0 (default), 2 (CSR product) or 3 (patch operator); 1, the round-2 element-wise operator, left the library with ABI 7");
    remo_stats_t local;
    if (!st) st = &local;
    std::memset(st, 0, sizeof *st);
    std::fill(b->u_out.begin(), b->u_out.end(), std::nan(""));
    b->has_system = false;
    b->amg64 = AmgT<double>{};
    b->amg32 = AmgT<float>{};
    b->run_id = ++ctx->run_id;
    const double t_start = now_ms();
    try {
        HIP_TRY(hipSetDevice(ctx->device));
        hipStream_t s = ctx->stream;
        const int dim = b->dim, N = (dim == 2) ? 10 : 20, NT = (dim == 2) ? 9 : 6;
        const int64_t nt = b->nt, nv = b->nv;
        const int kmax = std::min<int>(b->n_rhs, REMO_MAX_RHS);

        // ---- points of all RHS, chunk by chunk: [sources..., evals...] -------------------
        std::vector<double> pz, pI;
        std::vector<int32_t> prhs, chunk_pt_begin, eval_slot;  // eval_slot: u_out index or -1
        for (int c0 = 0; c0 < b->n_rhs; c0 += REMO_MAX_RHS) {
            chunk_pt_begin.push_back(int32_t(pz.size()));
            const int c1 = std::min(b->n_rhs, c0 + REMO_MAX_RHS);
            for (int r = c0; r < c1; ++r)
                for (int q = b->src_ptr[r]; q < b->src_ptr[r + 1]; ++q) {
                    pz.push_back(b->src_z[q]); pI.push_back(b->src_I[q]); prhs.push_back(r - c0); eval_slot.push_back(-1);
                }
            for (int r = c0; r < c1; ++r)
                for (int q = b->eval_ptr[r]; q < b->eval_ptr[r + 1]; ++q) {
                    pz.push_back(b->eval_z[q]); pI.push_back(0.0); prhs.push_back(r - c0); eval_slot.push_back(q);
                }
        }
        chunk_pt_begin.push_back(int32_t(pz.size()));
        const int npts = int(pz.size());
        for (double z : pz)
            if (!std::isfinite(z)) return fail(ctx, REMO_ERR_POINT, "non-finite point coordinate");

        // ---- device arena: numbering scratch + upper bounds of everything numeric ---------
        const int64_t ndof_max = nv + (dim == 2 ? 7 : 16) * nt, nnz_max = nt * int64_t(N) * N;
        size_t need = symbolic_gpu_arena_bytes(dim, nv, nt, b->nbf);
        need += size_t(nt) * NT * 8 + size_t(nnz_max) * 8 + size_t(ndof_max) * 8 * (1 + 5 * size_t(kmax)) + size_t(nv + 64) * 8 * 4 * size_t(kmax);
        // the patch operator is 3D only; a 2D batch always runs on the CSR product, whatever `op` says
        const bool want_patch = dim == 3 && (o.op == 3 || o.op == 0);
        if (want_patch) need += patch_arena_bytes(nt, ndof_max, kmax) + (size_t(nt) * 21 + 64) * size_t(kmax) * 8;   // tables + slab (upper bound: a row per element dof, and every patch's block padded to 16 rows)
        need += size_t(kMaxPartialBlocks) * 8 * 8 * 3 + size_t(npts) * (N + 8) * 8 + (1 << 20);
        need += size_t(nv + 64) * 200 * 20 + size_t(nv + 64) * 8;   // squared vertex block (paired Chebyshev steps)
        need += size_t(nv + 64) * kCompactPerRow * 16 + size_t(nv + 64) * 8;   // compact vertex block (+ its fp32 values)
        need += size_t(nv + 64) * kEllWidth * 12 + size_t(nv + 64) * 8;        // its fixed-width image
        need += size_t(nv + 64) * (4 * 4 * size_t(kmax) + 8);                    // fp32 Chebyshev chain of the fp64 solve
        const bool want_amg = o.preconditioner != 0 && g_amg != 1 && (g_amg == 2 || o.coarse == 2 || o.coarse == 3 || (o.coarse == 0 && dim == 2 && o.coarse_degree <= 0));   // coarse = 0: an explicit degree asks for the polynomial
        if (want_amg) need += size_t(nv + 64) * (dim == 2 ? 1536 : 3072) * 2 + (1 << 20);   // multigrid hierarchy of the vertex block + its scratch
        if (o.precision == 1)   // fp32 copies of the matrix values and of every PCG vector
            need += size_t(nv + 64) * 200 * 8;
        if (o.precision == 1)
            need += size_t(nnz_max) * 4 + size_t(ndof_max) * 4 * (1 + 5 * size_t(kmax)) + size_t(nv + 64) * 4 * 4 * size_t(kmax) + (1 << 16);
        ctx->reserve(need);

        // ---- dof numbering + CSR pattern (device) ------------------------------------------
        std::string err;
        DeviceSymbolic &sy = b->sym;
        // patch operator batches above 200 k tetrahedra (assemble = 2: any size) number only the P1 block of the matrix
        const bool want_patch0 = want_patch;
        const int64_t vertex_block_above = (want_patch0 && o.assemble != 1) ? (o.assemble == 2 ? 0 : 200000) : -1;
        int rc = build_symbolic_gpu(ctx->ar, s, dim, nv, nt, b->d_conn, b->nbf, b->d_bconn, b->d_bdir, o.condense != 0, ctx->d_err, sy, err, vertex_block_above);
        if (rc != REMO_OK) return fail(ctx, rc, err);
        st->ms_symbolic = now_ms() - t_start;
        const bool lite = sy.vertex_block_only;     // only the P1 block has a pattern: the operator is the patch operator
        st->n_dof = sy.ndof; st->n_free = sy.nfree; st->nnz = lite ? 0 : sy.nnz; st->n_edges = sy.ne; st->n_faces = sy.nf;
        st->n_rhs = b->n_rhs;
        const int64_t n = sy.nfree;

        double *d_C = ctx->take<double>(nt * NT);
        double *d_val = ctx->take<double>(sy.nnz + 2);   // + 16 bytes: the SpMM reads single rows' values as 16-byte pairs (CsrViewT)
        double *d_dinv = ctx->take<double>(n);
        double *d_f = ctx->take<double>(n * kmax);
        PcgBuffers buf{};
        buf.x = ctx->take<double>(n * kmax); buf.r = ctx->take<double>(n * kmax);
        buf.p = ctx->take<double>(n * kmax); buf.q = ctx->take<double>(n * kmax);
        buf.dinv = d_dinv;
        buf.part_pq = ctx->take<double>(kMaxPartialBlocks * 8);
        buf.part_rz = ctx->take<double>(kMaxPartialBlocks * 8 * 2);
        buf.rz0 = ctx->take<double>(kScalarSlots);
        const bool two_level = (o.preconditioner != 0) && sy.nvfree > 0;
        buf.nv_coarse = two_level ? sy.nvfree : 0;
        // Defaults from GPU scans (tools/scan_coarse2d.py, scan_coarse3d.py).  3D: the best degree / interval grow with the
        // vertex count (kappa of the P1 block ~ nv^(2/3), degree ~ sqrt(kappa)): (5, 90) at 12.6 k vertices, (8-10, 150-200) at
        // 24 k, (12-16, 300-600) at 80 k; fine scan after the first / last step lost their launches (tools/scan_coarse3d_fine.py):
        // (5, 70-90) at 12.8 k, (8, 120-160) at 27 k (7: +3 %, 9: +4.5 %).  2D (launch-bound steps, paired Chebyshev launches): (16, 600).
        const double nv_rel = double(sy.nvfree > 0 ? sy.nvfree : 1) / 12600.0;
        // 2D, round 2 (tools/run_2d_batches.py at the 80 k vertices of config 2 with the 0.35 default mesh scale): (16, 600) 173 ms per four
        // batches, (24, 1200) 162, (32, 2400) 155, 752 / 561 / 446 steps - the product of degree and steps grows slowly, a launch pair costs
        // 14 us; (16, 600) was the optimum at 25 k vertices: degree ~ sqrt(vertices), ratio ~ vertices, even degrees (paired launches).
        // The paired (root-product) form must stay in fp64: in fp32 storage it needs MORE steps at degree 16 and breaks down above.
        const double nv2 = double(sy.nvfree > 0 ? sy.nvfree : 1) / 25000.0;
        const int deg2 = 2 * int(std::min(16.0, std::max(8.0, std::floor(8.0 * std::sqrt(nv2) + 0.5))));
        // round 3, size L with the patch operator (83 k vertices, tools/scan_coarse3d_fine.py L, profiles/r03_scan_coarse_L.log): steps per
        // four batches 675 / 637 / 638 / 630 / 601 at degrees 11 / 12 / 13 / 14 / 16 - an odd degree above 7 buys nothing over the even one
        // below it (M: 8 best, 7 and 9 worse) - solve 300 / 289 / 295 / 297 / 294 ms: even degrees from 8 up
        int deg3 = int(std::min(16.0, std::max(5.0, std::floor(5.0 * std::sqrt(nv_rel) + 0.9))));
        if (deg3 > 8) deg3 &= ~1;
        const int deg_default = (dim == 3) ? deg3 : deg2;
        const double ratio_default = (dim == 3) ? std::min(1200.0, std::max(60.0, 90.0 * std::pow(nv_rel, 2.0 / 3.0))) : std::min(2400.0, std::max(600.0, 600.0 * nv2 * 1.25));
        buf.cheb_degree = two_level ? (o.coarse_degree > 0 ? o.coarse_degree : deg_default) : 0;
        buf.cheb_lmax = buf.cheb_lmin = 0.0;
        const size_t nc = size_t(buf.nv_coarse) * kmax + 2;
        buf.cz = ctx->take<double>(nc); buf.cres = ctx->take<double>(nc);
        buf.cd[0] = ctx->take<double>(nc); buf.cd[1] = ctx->take<double>(nc);
        unsigned long long *d_bound = ctx->take<unsigned long long>(2);
        double *d_pz = ctx->take<double>(npts + 1), *d_pI = ctx->take<double>(npts + 1);
        int32_t *d_prhs = ctx->take<int32_t>(npts + 1), *d_found = ctx->take<int32_t>(npts + 1);
        double *d_phi = ctx->take<double>(size_t(npts + 1) * N), *d_fint = ctx->take<double>(npts + 1), *d_out = ctx->take<double>(npts + 1);
        ctx->ensure_progress(o.maxsteps + 3);
        buf.progress = ctx->progress_dev;
        buf.progress_len = ctx->progress_len;
        if (o.time_kernels && ctx->spmv_ev.size() < 8192) {
            const size_t old = ctx->spmv_ev.size();
            ctx->spmv_ev.resize(8192);
            for (size_t i = old; i < ctx->spmv_ev.size(); ++i) HIP_TRY(hipEventCreate(&ctx->spmv_ev[i]));
        }

        // ---- small uploads (points) ---------------------------------------------------------
        HIP_TRY(hipEventRecord(ctx->ev[0], s));
        HIP_TRY(hipMemsetAsync(ctx->d_err, 0, sizeof(int32_t), s));
        std::vector<int32_t> found_init(npts, INT_MAX);
        if (npts > 0) {
            HIP_TRY(hipMemcpyAsync(d_pz, pz.data(), sizeof(double) * npts, hipMemcpyHostToDevice, s));
            HIP_TRY(hipMemcpyAsync(d_pI, pI.data(), sizeof(double) * npts, hipMemcpyHostToDevice, s));
            HIP_TRY(hipMemcpyAsync(d_prhs, prhs.data(), sizeof(int32_t) * npts, hipMemcpyHostToDevice, s));
            HIP_TRY(hipMemcpyAsync(d_found, found_init.data(), sizeof(int32_t) * npts, hipMemcpyHostToDevice, s));
        }
        HIP_TRY(hipEventRecord(ctx->ev[1], s));

        // ---- assembly ---------------------------------------------------------------------
        const double *d_M = (dim == 2) ? (o.quadrature == 1 ? ctx->d_M2q : ctx->d_M2) : ctx->d_M3;
        b->d_M_last = d_M;
        launch_metric_terms(dim, nt, b->d_coords, sy.conn, b->d_mat, sy.eperm, b->d_sigma, b->n_mat, d_C, ctx->d_err, s);
        int64_t pair_begin = 0, pair_end = 0;   // edge-dof rows: consecutive pairs with identical patterns, values interleaved
        if (sy.nvefree > sy.nvfree && ((sy.nvefree - sy.nvfree) & 1) == 0) { pair_begin = sy.nvfree; pair_end = sy.nvefree; }
        if (lite) {   // values of the P1 block (the generic row walk over the vertex rows: their columns are vertex dofs, other local dofs
                      // of an element fall behind the row's last column and are dropped) + the Jacobi factors of every other row
            pair_begin = pair_end = 0;
            launch_assemble(dim, sy.condense, sy.nvfree, 0, 0, sy.rowptr, sy.col, sy.adjptr, sy.adj, sy.eldof, d_C, d_M, d_val, d_dinv, s);
            launch_diag_rows(dim, sy.nvfree, n, sy.adjptr, sy.adj, d_C, d_M, d_dinv, s);
        } else
        launch_assemble(dim, sy.condense, n, pair_begin, pair_end, sy.rowptr, sy.col, sy.adjptr, sy.adj, sy.eldof, d_C, d_M, d_val, d_dinv, s);
        HIP_TRY(hipEventRecord(ctx->ev[2], s));

        // ---- point location + shapes (all points at once) ---------------------------------
        if (npts > 0) {
            for (int q0 = 0; q0 < npts; q0 += kMaxPoints)
                launch_locate(dim, nt, b->d_coords, sy.conn, std::min(kMaxPoints, npts - q0), d_pz + q0, d_found + q0, s);
            launch_point_shapes(dim, npts, d_pz, d_found, b->d_coords, sy.conn, d_phi, ctx->d_err, s);
        }
        HIP_TRY(hipEventRecord(ctx->ev[3], s));
        int32_t h_err = 0;
        unsigned long long h_bound = 0;
        int32_t h_sq[2] = {1, 0};   // flag, entries
        int32_t *sq_rowptr = nullptr, *sq_col = nullptr;
        double *sq_a = nullptr, *sq_b = nullptr;
        // paired steps pay off where B stays small: 2D (~19 entries per row: 81 vs 93 us per PCG step); in 3D B has ~65
        // entries per row and three launches on it cost more than six on A_vv (153 vs 149 us) - forced by tune value 2
        // compact copy of the vertex block for the Chebyshev launches (remo_debug_tune key 13: 0 = read A in place)
        int32_t h_vb[2] = {1, 0};   // flag, entries
        int32_t *vb_rowptr = nullptr, *vb_col = nullptr;
        double *vb_val = nullptr;
        const bool want_square = two_level && !want_amg && (buf.cheb_degree % 2 == 0) && ((g_square == 1 && dim == 2) || g_square == 2);
        // measured in the bench (--tune 13=0 against 13=1, one box): 538.9 -> 516.1 ms solve per step at 83 k vertices; at 12.8 k the
        // launches are latency, not bytes (714 -> 710 ms) and building the copy costs what it saves: larger blocks only (2 forces it)
        const bool want_compact = !lite && two_level && !want_square && (g_compact == 2 || (g_compact == 1 && buf.nv_coarse > 16384));
        if (two_level) {  // spectrum bound of the Jacobi-scaled vertex block for the Chebyshev interval
            HIP_TRY(hipMemsetAsync(d_bound, 0, sizeof(unsigned long long), s));
            launch_vblock_bound(buf.nv_coarse, CsrView{n, sy.nnz, sy.rowptr, sy.col, d_val}, d_dinv, d_bound, s);
            HIP_TRY(hipMemcpyAsync(&h_bound, d_bound, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        }
        if (want_square) {   // B = A_vv D^-1 A_vv for the paired Chebyshev steps (kernels.hip)
            const int64_t nvc = buf.nv_coarse, cap = nvc * 200;
            sq_rowptr = ctx->take<int32_t>(size_t(nvc) + 2);
            sq_col = ctx->take<int32_t>(size_t(cap));
            sq_a = ctx->take<double>(size_t(cap)); sq_b = ctx->take<double>(size_t(cap));
            int32_t *d_sqflag = ctx->take<int32_t>(1);
            HIP_TRY(hipMemsetAsync(d_sqflag, 0, sizeof(int32_t), s));
            launch_vblock_square(nvc, CsrView{n, sy.nnz, sy.rowptr, sy.col, d_val}, d_dinv, sq_rowptr, sq_col, sq_a, sq_b, cap, d_sqflag, s);
            HIP_TRY(hipMemcpyAsync(&h_sq[0], d_sqflag, sizeof(int32_t), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipMemcpyAsync(&h_sq[1], sq_rowptr + nvc, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        }
        if (want_compact) {
            const int64_t nvc = buf.nv_coarse, cap = (nvc + 64) * kCompactPerRow;
            vb_rowptr = ctx->take<int32_t>(size_t(nvc) + 2);
            vb_col = ctx->take<int32_t>(size_t(cap));
            vb_val = ctx->take<double>(size_t(cap));
            int32_t *d_vbflag = ctx->take<int32_t>(1);
            HIP_TRY(hipMemsetAsync(d_vbflag, 0, sizeof(int32_t), s));
            launch_vblock_compact(nvc, CsrView{n, sy.nnz, sy.rowptr, sy.col, d_val}, vb_rowptr, vb_col, vb_val, cap, d_vbflag, s);
            HIP_TRY(hipMemcpyAsync(&h_vb[0], d_vbflag, sizeof(int32_t), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipMemcpyAsync(&h_vb[1], vb_rowptr + nvc, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        }
        // patch operator (patch.hip): its tables are built beside the assembly; their overflow flag and largest patch come
        // back with the other small read-backs below
        int32_t h_patch[3] = {1, 0, 0};
        PatchTables ptab{};
        if (want_patch) {
            int32_t *d_pflag = ctx->take<int32_t>(4);
            build_patch_tables(ctx->ar, s, sy, d_C, kmax, ptab, d_pflag);
            HIP_TRY(hipMemcpyAsync(h_patch, d_pflag, 3 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        }
        HIP_TRY(hipMemcpyAsync(&h_err, ctx->d_err, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (want_compact && h_vb[0] == 0 && h_vb[1] > 0) { buf.vb_rowptr = vb_rowptr; buf.vb_col = vb_col; buf.vb_val = vb_val; }
        if (lite && two_level) { buf.vb_rowptr = sy.rowptr; buf.vb_col = sy.col; buf.vb_val = d_val; h_vb[0] = 0; h_vb[1] = int32_t(sy.nnz); }   // the assembled block IS the compact vertex block
        // fp32 Chebyshev chain inside the fp64 solve (remo_debug_tune key 15: 0 = off): where the chain's launches are HBM streams
        // (no folded first step: more than 32 k vertex rows) and the compact block exists
        if (g_chain32 && o.precision == 0 && buf.vb_rowptr && (buf.nv_coarse > 32768 || g_chain32 == 2)) {   // 2: forced (tests)
            float *v32c = ctx->take<float>(size_t(h_vb[1]) + 4), *d32c = ctx->take<float>(size_t(buf.nv_coarse) + 4);
            launch_to_float(h_vb[1], buf.vb_val, v32c, s);
            launch_to_float(buf.nv_coarse, d_dinv, d32c, s);
            buf.c32_val = v32c; buf.c32_dinv = d32c;
            buf.c32_z = ctx->take<float>(nc); buf.c32_res = ctx->take<float>(nc);
            buf.c32_d[0] = ctx->take<float>(nc); buf.c32_d[1] = ctx->take<float>(nc);
        }
        // fixed-width image of the vertex block for the polynomial's launches (kernels.hip k_vblock_ell; remo_debug_tune key 24: 0 = off)
        int32_t *ell_col = nullptr, *ell_tail = nullptr;
        if (g_ell && two_level && !want_amg && !want_square && dim == 3) {
            const int64_t nvc = buf.nv_coarse;
            const bool from_block = buf.vb_rowptr != nullptr;
            ell_col = ctx->take<int32_t>(size_t(nvc) * kEllWidth + 8);
            ell_tail = ctx->take<int32_t>(size_t(nvc) * 2 + 8);
            double *e64 = (o.precision == 0 && !buf.c32_val) ? ctx->take<double>(size_t(nvc) * kEllWidth + 8) : nullptr;
            float *e32 = (o.precision != 0 || buf.c32_val) ? ctx->take<float>(size_t(nvc) * kEllWidth + 8) : nullptr;
            launch_vblock_ell(nvc, from_block ? buf.vb_rowptr : sy.rowptr, from_block ? buf.vb_col : sy.col, from_block ? buf.vb_val : d_val,
                              ell_col, ell_tail, e64, e32, s);
            buf.ell_col = ell_col; buf.ell_tail = ell_tail; buf.ell_val = e64; buf.c32_ell_val = e32;
        }
        if (two_level && want_amg) {   // multigrid cycle on the vertex block instead of the polynomial (amg.hip)
            std::string why;
            if (amg_setup(ctx->ar, s, dim, buf.nv_coarse, sy.rowptr, sy.col, d_val, kmax, b->amg64, why)) buf.amg = &b->amg64;
            else if (o.coarse == 2 || g_amg == 2) return fail(ctx, REMO_ERR_NUMERIC, "multigrid hierarchy of the vertex block: " + why);
        }
        bool amg32_ready = false;
        if (buf.amg && (o.precision == 1 || g_amg32)) {   // fp32 image of the hierarchy: the mixed mode's inner solver, or the cycle of an fp64 solve
            amg_to_float(ctx->ar, s, b->amg64, kmax, b->amg32);
            amg32_ready = true;
            if (o.precision == 0) buf.amg32 = &b->amg32;
        }
        st->coarse_used = !two_level ? 0 : (buf.amg ? 2 : 1);
        if (two_level) {
            double lmax;
            std::memcpy(&lmax, &h_bound, sizeof lmax);
            if (!(lmax > 0.0) || !std::isfinite(lmax)) return fail(ctx, REMO_ERR_NUMERIC, "vertex block has no positive spectrum bound");
            buf.cheb_lmax = lmax;
            buf.cheb_lmin = lmax / (o.coarse_ratio > 0 ? double(o.coarse_ratio) : ratio_default);
        }
        if (want_square && h_sq[0] == 0) {   // otherwise (a vertex of very high valence) the one-step launches stay
            buf.sq_rowptr = sq_rowptr; buf.sq_col = sq_col; buf.sq_a = sq_a; buf.sq_b = sq_b;
            const double avg = double(h_sq[1]) / double(buf.nv_coarse > 0 ? buf.nv_coarse : 1);
            buf.sq_lanes = g_sq_lanes ? g_sq_lanes : (avg > 40.0 ? 32 : (avg > 24.0 ? 16 : 8));   // 2D rows of B hold ~19 entries: 8 lanes (two passes in flight) 145 vs 150 ms with 16
        }
        if (h_err & 1) return fail(ctx, REMO_ERR_MESH, "degenerate element or material index out of range");
        if (h_err & 2) return fail(ctx, REMO_ERR_POINT, "source or evaluation point outside the mesh");

        b->A = CsrView{n, sy.nnz, sy.rowptr, sy.col, d_val};
        b->A.pair_begin = pair_begin; b->A.pair_end = pair_end;
        b->A.vertex_block_only = lite;
        // (the patch kernel's buffer descriptors address the slab and x with 32-bit byte offsets: beyond 4 GB the CSR product stays)
        // (all_slab, the product's form: every patch addresses its own block of the slab through a descriptor of its own - only x is bound by this)
        const bool slab_fits = (ptab.all_slab || uint64_t(nt) * 20 * uint64_t(kmax) * 8 < 0xFFFFF000ull) && uint64_t(n) * uint64_t(kmax) * 8 < 0xFFFFF000ull;
        // patch operator: asked for, or (op = 0) whenever its tables fit; a patch with more distinct rows than the tables hold
        // (an element list without locality) sends op = 0 on to the CSR product and fails op = 3
        // (the kernel forms byte offsets of rows and slab slots with 24-bit multiplies and 32-bit buffer offsets)
        const size_t patch_lds = ptab.block > 0 ? patch_lds_bytes(h_patch[1], kmax, ptab.block, ptab.all_slab != 0) : 0;   // what k_patch_apply asks for (kernels.hip patch_applies)
        const bool patch_ok = want_patch && h_patch[0] == 0 && h_patch[1] > 0 && slab_fits && n < (int64_t(1) << 24) && h_patch[2] < (ptab.all_slab ? 0x7FFFFFF0 : (1 << 24)) && patch_lds <= kPatchLdsLimit;
        if (lite && !patch_ok) return fail(ctx, REMO_ERR_ARG, "only the P1 block was assembled but the patch operator cannot run on this batch: rerun with remo_opts_t.assemble = 1");
        if (o.op == 3 && dim == 3 && !patch_ok) return fail(ctx, REMO_ERR_ARG, "patch operator: a patch of the element list touches more distinct rows than its tables hold (or the mesh is too large)");
        const bool patch_op = patch_ok;
        st->op_used = patch_op ? 3 : 0;
        st->assembled = lite ? 2 : 1;
        if (patch_op) {
            ptab.nslot_cap = h_patch[2] > 0 ? h_patch[2] : 1;     // the slab holds the slots in use
            b->patch64 = PatchOpT<double>{ptab, ctx->take<double>(size_t(ptab.nslot_cap) * size_t(kmax) + 8), ctx->take<double>(size_t(ptab.npatch) * 8 + 8), h_patch[1]};
            b->A.patch = &b->patch64;
        }
        b->d_val = d_val;
        b->d_dinv = d_dinv;
        b->d_x = buf.x;
        b->d_f = d_f;
        b->d_C = d_C;
        b->k_last = 0;
        b->has_system = true;

        // ---- solve, chunk by chunk --------------------------------------------------------
        // serialize_solves: batches of other contexts may number and assemble beside this PCG, but not run theirs
        std::unique_lock<std::mutex> solve_turn(g_solve_mutex, std::defer_lock);
        if (o.serialize_solves) solve_turn.lock();
        const int lpr = choose_lanes_per_row(n, sy.nnz);
        buf.nb_spmv = spmv_grid(n, lpr);
        buf.nb_vec = vec_grid(n);
        buf.defer_q = patch_op && g_defer_q && !pcg_update_folds(buf);     // the update launch sums the shared rows of q = A p itself
        buf.x_in_direction = g_x_in_direction != 0;
        buf.pq_bins = buf.defer_q && g_dot_bins != 0;
        if (patch_op) b->patch64.dot_bins = buf.pq_bins;
        const bool mixed = (o.precision == 1);
        MixedBuffers mx;
        if (mixed) {   // fp32 images of the system for the inner solver
            float *v32 = ctx->take<float>(size_t(sy.nnz) + 2);   // same slack as d_val
            float *dinv32 = ctx->take<float>(size_t(n));
            launch_to_float(sy.nnz, d_val, v32, s);
            launch_to_float(n, d_dinv, dinv32, s);
            mx.A32 = CsrViewT<float>{n, sy.nnz, sy.rowptr, sy.col, v32};
            mx.A32.pair_begin = b->A.pair_begin; mx.A32.pair_end = b->A.pair_end; mx.A32.vertex_block_only = lite;
            if (patch_op) {   // the slab and the partial sums are scratch of one application: the fp32 operator shares them
                b->patch32 = PatchOpT<float>{ptab, reinterpret_cast<float *>(b->patch64.Yb), b->patch64.ppart, h_patch[1]};
                mx.A32.patch = &b->patch32;
            }
            PcgBuffersT<float> &f = mx.b32;
            f.x = ctx->take<float>(size_t(n) * kmax); f.r = ctx->take<float>(size_t(n) * kmax);
            f.p = ctx->take<float>(size_t(n) * kmax + 4); f.q = ctx->take<float>(size_t(n) * kmax);
            mx.f32 = ctx->take<float>(size_t(n) * kmax);
            f.dinv = dinv32;
            f.part_pq = buf.part_pq; f.part_rz = buf.part_rz; f.rz0 = buf.rz0;
            f.nv_coarse = buf.nv_coarse; f.cheb_degree = buf.cheb_degree; f.cheb_lmax = buf.cheb_lmax; f.cheb_lmin = buf.cheb_lmin;
            f.cz = ctx->take<float>(nc); f.cres = ctx->take<float>(nc);
            f.cd[0] = ctx->take<float>(nc); f.cd[1] = ctx->take<float>(nc);
            f.progress = buf.progress; f.progress_len = buf.progress_len;
            f.nb_spmv = buf.nb_spmv; f.nb_vec = buf.nb_vec;
            if (buf.amg && amg32_ready) f.amg = &b->amg32;
            if (buf.vb_rowptr) {
                float *vb32 = ctx->take<float>(size_t(h_vb[1]) + 1);
                launch_to_float(h_vb[1], buf.vb_val, vb32, s);
                f.vb_rowptr = buf.vb_rowptr; f.vb_col = buf.vb_col; f.vb_val = vb32;
            }
            f.ell_col = buf.ell_col; f.ell_tail = buf.ell_tail; f.ell_val = buf.c32_ell_val;
            if (buf.sq_rowptr) {
                float *a32 = ctx->take<float>(size_t(h_sq[1]) + 1), *b32 = ctx->take<float>(size_t(h_sq[1]) + 1);
                launch_to_float(h_sq[1], buf.sq_a, a32, s);
                launch_to_float(h_sq[1], buf.sq_b, b32, s);
                f.sq_rowptr = buf.sq_rowptr; f.sq_col = buf.sq_col; f.sq_a = a32; f.sq_b = b32; f.sq_lanes = buf.sq_lanes;
            }
        }
        if (mixed) mx.b32.defer_q = patch_op && g_defer_q && !pcg_update_folds(mx.b32);
        if (mixed) mx.b32.x_in_direction = g_x_in_direction != 0;
        if (mixed) { mx.b32.pq_bins = mx.b32.defer_q && g_dot_bins; if (patch_op) b->patch32.dot_bins = mx.b32.pq_bins; }
        std::vector<double> h_out(npts, std::nan(""));
        int ret = REMO_OK;
        size_t ev_used = 0;
        float ms_solve = 0.f, ms_eval = 0.f;
        int chunk = 0;
        for (int c0 = 0; c0 < b->n_rhs; c0 += REMO_MAX_RHS, ++chunk) {
            const int k = std::min(b->n_rhs - c0, REMO_MAX_RHS);
            const int q0 = chunk_pt_begin[chunk], nq = chunk_pt_begin[chunk + 1] - q0;
            HIP_TRY(hipEventRecord(ctx->ev[4], s));
            HIP_TRY(hipMemsetAsync(d_f, 0, sizeof(double) * n * k, s));
            if (nq > 0)
                launch_build_rhs(dim, sy.condense, nq, d_prhs + q0, d_pI + q0, d_found + q0, d_phi + size_t(q0) * N, sy.eldof, d_C, d_M, k,
                                 d_f, d_fint + q0, s);
            HIP_TRY(hipEventRecord(ctx->ev[5], s));
            ChunkResult cr = mixed ? run_pcg_mixed(ctx, b->A, k, d_f, buf, mx, o, st, ev_used) : run_pcg(ctx, b->A, k, d_f, buf, o, st, ev_used);
            HIP_TRY(hipEventRecord(ctx->ev[6], s));
            if (nq > 0)
                launch_eval(dim, sy.condense, nq, d_prhs + q0, d_pI + q0, d_found + q0, d_phi + size_t(q0) * N, sy.eldof, d_C, d_M, k, buf.x,
                            d_fint + q0, d_out + q0, s);
            HIP_TRY(hipEventRecord(ctx->ev[7], s));
            if (nq > 0) HIP_TRY(hipMemcpyAsync(h_out.data() + q0, d_out + q0, sizeof(double) * nq, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            float e1 = 0, e2 = 0, e3 = 0;
            (void)hipEventElapsedTime(&e1, ctx->ev[4], ctx->ev[5]);
            (void)hipEventElapsedTime(&e2, ctx->ev[5], ctx->ev[6]);
            (void)hipEventElapsedTime(&e3, ctx->ev[6], ctx->ev[7]);
            ms_eval += e1 + e3;
            ms_solve += e2;
            if (!cr.finite) return fail(ctx, REMO_ERR_NUMERIC, "non-finite residual in PCG");
            b->k_last = k;
            if (!cr.converged) ret = REMO_NOT_CONVERGED;
            for (int c = 0; c < k; ++c) {
                st->iterations[c] = cr.iters[c];
                st->relres[c] = cr.relres[c];
                st->max_iterations = std::max(st->max_iterations, cr.iters[c]);
            }
        }
        for (int q = 0; q < npts; ++q)
            if (eval_slot[q] >= 0) b->u_out[eval_slot[q]] = h_out[q];
        float m = 0;
        (void)hipEventElapsedTime(&m, ctx->ev[0], ctx->ev[1]); st->ms_h2d = m;
        (void)hipEventElapsedTime(&m, ctx->ev[1], ctx->ev[2]); st->ms_assemble = m;
        (void)hipEventElapsedTime(&m, ctx->ev[2], ctx->ev[3]); st->ms_eval = m + ms_eval;
        st->ms_solve = ms_solve;
        st->spmv_bytes = mixed ? 8.0 * double(sy.nnz) + 4.0 * double(n) + 8.0 * double(kmax) * double(n)   // fp32 values and vectors
                               : 12.0 * double(sy.nnz) + 4.0 * double(n) + 16.0 * double(kmax) * double(n);
        if (patch_op)   // the patch operator reads no stored entries: x and y once (k columns) + 40 bytes of local indices and 48 of metric terms per element
            st->spmv_bytes = (mixed ? 8.0 : 16.0) * double(kmax) * double(n) + 88.0 * double(nt);
        if (o.time_kernels) {
            // what an event bracket measures beyond the enclosed kernel: an empty pair on the same stream
            float overhead = 1e30f;
            for (int rep = 0; rep < 16; ++rep) {
                HIP_TRY(hipEventRecord(ctx->ev[0], s));
                HIP_TRY(hipEventRecord(ctx->ev[1], s));
                HIP_TRY(hipEventSynchronize(ctx->ev[1]));
                float e = 0;
                (void)hipEventElapsedTime(&e, ctx->ev[0], ctx->ev[1]);
                if (e < overhead) overhead = e;
            }
            if (!(overhead < 1e29f) || overhead < 0.f) overhead = 0.f;
            double sum = 0, raw = 0;
            for (size_t i = 0; i + 1 < ev_used; i += 2) {
                float e = 0;
                (void)hipEventElapsedTime(&e, ctx->spmv_ev[i], ctx->spmv_ev[i + 1]);
                raw += e;
                sum += (e > overhead) ? double(e - overhead) : 0.0;
            }
            st->spmv_ms = sum;
            st->spmv_ms_raw = raw;
            st->event_overhead_ms = overhead;
            st->spmv_launches = int64_t(ev_used / 2);
        }
        st->ms_total = now_ms() - t_start;
        if (ret == REMO_NOT_CONVERGED) ctx->err = "PCG did not reach rtol within maxsteps";
        return ret;
    } catch (const std::exception &ex) {
        std::fill(b->u_out.begin(), b->u_out.end(), std::nan(""));
        return fail(ctx, REMO_ERR_DEVICE, ex.what());
    }
}

int remo_batch_fetch(remo_ctx_t *ctx, remo_batch_t *b, double *u_out) {
    if (!ctx) return REMO_ERR_ARG;
    if (!b || (!u_out && !b->u_out.empty())) return fail(ctx, REMO_ERR_ARG, "null argument");
    if (!b->u_out.empty()) std::memcpy(u_out, b->u_out.data(), sizeof(double) * b->u_out.size());
    return REMO_OK;
}

int remo_solve_batch(remo_ctx_t *ctx, const remo_mesh_t *mesh, int32_t n_mat, const double *sigma, int32_t n_rhs,
                     const int32_t *src_ptr, const double *src_z, const double *src_I, const int32_t *eval_ptr,
                     const double *eval_z, double *u_out, const remo_opts_t *opts, remo_stats_t *stats) {
    if (!ctx) return REMO_ERR_ARG;
    if (u_out && eval_ptr && n_rhs > 0)
        for (int i = 0; i < eval_ptr[n_rhs]; ++i) u_out[i] = std::nan("");
    remo_batch_t *b = nullptr;
    int rc = batch_create(ctx, mesh, n_mat, sigma, n_rhs, src_ptr, src_z, src_I, eval_ptr, eval_z, &b, true);     // inputs into the context's pool: no hipMalloc / hipFree per batch
    if (rc != REMO_OK) return rc;
    rc = remo_batch_run(ctx, b, opts, stats);
    if (rc >= 0 && u_out) remo_batch_fetch(ctx, b, u_out);
    remo_batch_destroy(ctx, b);
    return rc;
}

int remo_batch_eval(remo_ctx_t *ctx, remo_batch_t *b, int32_t rhs, int32_t npts, const double *z, double *u_out) {
    if (!ctx) return REMO_ERR_ARG;
    if (!b || !z || !u_out || npts <= 0) return fail(ctx, REMO_ERR_ARG, "bad argument");
    for (int i = 0; i < npts; ++i) u_out[i] = std::nan("");
    if (!b->has_system || b->run_id != ctx->run_id || b->k_last <= 0 || b->n_rhs > REMO_MAX_RHS)
        return fail(ctx, REMO_ERR_ARG, "no resident solution for this batch (another batch ran on the context since)");
    if (rhs < 0 || rhs >= b->k_last) return fail(ctx, REMO_ERR_ARG, "rhs index out of range");
    void *scratch = nullptr;
    try {
        HIP_TRY(hipSetDevice(ctx->device));
        hipStream_t s = ctx->stream;
        const DeviceSymbolic &sy = b->sym;
        const int dim = b->dim, N = (dim == 2) ? 10 : 20;
        const size_t bytes = size_t(npts) * (sizeof(double) * (4 + N) + sizeof(int32_t) * 2) + 1024;
        HIP_TRY(hipMalloc(&scratch, bytes));
        double *d_z = static_cast<double *>(scratch), *d_I = d_z + npts, *d_phi = d_I + npts, *d_fint = d_phi + size_t(npts) * N, *d_out = d_fint + npts;
        int32_t *d_rhs = reinterpret_cast<int32_t *>(d_out + npts), *d_found = d_rhs + npts;
        std::vector<int32_t> h_rhs(npts, rhs), h_found(npts, INT_MAX);
        HIP_TRY(hipMemsetAsync(scratch, 0, bytes, s));   // strengths 0 (evaluation points), no bubble loads
        HIP_TRY(hipMemsetAsync(ctx->d_err, 0, sizeof(int32_t), s));
        HIP_TRY(hipMemcpyAsync(d_z, z, sizeof(double) * npts, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(d_rhs, h_rhs.data(), sizeof(int32_t) * npts, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(d_found, h_found.data(), sizeof(int32_t) * npts, hipMemcpyHostToDevice, s));
        for (int q0 = 0; q0 < npts; q0 += kMaxPoints)
            launch_locate(dim, b->nt, b->d_coords, sy.conn, std::min(kMaxPoints, npts - q0), d_z + q0, d_found + q0, s);
        launch_point_shapes(dim, npts, d_z, d_found, b->d_coords, sy.conn, d_phi, ctx->d_err, s);
        const double *d_M = b->d_M_last ? b->d_M_last : ((dim == 2) ? ctx->d_M2 : ctx->d_M3);
        launch_eval(dim, sy.condense, npts, d_rhs, d_I, d_found, d_phi, sy.eldof, b->d_C, d_M, b->k_last, b->d_x, d_fint, d_out, s);
        int32_t h_err = 0;
        HIP_TRY(hipMemcpyAsync(u_out, d_out, sizeof(double) * npts, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(&h_err, ctx->d_err, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        (void)hipFree(scratch);
        if (h_err & 2) return fail(ctx, REMO_ERR_POINT, "evaluation point outside the mesh");
        return REMO_OK;
    } catch (const std::exception &ex) {
        if (scratch) (void)hipFree(scratch);
        return fail(ctx, REMO_ERR_DEVICE, ex.what());
    }
}

int remo_batch_get_system(remo_ctx_t *ctx, remo_batch_t *b, int32_t *rowptr, int32_t *col, double *val, double *dinv,
                          int32_t *freeid) {
    if (!ctx) return REMO_ERR_ARG;
    if (!b || !b->has_system || b->run_id != ctx->run_id) return fail(ctx, REMO_ERR_ARG, "no assembled system on this batch (run it first)");
    if (b->A.vertex_block_only && (rowptr || col || val))
        return fail(ctx, REMO_ERR_ARG, "the last run assembled only the diagonal and the P1 block (remo_opts_t.assemble): run with assemble = 1 to inspect the matrix");
    try {
        HIP_TRY(hipSetDevice(ctx->device));
        const DeviceSymbolic &sy = b->sym;
        if (rowptr) HIP_TRY(hipMemcpy(rowptr, sy.rowptr, sizeof(int32_t) * (sy.nfree + 1), hipMemcpyDeviceToHost));
        if (col) HIP_TRY(hipMemcpy(col, sy.col, sizeof(int32_t) * sy.nnz, hipMemcpyDeviceToHost));
        if (freeid) HIP_TRY(hipMemcpy(freeid, sy.freeid, sizeof(int32_t) * sy.ndof, hipMemcpyDeviceToHost));
        if (val) {   // plain CSR order for the caller: undo the interleaving of the edge-pair rows
            std::vector<double> raw(size_t(sy.nnz));
            std::vector<int32_t> rp(size_t(sy.nfree) + 1);
            HIP_TRY(hipMemcpy(raw.data(), b->d_val, sizeof(double) * sy.nnz, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(rp.data(), sy.rowptr, sizeof(int32_t) * (sy.nfree + 1), hipMemcpyDeviceToHost));
            std::memcpy(val, raw.data(), sizeof(double) * sy.nnz);
            for (int64_t r = b->A.pair_begin; r + 1 < b->A.pair_end; r += 2) {
                const int32_t rs = rp[r], len = rp[r + 1] - rp[r];
                for (int32_t e = 0; e < len; ++e) { val[rs + e] = raw[rs + 2 * e]; val[rs + len + e] = raw[rs + 2 * e + 1]; }
            }
        }
        if (dinv) HIP_TRY(hipMemcpy(dinv, b->d_dinv, sizeof(double) * sy.nfree, hipMemcpyDeviceToHost));
        return REMO_OK;
    } catch (const std::exception &ex) {
        return fail(ctx, REMO_ERR_DEVICE, ex.what());
    }
}

int remo_batch_get_vectors(remo_ctx_t *ctx, remo_batch_t *b, double *x, double *f, int32_t *k_out) {
    if (!ctx) return REMO_ERR_ARG;
    if (!b || !b->has_system || b->run_id != ctx->run_id || b->k_last <= 0) return fail(ctx, REMO_ERR_ARG, "no resident solution on this batch (run it first)");
    try {
        HIP_TRY(hipSetDevice(ctx->device));
        const size_t bytes = sizeof(double) * size_t(b->A.n) * size_t(b->k_last);
        if (x) HIP_TRY(hipMemcpy(x, b->d_x, bytes, hipMemcpyDeviceToHost));
        if (f) HIP_TRY(hipMemcpy(f, b->d_f, bytes, hipMemcpyDeviceToHost));
        if (k_out) *k_out = b->k_last;
        return REMO_OK;
    } catch (const std::exception &ex) {
        return fail(ctx, REMO_ERR_DEVICE, ex.what());
    }
}

namespace {
__global__ void __launch_bounds__(256) k_stream_read(const double2 *__restrict__ x, int64_t n2, double *__restrict__ out) {
    double a = 0.0, b = 0.0;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n2; i += int64_t(gridDim.x) * blockDim.x) {
        const double2 v = x[i];
        a += v.x; b += v.y;
    }
    a += b;
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = a;
}
// scattered 16-byte reads from a 2 MiB buffer (resident in every XCD's 4 MiB L2 after the first touch): the L2 -> L1 -> lane path
// the SpMM's x gather lives on
__global__ void __launch_bounds__(256) k_l2_gather(const double2 *__restrict__ x, uint32_t mask, int iters, double *__restrict__ out) {
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    double a = 0.0;
    for (int i = 0; i < iters; ++i) {
        const double2 v = x[idx & mask];
        a += v.x + v.y;
        idx = idx * 1664525u + 1013904223u;
    }
    if (a == 1.2345e300) out[0] = a;
}

// a chain of dependent fp32 multiply-adds per wave, one wave per SIMD-sized slice of the chip: its rate follows the shader clock
__global__ void __launch_bounds__(64) k_clock_probe(int iters, float *__restrict__ out) {
    float a = float(threadIdx.x) * 1e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) a = __builtin_fmaf(a, 0.999999f, 0.5f);
    }
    if (a == 123.456f) out[blockIdx.x] = a;
}
}  // namespace

int remo_debug_cache_gather(remo_ctx_t *ctx, int64_t bytes, double *gbs) {
    if (!ctx || !gbs || bytes < (1 << 20) || bytes > (int64_t(1) << 32) || (bytes & (bytes - 1)) != 0) return REMO_ERR_ARG;   // a power of two
    double *a = nullptr, *o = nullptr;
    try {
        HIP_TRY(hipSetDevice(ctx->device));
        const int64_t n2 = bytes / 16;   // 16-byte elements
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&a), size_t(n2) * 16));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&o), 64));
        HIP_TRY(hipMemsetAsync(a, 0, size_t(n2) * 16, ctx->stream));
        const int blocks = 4096, iters = 64;
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            HIP_TRY(hipEventRecord(ctx->ev[0], ctx->stream));
            hipLaunchKernelGGL(k_l2_gather, dim3(blocks), dim3(256), 0, ctx->stream, reinterpret_cast<const double2 *>(a), uint32_t(n2 - 1), iters, o);
            HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            float ms = 0;
            (void)hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]);
            if (ms < best) best = ms;
        }
        *gbs = double(blocks) * 256.0 * iters * 16.0 / (double(best) * 1e6);
        (void)hipFree(a); (void)hipFree(o);
        return REMO_OK;
    } catch (const std::exception &ex) {
        if (a) (void)hipFree(a);
        if (o) (void)hipFree(o);
        return fail(ctx, REMO_ERR_DEVICE, ex.what());
    }
}

int remo_debug_clock(remo_ctx_t *ctx, double *gfma_per_wave) {
    if (!ctx || !gfma_per_wave) return REMO_ERR_ARG;
    float *o = nullptr;
    try {
        HIP_TRY(hipSetDevice(ctx->device));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&o), sizeof(float) * 1024));
        const int iters = 1 << 16;   // x 16 dependent multiply-adds
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            HIP_TRY(hipEventRecord(ctx->ev[0], ctx->stream));
            hipLaunchKernelGGL(k_clock_probe, dim3(1024), dim3(64), 0, ctx->stream, iters, o);
            HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            float ms = 0;
            (void)hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]);
            if (ms < best) best = ms;
        }
        *gfma_per_wave = double(iters) * 16.0 / (double(best) * 1e6);
        (void)hipFree(o);
        return REMO_OK;
    } catch (const std::exception &ex) {
        if (o) (void)hipFree(o);
        return fail(ctx, REMO_ERR_DEVICE, ex.what());
    }
}

namespace {
// which XCD (accelerator die) a workgroup runs on: hardware register XCC_ID (id 20, 4 bits) - the SpMM's row schedule assumes
// workgroup b runs on XCD b mod 8
__global__ void __launch_bounds__(64) k_xcc_probe(int32_t *__restrict__ out) {
    const int32_t id = __builtin_amdgcn_s_getreg((3 << 11) | 20);
    if (threadIdx.x == 0) out[blockIdx.x] = id;
}
}  // namespace

int remo_debug_xcc(remo_ctx_t *ctx, int32_t *out, int32_t nblocks) {
    if (!ctx || !out || nblocks < 1 || nblocks > 65536) return REMO_ERR_ARG;
    int32_t *d = nullptr;
    try {
        HIP_TRY(hipSetDevice(ctx->device));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d), sizeof(int32_t) * nblocks));
        hipLaunchKernelGGL(k_xcc_probe, dim3(nblocks), dim3(64), 0, ctx->stream, d);
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        HIP_TRY(hipMemcpy(out, d, sizeof(int32_t) * nblocks, hipMemcpyDeviceToHost));
        (void)hipFree(d);
        return REMO_OK;
    } catch (const std::exception &ex) {
        if (d) (void)hipFree(d);
        return fail(ctx, REMO_ERR_DEVICE, ex.what());
    }
}

int remo_debug_device(remo_ctx_t *ctx, int64_t *out8) {
    if (!ctx || !out8) return REMO_ERR_ARG;
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, ctx->device) != hipSuccess) return fail(ctx, REMO_ERR_DEVICE, "hipGetDeviceProperties failed");
    out8[0] = pr.multiProcessorCount;
    out8[1] = pr.clockRate;          // kHz
    out8[2] = pr.memoryClockRate;    // kHz
    out8[3] = pr.memoryBusWidth;
    out8[4] = pr.l2CacheSize;
    out8[5] = int64_t(pr.totalGlobalMem >> 20);
    out8[6] = pr.maxSharedMemoryPerMultiProcessor;
    out8[7] = pr.asicRevision;
    return REMO_OK;
}

int remo_debug_stream(remo_ctx_t *ctx, int64_t bytes, double *read_gbs, double *copy_gbs) {
    if (!ctx || bytes < (1 << 20)) return REMO_ERR_ARG;
    double *a = nullptr, *b = nullptr, *o = nullptr;
    try {
        HIP_TRY(hipSetDevice(ctx->device));
        const int64_t n = bytes / 16 * 2;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&a), sizeof(double) * n));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&b), sizeof(double) * n));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&o), sizeof(double) * 4096 * 4));
        HIP_TRY(hipMemsetAsync(a, 0, sizeof(double) * n, ctx->stream));
        HIP_TRY(hipMemsetAsync(b, 0, sizeof(double) * n, ctx->stream));
        float best_r = 1e30f, best_c = 1e30f;
        for (int rep = 0; rep < 6; ++rep) {
            HIP_TRY(hipEventRecord(ctx->ev[0], ctx->stream));
            hipLaunchKernelGGL(k_stream_read, dim3(4096), dim3(256), 0, ctx->stream, reinterpret_cast<const double2 *>(a), n / 2, o);
            HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));
            HIP_TRY(hipMemcpyAsync(b, a, sizeof(double) * n, hipMemcpyDeviceToDevice, ctx->stream));
            HIP_TRY(hipEventRecord(ctx->ev[2], ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            float r = 0, c = 0;
            (void)hipEventElapsedTime(&r, ctx->ev[0], ctx->ev[1]);
            (void)hipEventElapsedTime(&c, ctx->ev[1], ctx->ev[2]);
            if (r < best_r) best_r = r;
            if (c < best_c) best_c = c;
        }
        if (read_gbs) *read_gbs = double(n) * 8.0 / (double(best_r) * 1e6);
        if (copy_gbs) *copy_gbs = 2.0 * double(n) * 8.0 / (double(best_c) * 1e6);
        if (bytes <= (int64_t(192) << 20) && read_gbs) {   // a buffer that fits the 256 MB Infinity Cache: re-read it back to back
            float best = 1e30f;
            for (int rep = 0; rep < 6; ++rep) {
                hipLaunchKernelGGL(k_stream_read, dim3(4096), dim3(256), 0, ctx->stream, reinterpret_cast<const double2 *>(a), n / 2, o);   // refill
                HIP_TRY(hipEventRecord(ctx->ev[0], ctx->stream));
                for (int k = 0; k < 4; ++k)
                    hipLaunchKernelGGL(k_stream_read, dim3(4096), dim3(256), 0, ctx->stream, reinterpret_cast<const double2 *>(a), n / 2, o);
                HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));
                HIP_TRY(hipStreamSynchronize(ctx->stream));
                float r = 0;
                (void)hipEventElapsedTime(&r, ctx->ev[0], ctx->ev[1]);
                if (r < best) best = r;
            }
            *read_gbs = 4.0 * double(n) * 8.0 / (double(best) * 1e6);
        }
        (void)hipFree(a); (void)hipFree(b); (void)hipFree(o);
        return REMO_OK;
    } catch (const std::exception &ex) {
        if (a) (void)hipFree(a);
        if (b) (void)hipFree(b);
        if (o) (void)hipFree(o);
        return fail(ctx, REMO_ERR_DEVICE, ex.what());
    }
}

int remo_batch_apply_coarse(remo_ctx_t *ctx, remo_batch_t *b, int32_t k, const double *r, double *z, int32_t fp32, int64_t *nv_out) {
    if (!ctx) return REMO_ERR_ARG;
    if (!b || !b->has_system || b->run_id != ctx->run_id || k < 1 || k > REMO_MAX_RHS) return fail(ctx, REMO_ERR_ARG, "bad argument");
    if (b->amg64.levels < 2 || (fp32 && b->amg32.levels < 2)) return fail(ctx, REMO_ERR_ARG, "the last run on this batch built no multigrid hierarchy");
    if (k > b->amg64.kmax) return fail(ctx, REMO_ERR_ARG, "more columns than the hierarchy's level vectors hold (the batch's right-hand sides per chunk)");
    const int64_t nv = b->amg64.lev[0].n;
    if (nv_out) *nv_out = nv;
    if (!r || !z) return REMO_OK;
    double *dr = nullptr, *dz = nullptr, *dpart = nullptr;
    try {
        HIP_TRY(hipSetDevice(ctx->device));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dr), sizeof(double) * (nv * k + 2)));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dz), sizeof(double) * (nv * k + 2)));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dpart), sizeof(double) * kMaxPartialBlocks * 8));
        HIP_TRY(hipMemcpy(dr, r, sizeof(double) * nv * k, hipMemcpyHostToDevice));
        const int nb = cheb_grid(nv);
        if (fp32) launch_amg_cycle<float, double>(b->amg32, k, 0, (const double *)dr, dz, dpart, nb, (const double *)nullptr, ctx->stream);
        else launch_amg_cycle<double, double>(b->amg64, k, 0, (const double *)dr, dz, dpart, nb, (const double *)nullptr, ctx->stream);
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        std::vector<double> dinv(static_cast<size_t>(nv));
        HIP_TRY(hipMemcpy(z, dz, sizeof(double) * nv * k, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(dinv.data(), b->d_dinv, sizeof(double) * nv, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < nv; ++i)
            for (int c = 0; c < k; ++c) z[i * k + c] *= dinv[static_cast<size_t>(i)];   // the cycle stores z / dinv for the direction launch
        (void)hipFree(dr); (void)hipFree(dz); (void)hipFree(dpart);
        return REMO_OK;
    } catch (const std::exception &ex) {
        if (dr) (void)hipFree(dr);
        if (dz) (void)hipFree(dz);
        if (dpart) (void)hipFree(dpart);
        return fail(ctx, REMO_ERR_DEVICE, ex.what());
    }
}

int remo_batch_spmv(remo_ctx_t *ctx, remo_batch_t *b, int32_t k, const double *x, double *y, int32_t reps, double *ms_avg) {
    if (!ctx) return REMO_ERR_ARG;
    if (!b || !b->has_system || b->run_id != ctx->run_id || !x || !y || k < 1 || k > REMO_MAX_RHS || reps < 1)
        return fail(ctx, REMO_ERR_ARG, "bad argument");
    if (b->A.vertex_block_only && !patch_applies(b->A, k))
        return fail(ctx, REMO_ERR_ARG, "the last run assembled no matrix and its patch tables hold fewer columns than asked for (remo_opts_t.assemble = 1 keeps the matrix)");
    double *dx = nullptr, *dy = nullptr;
    try {
        HIP_TRY(hipSetDevice(ctx->device));
        const int64_t n = b->A.n;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dx), sizeof(double) * (n * k + 2)));  // 16 bytes of slack for chunk loads
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dy), sizeof(double) * n * k));
        HIP_TRY(hipMemcpy(dx, x, sizeof(double) * n * k, hipMemcpyHostToDevice));
        const int nb = spmv_grid(n, choose_lanes_per_row(n, b->A.nnz));
        launch_spmm(b->A, k, dx, dy, nullptr, nullptr, nb, ctx->stream);  // warm-up
        HIP_TRY(hipEventRecord(ctx->ev[0], ctx->stream));
        for (int r = 0; r < reps; ++r) launch_spmm(b->A, k, dx, dy, nullptr, nullptr, nb, ctx->stream);
        HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        float ms = 0;
        (void)hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]);
        if (ms_avg) *ms_avg = double(ms) / reps;
        HIP_TRY(hipMemcpy(y, dy, sizeof(double) * n * k, hipMemcpyDeviceToHost));
        (void)hipFree(dx); (void)hipFree(dy);
        return REMO_OK;
    } catch (const std::exception &ex) {
        if (dx) (void)hipFree(dx);
        if (dy) (void)hipFree(dy);
        return fail(ctx, REMO_ERR_DEVICE, ex.what());
    }
}

int remo_debug_grid_barrier(remo_ctx_t *ctx, int32_t nblocks, int32_t nbar, double *out3) {
    if (!ctx || !out3 || nblocks < 1 || nblocks > 2048 || nbar < 1 || nbar > 100000) return REMO_ERR_ARG;
    unsigned *counter = nullptr;
    float *buf = nullptr;
    try {
        HIP_TRY(hipSetDevice(ctx->device));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&counter), 64));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&buf), sizeof(float) * size_t(nblocks) * 256));
        int *fail_flag = reinterpret_cast<int *>(counter) + 4, *mismatch = reinterpret_cast<int *>(counter) + 8;
        hipEvent_t e0, e1;
        HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
        float ms = 0.f;
        for (int rep = 0; rep < 2; ++rep) {     // the second run is the measurement
            HIP_TRY(hipMemsetAsync(counter, 0, 64, ctx->stream));
            HIP_TRY(hipMemsetAsync(buf, 0, sizeof(float) * size_t(nblocks) * 256, ctx->stream));
            HIP_TRY(hipEventRecord(e0, ctx->stream));
            launch_barrier_probe(nblocks, nbar, counter, fail_flag, buf, mismatch, ctx->stream);
            HIP_TRY(hipEventRecord(e1, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        }
        int h[12];
        HIP_TRY(hipMemcpy(h, counter, sizeof h, hipMemcpyDeviceToHost));
        out3[0] = 1e3 * double(ms) / (2.0 * nbar);    // two barriers per iteration
        out3[1] = h[4]; out3[2] = h[8];
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        (void)hipFree(counter); (void)hipFree(buf);
        return REMO_OK;
    } catch (const std::exception &e) {
        (void)hipFree(counter); (void)hipFree(buf);
        return fail(ctx, REMO_ERR_DEVICE, e.what());
    }
}

int remo_debug_patch_phases(remo_ctx_t *ctx, remo_batch_t *b, int32_t fp32, double *out16) {
    if (!ctx || !b || !out16) return REMO_ERR_ARG;
#ifndef REMO_PROBES
    (void)fp32;
    return fail(ctx, REMO_ERR_ARG, "remo_debug_patch_phases: the library was built without -DREMO_PROBES (make -C remo3d_amd/csrc probes)");
#else
    if (!b->has_system || b->run_id != ctx->run_id || !b->A.patch) return fail(ctx, REMO_ERR_ARG, "the last run on this batch did not use the patch operator");
    const int k = 5;
    if (k * b->patch64.t.E > b->patch64.t.block) return fail(ctx, REMO_ERR_ARG, "the batch's patch tables are laid out for fewer than 5 columns");
    double *dx = nullptr, *dy = nullptr;
    long long *st = nullptr;
    try {
        HIP_TRY(hipSetDevice(ctx->device));
        const int64_t n = b->A.n, grid = (b->patch64.t.npatch + 7) / 8 * 8;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dx), sizeof(double) * (n * k + 2)));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dy), sizeof(double) * (n * k + 2)));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&st), sizeof(long long) * grid * 8));
        std::vector<double> hx(size_t(n) * k);
        for (size_t i = 0; i < hx.size(); ++i) hx[i] = double((i * 2654435761u) % 1000) * 1e-3 - 0.5;
        HIP_TRY(hipMemcpy(dx, hx.data(), sizeof(double) * hx.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMemset(st, 0, sizeof(long long) * grid * 8));
        const int nb = spmv_grid(n, choose_lanes_per_row(n, b->A.nnz));
        set_patch_persist(0);     // (this probe is about the one-workgroup-per-patch form; remo_debug_patch_phases_p is the persistent one's)
        set_patch_stamps(st); set_patch_mode(4);
        if (fp32) {     // the fp32 instantiation on the same tables (vectors reinterpreted: timing only)
            CsrViewT<float> A32{n, 0, nullptr, nullptr, nullptr};
            PatchOpT<float> P32{b->patch64.t, reinterpret_cast<float *>(b->patch64.Yb), b->patch64.ppart, b->patch64.lds_rows};
            A32.patch = &P32; A32.vertex_block_only = true;
            for (int rep = 0; rep < 3; ++rep) launch_spmm(A32, k, reinterpret_cast<const float *>(dx), reinterpret_cast<float *>(dy), nullptr, nullptr, nb, ctx->stream);
        } else {
            for (int rep = 0; rep < 3; ++rep) launch_spmm(b->A, k, dx, dy, nullptr, nullptr, nb, ctx->stream);
        }
        set_patch_mode(0); set_patch_stamps(nullptr);
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        // the whole application (apply + reduce launches) under the ablation modes 0 .. 3, microseconds per application
        for (int mode = 0; mode <= 3; ++mode) {
            set_patch_mode(mode);
            CsrViewT<float> A32{n, 0, nullptr, nullptr, nullptr};
            PatchOpT<float> P32{b->patch64.t, reinterpret_cast<float *>(b->patch64.Yb), b->patch64.ppart, b->patch64.lds_rows};
            A32.patch = &P32; A32.vertex_block_only = true;
            auto once = [&]() {
                if (fp32) launch_spmm(A32, k, reinterpret_cast<const float *>(dx), reinterpret_cast<float *>(dy), nullptr, nullptr, nb, ctx->stream);
                else launch_spmm(b->A, k, dx, dy, nullptr, nullptr, nb, ctx->stream);
            };
            once();
            HIP_TRY(hipEventRecord(ctx->ev[0], ctx->stream));
            for (int rep = 0; rep < 10; ++rep) once();
            HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            float ms = 0;
            (void)hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]);
            out16[10 + mode] = 1e3 * double(ms) / 10.0;
        }
        set_patch_mode(0);
        std::vector<long long> h(size_t(grid) * 8);
        HIP_TRY(hipMemcpy(h.data(), st, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
        (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(st);
        for (int i = 0; i < 10; ++i) out16[i] = 0.0;
        long long lo = LLONG_MAX, hi = 0;
        int64_t cnt = 0;
        for (int64_t w = 0; w < grid; ++w) {
            const long long *s8 = h.data() + w * 8;
            if (s8[0] == 0 || s8[7] == 0) continue;
            for (int q = 0; q < 7; ++q) out16[q] += double(s8[q + 1] - s8[q]);
            out16[7] += double(s8[7] - s8[0]);
            lo = std::min(lo, s8[0]); hi = std::max(hi, s8[7]);
            ++cnt;
        }
        for (int q = 0; q < 8; ++q) out16[q] /= double(cnt > 0 ? cnt : 1);
        out16[8] = double(hi - lo);     // first start to last end, clock ticks
        out16[9] = double(cnt);
        set_patch_persist(1);
        return REMO_OK;
    } catch (const std::exception &ex) {
        set_patch_mode(0); set_patch_stamps(nullptr); set_patch_persist(1);
        if (dx) (void)hipFree(dx);
        if (dy) (void)hipFree(dy);
        if (st) (void)hipFree(st);
        return fail(ctx, REMO_ERR_DEVICE, ex.what());
    }
#endif
}

int remo_debug_patch_phases_p(remo_ctx_t *ctx, remo_batch_t *b, int32_t fp32, double *out16) {
    if (!ctx || !b || !out16) return REMO_ERR_ARG;
#ifndef REMO_PROBES
    (void)fp32;
    return fail(ctx, REMO_ERR_ARG, "remo_debug_patch_phases_p: the library was built without -DREMO_PROBES (make -C remo3d_amd/csrc probes)");
#else
    if (!b->has_system || b->run_id != ctx->run_id || !b->A.patch) return fail(ctx, REMO_ERR_ARG, "the last run on this batch did not use the patch operator");
    const int k = 5;
    if (k * b->patch64.t.E > b->patch64.t.block) return fail(ctx, REMO_ERR_ARG, "the batch's patch tables are laid out for fewer than 5 columns");
    double *dx = nullptr, *dy = nullptr;
    long long *st = nullptr;
    const int64_t slots = 8192;      // workgroups the stamp buffer holds
    try {
        HIP_TRY(hipSetDevice(ctx->device));
        const int64_t n = b->A.n;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dx), sizeof(double) * (n * k + 2)));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dy), sizeof(double) * (n * k + 2)));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&st), sizeof(long long) * slots * 12));
        std::vector<double> hx(size_t(n) * k);
        for (size_t i = 0; i < hx.size(); ++i) hx[i] = double((i * 2654435761u) % 1000) * 1e-3 - 0.5;
        HIP_TRY(hipMemcpy(dx, hx.data(), sizeof(double) * hx.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMemset(st, 0, sizeof(long long) * slots * 12));
        const int nb = spmv_grid(n, choose_lanes_per_row(n, b->A.nnz));
        CsrViewT<float> A32{n, 0, nullptr, nullptr, nullptr};
        PatchOpT<float> P32{b->patch64.t, reinterpret_cast<float *>(b->patch64.Yb), b->patch64.ppart, b->patch64.lds_rows};
        A32.patch = &P32; A32.vertex_block_only = true;
        auto once = [&]() {
            if (fp32) launch_spmm(A32, k, reinterpret_cast<const float *>(dx), reinterpret_cast<float *>(dy), nullptr, nullptr, nb, ctx->stream);
            else launch_spmm(b->A, k, dx, dy, nullptr, nullptr, nb, ctx->stream);
        };
        set_patch_stamps(st); set_patch_mode(4);
        for (int rep = 0; rep < 3; ++rep) once();
        set_patch_mode(0); set_patch_stamps(nullptr);
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        once();
        HIP_TRY(hipEventRecord(ctx->ev[0], ctx->stream));
        for (int rep = 0; rep < 10; ++rep) once();
        HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        float ms = 0;
        (void)hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]);
        std::vector<long long> h(size_t(slots) * 12);
        HIP_TRY(hipMemcpy(h.data(), st, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
        (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(st);
        for (int i = 0; i < 16; ++i) out16[i] = 0.0;
        double patches = 0, wgs = 0, longest = 0;
        for (int64_t w = 0; w < slots; ++w) {
            const long long *s12 = h.data() + w * 12;
            if (s12[10] <= 0) continue;
            double tot = 0;
            for (int q = 0; q < 10; ++q) { out16[q] += double(s12[q]); tot += double(s12[q]); }
            patches += double(s12[10]); wgs += 1; longest = std::max(longest, tot);
        }
        for (int q = 0; q < 10; ++q) out16[q] /= (patches > 0 ? patches : 1);     // clock ticks per patch, phase by phase
        out16[10] = patches; out16[11] = wgs; out16[12] = longest; out16[13] = 1e3 * double(ms) / 10.0;
        return REMO_OK;
    } catch (const std::exception &ex) {
        set_patch_mode(0); set_patch_stamps(nullptr);
        if (dx) (void)hipFree(dx);
        if (dy) (void)hipFree(dy);
        if (st) (void)hipFree(st);
        return fail(ctx, REMO_ERR_DEVICE, ex.what());
    }
#endif
}

int remo_debug_tune(int32_t key, int32_t value) {
    // Keys that force ONE OF THE PRODUCT'S OWN PATHS - a choice the library makes by size or dimension, forced so that a small test
    // mesh reaches the code a large batch runs; every setting gives the same operator / preconditioner to rounding: always there.
    switch (key) {
        case 3: set_spmm_tuning(3, value); return 0;    // row schedule of the CSR product
        case 6: g_square = value; return 0;             // paired Chebyshev steps
        case 9: set_fold_first(value); return 0;        // first Chebyshev step inside the update launch
        case 13: g_compact = value; return 0;           // compact copy of the vertex block
        case 15: g_chain32 = value; return 0;           // fp32 Chebyshev chain inside fp64 solves
        case 16: g_amg = value; return 0;               // multigrid cycle on the vertex block
        case 17: g_amg32 = value; return 0;             // ... in fp32 storage
        case 18: set_element_order(value); return 0;    // elements in the caller's order
        case 31: set_tile_update(value); return 0;      // update launch: 64 rows per wave, a value per lane and pass / a k-wide row per lane
        case 30: set_flat_direction(value); return 0;   // direction launch: flat arrays, 16 bytes per lane / a k-wide row per lane
        case 29: set_slab_masked(value); return 0;      // slab slots a row does not have: not fetched / fetched and weighted by zero
        case 25: g_x_in_direction = value; return 0;    // x += alpha p in the direction / in the update launch
        case 22: g_defer_q = value; return 0;           // shared rows summed by k_patch_reduce / by the update launch
        case 24: g_ell = value; return 0;               // fixed-width image of the vertex block
        default: break;
    }
#ifdef REMO_PROBES
    // Keys of rejected experiments and ablations (some give wrong results on purpose): tools/ builds only (make probes)
    if (key == 7) g_sq_lanes = value;
    else if (key == 8) set_symbolic_tuning(value);
    else if (key == 19) set_patch_block(value);
    else if (key == 21) set_patch_mode(value);
    else if (key == 23) set_patch_slab_rows(value);
    else if (key == 26) set_patch_lean(value);
    else if (key == 27) set_slab_ahead(value);
    else if (key == 28) g_dot_bins = value;
    else if (key == 32) set_patch_spread(value);
    else if (key == 33) set_patch_trim(value);
    else if (key == 34) set_patch_persist(value);
    else if (key == 35) set_patch_wgs_per_xcd(value);
    else if (key == 36) g_extra_apply = value;
    else if (key == 37) set_patch_all_slab(value);
    else if (key == 38) set_patch_stagger(value);
    else set_spmm_tuning(key, value);
    return 0;
#else
    (void)value;
    return -1;      // not in this build
#endif
}

int remo_host_element_matrix(int32_t dim, const double *X, double sigma, double *K_out) {
    if ((dim != 2 && dim != 3) || !X || !K_out) return REMO_ERR_ARG;
    const double *M = ref_tables(dim);
    if (dim == 2) {
        double C[9];
        if (!metric_terms<2>(X, sigma, C)) return REMO_ERR_MESH;
        for (int i = 0; i < 10; ++i)
            for (int j = 0; j < 10; ++j) K_out[i * 10 + j] = kentry<2>(C, M, i, j);
    } else {
        double C[6];
        if (!metric_terms<3>(X, sigma, C)) return REMO_ERR_MESH;
        for (int i = 0; i < 20; ++i)
            for (int j = 0; j < 20; ++j) K_out[i * 20 + j] = kentry<3>(C, M, i, j);
    }
    return REMO_OK;
}

double remo_host_factor_error(void) { return ref_factors3_error(); }

int remo_host_symbolic(const remo_mesh_t *mesh, int32_t condense, int64_t *sizes, int32_t *rowptr, int32_t *col, int32_t *freeid) {
    if (!mesh || !sizes) return REMO_ERR_ARG;
    Symbolic sy;
    std::string err;
    const int rc = build_symbolic(*mesh, condense != 0, true, sy, err);
    if (rc != REMO_OK) { g_create_error = err; return rc; }
    sizes[0] = sy.ndof; sizes[1] = sy.nfree; sizes[2] = sy.nnz; sizes[3] = sy.ne; sizes[4] = sy.nf; sizes[5] = sy.nld;
    if (rowptr) std::memcpy(rowptr, sy.rowptr.data(), sizeof(int32_t) * (sy.nfree + 1));
    if (col) std::memcpy(col, sy.col.data(), sizeof(int32_t) * sy.nnz);
    if (freeid) std::memcpy(freeid, sy.freeid.data(), sizeof(int32_t) * sy.ndof);
    return REMO_OK;
}

}  // extern "C"
