// patch.hip — the patch operator (3D, remo_opts_t.op = 3): y = A x without the assembled matrix AND without a slab of element
// results.  Replaces the operator application inside CGSolver(a.mat, ...) (ngsolve_functions.py:50-51); same operator as the
// CSR product of the assembled matrix and as the element-wise operator of kernels.hip (k_elem_apply / k_elem_reduce) to rounding.
//
// The element list (sorted by smallest vertices, symbolic_gpu.hip: neighbours in the list are neighbours in the mesh) is cut into
// PATCHES of E consecutive tetrahedra, one 256-thread workgroup each.  A patch touches ~7.5 distinct matrix rows per element
// instead of 20 (E = 85), and 45 % of them belong to no other patch.  Per patch:
//   1. the x rows of the patch's distinct dofs are staged in LDS - every row read once per patch, whole k-wide rows;
//   2. lane (element, pair of right-hand sides) reads its 20 rows from LDS, runs the factorised reference tensors
//      (gen_elem_code.cpp: g = B x, h = c~ g, y = B^T h - 492 multiply-adds per column) and adds its 20 result rows into LDS
//      accumulators that REUSE the staging area (x lives in registers by then);
//   3. rows interior to the patch go straight to y (and leave their share of <x, y>); rows shared with other patches go to a
//      compact boundary slab, one slot per (row, patch), which k_patch_reduce sums in ascending patch order.
// HBM / L2 traffic per application: ~1.6 x-rows and ~2.5 y-rows of k values per matrix row plus 88 bytes per element, against
// 12 bytes per STORED ENTRY of the CSR product (48 entries per row) and against two passes over a 20-rows-per-element slab in the
// element-wise operator.  The LDS accumulation uses ds_add_f64 / ds_add_f32: the order of the adds inside a patch is not fixed,
// so results are reproducible to rounding (1e-16 relative per row), not bit for bit - the one kernel of the path for which that holds.
#include <hip/hip_runtime.h>
#include <limits.h>

#include <rocprim/device/device_scan.hpp>

#include "kernels.h"
#include "kutil.h"
#include "patch.h"
#include <type_traits>
#include "symbolic_gpu.h"
#include "wave_util.h"

#define REMO_ELEM_NS elem_tables_patch
#include "build/elem_apply.inc"

namespace remo {

namespace {

constexpr int kPatchDotBlocks = 32;                    // workgroups of k_patch_dot
constexpr uint32_t kSlotBits = 13;                      // element dof slots of a patch: E * 20 <= 4096 < 8192 (E <= 204: patch_elements_per_group)
constexpr uint64_t kNoRow = uint64_t(0x7FFFFFFF);      // constrained dof: sorts behind every row

// ---- tables ------------------------------------------------------------------------------------------------------------
// bcnt[r] = number of patches that touch row r if there are at least two (the row gets that many slots in the boundary slab),
// else 0.  The adjacency of a row lists its elements in ascending order, so its patches ascend too.
__global__ void __launch_bounds__(256) k_patch_row_slots(int64_t n, int E, const int32_t *__restrict__ adjptr, const uint32_t *__restrict__ adj,
                                                         int32_t *__restrict__ bcnt, int all_rows) {
    const int64_t r = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (r > n) return;
    int cnt = 0;
    if (r < n) {
        int32_t last = -1;
        for (int32_t a = adjptr[r]; a < adjptr[r + 1]; ++a) {
            const int32_t p = int32_t((adj[a] >> 5) / uint32_t(E));
            cnt += p != last;
            last = p;
        }
    }
    bcnt[r] = cnt >= (all_rows ? 1 : 2) ? cnt : 0;     // all_rows: every row goes through the slab, also the rows of one patch alone (g_patch_all_slab)
}

// One workgroup per patch: sort the patch's (row, element dof slot) pairs in LDS, number the distinct rows 0 .. M - 1 in
// ascending order, and write
//   lidx[element dof]  local row (0xFFFF: constrained dof)
//   prow[p][m]         global row of local row m
//   pout[p][m]         -1 if every element of the row is in this patch, else the row's slot in the boundary slab
//   pcount[p] = M      (raises flag bit 1 and records nothing beyond rows_cap if M > rows_cap)
__global__ void __launch_bounds__(256) k_patch_build(int64_t nt, int E, int rows_cap, int npad, const int32_t *__restrict__ eldof,
                                                     const int32_t *__restrict__ adjptr, const uint32_t *__restrict__ adj,
                                                     uint16_t *__restrict__ lidx, int32_t *__restrict__ pcount, int32_t *__restrict__ pbcnt,
                                                     int32_t *__restrict__ prow, int32_t *__restrict__ pout, int32_t *flag, int32_t *max_rows, int all_rows) {
    extern __shared__ uint64_t keys[];   // [npad]
    __shared__ int32_t cnts[256];
    const int tid = threadIdx.x;
    const int64_t p = blockIdx.x, e0 = p * E;
    const int ne = int(nt - e0 < E ? nt - e0 : E), nslots = ne * 20;
    for (int j = tid; j < npad; j += 256) {
        uint64_t k = ~uint64_t(0);
        if (j < nslots) {
            const int32_t r = eldof[e0 * 20 + j];
            k = ((r >= 0 ? uint64_t(uint32_t(r)) : kNoRow) << kSlotBits) | uint64_t(j);
        }
        keys[j] = k;
    }
    for (int size = 2; size <= npad; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = tid; t < npad / 2; t += 256) {
                const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                const bool up = (lo & size) == 0;
                const uint64_t a = keys[lo], b = keys[hi];
                if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
            }
        }
    __syncthreads();
    const int per = npad / 256, j0 = tid * per;
    auto row_at = [&](int j) -> uint64_t { return keys[j] >> kSlotBits; };
    int heads = 0;
    for (int j = j0; j < j0 + per; ++j) {
        const uint64_t r = row_at(j);
        heads += (j < nslots && r < kNoRow && (j == 0 || row_at(j - 1) != r)) ? 1 : 0;
    }
    cnts[tid] = heads;
    __syncthreads();
    int before = 0, total = 0;
    for (int t = 0; t < 256; ++t) { const int c = cnts[t]; before += t < tid ? c : 0; total += c; }
    if (tid == 0) {
        pcount[p] = total;
        atomicMax(max_rows, total);
        if (total > rows_cap) atomicOr(flag, 1);
    }
    int id = before - 1;     // local row of the most recent head at or before j
    int shared = 0;          // heads of this lane's chunk whose row is also touched by elements of other patches
    for (int j = j0; j < j0 + per && j < nslots; ++j) {
        const uint64_t r = row_at(j);
        const int slot = int(keys[j] & ((uint64_t(1) << kSlotBits) - 1));
        if (r >= kNoRow) { lidx[e0 * 20 + slot] = 0xFFFF; continue; }
        const bool head = (j == 0 || row_at(j - 1) != r);
        id += head ? 1 : 0;
        lidx[e0 * 20 + slot] = uint16_t(id < 0xFFFF ? id : 0xFFFE);
        if (head && id < rows_cap) {
            int run = 1;
            while (j + run < nslots && row_at(j + run) == r) ++run;
            const int32_t row = int32_t(r);
            const bool is_shared = all_rows != 0 || run != adjptr[row + 1] - adjptr[row];
            prow[p * rows_cap + id] = row;
            pout[p * rows_cap + id] = is_shared ? 0 : -1;     // k_patch_slots numbers the shared rows
            shared += is_shared ? 1 : 0;
        }
    }
    __syncthreads();
    cnts[tid] = shared;
    __syncthreads();
    if (tid == 0) {
        int t = 0;
        for (int q = 0; q < 256; ++q) t += cnts[q];
        // all_rows: a patch's block of the slab begins on a 64-byte line in fp64 and fp32 storage alike (16 rows of k values): the linear
        // output phase then stores whole lines only
        pbcnt[p] = all_rows ? ((t + 15) & ~15) : t;
    }
}

// Constrained dofs read (and add into) a row of zeros behind the staged rows: local row = the largest row count of any patch,
// known once every patch is built (device-side value: no host round trip) - so the kernel needs no test per dof.
__global__ void __launch_bounds__(256) k_patch_zero_rows(int64_t nslots, uint16_t *__restrict__ lidx, const int32_t *__restrict__ max_rows) {
    const uint16_t z = uint16_t(max_rows[0] < 0xFFFF ? max_rows[0] : 0xFFFE);
    for (int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; j < nslots; j += int64_t(gridDim.x) * blockDim.x)
        if (lidx[j] == 0xFFFF) lidx[j] = z;
}

// Slab slots.  The slab is PATCH-major: the shared rows of patch p own the slots pboff[p], pboff[p] + 1, ... in ascending row order,
// so a patch writes one contiguous block, and the rows of a block read by k_patch_reduce are neighbours too.  One workgroup per
// patch: pout[p][m] = slot of local row m (-1 stays: not shared), and the row's list bslot[bptr[row] + rank of this patch among the
// patches of the row] = that slot.
__global__ void __launch_bounds__(256) k_patch_slots(int E, int rows_cap, const int32_t *__restrict__ pcount, const int32_t *__restrict__ pboff,
                                                     const int32_t *__restrict__ prow, int32_t *__restrict__ pout, const int32_t *__restrict__ adjptr,
                                                     const uint32_t *__restrict__ adj, const int32_t *__restrict__ bptr, int32_t *__restrict__ bslot,
                                                     int row_major, int32_t *__restrict__ row4) {
    __shared__ int32_t cnts[256];
    const int tid = threadIdx.x;
    const int64_t p = blockIdx.x;
    const int mp = pcount[p] < rows_cap ? pcount[p] : rows_cap;
    const int per = (mp + 255) / 256, m0 = tid * per;
    int mine = 0;
    for (int m = m0; m < m0 + per && m < mp; ++m) mine += pout[p * rows_cap + m] >= 0 ? 1 : 0;
    cnts[tid] = mine;
    __syncthreads();
    int before = 0;
    for (int t = 0; t < tid; ++t) before += cnts[t];
    int32_t slot = pboff[p] + before;
    for (int m = m0; m < m0 + per && m < mp; ++m) {
        if (pout[p * rows_cap + m] < 0) continue;
        const int32_t row = prow[p * rows_cap + m];
        int32_t last = -1, rank = 0;
        for (int32_t a = adjptr[row]; a < adjptr[row + 1]; ++a) {     // patches of the row ascend with its elements
            const int32_t q = int32_t((adj[a] >> 5) / uint32_t(E));
            if (q >= p) break;
            rank += q != last;
            last = q;
        }
        // row_major: the slots of a row lie side by side instead (the readers stream, the patch's stores scatter)
        const int32_t at = row_major ? bptr[row] + rank : slot;
        pout[p * rows_cap + m] = at;
        bslot[bptr[row] + rank] = at;
        // row4[row] = the row's first four slots side by side (-1: none; preset), so that the update launch finds them with ONE load
        // instead of two dependent ones; a row of five or more patches (vertex rows) keeps three there and -2 = "the rest is in bslot"
        if (row4) {
            const int32_t cnt = bptr[row + 1] - bptr[row];
            if (rank < 3 || (rank == 3 && cnt <= 4)) row4[int64_t(row) * 4 + rank] = at;
            else if (rank == 3) row4[int64_t(row) * 4 + 3] = -2;
        }
        ++slot;
    }
}

// ---- apply ---------------------------------------------------------------------------------------------------------------
template <class T> __device__ __forceinline__ void lds_add(T *p, T v) {
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// One workgroup of BLK threads = one patch of E = BLK / K elements; lane = (element, right-hand side).
// The kernel is a chain of dependent memory round trips (tables -> x rows -> LDS -> ... -> stores) around ~1 us of arithmetic, so
// everything a lane will need from the tables is requested BEFORE the first wait - the rows it stages and later writes out (two
// per lane in registers; prow holds -1 behind a patch's last row, so no row count is needed first), where their results go, the
// local rows of its element, the metric terms - which leaves two round trips: tables, then x.
// <x, A x> is summed element by element as g . h (g = B x, h = c~ g: x^T B^T c~ B x): complete when this launch ends, no second
// look at x.  MODE != 0: ablations for tools/probe_patch.py (wrong results on purpose): 1 = plain stores instead of the LDS
// atomics, 2 = no tensor arithmetic (y = x), 3 = nothing leaves the workgroup.
// LEAN: the register-lean order of phase 2 (five waves per SIMD instead of four in fp64): the gradient chain runs input by input
// straight from the staged rows in LDS (no x[20] in registers), BEFORE the staging area is recycled; the metric terms are asked for
// after it; the divergence chain hands every group of outputs to the LDS accumulators as soon as it is complete (no y[20]).
#ifndef REMO_LEAN_WAVES
#define REMO_LEAN_WAVES 5
#endif
template <class T, int K, int BLK, int MODE = 0, bool LEAN = false>
__global__ void __launch_bounds__(BLK, (LEAN ? REMO_LEAN_WAVES * 256 / BLK : 1)) k_patch_apply(PatchTables tb, int rows, const T *__restrict__ x, T *__restrict__ y, T *__restrict__ Yb,
                                                     double *__restrict__ ppart, const double *__restrict__ scal, int step, long long *__restrict__ stamps,
                                                     double *__restrict__ pbins) {
    // every row of the patch goes to its block of the slab (PatchTables::all_slab; the product has no other form)
#ifdef REMO_PROBES
    const bool lin = tb.all_slab != 0;
#else
    constexpr bool lin = true;
#endif
    // lin: the "solve is over" word is asked for HERE but looked at when the row numbers are there (it travels with them, as a vector
    // load, so that no scalar load of the prologue queues behind a word another XCD wrote: one round trip instead of two)
    int done_word = 0;
    if (lin) { if (scal) done_word = load_as_vector(reinterpret_cast<const int *>(scal + kDoneSlot)); }
    else if (scal && solve_done(scal, step)) return;
    // MODE 4: wave 0 of every workgroup leaves the clock at the phase boundaries (remo_debug_patch_phases)
#define REMO_STAMP(k) if constexpr (MODE == 4) { if (threadIdx.x == 0) stamps[(int64_t(blockIdx.x) << 3) + (k)] = __builtin_readcyclecounter(); }
#ifdef REMO_PROBES
    if (tb.stagger > 0 && blockIdx.x < 2048) {     // probe (key 38): the first workgroups of a CU start one after the other, not together
        const int slot = (blockIdx.x >> 8) & 3;
        for (int i = 0; i < slot * tb.stagger; i += 100) __builtin_amdgcn_s_sleep(100);
    }
#endif
    REMO_STAMP(0)
    constexpr int NL = K;
    constexpr int U = kPatchPasses;                      // loads in flight per lane: one trip over up to U * (256 / K) rows
    constexpr int EK = BLK / K;                          // rows per pass of the staging / output phases
    constexpr uint32_t S = sizeof(T);
    const int rows_pad = (rows + U * EK - 1) / (U * EK) * (U * EK);
    extern __shared__ double lds_raw[];
    T *xs = reinterpret_cast<T *>(lds_raw);              // [(rows + 2)][K]: x rows (T), later the accumulators of y (double); row `rows` = zeros, row rows + 1 = slack
    int32_t *trow = reinterpret_cast<int32_t *>(lds_raw + size_t(rows + 2) * K);   // [rows_pad] matrix row of local row m (behind rows + 2 rows of K doubles)
    int32_t *tout = trow + rows_pad;                     // [rows_pad] -1 or slab slot
    __shared__ double smem[16 * K];
    const int tid = threadIdx.x;
    // workgroups b, b + 8, ... share an XCD and its L2: every XCD takes one contiguous eighth of the patches (neighbouring patches
    // share their boundary rows: the second reader finds them in that L2, and the slab rows of one matrix row are written through it)
    const int64_t per = (tb.npatch + 7) >> 3;
    const int64_t p = int64_t(blockIdx.x & 7) * per + int64_t(blockIdx.x >> 3);
    if (p >= tb.npatch) return;
    const int32_t *prow = tb.prow + p * tb.rows_cap, *pout = tb.pout + p * tb.rows_cap;
    // the staging and output phases walk THIS patch's rows, not the largest patch's (`rows`: it sizes LDS and names the zero row): the
    // average patch has 0.6 of the rows of the largest, and a pass without rows still cost its instructions.  The number of passes
    // is a compile-time constant of the code that runs (a switch over 1 .. U: guards inside one unrolled loop keep the loads from
    // going out together and cost what the shorter walk saves); remo_debug_tune key 33 = 0: the largest patch's count for all
    const int el = tid / NL, c0 = tid - el * NL;
    const int32_t off_mask = tid < EK * K ? 0 : int32_t(0x80000000);   // or-ed into a row number: negative = no row for this lane
    // lin: the patch's row count and the row numbers THIS lane stages (local rows el, el + EK, ...: the first U of them) are requested
    // together, straight into registers - no table in LDS, no barrier before the x rows can be asked for.  (The row count as a vector
    // load too: a scalar load would make every later scalar wait - kernel arguments, factor tables - wait for it.)
    const rsrc_t rpr = make_rsrc(prow, uint64_t(tb.rows_cap) * 4);
    int32_t r0[U];
    int rows_own_v = 0, slab0_v = 0;     // (slab0_v: first slab slot of the patch, wanted by the output phase)
    if (lin) {
        rows_own_v = load_as_vector(tb.pcount + p);
        slab0_v = load_as_vector(tb.pboff + p);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int32_t t[1];
            buf_load<int32_t, 1>(rpr, off_mask == 0 ? uint32_t(el + EK * u) * 4u : kOutOfRange, t);
            r0[u] = t[0];
        }
    }
    // the elements of a wave are taken from four runs of the patch's list instead of one (remo_debug_tune key 32: 0 = one run):
    // consecutive elements of the sorted list share their smallest vertices, i.e. they add into the same LDS rows in the same
    // instruction, which serialises - application 138.0 / 125.4 us against 141.3 / 128.2 at 443 k / 424 k tetrahedra.  Lane group el
    // takes element number (el >> 2) of run (el & 3); the runs have ceil((E - r) / 4) elements.
    // (R runs: lane group el takes element el / R of run el % R; run r has ceil((E - r) / R) elements and starts behind the runs before it:
    //  r * floor(E / R) + min(r, E % R))
    const int R = tb.spread > 1 ? tb.spread : 1;
    const int run = el % R, base = tb.E / R, extra = tb.E % R;
    const int elp = R > 1 ? (run * base + (run < extra ? run : extra) + el / R) : el;
    const int64_t e = p * tb.E + elp;
    const bool active = el < tb.E && e < tb.nt;
    uint32_t li[10];
    double cm[6];
#pragma unroll
    for (int q = 0; q < 10; ++q) li[q] = 0u;
#pragma unroll
    for (int q = 0; q < 6; ++q) cm[q] = 0.0;
    // 1a. the patch's row tables into LDS (one round trip; the output phase finds them there again).  prow holds -1 behind a
    // patch's last row, so no row count has to arrive first; the LDS copies are padded with -1 to whole passes of 1b.
    // The table loads are issued BEFORE the element's own data (local rows, metric terms), which is wanted much later: loads
    // return in order, and the wait in front of the LDS copies then covers the tables only.
    // (buffer loads: a lane without an element hands over an offset out of range instead of branching around the loads - with a branch
    // the compiler's count of loads in flight is the smaller one of the two paths, and every later wait for a row number would wait for
    // these, the slowest loads of the prologue, too)
    const rsrc_t rli = make_rsrc(tb.lidx, uint64_t(tb.nt) * 40), rcm = make_rsrc(tb.C, uint64_t(tb.nt) * 48);
    auto element_data = [&]() {
        buf_load<uint32_t, 10>(rli, active ? uint32_t(e) * 40u : kOutOfRange, li);   // 40-byte records: 8-byte aligned
        if constexpr (!(LEAN && MODE != 2)) buf_load<double, 6>(rcm, active ? uint32_t(e) * 48u : kOutOfRange, cm);   // metric terms (1,1) (1,2) (1,3) (2,2) (2,3) (3,3)
    };
    if (lin) element_data();
    const int rows_own = lin ? __builtin_amdgcn_readfirstlane(rows_own_v) : tb.pcount[p];
    if (lin && __builtin_amdgcn_readfirstlane(done_word) != 0 && __builtin_amdgcn_readfirstlane(done_word) <= step) return;   // (solve_done; uniform)
    const int rows_p = (tb.trim && rows_own < rows) ? rows_own : rows;
    const int npass = (rows_p + EK - 1) / EK;
    for (int m0 = tid; !lin && (m0 < rows_pad || m0 == tid); m0 += 4 * BLK) {
        int32_t r[4], o[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int m = m0 + BLK * u;
            r[u] = m < rows ? prow[m] : -1;
            o[u] = (!lin && m < rows) ? pout[m] : -1;
        }
        if (m0 == tid) element_data();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int m = m0 + BLK * u;
            if (m < rows_pad) { trow[m] = r[u]; if (!lin) tout[m] = o[u]; }
        }
    }
    if (!lin) __syncthreads();
    REMO_STAMP(1)
    // 1b. stage the patch's x rows, ONE VALUE per lane and load: in pass u lane (el, c0) takes column c0 of local row el + EK u
    // (EK = 256 / K rows per pass), i.e. value tid + EK K u of the staged image - neighbouring lanes read neighbouring addresses
    // inside a row and across consecutive rows (a patch's rows come in a few runs of consecutive matrix rows), no division per
    // value, and ALL loads of a lane are in flight together (U passes; the phase is one memory round trip, not one per pass: a
    // round trip is ~2-5 k clocks here, the arithmetic of a whole patch ~4 k).  Buffer loads with the hardware range check: a lane
    // without a row hands over an offset beyond the descriptor (no request, no branch); its LDS store goes to the slack row.
    // 24-bit multiplies: rows and slots are below 2^24 (checked by the caller).  Row `rows` is the zero row constrained dofs read.
    const rsrc_t rx = make_rsrc(x, uint64_t(tb.n) * K * S);
    const uint32_t slack = uint32_t((rows + 1) * K + c0);
    auto stage = [&](auto np_c, int m0) {
        constexpr int NP = decltype(np_c)::value;
        int32_t r[NP];
        T v[NP][1];
#pragma unroll
        for (int u = 0; u < NP; ++u) r[u] = trow[m0 + el + EK * u] | off_mask;     // lanes beyond the last whole row of a pass: no row
#pragma unroll
        for (int u = 0; u < NP; ++u)
            buf_load<T, 1>(rx, r[u] >= 0 ? __umul24(uint32_t(r[u]), K * S) + uint32_t(c0) * S : kOutOfRange, v[u]);
#pragma unroll
        for (int u = 0; u < NP; ++u) xs[r[u] >= 0 ? uint32_t((m0 + el + EK * u) * K + c0) : slack] = v[u][0];
    };
#define REMO_PASSES(fn)                                                                                                   \
    switch (npass) {                                                                                                      \
        case 1: fn(std::integral_constant<int, 1>{}, 0); break;   case 2: fn(std::integral_constant<int, 2>{}, 0); break;   \
        case 3: fn(std::integral_constant<int, 3>{}, 0); break;   case 4: fn(std::integral_constant<int, 4>{}, 0); break;   \
        case 5: fn(std::integral_constant<int, 5>{}, 0); break;   case 6: fn(std::integral_constant<int, 6>{}, 0); break;   \
        case 7: fn(std::integral_constant<int, 7>{}, 0); break;   case 8: fn(std::integral_constant<int, 8>{}, 0); break;   \
        case 9: fn(std::integral_constant<int, 9>{}, 0); break;   case 10: fn(std::integral_constant<int, 10>{}, 0); break; \
        case 11: fn(std::integral_constant<int, 11>{}, 0); break; case 12: fn(std::integral_constant<int, 12>{}, 0); break; \
        default:                                                                                                          \
            for (int m0 = 0; m0 < rows_p; m0 += U * EK) fn(std::integral_constant<int, U>{}, m0);                         \
    }
    static_assert(U == 12, "REMO_PASSES lists the cases 1 .. U");
    // lin: the same walk with the row numbers from registers (a patch with more than U passes of rows - the largest few of a batch -
    // asks for the numbers of its later chunks when it gets there)
    auto stage_lin = [&](auto np_c, int m0) {
        constexpr int NP = decltype(np_c)::value;
        int32_t r[NP];
        T v[NP][1];
        if (m0 == 0) {
#pragma unroll
            for (int u = 0; u < NP; ++u) r[u] = r0[u];
        } else {
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                int32_t t[1];
                buf_load<int32_t, 1>(rpr, off_mask == 0 ? uint32_t(m0 + el + EK * u) * 4u : kOutOfRange, t);
                r[u] = t[0];
            }
        }
#pragma unroll
        for (int u = 0; u < NP; ++u) r[u] = (off_mask == 0 && m0 + el + EK * u < rows_own) ? r[u] : -1;
#pragma unroll
        for (int u = 0; u < NP; ++u)
            buf_load<T, 1>(rx, r[u] >= 0 ? __umul24(uint32_t(r[u]), K * S) + uint32_t(c0) * S : kOutOfRange, v[u]);
#pragma unroll
        for (int u = 0; u < NP; ++u) xs[r[u] >= 0 ? uint32_t((m0 + el + EK * u) * K + c0) : slack] = v[u][0];
    };
    if (lin) { REMO_PASSES(stage_lin) }
    else { REMO_PASSES(stage) }
    if (tid < 2 * K) xs[rows * K + tid] = T(0);
    __syncthreads();
    REMO_STAMP(2)
    // 2. my element, my column
    double d0 = 0.0;
    double *ya = lds_raw;
#define REMO_PATCH_L(i) ((li[(i) >> 1] >> (16 * ((i) & 1))) & 0xFFFFu)      /* local row of dof i; constrained dofs: the zero row (k_patch_zero_rows) */
    if constexpr (LEAN && MODE != 2) {
        typedef const T __attribute__((address_space(4))) *ctab_t;
        ctab_t tgi = (ctab_t)ElemTables2<T>::grad_in(), tdo = (ctab_t)ElemTables2<T>::div_out();
        if constexpr (sizeof(T) == 8) { asm volatile("" : "+s"(tgi)); asm volatile("" : "+s"(tdo)); }
        T g[30];
#define REMO_PATCH_XL(i) xs[REMO_PATCH_L(i) * K + c0]
        // (lanes without an element read row 0 and drop the result.)  Scheduling barriers between the blocks of the chain: fp32
        // (factors are literals) 72 registers with them, 88 without; fp64 (factors arrive by scalar loads) must not wait per block
#define REMO_BAR_ON __builtin_amdgcn_sched_barrier(0);
#define REMO_BAR_OFF
        if constexpr (sizeof(T) == 4) { REMO_ELEM_GRAD_IN(T, REMO_PATCH_XL, g, tgi, REMO_BAR_ON) }
        else { REMO_ELEM_GRAD_IN(T, REMO_PATCH_XL, g, tgi, REMO_BAR_OFF) }
#undef REMO_PATCH_XL
        // (pin the chain HERE: its results are used only behind the barriers, inside a branch, and the compiler would sink the
        // arithmetic there - keeping all twenty x values and every factor alive across the barriers)
#pragma unroll
        for (int j = 0; j < 30; ++j) asm volatile("" : "+v"(g[j]));
        __builtin_amdgcn_sched_barrier(0);                   // the metric terms are NOT wanted in registers during the chain above
        if (active) {
            const double *ce = tb.C + e * 6;                 // in flight while the workgroup meets at the two barriers below
#pragma unroll
            for (int q = 0; q < 6; ++q) cm[q] = ce[q];
        }
        __syncthreads();    // every lane has read its x values: the staging area becomes the accumulators
        REMO_STAMP(3)
        for (int j = 2 * tid; j < rows_p * K; j += 2 * BLK) { ya[j] = 0.0; ya[j + 1] = 0.0; }     // (16 bytes per lane; an odd count clears one value of the next row: unused, or the zero row)
        if (tid < K) ya[rows * K + tid] = 0.0;
        __syncthreads();
        REMO_STAMP(4)
        if (active) {
            const T c11 = T(cm[0]), c12 = T(cm[1]), c13 = T(cm[2]), c22 = T(cm[3]), c23 = T(cm[4]), c33 = T(cm[5]);
            T dd = T(0);
#pragma unroll
            for (int m = 0; m < 10; ++m) {     // h = c~ g, in place; g . h on the way
                const T g1 = g[m], g2 = g[10 + m], g3 = g[20 + m];
                const T h1 = c11 * g1 + c12 * g2 + c13 * g3, h2 = c12 * g1 + c22 * g2 + c23 * g3, h3 = c13 * g1 + c23 * g2 + c33 * g3;
                dd += g1 * h1 + g2 * h2 + g3 * h3;
                g[m] = h1; g[10 + m] = h2; g[20 + m] = h3;
            }
            d0 = double(dd);
            if constexpr (sizeof(T) == 8) {      // fp64: the local rows are fetched again (an L1 / L2 hit) rather than held through h = c~ g
                const uint32_t *pl2 = reinterpret_cast<const uint32_t *>(tb.lidx + e * 20);
#pragma unroll
                for (int q = 0; q < 10; ++q) li[q] = __builtin_nontemporal_load(pl2 + q);
            }
#define REMO_PATCH_EMIT(i, v) { if constexpr (MODE == 1) ya[REMO_PATCH_L(i) * K + c0] = double(v); else lds_add(ya + REMO_PATCH_L(i) * K + c0, double(v)); }
            if constexpr (sizeof(T) == 4) { REMO_ELEM_DIV_OUT(T, g, tdo, REMO_PATCH_EMIT, REMO_BAR_ON) }
            else { REMO_ELEM_DIV_OUT(T, g, tdo, REMO_PATCH_EMIT, REMO_BAR_OFF) }
#undef REMO_PATCH_EMIT
#undef REMO_BAR_ON
#undef REMO_BAR_OFF
        }
    } else {
    T xv[20];
    if (active) {
#pragma unroll
        for (int i = 0; i < 20; ++i) xv[i] = xs[REMO_PATCH_L(i) * K + c0];
    }
    __syncthreads();        // every lane holds its x values: the staging area becomes the accumulators
    REMO_STAMP(3)
    // the accumulators are fp64 whatever T is: ds_add_f32 runs at about a lane per clock on this chip (measured: 204 of the 304 us
    // of the fp32 kernel at 443 k tetrahedra were its 20 atomics per lane; ds_add_f64 costs 4 us there)
    for (int j = 2 * tid; j < rows_p * K; j += 2 * BLK) { ya[j] = 0.0; ya[j + 1] = 0.0; }     // (16 bytes per lane; an odd count clears one value of the next row: unused, or the zero row)
    if (tid < K) ya[rows * K + tid] = 0.0;
    __syncthreads();
    REMO_STAMP(4)
    if (active) {
        const T c11 = T(cm[0]), c12 = T(cm[1]), c13 = T(cm[2]), c22 = T(cm[3]), c23 = T(cm[4]), c33 = T(cm[5]);
        T g[30], yv[20];
        if constexpr (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 20; ++i) yv[i] = xv[i] * c11;
        } else {
            // fp64: ONE base address of each factor table in scalar registers (left to itself the compiler forms the address anew,
            // three scalar instructions, before each of its 48 s_load_dwordx16)
            typedef const T __attribute__((address_space(4))) *ctab_t;
            ctab_t tgrad = (ctab_t)ElemTables<T>::grad(), tdiv = (ctab_t)ElemTables<T>::div();
            if constexpr (sizeof(T) == 8) { asm volatile("" : "+s"(tgrad)); asm volatile("" : "+s"(tdiv)); }
            REMO_ELEM_GRAD(T, xv, g, tgrad)
            T dd = T(0);
#pragma unroll
            for (int m = 0; m < 10; ++m) {     // h = c~ g, in place; g . h on the way
                const T g1 = g[m], g2 = g[10 + m], g3 = g[20 + m];
                const T h1 = c11 * g1 + c12 * g2 + c13 * g3, h2 = c12 * g1 + c22 * g2 + c23 * g3, h3 = c13 * g1 + c23 * g2 + c33 * g3;
                dd += g1 * h1 + g2 * h2 + g3 * h3;
                g[m] = h1; g[10 + m] = h2; g[20 + m] = h3;
            }
            d0 = double(dd);
            REMO_ELEM_DIV(T, g, yv, tdiv)
        }
        // the local rows are read a second time for the accumulation (an L1 / L2 hit) instead of being held in ten registers
        // through the tensor arithmetic: that is the difference between three and four waves per SIMD in fp64
        {
            const uint32_t *pl2 = reinterpret_cast<const uint32_t *>(tb.lidx + e * 20);
#pragma unroll
            for (int q = 0; q < 10; ++q) li[q] = __builtin_nontemporal_load(pl2 + q);
        }
#pragma unroll
        for (int i = 0; i < 20; ++i) {
            const uint32_t l = REMO_PATCH_L(i);
            if constexpr (MODE == 1) ya[l * K + c0] = double(yv[i]);
            else lds_add(ya + l * K + c0, double(yv[i]));
        }
    }
    }
#undef REMO_PATCH_L
    __syncthreads();
    REMO_STAMP(5)
    // 3. the accumulators -> the patch's block of the slab, one linear copy: value j of the image to value j of the block.  The block has
    // its own buffer descriptor (64-bit base, the patch's rows as its range): small offsets, no limit on the slab's size, and the
    // hardware drops what lies behind the patch's last row
    if (MODE != 3 && lin) {
        // 16 bytes per lane and store (two fp64 / four fp32 values; the accumulators come out of LDS 16 bytes at a time as well): half
        // the store instructions of a value per lane.  A store that straddles the end of the block loses its dwords beyond it
        // (buffer stores of several dwords are range-checked dword by dword).
        constexpr int VEC = 16 / int(S);
        const rsrc_t rp = make_rsrc(Yb + int64_t(__builtin_amdgcn_readfirstlane(slab0_v)) * K, uint64_t(rows_own) * K * S);
        auto put = [&](auto np_c, int m0) {
            constexpr int NP = decltype(np_c)::value;
            constexpr int NQ = (NP * EK * K + VEC * BLK - 1) / (VEC * BLK);      // NP passes of EK rows = NP EK K values from value m0 K on
            T v[NQ][VEC];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int j = m0 * K + VEC * (tid + BLK * q);                     // (behind the patch's rows: whatever LDS holds, never stored)
#pragma unroll
                for (int i = 0; i < VEC; ++i) v[q][i] = T(ya[j + i]);
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) buf_store<T, VEC>(rp, uint32_t(m0 * K + VEC * (tid + BLK * q)) * S, v[q]);
        };
        REMO_PASSES(put)
    }
    // (probe builds, PatchTables::all_slab = 0: rows of this patch alone -> y; shared rows -> the patch's block of the slab)
    if (MODE != 3 && !lin) {
        const rsrc_t ry = make_rsrc(y, uint64_t(tb.n) * K * S), rb = make_rsrc(Yb, uint64_t(tb.nslot_cap) * K * S);
        auto put = [&](auto np_c, int m0) {
            constexpr int NP = decltype(np_c)::value;
            int32_t r[NP], o[NP];
            T v[NP][1];
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                const int m = m0 + el + EK * u;
                r[u] = trow[m] | off_mask; o[u] = tout[m];
                v[u][0] = T(ya[m * K + c0]);    // (behind the staged rows: whatever LDS holds, never stored)
            }
#pragma unroll
            for (int u = 0; u < NP; ++u) {      // one of the two stores of a value is out of range: dropped by the hardware, no branch
                const bool have = r[u] >= 0;
                buf_store<T, 1>(ry, (have && o[u] < 0) ? __umul24(uint32_t(r[u]), K * S) + uint32_t(c0) * S : kOutOfRange, v[u]);
                buf_store<T, 1>(rb, (have && o[u] >= 0) ? __umul24(uint32_t(o[u]), K * S) + uint32_t(c0) * S : kOutOfRange, v[u]);
            }
        };
        REMO_PASSES(put)
    }
    REMO_STAMP(6)
    if (ppart) {
        double dot[K];
#pragma unroll
        for (int c = 0; c < K; ++c) dot[c] = (c == c0) ? d0 : 0.0;
        const double mine = block_sum_column<K>(dot, smem);     // (not block_sum + pick: that went through scratch memory, kutil.h)
        if (tid < K) {
            // pbins: straight into the consumer's rows (kPqBins of them, patches p, p + kPqBins, ... share one; return-less atomic
            // adds performed at the memory side, complete when the launch ends) instead of a row per patch that a launch folds
            if (pbins) (void)__hip_atomic_fetch_add(pbins + (p % kPqBins) * K + tid, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else ppart[p * K + tid] = mine;
        }
    }
    REMO_STAMP(7)
#undef REMO_STAMP
#undef REMO_PASSES
}

#ifdef REMO_PROBES     // the two persistent forms of the kernel: measured, parity-green, not faster: tools/ builds only
// ---- apply, persistent form --------------------------------------------------------------------------------------------------
// The kernel above is a chain of dependent memory round trips per workgroup: row tables, then x rows, then arithmetic, then stores;
// four workgroups per CU overlap each other's waits only partly (a workgroup spent 11 k of its 27 k cycles waiting for the two trips).
// Here a workgroup is PERSISTENT: it walks a contiguous run of patches (XCD-contiguous like the launch above), and everything patch
// i + 1 needs - its x rows, its row tables (i + 2: the row numbers are the addresses of the x rows), local indices and metric terms of
// its elements - travels while patch i is being computed, by LDS-DMA (`global_load_lds_dword`: per-lane global address, consecutive LDS
// words; no registers hold the data, so the arithmetic phase keeps its register budget):
//   LDS: two row images of (R + 1) k-wide fp64 rows (the image of patch i becomes its accumulators once its values are in
//        registers, the other one fills with patch i + 1), three slots of row numbers, two of slab slots, two of element data.
//   per patch: [zero row] barrier | issue the DMA of patch i + 1 (and the row numbers of i + 2) | x values into registers | barrier |
//        clear accumulators | barrier | tensor chains + ds_add_f64 | s_waitcnt vmcnt(0) (the DMA issued a whole arithmetic phase ago
//        and the stores of patch i - 1) | barrier | rows out (y or slab) - no wait on memory anywhere but that one, long satisfied.
// <x, A x> is kept in a register across the patches of the workgroup and leaves it once, at the end.
// R = the batch's largest row count; the image is sized by it, so two (small patches: three) workgroups share a CU's 160 KB.
#ifndef REMO_STAMP_WAVE
#define REMO_STAMP_WAVE 0        /* which wave of a persistent workgroup reports its phase clocks (probe builds) */
#endif
typedef const void __attribute__((address_space(1))) *dma_src_t;
typedef void __attribute__((address_space(3))) *dma_dst_t;

inline size_t patch_lds_bytes_p(int R, int K, int E) {
    const size_t Rp = size_t((R + 63) & ~63);
    return size_t(2) * size_t(R + 1) * size_t(K) * 8 + 5 * Rp * 4 + 2 * (size_t(E) * 40 + size_t(E) * 48) + 64;
}

// Workgroup barrier that waits for this wave's LDS operations only.  __syncthreads() is fence + barrier, and the fence drains the
// vector-memory counter too (`s_waitcnt vmcnt(0) lgkmcnt(0)` in the ISA): inside the persistent loop that would wait, at the first
// barrier of a patch, for the stores of the patch before it and, at the second, for the LDS-DMA just issued for the next one -
// exactly the two round trips the loop exists to hide (measured: 7.5 k of 20.6 k clock ticks per patch sat in that first barrier).
// Data that crosses the barrier here lives in LDS (ordered by lgkmcnt(0)); what the DMA brings is ordered by the explicit
// `s_waitcnt vmcnt(0)` before barrier B3; nothing a wave stores to global memory is read by another wave of the launch.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <class T, int K, bool STAMP = false>
__global__ void __launch_bounds__(256, 2) k_patch_apply_p(PatchTables tb, int R, const T *__restrict__ x, T *__restrict__ y, T *__restrict__ Yb,
                                                          double *__restrict__ ppart, const double *__restrict__ scal, int step, double *__restrict__ pbins,
                                                          long long *__restrict__ stamps) {
    if (scal && solve_done(scal, step)) return;
    // STAMP (probe builds, remo_debug_patch_phases): wave 0 sums the clock ticks of every phase over the workgroup's patches
    long long ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tk = 0;
#define REMO_PH(k) if constexpr (STAMP) { const long long now_ = __builtin_readcyclecounter(); ph[k] += now_ - tk; tk = now_; }
    constexpr int BLK = 256, EK = BLK / K, U = kPatchPasses;
    constexpr uint32_t S = sizeof(T);
    constexpr int DPR = int(K * S / 4);                  // 4-byte words of a staged row
    extern __shared__ double lds_raw[];
    __shared__ double smem[16 * K];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Rp = (R + 63) & ~63;
    const int E = tb.E;
    double *const Xb0 = lds_raw, *const Xb1 = lds_raw + size_t(R + 1) * K;
    int32_t *const trow0 = reinterpret_cast<int32_t *>(lds_raw + size_t(2) * size_t(R + 1) * K);   // [3][Rp]
    int32_t *const tout0 = trow0 + 3 * Rp;                                                          // [2][Rp]
    uint32_t *const eli0 = reinterpret_cast<uint32_t *>(tout0 + 2 * Rp);                             // [2][E * 10]: local rows, two per word
    double *const ecm0 = reinterpret_cast<double *>(eli0 + 2 * E * 10);                             // [2][E * 6]: metric terms
    // this workgroup's run of patches: XCD b & 7 owns one contiguous eighth of the list (neighbouring patches share boundary rows:
    // the second reader finds them in that L2), its workgroups contiguous runs of the eighth
    const int64_t per = (tb.npatch + 7) >> 3;
    const int64_t xb = int64_t(blockIdx.x & 7) * per, xe = (xb + per < tb.npatch) ? xb + per : tb.npatch;
    const int64_t cnt = xe > xb ? xe - xb : 0;
    const int64_t nwx = int64_t(gridDim.x >> 3), w = int64_t(blockIdx.x >> 3);
    const int64_t share = cnt / nwx, extra = cnt % nwx;
    const int64_t p0_ = xb + w * share + (w < extra ? w : extra);
    // (patch numbers in scalar registers: the row counts of the patches ahead are then scalar loads - a vector load here would be waited
    // for on the spot, together with every store still in flight)
    const int p0 = __builtin_amdgcn_readfirstlane(int(p0_)), p1 = __builtin_amdgcn_readfirstlane(int(p0_ + share + (w < extra ? 1 : 0)));
    if (p0 >= p1) return;                                 // (the whole workgroup)

    // `ndw` consecutive 4-byte words from src to LDS at dst (both wave-uniform), 64 words per wave instruction
    auto dma_copy = [&](const uint32_t *src, uint32_t *dst, int ndw) {
        for (int t = wave; t * 64 < ndw; t += 4) {
            const int j = t * 64 + lane;
            if (j < ndw) __builtin_amdgcn_global_load_lds((dma_src_t)(src + j), (dma_dst_t)(dst + t * 64), 4, 0, 0);
        }
    };
    // the same in 16-byte pieces (1 KB per wave instruction) for sources and destinations aligned to 16 bytes; the tail in 4-byte words
    auto dma_copy16 = [&](const uint32_t *src, uint32_t *dst, int ndw) {
        const int n16 = ndw >> 2;
        for (int t = wave; t * 64 < n16; t += 4) {
            const int j = t * 64 + lane;
            if (j < n16) __builtin_amdgcn_global_load_lds((dma_src_t)(src + 4 * j), (dma_dst_t)(dst + t * 256), 16, 0, 0);
        }
        const int tail = ndw & 3;
        if (wave == 3 && lane < tail) __builtin_amdgcn_global_load_lds((dma_src_t)(src + 4 * n16 + lane), (dma_dst_t)(dst + 4 * n16), 4, 0, 0);
    };
    // the k-wide x rows named by rowtab[0 .. rows) into the image X: word d of the image = word d % DPR of row rowtab[d / DPR].
    // The row numbers of UN pieces are read from LDS together, then the UN pieces leave back to back: one piece at a time, its row
    // number's LDS latency in front of every instruction, cost a wave ~300 clock ticks per piece (7 k ticks per patch: measured)
    auto dma_rows = [&](const int32_t *rowtab, double *X, int rows) {
        const int ndw = rows * DPR;
        const uint32_t *xw = reinterpret_cast<const uint32_t *>(x);
        constexpr int UN = 6;
        for (int t0 = wave; t0 * 64 < ndw; t0 += 4 * UN) {
            uint32_t r[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int d = (t0 + 4 * u) * 64 + lane;
                const int m = int(uint32_t(d) / uint32_t(DPR));
                r[u] = uint32_t(rowtab[m < rows ? m : rows - 1]);
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int t = t0 + 4 * u, d = t * 64 + lane;
                if (d < ndw) {
                    const int m = int(uint32_t(d) / uint32_t(DPR));
                    const uint32_t rr = r[u] < uint32_t(tb.n) ? r[u] : 0u;     // (never out of range by construction; a source address must not depend on that)
                    __builtin_amdgcn_global_load_lds((dma_src_t)(xw + size_t(rr) * DPR + (d - m * DPR)),
                                                     (dma_dst_t)(reinterpret_cast<uint32_t *>(X) + t * 64), 4, 0, 0);
                }
            }
        }
    };
    auto dma_elem = [&](int p, int slot) {
        const int64_t e0 = int64_t(p) * E;
        const int ne = int(tb.nt - e0 < E ? tb.nt - e0 : E);
        dma_copy(reinterpret_cast<const uint32_t *>(tb.lidx + e0 * 20), eli0 + slot * E * 10, ne * 10);
        dma_copy16(reinterpret_cast<const uint32_t *>(tb.C + e0 * 6), reinterpret_cast<uint32_t *>(ecm0 + slot * E * 6), ne * 12);
    };
    const bool tab16 = (tb.rows_cap & 3) == 0;          // a patch's tables start on 16 bytes
    auto dma_tab = [&](const uint32_t *src, uint32_t *dst, int n) { if (tab16) dma_copy16(src, dst, n); else dma_copy(src, dst, n); };
    auto prow_of = [&](int p) { return reinterpret_cast<const uint32_t *>(tb.prow + int64_t(p) * tb.rows_cap); };
    auto pout_of = [&](int p) { return reinterpret_cast<const uint32_t *>(tb.pout + int64_t(p) * tb.rows_cap); };
    auto clampR = [&](int c) { return c < R ? c : R; };

    // (the row counts through the constant address space: invariant during the launch, so the loads are scalar loads even behind stores)
    typedef const int32_t __attribute__((address_space(4))) *cint_t;
    const cint_t pcnt = (cint_t)tb.pcount;
    int rows_cur = clampR(pcnt[p0]);
    int rows_nxt = (p0 + 1 < p1) ? clampR(pcnt[p0 + 1]) : 0;
    // prologue: tables of the first two patches and the element data of the first, then the first patch's rows (the only exposed trips)
    dma_tab(prow_of(p0), reinterpret_cast<uint32_t *>(trow0), rows_cur);
    dma_tab(pout_of(p0), reinterpret_cast<uint32_t *>(tout0), rows_cur);
    if (p0 + 1 < p1) dma_tab(prow_of(p0 + 1), reinterpret_cast<uint32_t *>(trow0 + Rp), rows_nxt);
    dma_elem(p0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    dma_rows(trow0, Xb0, rows_cur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    const int el = tid / K, c0 = tid - el * K;
    const int32_t off_mask = tid < EK * K ? 0 : int32_t(0x80000000);
    const int Rr = tb.spread > 1 ? tb.spread : 1;
    const int run = el % Rr, rbase = E / Rr, rextra = E % Rr;
    const int elp = Rr > 1 ? (run * rbase + (run < rextra ? run : rextra) + el / Rr) : el;
    const rsrc_t ry = make_rsrc(y, uint64_t(tb.n) * K * S), rb = make_rsrc(Yb, uint64_t(tb.nslot_cap) * K * S);
    // y and the slab are two ranges of one arena: when both fit one buffer descriptor (4 GB of byte offsets) a value needs ONE store
    // instruction with the destination chosen by offset, not two of which the hardware drops one - the output phase is bound by the
    // ISSUE of its stores (the waves of a workgroup queue behind each other in it)
    const char *const lo_ = reinterpret_cast<const char *>(y) < reinterpret_cast<const char *>(Yb) ? reinterpret_cast<const char *>(y) : reinterpret_cast<const char *>(Yb);
    const uint64_t y_off64 = uint64_t(reinterpret_cast<const char *>(y) - lo_), b_off64 = uint64_t(reinterpret_cast<const char *>(Yb) - lo_);
    const uint64_t span = (y_off64 + uint64_t(tb.n) * K * S > b_off64 + uint64_t(tb.nslot_cap) * K * S) ? y_off64 + uint64_t(tb.n) * K * S : b_off64 + uint64_t(tb.nslot_cap) * K * S;
    const bool one_desc = span < 0xFFFFF000ull;
    const rsrc_t rboth = make_rsrc(lo_, one_desc ? span : 0);
    const uint32_t y_off = uint32_t(y_off64), b_off = uint32_t(b_off64);
    double d0 = 0.0;
    int slot3 = 0;                                        // slot of the current patch's row numbers (patch number mod 3, without the division)
    for (int p = p0; p < p1; ++p) {
        const int it = p - p0;
        double *const Xc = (it & 1) ? Xb1 : Xb0, *const Xn = (it & 1) ? Xb0 : Xb1;
        T *const xs = reinterpret_cast<T *>(Xc);
        const int s1 = slot3 == 2 ? 0 : slot3 + 1, s2 = s1 == 2 ? 0 : s1 + 1;
        int32_t *const trow_c = trow0 + slot3 * Rp, *const trow_n = trow0 + s1 * Rp, *const trow_nn = trow0 + s2 * Rp;
        int32_t *const tout_c = tout0 + (it & 1) * Rp, *const tout_n = tout0 + ((it + 1) & 1) * Rp;
        const int rows_nn = (p + 2 < p1) ? clampR(pcnt[p + 2]) : 0;           // (a scalar load: wanted at the next turn)
        if constexpr (STAMP) tk = __builtin_readcyclecounter();
        if (tid < K) xs[R * K + tid] = T(0);              // the row constrained dofs read
        if constexpr (STAMP) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); REMO_PH(9) }      // (probe: this wave's own LDS / scalar operations, apart from the barrier)
        lds_barrier();                                   // B0: X(p) is in LDS (every wave waited for its pieces); the other image and the oldest table slots are free
        REMO_PH(0)                                          // barrier B0
        if (p + 1 < p1) {
            dma_rows(trow_n, Xn, rows_nxt);
            dma_tab(pout_of(p + 1), reinterpret_cast<uint32_t *>(tout_n), rows_nxt);
            dma_elem(p + 1, (it + 1) & 1);
        }
        if (p + 2 < p1) dma_tab(prow_of(p + 2), reinterpret_cast<uint32_t *>(trow_nn), rows_nn);
        REMO_PH(1)                                          // DMA issue
        const int64_t e = int64_t(p) * E + elp;
        const bool active = el < E && e < tb.nt;
        const uint32_t *const eli = eli0 + (it & 1) * E * 10 + elp * 10;
        const double *const ecm = ecm0 + (it & 1) * E * 6 + elp * 6;
        uint32_t li[10];
#pragma unroll
        for (int q = 0; q < 10; ++q) li[q] = active ? eli[q] : 0u;
#define REMO_PATCH_L(i) ((li[(i) >> 1] >> (16 * ((i) & 1))) & 0xFFFFu)
        T xv[20];
#pragma unroll
        for (int i = 0; i < 20; ++i) xv[i] = xs[REMO_PATCH_L(i) * K + c0];
        REMO_PH(2)                                          // x values into registers
        lds_barrier();                                   // B1: every lane holds its x values: the image becomes the accumulators
        REMO_PH(3)
        double *const ya = Xc;
        for (int j = tid; j < rows_cur * K; j += BLK) ya[j] = 0.0;
        if (tid < K) ya[R * K + tid] = 0.0;
        lds_barrier();                                   // B2
        REMO_PH(4)                                          // clearing + B2
        if (active) {
            double cm[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) cm[q] = ecm[q];
            const T c11 = T(cm[0]), c12 = T(cm[1]), c13 = T(cm[2]), c22 = T(cm[3]), c23 = T(cm[4]), c33 = T(cm[5]);
            T g[30], yv[20];
            typedef const T __attribute__((address_space(4))) *ctab_t;
            ctab_t tgrad = (ctab_t)ElemTables<T>::grad(), tdiv = (ctab_t)ElemTables<T>::div();
            if constexpr (sizeof(T) == 8) { asm volatile("" : "+s"(tgrad)); asm volatile("" : "+s"(tdiv)); }
            REMO_ELEM_GRAD(T, xv, g, tgrad)
            T dd = T(0);
#pragma unroll
            for (int m = 0; m < 10; ++m) {     // h = c~ g, in place; g . h on the way
                const T g1 = g[m], g2 = g[10 + m], g3 = g[20 + m];
                const T h1 = c11 * g1 + c12 * g2 + c13 * g3, h2 = c12 * g1 + c22 * g2 + c23 * g3, h3 = c13 * g1 + c23 * g2 + c33 * g3;
                dd += g1 * h1 + g2 * h2 + g3 * h3;
                g[m] = h1; g[10 + m] = h2; g[20 + m] = h3;
            }
            d0 += double(dd);
            REMO_ELEM_DIV(T, g, yv, tdiv)
#pragma unroll
            for (int i = 0; i < 20; ++i) lds_add(ya + REMO_PATCH_L(i) * K + c0, double(yv[i]));
        }
#undef REMO_PATCH_L
        REMO_PH(5)                                          // tensor chains + LDS accumulation
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of patch p + 1 (issued an arithmetic phase ago) and its stores of patch p - 1
        REMO_PH(6)                                          // wait for the DMA / older stores
        lds_barrier();                                   // B3: accumulators complete; X(p + 1), tables and element data of p + 1 in LDS
        REMO_PH(7)
        auto rows_out = [&](auto one_c) {
            constexpr bool ONE_STORE = decltype(one_c)::value;
            auto put = [&](auto np_c, int m0) {
                constexpr int NP = decltype(np_c)::value;
                int32_t r[NP], o[NP];
                T v[NP][1];
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    const int m = m0 + el + EK * u;
                    const bool in = m < rows_cur;
                    r[u] = (in ? trow_c[m] : -1) | off_mask; o[u] = in ? tout_c[m] : -1;
                    v[u][0] = T(ya[(in ? m : 0) * K + c0]);
                }
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    if constexpr (ONE_STORE) {      // ONE store per value: y and the slab through one descriptor (base = the lower of the two)
                        const bool have = r[u] >= 0;
                        const uint32_t off = o[u] < 0 ? y_off + __umul24(uint32_t(r[u]), K * S) : b_off + __umul24(uint32_t(o[u]), K * S);
                        buf_store<T, 1>(rboth, have ? off + uint32_t(c0) * S : kOutOfRange, v[u]);
                    } else {                        // one of the two stores of a value is out of range: dropped by the hardware, no branch
                        const bool have = r[u] >= 0;
                        buf_store<T, 1>(ry, (have && o[u] < 0) ? __umul24(uint32_t(r[u]), K * S) + uint32_t(c0) * S : kOutOfRange, v[u]);
                        buf_store<T, 1>(rb, (have && o[u] >= 0) ? __umul24(uint32_t(o[u]), K * S) + uint32_t(c0) * S : kOutOfRange, v[u]);
                    }
                }
            };
            const int npass = (rows_cur + EK - 1) / EK;
            switch (npass) {
                case 1: put(std::integral_constant<int, 1>{}, 0); break;   case 2: put(std::integral_constant<int, 2>{}, 0); break;
                case 3: put(std::integral_constant<int, 3>{}, 0); break;   case 4: put(std::integral_constant<int, 4>{}, 0); break;
                case 5: put(std::integral_constant<int, 5>{}, 0); break;   case 6: put(std::integral_constant<int, 6>{}, 0); break;
                case 7: put(std::integral_constant<int, 7>{}, 0); break;   case 8: put(std::integral_constant<int, 8>{}, 0); break;
                case 9: put(std::integral_constant<int, 9>{}, 0); break;   case 10: put(std::integral_constant<int, 10>{}, 0); break;
                case 11: put(std::integral_constant<int, 11>{}, 0); break; case 12: put(std::integral_constant<int, 12>{}, 0); break;
                default:
                    for (int m0 = 0; m0 < rows_cur; m0 += U * EK) put(std::integral_constant<int, U>{}, m0);
            }
        };
        if (one_desc) rows_out(std::true_type{}); else rows_out(std::false_type{});
        REMO_PH(8)                                          // rows out
        rows_cur = rows_nxt; rows_nxt = rows_nn;
        slot3 = s1;
    }
    if constexpr (STAMP) {
        if (tid == 64 * REMO_STAMP_WAVE) {
            for (int k = 0; k < 10; ++k) stamps[int64_t(blockIdx.x) * 12 + k] = ph[k];
            stamps[int64_t(blockIdx.x) * 12 + 10] = p1 - p0;
        }
    }
#undef REMO_PH
    if (ppart) {
        double dot[K];
#pragma unroll
        for (int c = 0; c < K; ++c) dot[c] = (c == c0) ? d0 : 0.0;
        const double mine = block_sum_column<K>(dot, smem);
        if (tid < K) {
            if (pbins) (void)__hip_atomic_fetch_add(pbins + (blockIdx.x % kPqBins) * K + tid, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else ppart[int64_t(p0) * K + tid] = mine;
        }
        if (!pbins)      // one row of sums per PATCH is what the folding launches read: this workgroup's other patches contribute zero
            for (int64_t j = tid; j < int64_t(p1 - p0 - 1) * K; j += BLK) ppart[int64_t(p0 + 1) * K + j] = 0.0;
    }
}

// ---- apply, persistent form with the prefetch through REGISTERS ----------------------------------------------------------------
// Same idea as k_patch_apply_p - a workgroup walks a run of patches and fetches patch i + 1 while it computes patch i - with
// what the LDS-DMA version taught: the DMA instructions cost a wave ~240 ticks each to issue (25 per patch) and the second row
// image halves the workgroups per CU.  Here the next patch's x rows are ordinary buffer loads into registers (one value per lane and
// pass, up to 12 in flight: 24 VGPRs in fp64), issued before the tensor chains and written into the ONE row image once the
// current patch's results have been read out of it; row tables travel the same way one patch further ahead, the element's local
// rows and metric terms are requested behind the chains (their registers are free then) for the next turn.  LDS: one image + two
// slots of tables (~32 KB: the register budget - three waves per SIMD - decides the residency, not LDS).  Barriers wait for LDS
// only (lds_barrier): a wave never waits for memory except where the compiler counts the loads it needs next.
inline size_t patch_lds_bytes_r(int R, int K, int E) {
    const size_t Rp = size_t((R + 255) & ~255);
    return size_t(R + 2) * size_t(K) * 8 + 4 * Rp * 4 + size_t(E) * 88 + 64;
}

template <class T, int K, bool STAMP = false>
__global__ void __launch_bounds__(256, 3) k_patch_apply_r(PatchTables tb, int R, const T *__restrict__ x, T *__restrict__ y, T *__restrict__ Yb,
                                                          double *__restrict__ ppart, const double *__restrict__ scal, int step, double *__restrict__ pbins,
                                                          long long *__restrict__ stamps) {
    if (scal && solve_done(scal, step)) return;
    long long ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tk = 0;
#define REMO_PH(k) if constexpr (STAMP) { const long long now_ = __builtin_readcyclecounter(); ph[k] += now_ - tk; tk = now_; }
    constexpr int BLK = 256, EK = BLK / K, U = kPatchPasses, TJ = 4;     // TJ table words per lane: up to 1024 rows
    constexpr uint32_t S = sizeof(T);
    extern __shared__ double lds_raw[];
    __shared__ double smem[16 * K];
    const int tid = threadIdx.x;
    const int Rp = (R + 255) & ~255;
    const int E = tb.E;
    double *const ya = lds_raw;                                          // [(R + 2)][K] fp64 accumulators; the same bytes hold the x image (T) before
    T *const xs = reinterpret_cast<T *>(lds_raw);
    int32_t *const trow0 = reinterpret_cast<int32_t *>(lds_raw + size_t(R + 2) * K);   // [2][Rp]
    int32_t *const tout0 = trow0 + 2 * Rp;                                              // [2][Rp]
    double *const ecm = reinterpret_cast<double *>(tout0 + 2 * Rp);                    // [E][6] metric terms of the current patch's elements
    uint32_t *const eli = reinterpret_cast<uint32_t *>(ecm + size_t(E) * 6);           // [E][10] their local rows, two per word
    const int64_t per = (tb.npatch + 7) >> 3;
    const int64_t xb = int64_t(blockIdx.x & 7) * per, xe = (xb + per < tb.npatch) ? xb + per : tb.npatch;
    const int64_t cnt = xe > xb ? xe - xb : 0;
    const int64_t nwx = int64_t(gridDim.x >> 3), w = int64_t(blockIdx.x >> 3);
    const int64_t share = cnt / nwx, extra = cnt % nwx;
    const int64_t p0_ = xb + w * share + (w < extra ? w : extra);
    const int p0 = __builtin_amdgcn_readfirstlane(int(p0_)), p1 = __builtin_amdgcn_readfirstlane(int(p0_ + share + (w < extra ? 1 : 0)));
    if (p0 >= p1) return;                                 // (the whole workgroup)
    typedef const int32_t __attribute__((address_space(4))) *cint_t;
    const cint_t pcnt = (cint_t)tb.pcount;
    auto clampR = [&](int c) { return c < R ? c : R; };

    const int el = tid / K, c0 = tid - el * K;
    const int32_t off_mask = tid < EK * K ? 0 : int32_t(0x80000000);
    const int Rr = tb.spread > 1 ? tb.spread : 1;
    const int run = el % Rr, rbase = E / Rr, rextra = E % Rr;
    const int elp = Rr > 1 ? (run * rbase + (run < rextra ? run : rextra) + el / Rr) : el;
    const bool has_el = el < E;
    const rsrc_t rx = make_rsrc(x, uint64_t(tb.n) * K * S);
    const rsrc_t ry = make_rsrc(y, uint64_t(tb.n) * K * S), rb = make_rsrc(Yb, uint64_t(tb.nslot_cap) * K * S);
    const rsrc_t rpr = make_rsrc(tb.prow, uint64_t(tb.npatch) * uint64_t(tb.rows_cap) * 4), rpo = make_rsrc(tb.pout, uint64_t(tb.npatch) * uint64_t(tb.rows_cap) * 4);

    // this lane's table words of patch p (rows tid + 256 j): range-checked buffer loads, nothing beyond the patch's rows is requested
    auto load_tables = [&](int p, int rows, int32_t (&tr)[TJ], int32_t (&to)[TJ]) {
        const uint32_t base = uint32_t(p) * uint32_t(tb.rows_cap);
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int m = tid + 256 * j;
            const uint32_t off = m < rows ? (base + uint32_t(m)) * 4u : kOutOfRange;
            tr[j] = int32_t(__builtin_amdgcn_raw_buffer_load_b32(rpr, off, 0, 0));
            to[j] = int32_t(__builtin_amdgcn_raw_buffer_load_b32(rpo, off, 0, 0));
        }
    };
    auto store_tables = [&](int slot, int rows, const int32_t (&tr)[TJ], const int32_t (&to)[TJ]) {
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int m = tid + 256 * j;
            if (m < rows) { trow0[slot * Rp + m] = tr[j]; tout0[slot * Rp + m] = to[j]; }
        }
    };
    // x values of the patch whose row numbers are in table slot `slot`: pass u = row el + EK u, this lane's column
    auto load_x = [&](int slot, int rows, T (&xr)[U]) {
        const int32_t *const tr = trow0 + slot * Rp;
        int el_l = el;
        asm volatile("" : "+v"(el_l));
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int m = el_l + EK * u;
            int32_t r = -1;
            if (u * EK < rows) r = (m < rows ? tr[m] : -1) | off_mask;
            T v[1];
            buf_load<T, 1>(rx, r >= 0 ? __umul24(uint32_t(r), K * S) + uint32_t(c0) * S : kOutOfRange, v);
            xr[u] = v[0];
        }
    };
    auto store_x = [&](int rows, const T (&xr)[U]) {
        int el_s = el;
        asm volatile("" : "+v"(el_s));                   // (no loop-invariant addresses held across the patch loop: see the read-out of the results)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int m = el_s + EK * u;
            if (m < rows && off_mask == 0) xs[m * K + c0] = xr[u];
        }
    };
    // local rows and metric terms of this lane's element of patch p into registers (lanes of an element ask for the same words)
    auto load_elem = [&](int p, uint32_t (&li)[10], double (&cm)[6]) {
        const int64_t e = int64_t(p) * E + elp;
        const bool active = has_el && e < tb.nt;
#pragma unroll
        for (int q = 0; q < 10; ++q) li[q] = 0u;
#pragma unroll
        for (int q = 0; q < 6; ++q) cm[q] = 0.0;
        if (active) {
            const uint32_t *pl = reinterpret_cast<const uint32_t *>(tb.lidx + e * 20);
#pragma unroll
            for (int q = 0; q < 10; ++q) li[q] = pl[q];
            const double *ce = tb.C + e * 6;
#pragma unroll
            for (int q = 0; q < 6; ++q) cm[q] = ce[q];
        }
    };
    // ... and from registers into the workgroup's LDS copy (the k lanes of an element write the same words): they are wanted three
    // times per patch - x values, metric contraction, accumulation - and would otherwise sit in 22 registers through the tensor chains
    auto park_elem = [&](const uint32_t (&li)[10], const double (&cm)[6]) {
        if (has_el && c0 == 0) {
#pragma unroll
            for (int q = 0; q < 10; ++q) eli[elp * 10 + q] = li[q];
#pragma unroll
            for (int q = 0; q < 6; ++q) ecm[elp * 6 + q] = cm[q];
        }
    };

    int rows_cur = clampR(pcnt[p0]);
    int rows_nxt = (p0 + 1 < p1) ? clampR(pcnt[p0 + 1]) : 0;
    // in flight across the loop's back edge: the NEXT patch's element data and row tables (requested behind the tensor chains)
    uint32_t li_n[10];
    double cm_n[6];
    int32_t tr_n[TJ], to_n[TJ];
    {   // prologue: the first patch's tables, rows and element data (the only exposed round trips), the second patch's tables on their way
        int32_t tr[TJ], to[TJ];
        load_tables(p0, rows_cur, tr, to);
        load_elem(p0, li_n, cm_n);
        load_tables(p0 + 1 < p1 ? p0 + 1 : p0, rows_nxt, tr_n, to_n);
        store_tables(0, rows_cur, tr, to);
        __syncthreads();
        T xr[U];
        load_x(0, rows_cur, xr);
        store_x(rows_cur, xr);
    }
    double d0 = 0.0;
    for (int p = p0; p < p1; ++p) {
        const int it = p - p0, sc = it & 1, sn = sc ^ 1;
        const int32_t *const trow_c = trow0 + sc * Rp, *const tout_c = tout0 + sc * Rp;
        const int rows_nn = (p + 2 < p1) ? clampR(pcnt[p + 2]) : 0;
        if constexpr (STAMP) tk = __builtin_readcyclecounter();
        if (tid < K) xs[R * K + tid] = T(0);              // the row constrained dofs read
        park_elem(li_n, cm_n);                           // element data of patch p (requested a patch ago)
        store_tables(sn, rows_nxt, tr_n, to_n);          // tables of patch p + 1 into the slot patch p - 1 has left
        lds_barrier();                                   // B0: the image holds X(p), slot sc the tables of p, slot sn those of p + 1
        REMO_PH(0)
        const int64_t e = int64_t(p) * E + elp;
        const bool active = has_el && e < tb.nt;
        const uint32_t *const my_li = eli + (has_el ? elp : 0) * 10;
        const double *const my_cm = ecm + (has_el ? elp : 0) * 6;
        T xv[20];
        {
            uint32_t li[10];
#pragma unroll
            for (int q = 0; q < 10; ++q) li[q] = my_li[q];
#define REMO_PATCH_L(i) ((li[(i) >> 1] >> (16 * ((i) & 1))) & 0xFFFFu)
#pragma unroll
            for (int i = 0; i < 20; ++i) xv[i] = xs[REMO_PATCH_L(i) * K + c0];
        }
        REMO_PH(2)
        lds_barrier();                                   // B1: every lane holds its x values: the image becomes the accumulators
        REMO_PH(3)
        for (int j = tid; j < rows_cur * K; j += BLK) ya[j] = 0.0;
        if (tid < K) ya[R * K + tid] = 0.0;
        lds_barrier();                                   // B2
        REMO_PH(4)
        typedef const T __attribute__((address_space(4))) *ctab_t;
        T g[30];
#pragma unroll
        for (int j = 0; j < 30; ++j) g[j] = T(0);
        if (active) {
            ctab_t tgrad = (ctab_t)ElemTables<T>::grad();
            if constexpr (sizeof(T) == 8) asm volatile("" : "+s"(tgrad));
            REMO_ELEM_GRAD(T, xv, g, tgrad)
        }
        // patch p + 1's x values are requested HERE, between the two tensor chains: the twenty x values of this patch are dead, the
        // thirty gradients alive - the point of lowest register pressure that still leaves the loads half the arithmetic phase, two
        // barriers and the read-out of the results to arrive (nothing is requested behind the run's last patch: rows_nxt = 0)
#pragma unroll
        for (int j = 0; j < 30; ++j) asm volatile("" : "+v"(g[j]));      // (the first chain is complete here, not sunk behind the loads)
        T xr[U];
        load_x(sn, rows_nxt, xr);
        REMO_PH(1)
        if (active) {
            T yv[20];
            ctab_t tdiv = (ctab_t)ElemTables<T>::div();
            if constexpr (sizeof(T) == 8) asm volatile("" : "+s"(tdiv));
            {
                const T c11 = T(my_cm[0]), c12 = T(my_cm[1]), c13 = T(my_cm[2]), c22 = T(my_cm[3]), c23 = T(my_cm[4]), c33 = T(my_cm[5]);
                T dd = T(0);
#pragma unroll
                for (int m = 0; m < 10; ++m) {     // h = c~ g, in place; g . h on the way
                    const T g1 = g[m], g2 = g[10 + m], g3 = g[20 + m];
                    const T h1 = c11 * g1 + c12 * g2 + c13 * g3, h2 = c12 * g1 + c22 * g2 + c23 * g3, h3 = c13 * g1 + c23 * g2 + c33 * g3;
                    dd += g1 * h1 + g2 * h2 + g3 * h3;
                    g[m] = h1; g[10 + m] = h2; g[20 + m] = h3;
                }
                d0 += double(dd);
            }
            REMO_ELEM_DIV(T, g, yv, tdiv)
            uint32_t li[10];
#pragma unroll
            for (int q = 0; q < 10; ++q) li[q] = my_li[q];
#pragma unroll
            for (int i = 0; i < 20; ++i) lds_add(ya + REMO_PATCH_L(i) * K + c0, double(yv[i]));
#undef REMO_PATCH_L
        }
        REMO_PH(5)
        // the next patch's element data and the row tables of the one after it: requested here, behind the chains (their registers
        // are free now), parked in LDS at the top of the next turn
        load_elem(p + 1 < p1 ? p + 1 : p, li_n, cm_n);
        load_tables(p + 2 < p1 ? p + 2 : p, rows_nn, tr_n, to_n);
        lds_barrier();                                   // B3: accumulators complete
        REMO_PH(6)
        // results of patch p out of LDS into registers (the image is about to receive patch p + 1), stored after that
        int32_t orow[U], oslot[U];
        T ov[U];
        const int npass = (rows_cur + EK - 1) / EK;
        // (the LDS addresses of the passes are loop invariants; hoisted out of the patch loop they would sit in a dozen registers through
        // the tensor chains - i.e. in scratch, and a scratch reload waits for every load still in flight: formed anew from an opaque copy)
        int el_o = el;
        asm volatile("" : "+v"(el_o));
#pragma unroll
        for (int u = 0; u < U; ++u) {
            orow[u] = -1; oslot[u] = -1; ov[u] = T(0);
            if (u < npass) {
                const int m = el_o + EK * u;
                const bool in = m < rows_cur;
                orow[u] = (in ? trow_c[m] : -1) | off_mask; oslot[u] = in ? tout_c[m] : -1;
                ov[u] = T(ya[(in ? m : 0) * K + c0]);
            }
        }
        lds_barrier();                                   // B4: every lane has its results: the image is free
        REMO_PH(7)
        store_x(rows_nxt, xr);                           // X(p + 1) (the compiler waits for exactly these loads: what was requested after them stays in flight)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (u < npass) {                             // one of the two stores of a value is out of range: dropped by the hardware, no branch
                const bool have = orow[u] >= 0;
                T v[1] = {ov[u]};
                buf_store<T, 1>(ry, (have && oslot[u] < 0) ? __umul24(uint32_t(orow[u]), K * S) + uint32_t(c0) * S : kOutOfRange, v);
                buf_store<T, 1>(rb, (have && oslot[u] >= 0) ? __umul24(uint32_t(oslot[u]), K * S) + uint32_t(c0) * S : kOutOfRange, v);
            }
        }
        REMO_PH(8)
        rows_cur = rows_nxt; rows_nxt = rows_nn;
    }
    if constexpr (STAMP) {
        if (tid == 64 * REMO_STAMP_WAVE) {
            for (int k = 0; k < 10; ++k) stamps[int64_t(blockIdx.x) * 12 + k] = ph[k];
            stamps[int64_t(blockIdx.x) * 12 + 10] = p1 - p0;
        }
    }
#undef REMO_PH
    if (ppart) {
        double dot[K];
#pragma unroll
        for (int c = 0; c < K; ++c) dot[c] = (c == c0) ? d0 : 0.0;
        const double mine = block_sum_column<K>(dot, smem);
        if (tid < K) {
            if (pbins) (void)__hip_atomic_fetch_add(pbins + (blockIdx.x % kPqBins) * K + tid, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else ppart[int64_t(p0) * K + tid] = mine;
        }
        if (!pbins)
            for (int64_t j = tid; j < int64_t(p1 - p0 - 1) * K; j += BLK) ppart[int64_t(p0 + 1) * K + j] = 0.0;
    }
}

#endif   // REMO_PROBES

// Rows shared by several patches: sum of the row's slab slots in ascending patch order.  DOT: the patches' <x, A x> are folded
// into <= 1024 partial rows for the consumer (every workgroup takes a fixed subset: deterministic given the patches' sums).
template <class T, int K, bool DOT>
__global__ void __launch_bounds__(256) k_patch_reduce(int64_t n, int64_t npatch, const int32_t *__restrict__ bptr, const int32_t *__restrict__ bslot,
                                                      const T *__restrict__ Yb,
                                                      T *__restrict__ y, const double *__restrict__ ppart,
                                                      double *__restrict__ part, const double *__restrict__ scal, int step) {
    if (scal && solve_done(scal, step)) return;
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t row = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; row < n; row += stride) {
        const int32_t b0 = bptr[row], b1 = bptr[row + 1];
        if (b1 <= b0) continue;
        T acc[K];
#pragma unroll
        for (int c = 0; c < K; ++c) acc[c] = T(0);
        for (int32_t s0 = b0; s0 < b1; s0 += 2) {     // two slots in flight (a row has 2-4), ascending patches
            T v0[K], v1[K];
            const bool second = s0 + 1 < b1;
            const int64_t a0 = bslot[s0], a1 = second ? bslot[s0 + 1] : 0;
#pragma unroll
            for (int c = 0; c < K; ++c) v0[c] = Yb[a0 * K + c];
#pragma unroll
            for (int c = 0; c < K; ++c) v1[c] = second ? Yb[a1 * K + c] : T(0);
#pragma unroll
            for (int c = 0; c < K; ++c) acc[c] = (acc[c] + v0[c]) + v1[c];
        }
#pragma unroll
        for (int c = 0; c < K; ++c) y[row * K + c] = acc[c];
    }
    if (DOT) {
        double dot[K];
#pragma unroll
        for (int c = 0; c < K; ++c) dot[c] = 0.0;
        for (int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; p < npatch; p += stride)
#pragma unroll
            for (int c = 0; c < K; ++c) dot[c] += ppart[p * K + c];
        __shared__ double smem[16 * K];
        block_sum<K>(dot, smem);
        if (threadIdx.x < K) part[blockIdx.x * K + threadIdx.x] = pick<K>(dot, threadIdx.x);
    }
}

// <x, A x> of the patches folded into <= 1024 partial rows for the consumer (every workgroup takes a fixed subset): what
// k_patch_reduce<DOT> does at its end, alone - for the PCG, whose update launch sums the shared rows itself
template <int K>
__global__ void __launch_bounds__(256) k_patch_dot(int64_t npatch, const double *__restrict__ ppart, double *__restrict__ part,
                                                   const double *__restrict__ scal, int step) {
    if (scal && solve_done(scal, step)) return;
    double dot[K];
#pragma unroll
    for (int c = 0; c < K; ++c) dot[c] = 0.0;
    for (int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; p < npatch; p += int64_t(gridDim.x) * blockDim.x)
#pragma unroll
        for (int c = 0; c < K; ++c) dot[c] += ppart[p * K + c];
    __shared__ double smem[16 * K];
    block_sum<K>(dot, smem);
    if (threadIdx.x < K) part[blockIdx.x * K + threadIdx.x] = pick<K>(dot, threadIdx.x);
}

int g_patch_mode = 0;  // key 21: ablation mode of k_patch_apply (fp64, k = 5 only)
long long *g_patch_stamps = nullptr;   // mode 4: device buffer [grid][8] of phase time stamps (remo_debug_patch_phases)

}  // namespace

void set_patch_mode(int mode) { g_patch_mode = mode; }
void set_patch_stamps(long long *buf) { g_patch_stamps = buf; }

// remo_debug_tune key 26: the register-lean order of the kernel's arithmetic phase (k_patch_apply LEAN): -1 = in fp32 storage only
// (default), 0 = never, 1 = always.  fp32: 72-80 registers against 104 at first (111 against 124 us at 443 k tetrahedra); once the
// staging / output phases had their pass count as a compile-time constant the plain order needed 67 registers by itself and was
// level at 443 k tetrahedra (87 against 90 us) but not on the batches of the headline sweep (370 k / 320 k: 86 against 77 us, solve
// 158.3 against 153.1 ms per four batches): lean stays the fp32 default.  fp64: 96 registers with spills, five waves: 160 against 156 us.
int g_patch_lean = -1;
void set_patch_lean(int v) { g_patch_lean = v; }
// remo_debug_tune key 34: 1 = persistent workgroups with LDS-DMA prefetch of the next patch, 0 = one workgroup per patch (default).
// Measured at size L (370 k tetrahedra, k = 5, fp64; profiles/r04_h_*): the persistent form is parity-green and removes every wait on
// memory from the loop (the wait before barrier B3 costs 8 clock ticks) but takes 138 us per application against 112: its LDS (two
// row images + tables: 65 KB) leaves two workgroups per CU, a wave spends 5.9 k of its 18.7 k ticks per patch ISSUING the ~25 LDS-DMA
// instructions of the next patch (4-byte pieces - 40-byte rows do not tile into 16-byte ones - cost ~240 ticks each), and the tensor
// chains (7.5 k) are bound by the fp64 rate whenever the CU's two workgroups are in them together.  With two contexts on the GPU the
// two forms give the same points/s (115.9 against 115.3).  Kept for that record and as the starting point of a version with
// 16-byte-tiled rows; not the default.
int g_patch_persist = 0;
void set_patch_persist(int v) { g_patch_persist = (v == 1 || v == 2) ? v : 0; }
int g_patch_wgs_per_xcd = 0;   // key 35: workgroups per XCD of the persistent kernel (0 = as many as stay resident); tests make small meshes walk several patches per workgroup
void set_patch_wgs_per_xcd(int v) { g_patch_wgs_per_xcd = v > 0 ? v : 0; }
// EVERY row of a patch goes to the patch's contiguous block of the slab, also the rows no other patch touches: the kernel's output phase is one
// linear copy (whole lines; no slot table, one store per value instead of two of which the hardware drops one), and the PCG's update
// launch gathers every row the same way (no divergence between rows with and without slots).  Same number of rows written and read as
// with the shared rows only; measured at the headline size: application 112.0 -> 111.0 us, solve of a batch 108.0 -> 106.5 ms, three
// contexts 126.3 -> 128.6 points/s (profiles/r04_aa_*).  remo_debug_tune key 37 = 0 (probe builds): shared rows only.
int g_patch_all_slab = 1;
void set_patch_all_slab(int v) { g_patch_all_slab = v ? 1 : 0; }
#ifdef REMO_PROBES
int g_patch_stagger = 0;       // key 38: the workgroups of the first round wait (blockIdx / 256) x this many 64-clock units before they start (de-phasing the four workgroups of a CU)
void set_patch_stagger(int v) { g_patch_stagger = v > 0 ? v : 0; }
#endif
int g_patch_trim = 1;
void set_patch_trim(int v) { g_patch_trim = v; }
int g_patch_spread = 4;
void set_patch_spread(int v) { g_patch_spread = v; }
int g_patch_slab_rows = 0;  // remo_debug_tune key 23: 1 = boundary slab row-major
void set_patch_slab_rows(int v) { g_patch_slab_rows = v ? 1 : 0; }
int g_patch_block = 256;   // remo_debug_tune key 19: threads per workgroup of the patch kernel, 256 or 512 (the tables are laid out for it)
#ifdef REMO_PROBES
void set_patch_block(int b) { g_patch_block = (b == 512) ? 512 : 256; }
#else
void set_patch_block(int) {}
#endif
// one lane per (element, right-hand side); at most 204 elements (the table builder sorts 20 slots per element in 32 KB of LDS)
int patch_elements_per_group(int kmax) { const int e = g_patch_block / (kmax > 0 ? kmax : 1); return e < 204 ? e : 204; }   // 204 x 20 slots sort in 4096 LDS keys

size_t patch_arena_bytes(int64_t nt, int64_t n_max, int kmax) {
    const int E = patch_elements_per_group(kmax);
    const int64_t npatch = (nt + E - 1) / E;
    const int64_t cap = int64_t(E) * 20;
    return size_t(nt) * 40 + size_t(npatch) * (16 + 12 * size_t(cap)) + size_t(n_max + 2) * 8 + size_t(n_max + 1) * 16 + size_t(npatch) * 8 * 8 + (1 << 20);
}

// Everything is enqueued on s; flag_and_max (device, two ints) must be read by the caller after its next synchronisation:
// [0] != 0 -> a patch has more distinct rows than the tables hold (the caller falls back to another operator), [1] = the
// largest row count (sizes the kernel's LDS).
void build_patch_tables(Arena &ar, hipStream_t s, const DeviceSymbolic &sy, const double *C, int kmax, PatchTables &out, int32_t *flag_and_max) {
    out = PatchTables{};
    const int64_t nt = sy.nt, n = sy.nfree;
    const int E = patch_elements_per_group(kmax);
    // rows a patch may hold: what fits the LDS budget of one workgroup (48 KB of k-wide fp64 rows), at most every dof distinct
    int rows_cap = int((48 * 1024) / (size_t(kmax) * 8)) - 2;
    if (rows_cap > E * 20) rows_cap = E * 20;
    int npad = 256;
    while (npad < E * 20) npad <<= 1;
    out.nt = nt; out.n = n; out.E = E; out.rows_cap = rows_cap; out.block = g_patch_block; out.spread = g_patch_spread; out.trim = g_patch_trim;
    out.npatch = (nt + E - 1) / E;
    out.C = C;
    uint16_t *lidx = ar.lo<uint16_t>(size_t(nt) * 20 + 8);
    int32_t *pcount = ar.lo<int32_t>(size_t(out.npatch) + 1);
    int32_t *prow = ar.lo<int32_t>(size_t(out.npatch) * rows_cap + 1);
    int32_t *pout = ar.lo<int32_t>(size_t(out.npatch) * rows_cap + 1);
    int32_t *bptr = ar.lo<int32_t>(size_t(n) + 2);
    out.all_slab = (g_patch_all_slab && !g_patch_slab_rows) ? 1 : 0;
    out.nslot_cap = out.npatch * int64_t(rows_cap) < nt * 20 ? out.npatch * int64_t(rows_cap) : nt * 20;
    int32_t *bslot = ar.lo<int32_t>(size_t(out.nslot_cap) + 2);
    int32_t *row4 = out.all_slab ? ar.lo<int32_t>(size_t(n) * 4 + 4) : nullptr;
    if (row4) (void)hipMemsetAsync(row4, 0xFF, sizeof(int32_t) * size_t(n) * 4, s);
    const size_t mark = ar.hi_mark();
    int32_t *bcnt = ar.hi<int32_t>(size_t(n) + 2);
    int32_t *pboff = ar.lo<int32_t>(size_t(out.npatch) + 2);
    int32_t *pbcnt = ar.hi<int32_t>(size_t(out.npatch) + 2);
    (void)hipMemsetAsync(flag_and_max, 0, 3 * sizeof(int32_t), s);
    (void)hipMemsetAsync(prow, 0xFF, sizeof(int32_t) * (size_t(out.npatch) * rows_cap + 1), s);   // -1 behind a patch's last row
    (void)hipMemsetAsync(pbcnt + out.npatch, 0, sizeof(int32_t), s);
    hipLaunchKernelGGL(k_patch_row_slots, dim3(int((n + 1 + 255) / 256)), dim3(256), 0, s, n, E, sy.adjptr, sy.adj, bcnt, out.all_slab);
    hipLaunchKernelGGL(k_patch_build, dim3(int(out.npatch)), dim3(256), size_t(npad) * 8, s, nt, E, rows_cap, npad, sy.eldof, sy.adjptr, sy.adj,
                       lidx, pcount, pbcnt, prow, pout, flag_and_max, flag_and_max + 1, out.all_slab);
    size_t tb1 = 0, tb2 = 0;
    (void)rocprim::exclusive_scan(nullptr, tb1, bcnt, bptr, int32_t(0), size_t(n + 1), rocprim::plus<int32_t>(), s);
    (void)rocprim::exclusive_scan(nullptr, tb2, pbcnt, pboff, int32_t(0), size_t(out.npatch + 1), rocprim::plus<int32_t>(), s);
    void *tmp = ar.hi<char>((tb1 > tb2 ? tb1 : tb2) + 256);
    (void)rocprim::exclusive_scan(tmp, tb1, bcnt, bptr, int32_t(0), size_t(n + 1), rocprim::plus<int32_t>(), s);
    (void)rocprim::exclusive_scan(tmp, tb2, pbcnt, pboff, int32_t(0), size_t(out.npatch + 1), rocprim::plus<int32_t>(), s);
    hipLaunchKernelGGL(k_patch_zero_rows, dim3(1024), dim3(256), 0, s, nt * 20, lidx, (const int32_t *)(flag_and_max + 1));
    hipLaunchKernelGGL(k_patch_slots, dim3(int(out.npatch)), dim3(256), 0, s, E, rows_cap, (const int32_t *)pcount, (const int32_t *)pboff, (const int32_t *)prow, pout,
                       sy.adjptr, sy.adj, (const int32_t *)bptr, bslot, g_patch_slab_rows, row4);
    (void)hipMemcpyAsync(flag_and_max + 2, out.all_slab ? pboff + out.npatch : bptr + n, sizeof(int32_t), hipMemcpyDeviceToDevice, s);   // slab slots in use (all_slab: blocks padded to 16 rows)
    ar.hi_release(mark);     // the stream orders later users of this scratch behind these launches
    out.lidx = lidx; out.pcount = pcount; out.prow = prow; out.pout = pout; out.pboff = pboff; out.row4 = row4; out.bptr = bptr; out.bslot = bslot;
}

template <class T, int K> static void patch_dispatch(const CsrViewT<T> &A, const T *x, T *y, double *part, const double *scal, int step, int nb, hipStream_t s, bool defer) {
    const PatchOpT<T> &P = *A.patch;
#ifdef REMO_PROBES
    PatchTables tb = P.t;
    tb.stagger = g_patch_stagger;
#else
    const PatchTables &tb = P.t;
#endif
    const int64_t per = (tb.npatch + 7) / 8;
    double *pp = part ? P.ppart : nullptr;
    const dim3 grid(int(per * 8));
    double *pp2 = pp;
    double *bins = (part && defer && P.dot_bins) ? part + (step & 1) * (kPqBins * 8) : nullptr;
    auto launch = [&](auto kernel, int blk) {
        const size_t bytes = patch_lds_bytes(P.lds_rows, K, blk, tb.all_slab != 0);   // staged rows, later fp64 accumulators (+ the two padded row tables of the probe form)
        hipLaunchKernelGGL(kernel, grid, dim3(blk), bytes, s, tb, P.lds_rows, x, y, P.Yb, pp2, scal, step, g_patch_stamps, bins);
    };
    bool launched = false;
#ifdef REMO_PROBES
    if (g_patch_persist == 2 && tb.block == 256 && (g_patch_mode == 0 || g_patch_mode == 4) && P.lds_rows <= kPatchPasses * (256 / K) && P.lds_rows <= 1024) {
        // persistent form with the prefetch through registers: three workgroups per CU by its registers (LDS would allow four or five)
        const size_t bytes = patch_lds_bytes_r(P.lds_rows, K, tb.E);
        int wpc = int((160 * 1024) / (bytes + 16 * K * 8 + 1024));
        if (wpc > 3) wpc = 3;
        if (wpc >= 1 && bytes <= 64 * 1024 - 2048) {
            int64_t nwx = int64_t(32) * wpc;
            if (g_patch_wgs_per_xcd > 0 && g_patch_wgs_per_xcd < nwx) nwx = g_patch_wgs_per_xcd;
            if (nwx > per) nwx = per;
            if (nwx < 1) nwx = 1;
#ifdef REMO_PROBES
            if (g_patch_stamps && K == 5)
                hipLaunchKernelGGL((k_patch_apply_r<T, K, true>), dim3(int(nwx * 8)), dim3(256), bytes, s, tb, P.lds_rows, x, y, P.Yb, pp2, scal, step, bins, g_patch_stamps);
            else
#endif
            hipLaunchKernelGGL((k_patch_apply_r<T, K>), dim3(int(nwx * 8)), dim3(256), bytes, s, tb, P.lds_rows, x, y, P.Yb, pp2, scal, step, bins, (long long *)nullptr);
            launched = true;
        }
    }
    if (!launched && g_patch_persist == 1 && tb.block == 256 && (g_patch_mode == 0 || g_patch_mode == 4)) {
        // persistent form: as many workgroups as stay resident (LDS: two row images + tables per workgroup), 32 CUs per XCD
        const size_t bytes = patch_lds_bytes_p(P.lds_rows, K, tb.E);
        const size_t lds_cu = 160 * 1024, per_wg = bytes + 16 * K * 8 + 1024;
        int wpc = int(lds_cu / per_wg);
        if (wpc > 4) wpc = 4;
        if (wpc >= 1 && bytes <= 150 * 1024) {
            static bool attr_ok[2][9] = {};
            auto kernel = k_patch_apply_p<T, K>;
            bool &ok = attr_ok[sizeof(T) == 4 ? 1 : 0][K];
            if (!ok) ok = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(150 * 1024)) == hipSuccess;
            if (ok) {
                int64_t nwx = int64_t(32) * wpc;
                if (g_patch_wgs_per_xcd > 0 && g_patch_wgs_per_xcd < nwx) nwx = g_patch_wgs_per_xcd;
                if (nwx > per) nwx = per;
                if (nwx < 1) nwx = 1;
#ifdef REMO_PROBES
                if (g_patch_stamps && K == 5) {
                    auto ks = k_patch_apply_p<T, K, true>;
                    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, int(150 * 1024));
                    hipLaunchKernelGGL(ks, dim3(int(nwx * 8)), dim3(256), bytes, s, tb, P.lds_rows, x, y, P.Yb, pp2, scal, step, bins, g_patch_stamps);
                } else
#endif
                hipLaunchKernelGGL(kernel, dim3(int(nwx * 8)), dim3(256), bytes, s, tb, P.lds_rows, x, y, P.Yb, pp2, scal, step, bins, (long long *)nullptr);
                launched = true;
            }
        }
    }
#endif
#ifdef REMO_PROBES      // ablations (wrong results on purpose), the phase probe and 512-thread workgroups: tools/ builds only (make probes)
    if constexpr (K == 5) {     // ablations and the phase probe (tools/probe_patch.py)
        if (!launched && g_patch_mode >= 1 && g_patch_mode <= 3 + (g_patch_stamps ? 1 : 0)) {
            launched = true;
            if (tb.block == 512) {
                if (g_patch_mode == 1) launch(k_patch_apply<T, 5, 512, 1>, 512);
                else if (g_patch_mode == 2) launch(k_patch_apply<T, 5, 512, 2>, 512);
                else if (g_patch_mode == 3) launch(k_patch_apply<T, 5, 512, 3>, 512);
                else launch(k_patch_apply<T, 5, 512, 4>, 512);
            } else {
                if (g_patch_mode == 1) launch(k_patch_apply<T, 5, 256, 1>, 256);
                else if (g_patch_mode == 2) launch(k_patch_apply<T, 5, 256, 2>, 256);
                else if (g_patch_mode == 3) launch(k_patch_apply<T, 5, 256, 3>, 256);
                else launch(k_patch_apply<T, 5, 256, 4>, 256);
            }
        }
    }
    if (!launched && tb.block == 512) { launch(k_patch_apply<T, K, 512, 0>, 512); launched = true; }
    if (!launched && ((g_patch_lean == 1) != (sizeof(T) == 4)) && g_patch_lean >= 0) {     // the other order than the product's
        if (sizeof(T) == 4) launch(k_patch_apply<T, K, 256, 0>, 256); else launch(k_patch_apply<T, K, 256, 0, true>, 256);
        launched = true;
    }
#endif
    if (!launched) {     // the product: plain order in fp64 (124 registers, four waves per SIMD), register-lean order in fp32 storage (seven)
        if constexpr (sizeof(T) == 4) launch(k_patch_apply<T, K, 256, 0, true>, 256);
        else launch(k_patch_apply<T, K, 256, 0>, 256);
    }
    if (bins) return;     // the patches have added their sums into the update launch's rows themselves
    if (part && defer) {
        // (a few workgroups: the rows of `part` behind them stay zero - cleared by the solver once per solve)
        hipLaunchKernelGGL(k_patch_dot<K>, dim3(nb < kPatchDotBlocks ? nb : kPatchDotBlocks), dim3(256), 0, s, tb.npatch, (const double *)P.ppart, part, scal, step);
        return;
    }
    if (part) hipLaunchKernelGGL((k_patch_reduce<T, K, true>), dim3(nb), dim3(256), 0, s, A.n, tb.npatch, tb.bptr, tb.bslot, (const T *)P.Yb, y, (const double *)P.ppart, part, scal, step);
    else hipLaunchKernelGGL((k_patch_reduce<T, K, false>), dim3(nb), dim3(256), 0, s, A.n, tb.npatch, tb.bptr, tb.bslot, (const T *)P.Yb, y, (const double *)nullptr, part, scal, step);
}

template <class T> void launch_patch_spmm(const CsrViewT<T> &A, int k, const T *x, T *y, double *part, const double *scal, int nb, hipStream_t s, int step, bool defer) {
    switch (k) {
        case 1: patch_dispatch<T, 1>(A, x, y, part, scal, step, nb, s, defer); break;
        case 2: patch_dispatch<T, 2>(A, x, y, part, scal, step, nb, s, defer); break;
        case 3: patch_dispatch<T, 3>(A, x, y, part, scal, step, nb, s, defer); break;
        case 4: patch_dispatch<T, 4>(A, x, y, part, scal, step, nb, s, defer); break;
        case 5: patch_dispatch<T, 5>(A, x, y, part, scal, step, nb, s, defer); break;
        case 6: patch_dispatch<T, 6>(A, x, y, part, scal, step, nb, s, defer); break;
        case 7: patch_dispatch<T, 7>(A, x, y, part, scal, step, nb, s, defer); break;
        default: patch_dispatch<T, 8>(A, x, y, part, scal, step, nb, s, defer); break;
    }
}
template void launch_patch_spmm<double>(const CsrViewT<double> &, int, const double *, double *, double *, const double *, int, hipStream_t, int, bool);
template void launch_patch_spmm<float>(const CsrViewT<float> &, int, const float *, float *, double *, const double *, int, hipStream_t, int, bool);

}  // namespace remo
