/*
 * remo3d_hip_debug.h — probes and tuning knobs of libremo3d_hip.so for tests/, tools/ and bench.py's `box` record.
 * NOT part of the drop-in boundary (include/remo3d_hip.h is: the surface SURVEY.md section 8b describes, replacing the seam
 * of remo3d/workers/worker.py:32-35, 110); nothing here is needed to run a batch, and the knobs are process-global.
 */
#ifndef REMO3D_HIP_DEBUG_H
#define REMO3D_HIP_DEBUG_H

#include "remo3d_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* What this GPU streams (bench.py `box`): a read of `bytes` through a plain 16-byte-per-lane summing kernel and a device-to-device
 * copy of them, HIP events, best of six; GB/s (the copy counts read + write).  With bytes <= 192 MiB the read figure is that of four
 * back-to-back re-reads of the buffer, i.e. of the 256 MB Infinity Cache rather than of HBM. */
int remo_debug_stream(remo_ctx_t *ctx, int64_t bytes, double *read_gbs, double *copy_gbs);
/* Rate of a chain of DEPENDENT fp32 multiply-adds of one wave (1024 waves over the chip at once), in 1e9 per second: follows the
 * shader clock under load - the part of the box-to-box spread that the stream figures do not show. */
int remo_debug_clock(remo_ctx_t *ctx, double *gfma_per_wave);
/* Scattered 16-byte reads from a buffer of `bytes` (a power of two) by 4096 workgroups, useful GB/s: 2 MiB stays in every XCD's L2 (the
 * path of the SpMM's x gather); 256 MiB adds the address translation of pages scattered over the memory. */
int remo_debug_cache_gather(remo_ctx_t *ctx, int64_t bytes, double *gbs);
/* hipDeviceProp_t of the context's GPU: compute units, clock kHz, memory clock kHz, bus width, L2 bytes, memory MiB, LDS bytes per CU, revision. */
int remo_debug_device(remo_ctx_t *ctx, int64_t *out8);
/* XCD (hardware register XCC_ID) of workgroups 0 .. nblocks-1 of a probe launch: the SpMM's row schedule assumes b mod 8. */
int remo_debug_xcc(remo_ctx_t *ctx, int32_t *out, int32_t nblocks);


/* Where a workgroup of the patch operator (remo_opts_t.op = 3, k = 5, fp64) spends its time: clock ticks between the phase
 * boundaries of k_patch_apply averaged over the workgroups of one launch - out16[0..6] = row tables into LDS | staging of x | x into registers |
 * zeroing | tensor arithmetic + LDS accumulation | output | partial sums; [7] whole workgroup; [8] first start to last end of the
 * launch; [9] workgroups.  The last run on the batch must have used the patch operator. */
int remo_debug_patch_phases(remo_ctx_t *ctx, remo_batch_t *batch, int32_t fp32 /* the fp32 instantiation instead */, double *out16);

/* Cost of a grid-wide barrier between the resident workgroups of one launch (nblocks <= 2048 workgroups of 256 threads; nbar
 * iterations of: agent-scope store, barrier, agent-scope load of another workgroup's store, barrier): out3[0] = microseconds per
 * barrier, [1] = 1 if a wait gave up, [2] = loads that did not see the store. */
int remo_debug_grid_barrier(remo_ctx_t *ctx, int32_t nblocks, int32_t nbar, double *out3);

/* Kernel tuning knob for the probe scripts (tools/probe_spmm.py): key 0 SpMM variant (1 lane per stored
 * entry, 3 edge row pairs = default), 1 lanes per row, 2 threads per workgroup, 3 row schedule of
 * the pair kernel (0 grid-stride, 1 XCD windows, 16 * nc XCD regions of nc chunks; default by size), 4 grid size;
 * value 0 (mapping: -1) restores the default; 5 ablation mode of the pair kernel; 6: 0 = one launch per Chebyshev
 * step, 1 = paired steps on the squared vertex block in 2D (default), 2 = paired steps also in 3D; 7 lanes per row of the
 * paired kernel; 8: 0 = CSR pattern by the global sort instead of row by row; 9: 0 = first
 * Chebyshev step as a launch of its own instead of inside the update launch; 13: 0 = Chebyshev launches read the vertex block inside A, 1 = from a compact copy above 16 k vertices (default), 2 = always; 15: 0 = the Chebyshev chain of an fp64 solve stays in fp64 also above 32 k vertex rows (default there: fp32 storage, the preconditioner may be inexact); 16: 1 = never the multigrid cycle on the vertex block, 2 = always, any dimension (0: remo_opts_t.coarse decides); 17: 0 = the multigrid cycle of an fp64 solve stays in fp64 (default: fp32 storage);
 * round 3: 18: 0 = elements taken in the caller's order (default 1: sorted by their two smallest vertices); 19: threads per workgroup of
 * the patch kernel, 256 (default) or 512; 20: 0 = op 0 chooses the operator by stored entries as in round 2 (default 1: the patch
 * operator wherever its tables fit); 21: ablation mode of the patch kernel (1 no LDS atomics, 2 no arithmetic, 3 no output: wrong
 * results on purpose); 22: 0 = shared rows summed by k_patch_reduce instead of the update launch; 23: 1 = boundary slab row-major
 * (measured slower); 24: 0 = Chebyshev launches walk the vertex block as CSR instead of its fixed-width image; 26: register-lean order
 * of the patch kernel's arithmetic phase: -1 in fp32 storage only (default), 0 never, 1 always; 27: 0 = slab slots of a shared row
 * fetched one by one in the update launch (default 1: four in flight); 28: 0 = the patches leave a row of <p, A p> each and a launch
 * folds them (default 1: atomic adds into the update launch's rows); 32: number of runs of the patch's list from which the lanes of a wave of the
 * patch kernel take their elements (default 4; 0 / 1: consecutive elements); 33: 0 = every workgroup of the patch kernel walks
 * the largest patch's row count in its staging and output phases (default 1: its own patch's).  Process-global. */
void remo_debug_tune(int32_t key, int32_t value);

#ifdef __cplusplus
}
#endif
#endif /* REMO3D_HIP_DEBUG_H */
