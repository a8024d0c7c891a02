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

/* The same for the PERSISTENT form of the kernel (k_patch_apply_p): clock ticks per patch of wave 0, summed by the workgroups over their
 * patches and averaged - out16[0..8] = barrier B0 | issue of the next patch's LDS-DMA | x values into registers | barrier B1 | clearing + B2 |
 * tensor chains + LDS accumulation | wait for the DMA and older stores | barrier B3 | rows out; [10] patches, [11] workgroups, [12] ticks of the
 * busiest workgroup (sum of its phases), [13] microseconds per application without the stamps.  -DREMO_PROBES builds only. */
int remo_debug_patch_phases_p(remo_ctx_t *ctx, remo_batch_t *batch, int32_t fp32, double *out16);

/* Cost of a grid-wide barrier between the resident workgroups of one launch (nblocks <= 2048 workgroups of 256 threads; nbar
 * iterations of: agent-scope store, barrier, agent-scope load of another workgroup's store, barrier): out3[0] = microseconds per
 * barrier, [1] = 1 if a wait gave up, [2] = loads that did not see the store. */
int remo_debug_grid_barrier(remo_ctx_t *ctx, int32_t nblocks, int32_t nbar, double *out3);

/* Process-global knobs, two kinds.  Returns 0, or -1 for a key this build does not have.
 *
 * (a) ALWAYS THERE - keys that force one of the product's own paths, i.e. a choice the library makes by size or dimension, so that
 * a small test mesh reaches the code a large batch runs.  Every setting gives the same operator / preconditioner to rounding:
 *    3  row schedule of the CSR product (0 grid-stride, 1 XCD windows, 16 * nc XCD regions of nc chunks; -1 = by size);
 *    6  0 = one launch per Chebyshev step, 1 = paired steps on the squared vertex block in 2D (default), 2 = also in 3D;
 *    9  0 = first Chebyshev step as a launch of its own instead of inside the update launch;
 *   13  0 = Chebyshev launches read the vertex block inside A, 1 = from a compact copy above 16 k vertices (default), 2 = always;
 *   15  0 = the Chebyshev chain of an fp64 solve stays in fp64 also above 32 k vertex rows (default there: fp32 storage), 2 = fp32 always;
 *   16  1 = never the multigrid cycle on the vertex block, 2 = always, any dimension (0: remo_opts_t.coarse decides);
 *   17  0 = the multigrid cycle of an fp64 solve stays in fp64 (default: fp32 storage);
 *   18  0 = elements taken in the caller's order (default 1: sorted by their two smallest vertices);
 *   22  0 = rows shared by patches summed by k_patch_reduce instead of the PCG's update launch;
 *   24  0 = Chebyshev launches walk the vertex block as CSR instead of its fixed-width image;
 *   25  0 = x += alpha p formed by the update launch instead of the direction launch of the step (bit-identical x);
 *   29  0 = the update launch fetches four slab slots for every row and weights the ones the row does not have by zero;
 *   30  0 = the direction launch takes a k-wide row per lane instead of walking its vectors as flat arrays, 16 bytes per lane;
 *   31  0 = the update launch takes a k-wide row per lane instead of 64 rows per wave with a value per lane and pass (fp64 storage).
 *
 * (b) ONLY IN A LIBRARY BUILT WITH -DREMO_PROBES (`make -C remo3d_amd/csrc probes` -> libremo3d_hip_probes.so, loaded by the tools
 * through REMO_LIB=...): rejected experiments and ablations, some of which give WRONG RESULTS ON PURPOSE.  The product ignores them:
 *    0 SpMM variant, 1 lanes per row, 2 threads per workgroup, 4 grid size, 5 ablation mode of the pair kernel; 7 lanes per row of
 *    the paired Chebyshev kernel; 8: 0 = CSR pattern by the global sort; 19 threads per workgroup of the patch kernel (512); 21 ablation
 *    mode of the patch kernel (1 no LDS atomics, 2 no arithmetic, 3 no output); 23: 1 = boundary slab row-major; 26 register-lean order
 *    of the patch kernel (0 never, 1 always; product: fp32 storage only); 27: 0 = slab slots of a shared row fetched one by one;
 *    28: 0 = a row of <p, A p> per patch + a folding launch; 32 runs of the patch's list per wave (product: 4); 33: 0 = every
 *    workgroup walks the largest patch's row count; 34: 1 / 2 = the patch operator as persistent workgroups that prefetch the next patch by
 *    LDS-DMA / through registers (round 4: parity-green, 138 / 111.5 us against 112 at size L: DESIGN.md section 8), 35 their number per XCD
 *    (0 = as many as stay resident); 36 extra operator applications per PCG step (results discarded: what more applications would cost);
 *    37: 0 = only the rows shared by several patches go through the slab, the others straight to y (the form before round 4's last build).
 *    38: the workgroups of the patch kernel's first round start (slot on the CU) x this many 64-clock units apart (de-phasing probe: no gain).
 *    remo_debug_patch_phases[_p] and remo_debug_grid_barrier also need that build. */
int remo_debug_tune(int32_t key, int32_t value);

#ifdef __cplusplus
}
#endif
#endif /* REMO3D_HIP_DEBUG_H */
