// kutil.h — device helpers shared by the kernel files (kernels.hip, patch.hip): fixed-order block reductions, the early-exit
// test of queued PCG launches, and buffer accesses with the hardware range check.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"
#include "wave_util.h"

namespace remo {

// Sum K per-thread values over the block (blockDim.x multiple of 64, <= 1024).  Result valid in
// every thread.  Deterministic: fixed tree.
template <int K> __device__ __forceinline__ void block_sum(double (&v)[K], double *smem /* [16*K] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int c = 0; c < K; ++c) v[c] = wave_sum(v[c]);
    __syncthreads();
    if (lane == 0)
#pragma unroll
        for (int c = 0; c < K; ++c) smem[wave * K + c] = v[c];
    __syncthreads();
#pragma unroll
    for (int c = 0; c < K; ++c) {
        double s = 0.0;
        for (int w = 0; w < nw; ++w) s += smem[w * K + c];
        v[c] = s;
    }
}

// The same sums (same order: bit-identical), handed out one per thread: thread c < K returns the sum of v[c] over the block, the others 0.
// For a writer that wants "column threadIdx.x": choosing among the K registers by a run-time index (pick below) makes the compiler
// put the array into scratch memory when K >= 3 - 48 bytes stored per lane, 74 MB per application of the patch operator at size L.
template <int K> __device__ __forceinline__ double block_sum_column(double (&v)[K], double *smem /* [16*K] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int c = 0; c < K; ++c) v[c] = wave_sum(v[c]);
    __syncthreads();
    if (lane == 0)
#pragma unroll
        for (int c = 0; c < K; ++c) smem[wave * K + c] = v[c];
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x < K)
        for (int w = 0; w < nw; ++w) s += smem[w * K + threadIdx.x];
    return s;
}

// Sum of per-block partials part[nb][K] in a fixed order; result in every thread.
template <int K> __device__ __forceinline__ void reduce_partials(const double *part, int nb, double (&out)[K], double *smem) {
    double v[K];
#pragma unroll
    for (int c = 0; c < K; ++c) v[c] = 0.0;
    for (int b = threadIdx.x; b < nb; b += blockDim.x)
#pragma unroll
        for (int c = 0; c < K; ++c) v[c] += part[b * K + c];
    block_sum<K>(v, smem);
#pragma unroll
    for (int c = 0; c < K; ++c) out[c] = v[c];
}

// Up to three partial arrays reduced in ONE pass (one barrier pair instead of three): the loads of all
// arrays are in flight together.  Arrays with n = 0 are skipped.  Fixed order: deterministic.
template <int K>
__device__ __forceinline__ void reduce_partials3(const double *pa, int na, const double *pb, int nb, const double *pc, int nc,
                                                 double (&oa)[K], double (&ob)[K], double (&oc)[K], double *smem /* [16*3*K] */) {
    double v[3 * K];
#pragma unroll
    for (int c = 0; c < 3 * K; ++c) v[c] = 0.0;
    for (int b = threadIdx.x; b < na; b += blockDim.x)
#pragma unroll
        for (int c = 0; c < K; ++c) v[c] += pa[b * K + c];
    for (int b = threadIdx.x; b < nb; b += blockDim.x)
#pragma unroll
        for (int c = 0; c < K; ++c) v[K + c] += pb[b * K + c];
    for (int b = threadIdx.x; b < nc; b += blockDim.x)
#pragma unroll
        for (int c = 0; c < K; ++c) v[2 * K + c] += pc[b * K + c];
    block_sum<3 * K>(v, smem);
#pragma unroll
    for (int c = 0; c < K; ++c) { oa[c] = v[c]; ob[c] = v[K + c]; oc[c] = v[2 * K + c]; }
}

template <int K> __device__ __forceinline__ double pick(const double (&a)[K], int c) {
    double r = a[0];
#pragma unroll
    for (int j = 1; j < K; ++j) r = (c == j) ? a[j] : r;
    return r;
}

// scal[kDoneSlot] (as int) is set to s + 1 by the update launch of step s once every column is frozen; the launches of
// LATER steps that the host has already queued (it runs a few steps ahead of the device) then return at once.  The
// launches of step s itself (the update that raises the flag included) never act on it: a workgroup whose waves start
// on both sides of the store would otherwise split, the early leavers missing from the block sums of the rest.  They
// run a harmless step instead (alpha = beta = 0 for every column).
__device__ __forceinline__ bool solve_done(const double *scal, int step) {
    const int d = reinterpret_cast<const int *>(scal + kDoneSlot)[0];
    return d != 0 && d <= step;
}

// A load the compiler sends down the VECTOR-memory path although its address is the same for every lane (the address passes through
// vector registers it cannot see into).  For a word that would otherwise make a wave's scalar loads (which return out of order: any use
// waits for all of them) queue behind a slow one.
template <class V> __device__ __forceinline__ V load_as_vector(const V *p) {
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    uint32_t lo = uint32_t(a), hi = uint32_t(a >> 32);
    asm volatile("" : "+v"(lo), "+v"(hi));
    return *reinterpret_cast<const V __attribute__((address_space(1))) *>((uint64_t(hi) << 32) | lo);   // (global, not flat: a flat load counts as an LDS access too)
}

// ---- buffer accesses with the hardware range check: a lane with nothing to load / store hands the instruction an offset
// beyond the descriptor's range - the load returns 0, the store is dropped, and neither sends a request down the
// vector-memory path.  No branch around the access, so the compiler keeps all of them in flight together.
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr uint32_t kOutOfRange = 0xFFFFF000u;   // beyond any descriptor the launcher accepts
__device__ __forceinline__ rsrc_t make_rsrc(const void *p, uint64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, int(uint32_t(bytes)), 0x00020000);
}
template <class T, int N> __device__ __forceinline__ void buf_load(rsrc_t r, uint32_t off, T (&out)[N]) {   // N values from byte offset off (multiple of 4)
    constexpr int W = N * int(sizeof(T)) / 4;
    unsigned int w[W];
    int d = 0;
#pragma unroll
    for (; d + 4 <= W; d += 4) {
        const u32x4_t q = __builtin_amdgcn_raw_buffer_load_b128(r, off + 4u * d, 0, 0);
        w[d] = q.x; w[d + 1] = q.y; w[d + 2] = q.z; w[d + 3] = q.w;
    }
    if constexpr ((W & 3) >= 2) {
        const u32x2_t q = __builtin_amdgcn_raw_buffer_load_b64(r, off + 4u * (W & ~3), 0, 0);
        w[W & ~3] = q.x; w[(W & ~3) + 1] = q.y;
    }
    if constexpr (W & 1) w[W - 1] = __builtin_amdgcn_raw_buffer_load_b32(r, off + 4u * (W - 1), 0, 0);
    __builtin_memcpy(out, w, sizeof(T) * N);
}
template <class T, int N> __device__ __forceinline__ void buf_store(rsrc_t r, uint32_t off, const T (&in)[N]) {
    constexpr int W = N * int(sizeof(T)) / 4;
    unsigned int w[W];
    __builtin_memcpy(w, in, sizeof(T) * N);
    int d = 0;
#pragma unroll
    for (; d + 4 <= W; d += 4) {
        u32x4_t q; q.x = w[d]; q.y = w[d + 1]; q.z = w[d + 2]; q.w = w[d + 3];
        __builtin_amdgcn_raw_buffer_store_b128(q, r, off + 4u * d, 0, 0);
    }
    if constexpr ((W & 3) >= 2) {
        u32x2_t q; q.x = w[W & ~3]; q.y = w[(W & ~3) + 1];
        __builtin_amdgcn_raw_buffer_store_b64(q, r, off + 4u * (W & ~3), 0, 0);
    }
    if constexpr (W & 1) __builtin_amdgcn_raw_buffer_store_b32(w[W - 1], r, off + 4u * (W - 1), 0, 0);
}

}  // namespace remo
