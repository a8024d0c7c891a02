// wave_util.h — cross-lane reductions shared by the kernel files (wave = 64 lanes, gfx950).
#pragma once
#include <hip/hip_runtime.h>

namespace remo {

// ------------------------------------------------------------------------------------------
// wave / block reductions (wave = 64 lanes)

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int W, class T> __device__ __forceinline__ T group_sum(T v) {
#pragma unroll
    for (int off = W / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ---- transposing reduction ---------------------------------------------------------------
// A group of W lanes holds M partial values per lane and needs the M group sums.  Summing each value
// with its own butterfly costs M log2(W) lane exchanges, and on gfx9 a __shfl is a ds_bpermute through
// the LDS pipe that all four SIMDs share: for the SpMM (M = 10 values per edge-row pair) that pipe,
// not HBM, set the kernel time.  Here every exchange step HALVES the value list instead: the two
// halves of the group keep one value of each pair and send the other, so the whole reduction costs
// M - 1 + (odd leftovers) exchanges, all of them DPP moves inside a 16-lane row (VALU, no LDS), and
// ends with the sums spread over the lanes, one (or ceil(M / W)) per lane, which also turns the M
// serial stores of lane 0 into one coalesced store.
template <int CTRL> __device__ __forceinline__ double dpp_mov(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, false));
}
// value of the partner lane in the other half of the W-lane group (a bijection between the halves)
template <int W, class T> __device__ __forceinline__ T partner(T v) {
    if constexpr (W == 16) return dpp_mov<0x140>(v);      // row_mirror: i <-> 15 - i
    else if constexpr (W == 8) return dpp_mov<0x141>(v);  // row_half_mirror: i <-> 7 - i
    else if constexpr (W == 4) return dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    else if constexpr (W == 2) return dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    else return __shfl_xor(v, W / 2, 64);
}
constexpr int treduce_out(int m, int w) { return w < 2 ? m : treduce_out((m + 1) / 2, w / 2); }
// In place: on return v[0 .. treduce_out(M, W)) are complete group sums; which ones, per lane, is
// given by TOwner below (same recursion on indices).  Fixed order: deterministic.
template <int M, int W> struct TReduce {
    template <class T> static __device__ __forceinline__ void run(T *v, int sub) {
        if constexpr (W >= 2) {
            const bool hi = (sub & (W / 2)) != 0;
            constexpr int P = M / 2;
#pragma unroll
            for (int j = 0; j < P; ++j) {
                const T a = v[2 * j], b = v[2 * j + 1];
                v[j] = (hi ? b : a) + partner<W, T>(hi ? a : b);
            }
            if constexpr (M & 1) {
                const T l = v[M - 1];
                v[P] = l + partner<W, T>(l);
            }
            TReduce<(M + 1) / 2, W / 2>::run(v, sub);
        }
    }
};
// idx[f] = index (0 .. M-1) of the original value whose sum lane `sub` holds in v[f] after TReduce;
// an odd leftover is carried by both halves, and only the lane that took the low half every time it
// was carried is its owner (own[f]), so that each sum is stored exactly once.
template <int M, int W> struct TOwner {
    static __device__ __forceinline__ void run(int *idx, int *own, int sub) {
        if constexpr (W >= 2) {
            const int hi = (sub & (W / 2)) != 0;   // arithmetic blends: keeps the arrays in registers
            constexpr int P = M / 2;
#pragma unroll
            for (int j = 0; j < P; ++j) {
                idx[j] = idx[2 * j] + hi * (idx[2 * j + 1] - idx[2 * j]);
                own[j] = own[2 * j] + hi * (own[2 * j + 1] - own[2 * j]);
            }
            if constexpr (M & 1) {
                idx[P] = idx[M - 1];
                own[P] = own[M - 1] * (1 - hi);
            }
            TOwner<(M + 1) / 2, W / 2>::run(idx, own, sub);
        }
    }
};
template <int M, int W> __device__ __forceinline__ void towner_init(int (&idx)[M], int (&own)[M], int sub) {
#pragma unroll
    for (int j = 0; j < M; ++j) { idx[j] = j; own[j] = 1; }
    TOwner<M, W>::run(idx, own, sub);
}

}  // namespace remo
