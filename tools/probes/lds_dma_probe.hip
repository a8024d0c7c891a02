// lds_dma_probe.hip - does `global_load_lds_dword` reach LDS addresses beyond 64 KB on this chip (160 KB of LDS per CU)?
// One workgroup asks for `bytes` of dynamic LDS, every wave DMA-copies a pattern to the LAST kilobyte of it and to the first,
// reads both back with ds_read and reports mismatches.   hipcc --offload-arch=gfx950 -O2 lds_dma_probe.hip -o lds_dma_probe && ./lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef const void __attribute__((address_space(1))) *gsrc_t;
typedef void __attribute__((address_space(3))) *ldst_t;
__global__ void __launch_bounds__(256) k(const unsigned *src, unsigned *out, int words_total) {
    extern __shared__ unsigned lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int j = tid; j < words_total; j += 256) lds[j] = 0xDEADBEEFu;
    __syncthreads();
    const int hi = words_total - 256;                       // last kilobyte
    __builtin_amdgcn_global_load_lds((gsrc_t)(src + wave * 64 + lane), (ldst_t)(lds + wave * 64), 4, 0, 0);
    __builtin_amdgcn_global_load_lds((gsrc_t)(src + 256 + wave * 64 + lane), (ldst_t)(lds + hi + wave * 64), 4, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[tid] = lds[tid];
    out[256 + tid] = lds[hi + tid];
    // anything landed somewhere else?
    unsigned stray = 0;
    for (int j = 256 + tid; j < hi; j += 256) stray += lds[j] != 0xDEADBEEFu;
    atomicAdd(out + 512, stray);
}
int main() {
    unsigned *src, *out;
    hipMalloc(&src, 512 * 4); hipMalloc(&out, 513 * 4);
    std::vector<unsigned> h(512);
    for (int i = 0; i < 512; ++i) h[i] = 0x1000u + i;
    hipMemcpy(src, h.data(), 512 * 4, hipMemcpyHostToDevice);
    for (int kb : {32, 60, 64, 66, 80, 120, 150}) {
        const int bytes = kb * 1024;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        hipMemset(out, 0, 513 * 4);
        hipLaunchKernelGGL(k, dim3(1), dim3(256), bytes, 0, src, out, bytes / 4);
        hipError_t e2 = hipDeviceSynchronize();
        std::vector<unsigned> r(513);
        hipMemcpy(r.data(), out, 513 * 4, hipMemcpyDeviceToHost);
        int bad_lo = 0, bad_hi = 0;
        for (int i = 0; i < 256; ++i) { bad_lo += r[i] != h[i]; bad_hi += r[256 + i] != h[256 + i]; }
        printf("dynamic LDS %3d KB: attribute %s, launch %s, mismatches low %d high %d, stray words %u\n", kb, hipGetErrorString(e), hipGetErrorString(e2), bad_lo, bad_hi, r[512]);
    }
    return 0;
}
