// fill_probe.hip — how fast can ONE CU pull L2-resident bytes on chip, by path?  (decides the operand paths of the shadow-plane GEMM)
//   A  global_load_dwordx4 -> VGPR (consumed by an empty asm)
//   B  global_load_lds_dwordx4 (LDS-DMA)
//   C  global_load_dwordx4 -> VGPR -> ds_write_b128
// 256 workgroups x 512 threads (one per CU, 8 waves), every wave streams 1 KB pieces from a 4 MB window (L2-resident after the first touch),
// 8 pieces in flight per wave.  Prints bytes / clock / CU at the measured wall time and 2.1 GHz.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/fill_probe.hip -o scripts/ubench/fill_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void dma16(const void * g, const uint32_t lds) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}

template <int MODE>
__global__ void __launch_bounds__(512) k_fill(const char * __restrict__ src, const size_t window, const int iters, int * sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lds0 = (uint32_t) (uintptr_t) smem + wave * 8192;
    size_t off = ((size_t) blockIdx.x * 8 + wave) * 65536 % window;
    int acc = 0;
    for (int it = 0; it < iters; ++it) {
        const char * p = src + off + lane * 16;
        if constexpr (MODE == 0) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *(const uint4 *) (p + u * 1024);
#pragma unroll
            for (int u = 0; u < 8; ++u) asm volatile("" ::"v"(v[u].x), "v"(v[u].y), "v"(v[u].z), "v"(v[u].w));
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u) dma16(p + u * 1024, lds0 + u * 1024);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *(const uint4 *) (p + u * 1024);
#pragma unroll
            for (int u = 0; u < 8; ++u) *(uint4 *) (smem + wave * 8192 + u * 1024 + lane * 16) = v[u];
        }
        off += 8192;
        if (off + 8192 > window) off = 0;
    }
    if (MODE != 0) acc = *(const int *) (smem + threadIdx.x * 4);
    if (acc == 0x7fffffff) *sink = acc;
}

int main() {
    const size_t window = (size_t) 4 << 20;
    char * src; int * sink;
    CK(hipMalloc(&src, window + 65536)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(src, 1, window + 65536));
    const int iters = 2000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k_fill<0>, dim3(256), dim3(512), 65536, 0, src, window, iters, sink);
            else if (mode == 1) hipLaunchKernelGGL(k_fill<1>, dim3(256), dim3(512), 65536, 0, src, window, iters, sink);
            else hipLaunchKernelGGL(k_fill<2>, dim3(256), dim3(512), 65536, 0, src, window, iters, sink);
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double bytes = 256.0 * 8 * iters * 8192.0;
            if (rep) printf("%s: %.1f us, %.2f TB/s aggregate, %.1f GB/s per CU, %.1f B/clk/CU at 2.1 GHz\n", mode == 0 ? "A  global_load_dwordx4 -> VGPR      " : (mode == 1 ? "B  global_load_lds_dwordx4 (LDS-DMA)" : "C  global_load -> VGPR -> ds_write  "),
                            ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / 256, bytes / (ms * 1e-3) / 256 / 2.1e9);
        }
    }
    return 0;
}
