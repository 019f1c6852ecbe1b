// valu_probe.hip — issue cost (clocks per wave-instruction, one wave per SIMD and two) of the instruction kinds the skinny
// matrix-core kernel's unit is made of: v_mad_i32_i24 (the sub-block fold), bit-field unpacking, int->float + fma (the super-block
// fold), the int8 MFMA alone and an MFMA followed by the 16 dependent folds — to see what a unit of 8 MFMAs + ~270 VALU can cost.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef int int4v __attribute__((ext_vector_type(4)));
typedef int int16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned long long clk() { return __builtin_readcyclecounter(); }

template <int KIND> __global__ void __launch_bounds__(512) k_probe(unsigned long long * out, int * sink, const int iters, const int seed) {
    int a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 7 + i + seed;
    int4v fa = {seed, seed + 1, seed + 2, seed + 3}, fb = {seed + 4, seed + 5, seed + 6, seed + 7};
    int16v acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    float f[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) f[i] = (float) a[i];
    const int sc = seed & 63;
    __syncthreads();
    const unsigned long long t0 = clk();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {  // 16 independent v_mad_i32_i24
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = __mul24(a[i], sc) + a[(i + 1) & 15];
        } else if (KIND == 1) {  // unpack: and + shift-and
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                a[i] = (a[i + 1] & 0x0F0F0F0F) + it;
                a[i + 1] = ((a[i] >> 4) & 0x0F0F0F0F) ^ it;
            }
        } else if (KIND == 2) {  // cvt + fma
#pragma unroll
            for (int i = 0; i < 16; ++i) f[i] = __builtin_fmaf((float) a[i], 1.0001f, f[i]);
        } else if (KIND == 3) {  // independent MFMAs (zero C), results summed rarely
            const int16v t = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, acc, 0, 0, 0);
            acc = t;
        } else if (KIND == 4) {  // MFMA with zero C, then 16 dependent mads
            const int16v z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            fa[0] += it;
            const int16v t = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, z, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] += __mul24(t[i], sc);
        } else if (KIND == 5) {  // two MFMAs, then their 32 mads (the kernel's pair step)
            const int16v z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            fa[0] += it;
            fb[1] ^= it;
            const int16v t = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, z, 0, 0, 0);
            const int16v u = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb, fa, z, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] += __mul24(t[i], sc);
#pragma unroll
            for (int i = 0; i < 16; ++i) f[i] += (float) __mul24(u[i], sc);
        }
    }
    const unsigned long long t1 = clk();
    int s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i] + acc[i] + (int) f[i];
    if (s == 0x7fffffff) sink[threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int KIND> static void run(const char * name, int per_iter, unsigned long long * out, int * sink) {
    for (int threads : {256, 512}) {
        k_probe<KIND><<<256, threads>>>(out, sink, 2000, 3);
        CK(hipDeviceSynchronize());
        unsigned long long c;
        CK(hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost));
        // s_memtime counts at 100 MHz: convert with the shader clock (2.4 GHz assumed)
        printf("%-44s %d waves/SIMD: %8.2f counter ticks per iteration (%d instr) = %6.3f per instruction\n", name, threads / 256, c / 2000.0, per_iter, c / 2000.0 / per_iter);
    }
}

int main() {
    unsigned long long * out;
    int * sink;
    CK(hipMalloc(&out, 64));
    CK(hipMalloc(&sink, 4096));
    run<0>("16 x v_mad_i32_i24 (independent)", 16, out, sink);
    run<1>("16 x and / shift+and", 16 + 8, out, sink);
    run<2>("16 x (v_cvt_f32_i32 + v_fma)", 32, out, sink);
    run<3>("1 x v_mfma_i32_32x32x32_i8 (chained C)", 1, out, sink);
    run<4>("1 MFMA + 16 dependent mads", 17, out, sink);
    run<5>("2 MFMA + 32 dependent mads (+16 cvt)", 50, out, sink);
    return 0;
}
