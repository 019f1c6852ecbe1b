// dev_util.h — device-side helpers for the gfx950 kernels (wave64 everywhere; no CUDA idioms).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace mi355x {

#define MI_WAVE 64

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// IEEE binary16 <-> f32, round-to-nearest-even (v_cvt_f16_f32 / v_cvt_f32_f16), bit-identical to the CPU path
__device__ __forceinline__ float h2f(uint16_t h) {
    _Float16 x;
    __builtin_memcpy(&x, &h, 2);
    return (float) x;
}
__device__ __forceinline__ uint16_t f2h(float f) {
    _Float16 x = (_Float16) f;
    uint16_t h;
    __builtin_memcpy(&h, &x, 2);
    return h;
}

// 4 x int8 dot-accumulate (v_dot4_i32_i8 on CDNA)
__device__ __forceinline__ int dot4(int a, int b, int c) { return __builtin_amdgcn_sdot4(a, b, c, false); }

// loads with the TRUE (2-byte) alignment of Q6_K / Q8_0 block fields: the compiler may not assume more
typedef uint32_t __attribute__((aligned(2))) u32_a2;
typedef uint16_t __attribute__((aligned(2))) u16_a2;
__device__ __forceinline__ uint32_t ld32_a2(const void * p) { return *(const u32_a2 *) p; }
__device__ __forceinline__ uint16_t ld16(const void * p) { return *(const u16_a2 *) p; }

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }

}  // namespace mi355x
