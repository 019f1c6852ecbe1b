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

// One wave64 quantises one 256-value block to Q8_K exactly as ggml-cpu's quantize_row_q8_K_ref does (first-index
// arg-max of |x| defines the sign of the scale, iscale = -127/max, round-half-even, clamp to 127, 16-value bsums).
// v[0..3] are the lane's four consecutive values; y may point to LDS or global memory.
template <typename Q8K> __device__ __forceinline__ void wave_quantize_q8_K(const float (&v)[4], const int lane, Q8K * y) {
    float amax = 0.0f, mx = 0.0f;
    int idx = lane * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float ax = fabsf(v[k]);
        if (ax > amax) { amax = ax; mx = v[k]; idx = lane * 4 + k; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float oa = __shfl_xor(amax, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        const float om = __shfl_xor(mx, o, 64);
        const bool take = oa > amax || (oa == amax && oi < idx);
        if (take) { amax = oa; idx = oi; mx = om; }
    }
    int q[4] = {0, 0, 0, 0};
    float iscale = 0.0f;
    if (amax != 0.0f) {
        iscale = -127.f / mx;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = (int) rintf(iscale * v[k]);  // round-half-even == ggml's nearest_int()
            q[k] = r < 127 ? r : 127;
        }
    }
    const uint32_t packed = (uint32_t) (q[0] & 0xFF) | ((uint32_t) (q[1] & 0xFF) << 8) | ((uint32_t) (q[2] & 0xFF) << 16) | ((uint32_t) (q[3] & 0xFF) << 24);
    ((uint32_t *) y->qs)[lane] = packed;
    int s = q[0] + q[1] + q[2] + q[3];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if ((lane & 3) == 0) y->bsums[lane >> 2] = (int16_t) s;
    if (lane == 0) {
        y->d = amax != 0.0f ? 1.0f / iscale : 0.0f;
        y->pad[0] = y->pad[1] = y->pad[2] = 0.0f;
    }
}

// rotary-embedding angle for pair `ip` of a token at position pos_f: the same chain of f32 multiplies as ggml-cpu's
// rope cache loop (theta *= theta_scale per pair), then the YaRN mix and the accurate cosf/sinf
struct rope_consts {
    float theta_scale, freq_scale, ext_factor, attn_factor, corr0, corr1;
};
__device__ __forceinline__ void rope_cos_sin(const int ip, const float pos_f, const float * __restrict__ ff, const rope_consts rc, float & cs, float & sn) {
    float theta = pos_f;
    for (int k = 0; k < ip; ++k) theta *= rc.theta_scale;
    const float fq = ff ? ff[ip] : 1.0f;
    const float theta_extrap = theta / fq;
    const float theta_interp = rc.freq_scale * theta_extrap;
    float th = theta_interp, mscale = rc.attn_factor;
    if (rc.ext_factor != 0.0f) {
        const float y = ((float) ip - rc.corr0) / fmaxf(0.001f, rc.corr1 - rc.corr0);
        const float ramp_mix = (1.0f - fminf(1.0f, fmaxf(0.0f, y))) * rc.ext_factor;
        th = theta_interp * (1.0f - ramp_mix) + theta_extrap * ramp_mix;
        mscale *= 1.0f + 0.1f * logf(1.0f / rc.freq_scale);
    }
    cs = cosf(th) * mscale;
    sn = sinf(th) * mscale;
}

}  // namespace mi355x
