// dev_util.h — device-side helpers for the gfx950 kernels (wave64 everywhere; no CUDA idioms).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace mi355x {

#define MI_WAVE 64

// ---- wave64 reductions on the DPP data path (no LDS crossbar): the profile showed every __shfl_xor lowered to
// ds_bpermute_b32 (~100+ cycles each, 6 in a dependent chain per row); DPP row rotates cost a few cycles.
// Pattern: all-reduce inside each row of 16 lanes with row_ror 8/4/2/1, then combine the four rows through SGPRs.
template <int CTRL> __device__ __forceinline__ float dpp_f32(const float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL> __device__ __forceinline__ int dpp_i32(const int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
template <int CTRL> __device__ __forceinline__ double dpp_f64(const double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned) __builtin_amdgcn_update_dpp(0, (int) (unsigned) u, CTRL, 0xF, 0xF, true);
    const unsigned hi = (unsigned) __builtin_amdgcn_update_dpp(0, (int) (unsigned) (u >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((unsigned long long) hi << 32) | lo);
}
#define MI_DPP_ROR8 0x128
#define MI_DPP_ROR4 0x124
#define MI_DPP_ROR2 0x122
#define MI_DPP_ROR1 0x121
#define MI_DPP_QUAD_XOR1 0xB1  /* quad_perm [1,0,3,2] */
#define MI_DPP_QUAD_XOR2 0x4E  /* quad_perm [2,3,0,1] */
#define MI_DPP_HALF_MIRROR 0x141

__device__ __forceinline__ float readlane_f32(const float v, const int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
// sum over the 16 lanes of each DPP row, result in every lane of the row
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f32<MI_DPP_ROR8>(v);
    v += dpp_f32<MI_DPP_ROR4>(v);
    v += dpp_f32<MI_DPP_ROR2>(v);
    v += dpp_f32<MI_DPP_ROR1>(v);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    return ((readlane_f32(v, 0) + readlane_f32(v, 16)) + readlane_f32(v, 32)) + readlane_f32(v, 48);
}
__device__ __forceinline__ double wave_sum_d(double v) {
    v += dpp_f64<MI_DPP_ROR8>(v);
    v += dpp_f64<MI_DPP_ROR4>(v);
    v += dpp_f64<MI_DPP_ROR2>(v);
    v += dpp_f64<MI_DPP_ROR1>(v);
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    double r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned lo = (unsigned) __builtin_amdgcn_readlane((int) (unsigned) u, 16 * i);
        const unsigned hi = (unsigned) __builtin_amdgcn_readlane((int) (unsigned) (u >> 32), 16 * i);
        r[i] = __builtin_bit_cast(double, ((unsigned long long) hi << 32) | lo);
    }
    return ((r[0] + r[1]) + r[2]) + r[3];
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f32<MI_DPP_ROR8>(v));
    v = fmaxf(v, dpp_f32<MI_DPP_ROR4>(v));
    v = fmaxf(v, dpp_f32<MI_DPP_ROR2>(v));
    v = fmaxf(v, dpp_f32<MI_DPP_ROR1>(v));
    return fmaxf(fmaxf(readlane_f32(v, 0), readlane_f32(v, 16)), fmaxf(readlane_f32(v, 32), readlane_f32(v, 48)));
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

// One wave64 quantises 256 values to EIGHT Q8_0 blocks exactly as ggml-cpu's quantize_row_q8_0_ref does (8 lanes per 32-value
// block: d = amax / 127 stored through fp16, q = roundf(x / d) with the unrounded d).  y points at the first of the 8 blocks.
template <typename Q80> __device__ __forceinline__ void wave_quantize_q8_0(const float (&v)[4], const int lane, Q80 * y) {
    float amax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
    amax = fmaxf(amax, dpp_f32<MI_DPP_QUAD_XOR1>(amax));
    amax = fmaxf(amax, dpp_f32<MI_DPP_QUAD_XOR2>(amax));
    amax = fmaxf(amax, dpp_f32<MI_DPP_HALF_MIRROR>(amax));
    const float d = amax / 127.0f;
    const float id = d != 0.0f ? 1.0f / d : 0.0f;
    uint32_t packed = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) packed |= (uint32_t) ((int) roundf(v[k] * id) & 0xFF) << (8 * k);
    Q80 * blk = y + (lane >> 3);
    ((uint32_t *) blk->qs)[lane & 7] = packed;
    if ((lane & 7) == 0) blk->d = h2f(f2h(d));
}

// One wave64 quantises one 256-value block to Q8_K exactly as ggml-cpu's quantize_row_q8_K_ref does (first-index
// arg-max of |x| defines the sign of the scale, iscale = -127/max, round-half-even, clamp to 127, 16-value bsums).
// v[0..3] are the lane's four consecutive values; y may point to LDS or global memory.
template <typename Q8K> __device__ __forceinline__ void wave_quantize_q8_K(const float (&v)[4], const int lane, Q8K * y) {
    // lane-local: largest |x| and the FIRST element attaining it (the CPU scans with a strict `ax > amax`)
    float amax_l = 0.0f, mx_l = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float ax = fabsf(v[k]);
        if (ax > amax_l) { amax_l = ax; mx_l = v[k]; }
    }
    const float amax = wave_max(amax_l);
    int q[4] = {0, 0, 0, 0};
    float iscale = 0.0f;
    if (amax != 0.0f) {
        // lanes hold consecutive groups of four values, so the lowest lane attaining amax holds the first index
        const unsigned long long hit = __ballot(amax_l == amax);
        const int first = __ffsll((long long) hit) - 1;
        const float mx = readlane_f32(mx_l, first);
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
    s += dpp_i32<MI_DPP_QUAD_XOR1>(s);
    s += dpp_i32<MI_DPP_QUAD_XOR2>(s);
    if ((lane & 3) == 0) y->bsums[lane >> 2] = f2h((float) s);  // |s| <= 2032: exact in f16
    const int s32 = s + dpp_i32<MI_DPP_HALF_MIRROR>(s);  // every lane of a quad holds the quad sum: add the other quad of the 8-lane group
    if ((lane & 7) == 0) y->bs32[lane >> 3] = (int16_t) s32;
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
// rope_yarn for pair ip at angle theta
__device__ __forceinline__ void rope_yarn_dev(const float theta, const int ip, const float * __restrict__ ff, const rope_consts rc, float & cs, float & sn) {
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
__device__ __forceinline__ void rope_cos_sin(const int ip, const float pos_f, const float * __restrict__ ff, const rope_consts rc, float & cs, float & sn) {
    float theta = pos_f;
    for (int k = 0; k < ip; ++k) theta *= rc.theta_scale;
    rope_yarn_dev(theta, ip, ff, rc, cs, sn);
}

}  // namespace mi355x
