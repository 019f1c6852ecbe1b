// kv_dequant.h — device helpers shared by kv_types.hip (the f16 image of a K / V view) and fattn.hip (the decode attention kernel reading a cache
// kept in another type in place): one OCTET = eight consecutive values of a 32-value block, what one lane of those kernels owns.
//
// Layouts (include/ggml_abi.h): q8_0 {d, qs[32]}; q4_0 / iq4_nl {d, qs[16]}; q4_1 {d, m, qs[16]}; q5_0 {d, qh[4], qs[16]}; q5_1 {d, m, qh[4], qs[16]}.
// In the 4- and 5-bit formats value j < 16 is the LOW nibble of byte j and value j + 16 the HIGH nibble of the same byte, the fifth bit of value j is bit
// j of qh: octet o = 0..3 therefore reads bytes 8 (o & 1) .. + 7, their low (o < 2) or high nibbles, and qh bits 8 o .. 8 o + 7.
#pragma once
#include "dev_util.h"

namespace mi355x {

__device__ static const int8_t k_iq4nl_values[16] = {-127, -104, -83, -65, -49, -35, -22, -10, 1, 13, 25, 38, 53, 69, 89, 113};

__device__ __forceinline__ int kv_block_bytes_any(const int type) {
    return type == GGML_TYPE_F32 ? 128 : (type == GGML_TYPE_BF16 || type == GGML_TYPE_F16) ? 64 : type == GGML_TYPE_Q8_0 ? 34 : (type == GGML_TYPE_Q4_0 || type == GGML_TYPE_IQ4_NL) ? 18 :
           type == GGML_TYPE_Q4_1 ? 20 : type == GGML_TYPE_Q5_0 ? 22 : 24;
}

// the raw bytes of an octet, as four dwords that can sit in registers while the loads are in flight (NOT for f32: eight dwords):
//   f16 / bf16: the eight values;  q8_0: x, y = the 8 quants, z = d;  4 / 5-bit formats: x, y = the 8 nibble bytes, z = d | m << 16, w = qh
__device__ __forceinline__ uint4 kv_load_octet_raw(const int type, const char * blk, const int o) {
    switch (type) {
        case GGML_TYPE_F16: case GGML_TYPE_BF16: return *(const uint4 *) (blk + 16 * o);
        case GGML_TYPE_Q8_0: return make_uint4(ld32_a2(blk + 2 + 8 * o), ld32_a2(blk + 6 + 8 * o), (uint32_t) ld16(blk), 0u);
        case GGML_TYPE_Q4_1: return make_uint4(ld32_a2(blk + 4 + 8 * (o & 1)), ld32_a2(blk + 8 + 8 * (o & 1)), ld32_a2(blk), 0u);
        case GGML_TYPE_Q5_0: return make_uint4(ld32_a2(blk + 6 + 8 * (o & 1)), ld32_a2(blk + 10 + 8 * (o & 1)), (uint32_t) ld16(blk), ld32_a2(blk + 2));
        case GGML_TYPE_Q5_1: return make_uint4(ld32_a2(blk + 8 + 8 * (o & 1)), ld32_a2(blk + 12 + 8 * (o & 1)), ld32_a2(blk), ld32_a2(blk + 4));
        default: /* Q4_0, IQ4_NL */ return make_uint4(ld32_a2(blk + 2 + 8 * (o & 1)), ld32_a2(blk + 6 + 8 * (o & 1)), (uint32_t) ld16(blk), 0u);
    }
}
// ... -> the eight values in f32: level * d (+ m), one rounding per operation (dequantize_row_* of ggml-quants.c)
__device__ __forceinline__ void kv_octet_f32(const int type, const uint4 raw, const int o, float (&y)[8]) {
    const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
    if (type == GGML_TYPE_F16) {
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = h2f((uint16_t) (w[j >> 1] >> (16 * (j & 1))));
        return;
    }
    if (type == GGML_TYPE_BF16) {
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = __uint_as_float((w[j >> 1] >> (16 * (j & 1))) << 16);
        return;
    }
    const float d = h2f((uint16_t) (raw.z & 0xFFFFu));
    if (type == GGML_TYPE_Q8_0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = d * (float) (int8_t) (w[j >> 2] >> (8 * (j & 3)));
        return;
    }
    const bool offset = type == GGML_TYPE_Q4_1 || type == GGML_TYPE_Q5_1, five = type == GGML_TYPE_Q5_0 || type == GGML_TYPE_Q5_1;
    const float m = offset ? h2f((uint16_t) (raw.z >> 16)) : 0.0f;
    const int sh = (o >> 1) * 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        int q = (int) ((w[j >> 2] >> (8 * (j & 3) + sh)) & 0x0Fu);
        if (five) q |= (int) ((raw.w >> (8 * o + j)) & 1u) << 4;
        if (type == GGML_TYPE_IQ4_NL) y[j] = d * (float) k_iq4nl_values[q];
        else if (offset) y[j] = (float) q * d + m;
        else y[j] = (float) (q - (five ? 16 : 8)) * d;
    }
}
// ... -> packed f16 (ONE rounding of the exact f32 value: the form the f16 attention kernels read, in the image and in place alike)
__device__ __forceinline__ void kv_octet_f16(const int type, const uint4 raw, const int o, uint32_t (&h)[4]) {
    if (type == GGML_TYPE_F16) {
        h[0] = raw.x; h[1] = raw.y; h[2] = raw.z; h[3] = raw.w;
        return;
    }
    float y[8];
    kv_octet_f32(type, raw, o, y);
#pragma unroll
    for (int j = 0; j < 4; ++j) h[j] = (uint32_t) f2h(y[2 * j]) | ((uint32_t) f2h(y[2 * j + 1]) << 16);
}

// ---- compile-time forms for the decode attention kernel (fattn.hip, KVT): q4_0, q4_1, q5_0, q5_1 (levels are plain integers) and iq4_nl (a 16-entry table).
// 0x6400 | n is the f16 number 1024 + n: the eight levels of an octet become packed f16 with two byte permutes per dword, one packed subtraction
// (1024 + n - (1024 + 8) = n - 8, exact) and one packed multiply by the block's f16 scale — ONE rounding of the exact product, the number
// f2h((float) level * d) is; the offset formats take a packed fma (level * d + m rounded once: within half an f16 ulp of the image's f32 sum).
typedef _Float16 kv_half2 __attribute__((ext_vector_type(2)));
template <int T> __device__ __forceinline__ uint4 kv_load_octet_raw_t(const char * blk, const int o) {
    if constexpr (T == GGML_TYPE_Q4_0 || T == GGML_TYPE_IQ4_NL) return make_uint4(ld32_a2(blk + 2 + 8 * (o & 1)), ld32_a2(blk + 6 + 8 * (o & 1)), (uint32_t) ld16(blk), 0u);
    else if constexpr (T == GGML_TYPE_Q4_1) return make_uint4(ld32_a2(blk + 4 + 8 * (o & 1)), ld32_a2(blk + 8 + 8 * (o & 1)), ld32_a2(blk), 0u);
    else if constexpr (T == GGML_TYPE_Q5_0) return make_uint4(ld32_a2(blk + 6 + 8 * (o & 1)), ld32_a2(blk + 10 + 8 * (o & 1)), (uint32_t) ld16(blk), ld32_a2(blk + 2));
    else return make_uint4(ld32_a2(blk + 8 + 8 * (o & 1)), ld32_a2(blk + 12 + 8 * (o & 1)), ld32_a2(blk), ld32_a2(blk + 4));  // Q5_1
}
// the eight integer LEVELS of an octet as the bytes of two dwords — what the reference's integer block dots multiply (ggml_vec_dot_q4_0_q8_0 ...:
// oracle/ggml_cpu_ref.c restates them) and what dequantize_row_* scales.  q4_0 / q4_1: the nibble 0..15; q5_0 / q5_1: nibble | fifth bit << 4 = 0..31 (the
// "- 8" / "- 16" of the symmetric formats is applied to the block's integer sum by the caller: sum((n - 8) q) = sum(n q) - 8 sum(q));
// iq4_nl: the table value itself as a SIGNED byte (SIGNED = true: the dot's operand) or + 128 as an unsigned one (the V side: cvt_f32_ubyte, then - 128 d).
template <int T, bool SIGNED> __device__ __forceinline__ void kv_octet_levels_t(const uint4 raw, const int o, uint32_t & t0, uint32_t & t1) {
    constexpr bool FIVE = T == GGML_TYPE_Q5_0 || T == GGML_TYPE_Q5_1;
    const int sh = (o >> 1) * 4;
    t0 = (raw.x >> sh) & 0x0F0F0F0Fu;
    t1 = (raw.y >> sh) & 0x0F0F0F0Fu;
    if constexpr (FIVE) {
        const uint32_t b = raw.w >> (8 * o);
        t0 |= (((b & 0xFu) * 0x00204081u) & 0x01010101u) << 4;
        t1 |= ((((b >> 4) & 0xFu) * 0x00204081u) & 0x01010101u) << 4;
    }
    if constexpr (T == GGML_TYPE_IQ4_NL) {
        // {-127, -104, -83, -65 | -49, -35, -22, -10 || 1, 13, 25, 38 | 53, 69, 89, 113} as two's-complement bytes, or + 128
        constexpr uint32_t A0 = SIGNED ? 0xBFAD9881u : 0x3F2D1801u, A1 = SIGNED ? 0xF6EADDCFu : 0x766A5D4Fu, B0 = SIGNED ? 0x26190D01u : 0xA6998D81u, B1 = SIGNED ? 0x71594535u : 0xF1D9C5B5u;
        const uint32_t s0 = t0 & 0x07070707u, s1 = t1 & 0x07070707u;
        const uint32_t m0 = ((t0 >> 3) & 0x01010101u) * 0xFFu, m1 = ((t1 >> 3) & 0x01010101u) * 0xFFu;
        t0 = (__builtin_amdgcn_perm(A1, A0, s0) & ~m0) | (__builtin_amdgcn_perm(B1, B0, s0) & m0);
        t1 = (__builtin_amdgcn_perm(A1, A0, s1) & ~m1) | (__builtin_amdgcn_perm(B1, B0, s1) & m1);
    }
}
// ... and the eight VALUES in f32 exactly as dequantize_row_* leaves them (v_to_float of ggml-cpu's flash attention: a block-format V row is accumulated in f32):
// symmetric formats (level - z) * d — computed as fma(level, d, -z d): z d is exact, the fma rounds the exact product once, the same number; offset formats
// level * d, rounded, + m, rounded (the reference is compiled without contraction, and so is this file)
template <int T> __device__ __forceinline__ void kv_octet_f32_t(const uint4 raw, const int o, float (&y)[8]) {
    constexpr bool OFFSET = T == GGML_TYPE_Q4_1 || T == GGML_TYPE_Q5_1, FIVE = T == GGML_TYPE_Q5_0 || T == GGML_TYPE_Q5_1;
    uint32_t t0, t1;
    kv_octet_levels_t<T, false>(raw, o, t0, t1);
    const float d = h2f((uint16_t) (raw.z & 0xFFFFu));
    if constexpr (OFFSET) {
        const float m = h2f((uint16_t) (raw.z >> 16));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            y[i] = (float) ((t0 >> (8 * i)) & 0xFFu) * d + m;
            y[4 + i] = (float) ((t1 >> (8 * i)) & 0xFFu) * d + m;
        }
    } else {
        constexpr float Z = T == GGML_TYPE_IQ4_NL ? 128.0f : FIVE ? 16.0f : 8.0f;
        const float zd = -Z * d;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            y[i] = __builtin_fmaf((float) ((t0 >> (8 * i)) & 0xFFu), d, zd);
            y[4 + i] = __builtin_fmaf((float) ((t1 >> (8 * i)) & 0xFFu), d, zd);
        }
    }
}
template <int T> __device__ __forceinline__ constexpr int kv_block_bytes_t() { return (T == GGML_TYPE_Q4_0 || T == GGML_TYPE_IQ4_NL) ? 18 : T == GGML_TYPE_Q4_1 ? 20 : T == GGML_TYPE_Q5_0 ? 22 : 24; }
template <int T> __device__ __forceinline__ void kv_octet_f16_t(const uint4 raw, const int o, uint32_t (&h)[4]) {
    constexpr bool OFFSET = T == GGML_TYPE_Q4_1 || T == GGML_TYPE_Q5_1, FIVE = T == GGML_TYPE_Q5_0 || T == GGML_TYPE_Q5_1;
    const int sh = (o >> 1) * 4;
    uint32_t t0 = (raw.x >> sh) & 0x0F0F0F0Fu, t1 = (raw.y >> sh) & 0x0F0F0F0Fu;
    if constexpr (FIVE) {  // the fifth bits of values 8 o .. 8 o + 7: bit i of a nibble of them to bit 4 of byte i (x * (1 + 2^7 + 2^14 + 2^21) puts bit i at 8 i)
        const uint32_t b = raw.w >> (8 * o);
        t0 |= (((b & 0xFu) * 0x00204081u) & 0x01010101u) << 4;
        t1 |= ((((b >> 4) & 0xFu) * 0x00204081u) & 0x01010101u) << 4;
    }
    if constexpr (T == GGML_TYPE_IQ4_NL) {
        // the 16 non-linear levels through two 8-entry byte tables (level + 128, so that the byte is unsigned): a byte permute looks four nibbles up at once,
        // bit 3 of each nibble picks the table; 1024 + (level + 128) - 1152 = level, exact
        constexpr uint32_t A0 = 0x3F2D1801u, A1 = 0x766A5D4Fu, B0 = 0xA6998D81u, B1 = 0xF1D9C5B5u;  // {1, 24, 45, 63 | 79, 93, 106, 118 || 129, 141, 153, 166 | 181, 197, 217, 241}
        const uint32_t s0 = t0 & 0x07070707u, s1 = t1 & 0x07070707u;
        const uint32_t m0 = ((t0 >> 3) & 0x01010101u) * 0xFFu, m1 = ((t1 >> 3) & 0x01010101u) * 0xFFu;
        t0 = (__builtin_amdgcn_perm(A1, A0, s0) & ~m0) | (__builtin_amdgcn_perm(B1, B0, s0) & m0);
        t1 = (__builtin_amdgcn_perm(A1, A0, s1) & ~m1) | (__builtin_amdgcn_perm(B1, B0, s1) & m1);
    }
    const uint32_t k64 = 0x64646464u;
    const uint32_t p[4] = {__builtin_amdgcn_perm(k64, t0, 0x05010400u), __builtin_amdgcn_perm(k64, t0, 0x07030602u), __builtin_amdgcn_perm(k64, t1, 0x05010400u),
                           __builtin_amdgcn_perm(k64, t1, 0x07030602u)};  // [n0, 0x64, n1, 0x64]: the halves 1024 + n
    const _Float16 d = __builtin_bit_cast(_Float16, (uint16_t) (raw.z & 0xFFFFu));
    const kv_half2 d2 = {d, d};
    constexpr float zero = 1024.0f + (T == GGML_TYPE_IQ4_NL ? 128.0f : OFFSET ? 0.0f : FIVE ? 16.0f : 8.0f);
    const kv_half2 z2 = {(_Float16) zero, (_Float16) zero};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const kv_half2 lv = __builtin_bit_cast(kv_half2, p[i]) - z2;
        if constexpr (OFFSET) {
            const _Float16 m = __builtin_bit_cast(_Float16, (uint16_t) (raw.z >> 16));
            const kv_half2 m2 = {m, m};
            h[i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_fma(lv, d2, m2));
        } else {
            h[i] = __builtin_bit_cast(uint32_t, lv * d2);
        }
    }
}

}  // namespace mi355x
