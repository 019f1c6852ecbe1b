// kv_quant.h — device code that turns 32 f32 values into one block of a KV-cache format (q4_0, q4_1, q5_0, q5_1, iq4_nl; bf16 conversion), byte for byte what
// ggml's quantize_row_*_ref leaves (one f32 rounding per operation, the FIRST element of largest magnitude sets the sign of the scale, truncating
// conversions).  Two forms: a thread per block (kv_types.hip: SET_ROWS / CPY of whole rows) and a LANE per value (a block = 32 lanes of a wave: the fused
// Q/K/V launch's epilogue, qkv.hip, and SET_ROWS into iq4_nl, whose level search is ~90 VALU per value).
#pragma once
#include "dev_util.h"

namespace mi355x {

__device__ __forceinline__ uint16_t f2bf(const float f) {  // ggml_compute_fp32_to_bf16: nearest even, NaN kept quiet
    const uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t) ((u >> 16) | 64);
    return (uint16_t) ((u + (0x7fffu + ((u >> 16) & 1u))) >> 16);
}
__device__ __forceinline__ void st16(char * p, const uint16_t v) { *(uint16_t *) p = v; }  // (blocks are 2-byte aligned, no more)

__device__ __forceinline__ float iq4nl_level(const int l) {  // kvalues_iq4nl[l] from constants (two selects deep, no memory)
    constexpr float V[16] = {-127.f, -104.f, -83.f, -65.f, -49.f, -35.f, -22.f, -10.f, 1.f, 13.f, 25.f, 38.f, 53.f, 69.f, 89.f, 113.f};
    float r = V[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) r = l == k ? V[k] : r;
    return r;
}
// best_index_int8(16, kvalues_iq4nl, x) without its table walk (a dependent chain of memory loads per value: SET_ROWS to iq4_nl ran at 25 us per layer):
// the bracketing levels lo <= x < hi come out of 15 compares against constants, the choice between them is the reference's own expression
__device__ __forceinline__ int iq4nl_best_index(const float x) {
    constexpr float V[16] = {-127.f, -104.f, -83.f, -65.f, -49.f, -35.f, -22.f, -10.f, 1.f, 13.f, 25.f, 38.f, 53.f, 69.f, 89.f, 113.f};
    if (x <= V[0]) return 0;
    if (x >= V[15] || x != x) return 15;  // (a NaN fails every `x < val[mid]` of the reference's search and ends at the top)
    int ml = 0;
    float lo = V[0], hi = V[15];
#pragma unroll
    for (int k = 1; k < 16; ++k) {
        const bool ge = x >= V[k];
        ml += ge ? 1 : 0;
        lo = ge ? V[k] : lo;             // the largest level <= x (levels ascend)
        hi = (!ge && V[k] < hi) ? V[k] : hi;  // the smallest level > x
    }
    return x - lo < hi - x ? ml : ml + 1;
}

// the element of largest magnitude with its sign, the first one on a tie (quantize_row_q4_0_ref: `if (amax < fabsf(v))`)
__device__ __forceinline__ float signed_extreme(const float (&x)[32]) {
    float amax = 0.0f, mx = 0.0f;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const float v = x[j];
        if (amax < fabsf(v)) {
            amax = fabsf(v);
            mx = v;
        }
    }
    return mx;
}

// one block of 32 values -> its bytes at `out` (quantize_row_{q4_0,q4_1,q5_0,q5_1,iq4_nl}_ref)
template <int TYPE> __device__ __forceinline__ void quantize_block(const float (&x)[32], char * out) {
    if constexpr (TYPE == GGML_TYPE_Q4_0 || TYPE == GGML_TYPE_Q5_0) {
        constexpr bool Q5 = TYPE == GGML_TYPE_Q5_0;
        const float d = signed_extreme(x) / (Q5 ? -16.0f : -8.0f);
        const float id = d != 0.0f ? 1.0f / d : 0.0f;
        // (an all-zero block: 0 / -8 = -0.0, stored as 0x8000 by the reference.  The compiler folds the scaling and the conversion into ONE
        // v_fma_mixlo_f16 with a +0 addend, and (-0) + (+0) = +0 loses the sign — the zero is stored from its own bits)
        st16(out, d == 0.0f ? (uint16_t) (__float_as_uint(d) >> 16) : f2h(d));
        uint32_t qh = 0;
        char * qs = out + (Q5 ? 6 : 2);
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
            uint32_t pair = 0;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float x0 = x[j + e] * id, x1 = x[16 + j + e] * id;
                const int xi0 = min(Q5 ? 31 : 15, (int) (x0 + (Q5 ? 16.5f : 8.5f))), xi1 = min(Q5 ? 31 : 15, (int) (x1 + (Q5 ? 16.5f : 8.5f)));
                pair |= (uint32_t) ((xi0 & 0x0F) | ((xi1 & 0x0F) << 4)) << (8 * e);
                if (Q5) {
                    qh |= (uint32_t) ((xi0 & 0x10) >> 4) << (j + e);
                    qh |= (uint32_t) ((xi1 & 0x10) >> 4) << (j + e + 16);
                }
            }
            st16(qs + j, (uint16_t) pair);
        }
        if (Q5) {
            st16(out + 2, (uint16_t) (qh & 0xFFFF));
            st16(out + 4, (uint16_t) (qh >> 16));
        }
    } else if constexpr (TYPE == GGML_TYPE_Q4_1 || TYPE == GGML_TYPE_Q5_1) {
        constexpr bool Q5 = TYPE == GGML_TYPE_Q5_1;
        float mn = 3.402823466e+38f, mx = -3.402823466e+38f;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            if (x[j] < mn) mn = x[j];
            if (x[j] > mx) mx = x[j];
        }
        const float d = (mx - mn) / (Q5 ? 31.0f : 15.0f);
        const float id = d != 0.0f ? 1.0f / d : 0.0f;
        st16(out, f2h(d));
        st16(out + 2, f2h(mn));
        uint32_t qh = 0;
        char * qs = out + (Q5 ? 8 : 4);
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
            uint32_t pair = 0;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float x0 = (x[j + e] - mn) * id, x1 = (x[16 + j + e] - mn) * id;
                // Q4_1: MIN(15, (int8_t)(x0 + 0.5f)); Q5_1: (uint8_t)(x0 + 0.5f) — x0 lies in [0, levels], so both conversions are the truncation
                const int xi0 = Q5 ? (int) (x0 + 0.5f) : min(15, (int) (x0 + 0.5f)), xi1 = Q5 ? (int) (x1 + 0.5f) : min(15, (int) (x1 + 0.5f));
                pair |= (uint32_t) ((xi0 & 0x0F) | ((xi1 & 0x0F) << 4)) << (8 * e);
                if (Q5) {
                    qh |= (uint32_t) ((xi0 & 0x10) >> 4) << (j + e);
                    qh |= (uint32_t) ((xi1 & 0x10) >> 4) << (j + e + 16);
                }
            }
            st16(qs + j, (uint16_t) pair);
        }
        if (Q5) {
            st16(out + 4, (uint16_t) (qh & 0xFFFF));
            st16(out + 6, (uint16_t) (qh >> 16));
        }
    } else {  // IQ4_NL: quantize_row_iq4_nl_impl(32, 32, ..., ntry = -1): levels against max / -127, then the least-squares scale under the weights x^2
        float amax = 0.0f, mx = 0.0f;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const float ax = fabsf(x[j]);
            if (ax > amax) {
                amax = ax;
                mx = x[j];
            }
        }
        uint32_t lv[4] = {0, 0, 0, 0};  // 32 four-bit levels, element j in nibble j
        float scale = 0.0f;
        if (amax >= 1e-15f) {
            const float d0 = mx / -127.0f;
            const float id = 1.0f / d0;
            float sumqx = 0.0f, sumq2 = 0.0f;
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const int l = iq4nl_best_index(id * x[j]);
                lv[j >> 3] |= (uint32_t) l << (4 * (j & 7));
                const float q = iq4nl_level(l), w = x[j] * x[j];
                sumqx += w * q * x[j];
                sumq2 += w * q * q;
            }
            scale = sumqx / sumq2;
        }
        st16(out, f2h(scale));
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
            uint32_t pair = 0;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const uint32_t lo = (lv[(j + e) >> 3] >> (4 * ((j + e) & 7))) & 15u, hi = (lv[(16 + j + e) >> 3] >> (4 * ((16 + j + e) & 7))) & 15u;
                pair |= (lo | (hi << 4)) << (8 * e);
            }
            st16(out + 2 + j, (uint16_t) pair);
        }
    }
}


// ---- a lane per value: lanes base .. base + 31 (base = lane & 32) hold the block's 32 values in order.  `store`: this half-wave writes the block.
// first lane (lowest index) whose key is the best under `better(candidate, incumbent)`, by a butterfly that carries the index
#define KVQ_ARGBEST(KEY, AT, BETTER)                                                \
    _Pragma("unroll") for (int off_ = 16; off_ >= 1; off_ >>= 1) {                 \
        const float ok_ = __shfl_xor(KEY, off_);                                    \
        const int oa_ = __shfl_xor(AT, off_);                                       \
        const bool take_ = BETTER(ok_, KEY) || (ok_ == KEY && oa_ < AT);            \
        KEY = take_ ? ok_ : KEY;                                                    \
        AT = take_ ? oa_ : AT;                                                      \
    }
#define KVQ_GT(a, b) ((a) > (b))
#define KVQ_LT(a, b) ((a) < (b))
template <int TYPE> __device__ __forceinline__ void quantize_block_lanes(const float xv, const int lane, char * out, const bool store) {
    const int l32 = lane & 31, base = lane & 32;
    int level = 0;          // this value's 4- or 5-bit level
    uint16_t h0 = 0, h1 = 0;  // the block's f16 header field(s)
    if constexpr (TYPE == GGML_TYPE_Q4_0 || TYPE == GGML_TYPE_Q5_0) {
        constexpr bool Q5 = TYPE == GGML_TYPE_Q5_0;
        float ax = fabsf(xv);
        int at = l32;
        KVQ_ARGBEST(ax, at, KVQ_GT)  // the FIRST largest magnitude (quantize_row_q4_0_ref: `if (amax < fabsf(v))`)
        const float mx = __shfl(xv, base + at);
        const float d = (ax > 0.0f ? mx : 0.0f) / (Q5 ? -16.0f : -8.0f);  // (no magnitude above zero: the reference's max stays 0.0f, whatever the signs of the zeros)
        const float id = d != 0.0f ? 1.0f / d : 0.0f;
        level = min(Q5 ? 31 : 15, (int) (xv * id + (Q5 ? 16.5f : 8.5f)));
        h0 = d == 0.0f ? (uint16_t) (__float_as_uint(d) >> 16) : f2h(d);  // (an all-zero block stores -0.0: see quantize_block)
    } else if constexpr (TYPE == GGML_TYPE_Q4_1 || TYPE == GGML_TYPE_Q5_1) {
        constexpr bool Q5 = TYPE == GGML_TYPE_Q5_1;
        float mn = xv, mxv = xv;
        int an = l32, ax_ = l32;
        KVQ_ARGBEST(mn, an, KVQ_LT)   // first minimum / first maximum, as the reference's scan meets them (the sign of a zero minimum is stored)
        KVQ_ARGBEST(mxv, ax_, KVQ_GT)
        mn = __shfl(xv, base + an);
        mxv = __shfl(xv, base + ax_);
        const float d = (mxv - mn) / (Q5 ? 31.0f : 15.0f);
        const float id = d != 0.0f ? 1.0f / d : 0.0f;
        const float x0 = (xv - mn) * id;
        level = Q5 ? (int) (x0 + 0.5f) : min(15, (int) (x0 + 0.5f));
        h0 = f2h(d);
        h1 = f2h(mn);
    } else {  // IQ4_NL
        float ax = fabsf(xv);
        int at = l32;
        KVQ_ARGBEST(ax, at, KVQ_GT)
        const float mx = __shfl(xv, base + at);
        float scale = 0.0f;
        if (ax >= 1e-15f) {
            const float d0 = mx / -127.0f;
            const float id = 1.0f / d0;
            level = iq4nl_best_index(id * xv);
            const float q = iq4nl_level(level), w = xv * xv;
            const float pqx = w * q * xv, pq2 = w * q * q;
            float sumqx = 0.0f, sumq2 = 0.0f;
#pragma unroll
            for (int j = 0; j < 32; ++j) {  // the reference's order of additions (64 dependent adds per block, once)
                sumqx += __shfl(pqx, base + j);
                sumq2 += __shfl(pq2, base + j);
            }
            scale = sumqx / sumq2;
        }
        h0 = f2h(scale);
    }
    constexpr bool OFFSET = TYPE == GGML_TYPE_Q4_1 || TYPE == GGML_TYPE_Q5_1, FIVE = TYPE == GGML_TYPE_Q5_0 || TYPE == GGML_TYPE_Q5_1;
    // byte j of qs = level j | level (j + 16) << 4; bit j of qh = the fifth bit of level j
    const int hi = __shfl(level, base + ((l32 + 16) & 31));
    const uint32_t byte = (uint32_t) (level & 0x0F) | ((uint32_t) (hi & 0x0F) << 4);
    const uint32_t next = (uint32_t) __shfl((int) byte, lane + 1);
    const uint64_t fifth = __ballot((level & 0x10) != 0);
    if (!store) return;
    char * qs = out + 2 + (OFFSET ? 2 : 0) + (FIVE ? 4 : 0);
    if (l32 < 16 && (l32 & 1) == 0) st16(qs + l32, (uint16_t) (byte | (next << 8)));
    if (l32 == 0) {
        st16(out, h0);
        if constexpr (OFFSET) st16(out + 2, h1);
        if constexpr (FIVE) {
            const uint32_t qh = (uint32_t) (fifth >> base);
            st16(out + (OFFSET ? 4 : 2), (uint16_t) (qh & 0xFFFF));
            st16(out + (OFFSET ? 6 : 4), (uint16_t) (qh >> 16));
        }
    }
}
#undef KVQ_ARGBEST
#undef KVQ_GT
#undef KVQ_LT

}  // namespace mi355x
