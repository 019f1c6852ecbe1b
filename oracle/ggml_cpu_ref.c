/*
 * ggml_cpu_ref.c — plain-C restatement of the ggml-cpu *generic* algorithms for the hot path.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  PARITY UNPINNED (see oracle.h).
 *
 * Every function names the upstream routine it restates.  The upstream sources are absent from
 * /root/reference (un-vendored llama.cpp submodule); anchors are the patch-pinned lines where a
 * hunk exists and otherwise the published algorithm.
 *
 * Build: see oracle/Makefile (-O3 -ffp-contract=off so that float expressions are evaluated
 * exactly as written: one rounding per operation, no FMA contraction).
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef double ggml_float; /* ggml-cpu accumulates scalar reductions in double */

/* Summation-order variant.  0 = the generic ggml-cpu order (blocks first to last).  1 = blocks last to first: an
 * equally valid f32 evaluation order (every SIMD build of ggml-cpu has its own), used by the tests to measure how far
 * the network's logits move under a change of summation order ALONE — the yardstick for the GPU-vs-oracle gates. */
/* Sensitivity probes for the tests (variant 0 = the generic ggml-cpu algorithm, the only one used as a checker).  Each probe
 * changes ONE thing that two correct implementations of the same graph legitimately differ in, by about an ulp:
 *   1  block dots summed last block first (f32 summation order of the mat-muls)
 *   2  RMS_NORM sum of squares accumulated in f32 instead of double
 *   3  expf() results moved one ulp up or down, by a bit of the argument (what a different libm / a hardware exp is
 *      entitled to; a uniform shift would cancel in soft_max's ratio)
 *   4  FLASH_ATTN_EXT over a block-format K (q8_0, q4_0, q4_1, q5_0, q5_1, iq4_nl): the logit K.q from the dequantised row and the
 *      UNQUANTISED query — how far ggml-cpu's 8-bit query (Q8_0 / Q8_1) alone moves the result; an implementation that keeps the
 *      query in f16 (csrc/kv_types.hip) sits on this side of that distance                                          */
static int g_variant = 0;
void oracle_set_variant(int v) { g_variant = v; }
#define BLK_IDX(i, nb) (g_variant == 1 ? ((nb) - 1 - (i)) : (i))
static inline float oracle_expf(float x) {
    const float r = expf(x);
    if (g_variant != 3 || !(r > 0.0f) || !(r < INFINITY)) return r;
    uint32_t u;
    memcpy(&u, &x, 4);
    return ((u ^ (u >> 7) ^ (u >> 13)) & 1u) ? nextafterf(r, INFINITY) : nextafterf(r, 0.0f);
}

/* ------------------------------------------------------------------------------------------ */
/* fp16 <-> fp32 (IEEE binary16, round-to-nearest-even; same results as F16C / ggml's fallback) */
/* ------------------------------------------------------------------------------------------ */
float oracle_fp16_to_fp32(ggml_fp16_t h) {
    const uint32_t sign = (uint32_t) (h & 0x8000) << 16;
    const uint32_t exp = (h >> 10) & 0x1F;
    const uint32_t man = h & 0x3FF;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else { /* subnormal: value = man * 2^-24 */
            float f = (float) man * 5.9604644775390625e-08f;
            memcpy(&bits, &f, 4);
            bits |= sign;
        }
    } else if (exp == 31) {
        bits = sign | 0x7F800000u | (man << 13);
    } else {
        bits = sign | ((exp + 112) << 23) | (man << 13);
    }
    float out;
    memcpy(&out, &bits, 4);
    return out;
}

ggml_fp16_t oracle_fp32_to_fp16(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint16_t sign = (uint16_t) ((x >> 16) & 0x8000);
    const uint32_t ax = x & 0x7FFFFFFFu;
    if (ax >= 0x7F800000u) { /* inf / nan */
        return (ggml_fp16_t) (sign | 0x7C00 | ((ax > 0x7F800000u) ? (0x200 | ((ax >> 13) & 0x3FF)) : 0));
    }
    if (ax >= 0x477FF000u) { /* >= 65520 rounds to inf */
        return (ggml_fp16_t) (sign | 0x7C00);
    }
    if (ax < 0x33000001u) { /* <= 2^-25 rounds to zero (2^-25 exactly ties to even = 0) */
        return sign;
    }
    const int32_t e = (int32_t) (ax >> 23) - 127;
    uint32_t man = (ax & 0x7FFFFFu) | 0x800000u;
    uint32_t shift;
    uint32_t base;
    if (e < -14) { /* subnormal half */
        shift = (uint32_t) (13 + (-14 - e));
        base = 0;
    } else {
        shift = 13;
        base = (uint32_t) (e + 15) << 10;
        man &= 0x7FFFFFu;
    }
    const uint32_t halfway = 1u << (shift - 1);
    const uint32_t rem = man & ((1u << shift) - 1);
    uint32_t q = man >> shift;
    if (rem > halfway || (rem == halfway && (q & 1))) q++;
    return (ggml_fp16_t) (sign | (base + q)); /* carry into exponent is the correct behaviour */
}

#define F16(x) oracle_fp16_to_fp32(x)

/* ------------------------------------------------------------------------------------------ */
/* dequantisation: dequantize_row_{q8_0,q4_K,q5_K,q6_K} (ggml-quants.c) */
/* ------------------------------------------------------------------------------------------ */
static inline void get_scale_min_k4(int j, const uint8_t * q, uint8_t * d, uint8_t * m) {
    if (j < 4) {
        *d = q[j] & 63;
        *m = q[j + 4] & 63;
    } else {
        *d = (uint8_t) ((q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4));
        *m = (uint8_t) ((q[j + 4] >> 4) | ((q[j - 0] >> 6) << 4));
    }
}

static void dequantize_row_q8_0(const block_q8_0 * x, float * y, int64_t k) {
    const int64_t nb = k / QK8_0;
    for (int64_t i = 0; i < nb; i++) {
        const float d = F16(x[i].d);
        for (int j = 0; j < QK8_0; ++j) y[i * QK8_0 + j] = x[i].qs[j] * d;
    }
}

static void dequantize_row_q4_K(const block_q4_K * x, float * y, int64_t k) {
    const int64_t nb = k / QK_K;
    for (int64_t i = 0; i < nb; i++) {
        const uint8_t * q = x[i].qs;
        const float d = F16(x[i].d);
        const float min = F16(x[i].dmin);
        int is = 0;
        uint8_t sc, m;
        for (int j = 0; j < QK_K; j += 64) {
            get_scale_min_k4(is + 0, x[i].scales, &sc, &m);
            const float d1 = d * sc;
            const float m1 = min * m;
            get_scale_min_k4(is + 1, x[i].scales, &sc, &m);
            const float d2 = d * sc;
            const float m2 = min * m;
            for (int l = 0; l < 32; ++l) *y++ = d1 * (q[l] & 0xF) - m1;
            for (int l = 0; l < 32; ++l) *y++ = d2 * (q[l] >> 4) - m2;
            q += 32;
            is += 2;
        }
    }
}

static void dequantize_row_q5_K(const block_q5_K * x, float * y, int64_t k) {
    const int64_t nb = k / QK_K;
    for (int64_t i = 0; i < nb; i++) {
        const uint8_t * ql = x[i].qs;
        const uint8_t * qh = x[i].qh;
        const float d = F16(x[i].d);
        const float min = F16(x[i].dmin);
        int is = 0;
        uint8_t sc, m;
        uint8_t u1 = 1, u2 = 2;
        for (int j = 0; j < QK_K; j += 64) {
            get_scale_min_k4(is + 0, x[i].scales, &sc, &m);
            const float d1 = d * sc;
            const float m1 = min * m;
            get_scale_min_k4(is + 1, x[i].scales, &sc, &m);
            const float d2 = d * sc;
            const float m2 = min * m;
            for (int l = 0; l < 32; ++l) *y++ = d1 * ((ql[l] & 0xF) + (qh[l] & u1 ? 16 : 0)) - m1;
            for (int l = 0; l < 32; ++l) *y++ = d2 * ((ql[l] >> 4) + (qh[l] & u2 ? 16 : 0)) - m2;
            ql += 32;
            is += 2;
            u1 <<= 2;
            u2 <<= 2;
        }
    }
}

static void dequantize_row_q6_K(const block_q6_K * x, float * y, int64_t k) {
    const int64_t nb = k / QK_K;
    for (int64_t i = 0; i < nb; i++) {
        const float d = F16(x[i].d);
        const uint8_t * ql = x[i].ql;
        const uint8_t * qh = x[i].qh;
        const int8_t * sc = x[i].scales;
        for (int n = 0; n < QK_K; n += 128) {
            for (int l = 0; l < 32; ++l) {
                const int is = l / 16;
                const int8_t q1 = (int8_t) ((ql[l + 0] & 0xF) | (((qh[l] >> 0) & 3) << 4)) - 32;
                const int8_t q2 = (int8_t) ((ql[l + 32] & 0xF) | (((qh[l] >> 2) & 3) << 4)) - 32;
                const int8_t q3 = (int8_t) ((ql[l + 0] >> 4) | (((qh[l] >> 4) & 3) << 4)) - 32;
                const int8_t q4 = (int8_t) ((ql[l + 32] >> 4) | (((qh[l] >> 6) & 3) << 4)) - 32;
                y[l + 0] = d * sc[is + 0] * q1;
                y[l + 32] = d * sc[is + 2] * q2;
                y[l + 64] = d * sc[is + 4] * q3;
                y[l + 96] = d * sc[is + 6] * q4;
            }
            y += 128;
            ql += 64;
            qh += 32;
            sc += 8;
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* The legacy 32-value formats a KV cache may be kept in (llama-box/engine_param.hpp:51-54: -ctk / -ctv f32, f16, bf16, q8_0, q4_0, */
/* q4_1, iq4_nl, q5_0, q5_1): quantize_row_*_ref / dequantize_row_* of ggml-quants.c, ggml_compute_fp32_to_bf16 of ggml-impl.h.       */
/* SET_ROWS stores a cache row through the destination type's from_float; FLASH_ATTN_EXT reads K through vec_dot and V through        */
/* to_float (below).                                                                                                                  */
/* ------------------------------------------------------------------------------------------ */
static inline float bf16_to_fp32(uint16_t h) {
    union { uint32_t u; float f; } v;
    v.u = (uint32_t) h << 16;
    return v.f;
}
static inline uint16_t fp32_to_bf16(float f) { /* ggml_compute_fp32_to_bf16: round to nearest even, NaNs kept quiet */
    union { uint32_t u; float f; } v;
    v.f = f;
    if ((v.u & 0x7fffffffu) > 0x7f800000u) return (uint16_t) ((v.u >> 16) | 64);
    return (uint16_t) ((v.u + (0x7fffu + ((v.u >> 16) & 1u))) >> 16);
}
uint16_t oracle_fp32_to_bf16(float f) { return fp32_to_bf16(f); }
float oracle_bf16_to_fp32(uint16_t h) { return bf16_to_fp32(h); }

static void quantize_row_q4_0_ref(const float * x, block_q4_0 * y, int64_t k) {
    const int64_t nb = k / 32;
    for (int64_t i = 0; i < nb; i++) {
        float amax = 0.0f, max = 0.0f; /* the value of largest magnitude, with its sign */
        for (int j = 0; j < 32; j++) {
            const float v = x[i * 32 + j];
            if (amax < fabsf(v)) { amax = fabsf(v); max = v; }
        }
        const float d = max / -8;
        const float id = d ? 1.0f / d : 0.0f;
        y[i].d = oracle_fp32_to_fp16(d);
        for (int j = 0; j < 16; ++j) {
            const float x0 = x[i * 32 + 0 + j] * id, x1 = x[i * 32 + 16 + j] * id;
            const uint8_t xi0 = (uint8_t) ((int8_t) (x0 + 8.5f) < 15 ? (int8_t) (x0 + 8.5f) : 15); /* MIN(15, (int8_t)(x0 + 8.5f)) */
            const uint8_t xi1 = (uint8_t) ((int8_t) (x1 + 8.5f) < 15 ? (int8_t) (x1 + 8.5f) : 15);
            y[i].qs[j] = (uint8_t) (xi0 | (xi1 << 4));
        }
    }
}
static void dequantize_row_q4_0(const block_q4_0 * x, float * y, int64_t k) {
    const int64_t nb = k / 32;
    for (int64_t i = 0; i < nb; i++) {
        const float d = F16(x[i].d);
        for (int j = 0; j < 16; ++j) {
            const int x0 = (x[i].qs[j] & 0x0F) - 8, x1 = (x[i].qs[j] >> 4) - 8;
            y[i * 32 + j] = x0 * d;
            y[i * 32 + j + 16] = x1 * d;
        }
    }
}
static void quantize_row_q4_1_ref(const float * x, block_q4_1 * y, int64_t k) {
    const int64_t nb = k / 32;
    for (int64_t i = 0; i < nb; i++) {
        float min = 3.402823466e+38f, max = -3.402823466e+38f;
        for (int j = 0; j < 32; j++) {
            const float v = x[i * 32 + j];
            if (v < min) min = v;
            if (v > max) max = v;
        }
        const float d = (max - min) / ((1 << 4) - 1);
        const float id = d ? 1.0f / d : 0.0f;
        y[i].d = oracle_fp32_to_fp16(d);
        y[i].m = oracle_fp32_to_fp16(min);
        for (int j = 0; j < 16; ++j) {
            const float x0 = (x[i * 32 + 0 + j] - min) * id, x1 = (x[i * 32 + 16 + j] - min) * id;
            const uint8_t xi0 = (uint8_t) ((int8_t) (x0 + 0.5f) < 15 ? (int8_t) (x0 + 0.5f) : 15);
            const uint8_t xi1 = (uint8_t) ((int8_t) (x1 + 0.5f) < 15 ? (int8_t) (x1 + 0.5f) : 15);
            y[i].qs[j] = (uint8_t) (xi0 | (xi1 << 4));
        }
    }
}
static void dequantize_row_q4_1(const block_q4_1 * x, float * y, int64_t k) {
    const int64_t nb = k / 32;
    for (int64_t i = 0; i < nb; i++) {
        const float d = F16(x[i].d), m = F16(x[i].m);
        for (int j = 0; j < 16; ++j) {
            const int x0 = (x[i].qs[j] & 0x0F), x1 = (x[i].qs[j] >> 4);
            y[i * 32 + j] = x0 * d + m;
            y[i * 32 + j + 16] = x1 * d + m;
        }
    }
}
static void quantize_row_q5_0_ref(const float * x, block_q5_0 * y, int64_t k) {
    const int64_t nb = k / 32;
    for (int64_t i = 0; i < nb; i++) {
        float amax = 0.0f, max = 0.0f;
        for (int j = 0; j < 32; j++) {
            const float v = x[i * 32 + j];
            if (amax < fabsf(v)) { amax = fabsf(v); max = v; }
        }
        const float d = max / -16;
        const float id = d ? 1.0f / d : 0.0f;
        y[i].d = oracle_fp32_to_fp16(d);
        uint32_t qh = 0;
        for (int j = 0; j < 16; ++j) {
            const float x0 = x[i * 32 + 0 + j] * id, x1 = x[i * 32 + 16 + j] * id;
            const uint8_t xi0 = (uint8_t) ((int8_t) (x0 + 16.5f) < 31 ? (int8_t) (x0 + 16.5f) : 31); /* MIN(31, (int8_t)(x0 + 16.5f)) */
            const uint8_t xi1 = (uint8_t) ((int8_t) (x1 + 16.5f) < 31 ? (int8_t) (x1 + 16.5f) : 31);
            y[i].qs[j] = (uint8_t) ((xi0 & 0x0F) | ((xi1 & 0x0F) << 4));
            qh |= ((xi0 & 0x10u) >> 4) << (j + 0); /* the fifth bits */
            qh |= ((xi1 & 0x10u) >> 4) << (j + 16);
        }
        memcpy(&y[i].qh, &qh, sizeof(qh));
    }
}
static void dequantize_row_q5_0(const block_q5_0 * x, float * y, int64_t k) {
    const int64_t nb = k / 32;
    for (int64_t i = 0; i < nb; i++) {
        const float d = F16(x[i].d);
        uint32_t qh;
        memcpy(&qh, x[i].qh, sizeof(qh));
        for (int j = 0; j < 16; ++j) {
            const uint8_t xh_0 = ((qh >> (j + 0)) << 4) & 0x10, xh_1 = ((qh >> (j + 12))) & 0x10;
            const int32_t x0 = ((x[i].qs[j] & 0x0F) | xh_0) - 16, x1 = ((x[i].qs[j] >> 4) | xh_1) - 16;
            y[i * 32 + j] = x0 * d;
            y[i * 32 + j + 16] = x1 * d;
        }
    }
}
static void quantize_row_q5_1_ref(const float * x, block_q5_1 * y, int64_t k) {
    const int64_t nb = k / 32;
    for (int64_t i = 0; i < nb; i++) {
        float min = 3.402823466e+38f, max = -3.402823466e+38f;
        for (int j = 0; j < 32; j++) {
            const float v = x[i * 32 + j];
            if (v < min) min = v;
            if (v > max) max = v;
        }
        const float d = (max - min) / ((1 << 5) - 1);
        const float id = d ? 1.0f / d : 0.0f;
        y[i].d = oracle_fp32_to_fp16(d);
        y[i].m = oracle_fp32_to_fp16(min);
        uint32_t qh = 0;
        for (int j = 0; j < 16; ++j) {
            const float x0 = (x[i * 32 + 0 + j] - min) * id, x1 = (x[i * 32 + 16 + j] - min) * id;
            const uint8_t xi0 = (uint8_t) (x0 + 0.5f), xi1 = (uint8_t) (x1 + 0.5f);
            y[i].qs[j] = (uint8_t) ((xi0 & 0x0F) | ((xi1 & 0x0F) << 4));
            qh |= ((xi0 & 0x10u) >> 4) << (j + 0);
            qh |= ((xi1 & 0x10u) >> 4) << (j + 16);
        }
        memcpy(&y[i].qh, &qh, sizeof(qh));
    }
}
static void dequantize_row_q5_1(const block_q5_1 * x, float * y, int64_t k) {
    const int64_t nb = k / 32;
    for (int64_t i = 0; i < nb; i++) {
        const float d = F16(x[i].d), m = F16(x[i].m);
        uint32_t qh;
        memcpy(&qh, x[i].qh, sizeof(qh));
        for (int j = 0; j < 16; ++j) {
            const uint8_t xh_0 = ((qh >> (j + 0)) << 4) & 0x10, xh_1 = ((qh >> (j + 12))) & 0x10;
            const int x0 = (x[i].qs[j] & 0x0F) | xh_0, x1 = (x[i].qs[j] >> 4) | xh_1;
            y[i * 32 + j] = x0 * d + m;
            y[i * 32 + j + 16] = x1 * d + m;
        }
    }
}
/* IQ4_NL: 16 non-linear levels, one f16 scale per 32 values.  quantize_row_iq4_nl_ref -> quantize_row_iq4_nl_impl(super-block 32, block 32, no
   importance weights, ntry = -1 — the run-time form; the offline quantiser searches 15 scales with ntry = 7): levels chosen against max / -127, then
   the least-squares scale for those levels under the weights x^2.  (An all-zero block leaves upstream's level buffer as the previous block left
   it; d = 0 there, so the values are 0 either way — the levels are written as 0 here.) */
static const int8_t kvalues_iq4nl[16] = {-127, -104, -83, -65, -49, -35, -22, -10, 1, 13, 25, 38, 53, 69, 89, 113};
static inline int best_index_int8(int n, const int8_t * val, float x) {
    if (x <= val[0]) return 0;
    if (x >= val[n - 1]) return n - 1;
    int ml = 0, mu = n - 1;
    while (mu - ml > 1) {
        const int mav = (ml + mu) / 2;
        if (x < val[mav]) mu = mav; else ml = mav;
    }
    return x - val[mu - 1] < val[mu] - x ? mu - 1 : mu;
}
static void quantize_row_iq4_nl_ref(const float * x, block_iq4_nl * y, int64_t k) {
    const int64_t nb = k / 32;
    const int8_t * values = kvalues_iq4nl;
    for (int64_t ib = 0; ib < nb; ++ib) {
        const float * xb = x + ib * 32;
        uint8_t L[32];
        memset(L, 0, sizeof(L));
        float amax = 0, max = 0;
        for (int j = 0; j < 32; ++j) {
            const float ax = fabsf(xb[j]);
            if (ax > amax) { amax = ax; max = xb[j]; }
        }
        float scale = 0.0f;
        if (amax >= 1e-15f) { /* GROUP_MAX_EPS */
            float d = max / values[0];
            const float id = 1 / d;
            float sumqx = 0, sumq2 = 0;
            for (int j = 0; j < 32; ++j) {
                const float al = id * xb[j];
                const int l = best_index_int8(16, values, al);
                L[j] = (uint8_t) l;
                const float q = values[l], w = xb[j] * xb[j];
                sumqx += w * q * xb[j];
                sumq2 += w * q * q;
            }
            d = sumqx / sumq2;
            scale = d;
        }
        y[ib].d = oracle_fp32_to_fp16(scale);
        for (int j = 0; j < 16; ++j) y[ib].qs[j] = (uint8_t) (L[j] | (L[16 + j] << 4));
    }
}
static void dequantize_row_iq4_nl(const block_iq4_nl * x, float * y, int64_t k) {
    const int64_t nb = k / 32;
    for (int64_t i = 0; i < nb; i++) {
        const float d = F16(x[i].d);
        for (int j = 0; j < 16; ++j) {
            y[i * 32 + j] = d * kvalues_iq4nl[x[i].qs[j] & 0xf];
            y[i * 32 + j + 16] = d * kvalues_iq4nl[x[i].qs[j] >> 4];
        }
    }
}
/* quantize_row_q8_1_ref: Q8_0 plus s = d * sum(qs), the term the offset formats (Q4_1, Q5_1) multiply their minimum by */
static void quantize_row_q8_1_ref(const float * x, block_q8_1 * y, int64_t k) {
    const int64_t nb = k / 32;
    for (int64_t i = 0; i < nb; i++) {
        float amax = 0.0f;
        for (int j = 0; j < 32; j++) {
            const float v = fabsf(x[i * 32 + j]);
            if (v > amax) amax = v;
        }
        const float d = amax / ((1 << 7) - 1);
        const float id = d ? 1.0f / d : 0.0f;
        y[i].d = oracle_fp32_to_fp16(d);
        int sum = 0;
        for (int j = 0; j < 16; ++j) {
            const float v0 = x[i * 32 + j] * id, v1 = x[i * 32 + 16 + j] * id;
            y[i].qs[j] = (int8_t) roundf(v0);
            y[i].qs[16 + j] = (int8_t) roundf(v1);
            sum += y[i].qs[j];
            sum += y[i].qs[16 + j];
        }
        y[i].s = oracle_fp32_to_fp16(sum * d);
    }
}
/* the generic ggml_vec_dot_{q4_0,q5_0,iq4_nl}_q8_0 and ggml_vec_dot_{q4_1,q5_1}_q8_1: integer sums per block, one f32 product per block */
static float vec_dot_q4_0_q8_0(int64_t n, const block_q4_0 * x, const block_q8_0 * y) {
    float sumf = 0;
    for (int64_t ib = 0; ib < n / 32; ++ib) {
        int sumi0 = 0, sumi1 = 0;
        for (int j = 0; j < 16; ++j) {
            const int v0 = (x[ib].qs[j] & 0x0F) - 8, v1 = (x[ib].qs[j] >> 4) - 8;
            sumi0 += v0 * y[ib].qs[j];
            sumi1 += v1 * y[ib].qs[j + 16];
        }
        sumf += (sumi0 + sumi1) * F16(x[ib].d) * F16(y[ib].d);
    }
    return sumf;
}
static float vec_dot_q4_1_q8_1(int64_t n, const block_q4_1 * x, const block_q8_1 * y) {
    float sumf = 0;
    for (int64_t ib = 0; ib < n / 32; ++ib) {
        int sumi0 = 0, sumi1 = 0;
        for (int j = 0; j < 16; ++j) {
            const int v0 = (x[ib].qs[j] & 0x0F), v1 = (x[ib].qs[j] >> 4);
            sumi0 += v0 * y[ib].qs[j];
            sumi1 += v1 * y[ib].qs[j + 16];
        }
        sumf += (F16(x[ib].d) * F16(y[ib].d)) * (sumi0 + sumi1) + F16(x[ib].m) * F16(y[ib].s);
    }
    return sumf;
}
static float vec_dot_q5_0_q8_0(int64_t n, const block_q5_0 * x, const block_q8_0 * y) {
    float sumf = 0;
    for (int64_t ib = 0; ib < n / 32; ++ib) {
        uint32_t qh;
        memcpy(&qh, x[ib].qh, sizeof(qh));
        int sumi0 = 0, sumi1 = 0;
        for (int j = 0; j < 16; ++j) {
            const uint8_t xh_0 = ((qh & (1u << (j + 0))) >> (j + 0)) << 4, xh_1 = ((qh & (1u << (j + 16))) >> (j + 12));
            const int32_t x0 = (int8_t) (((x[ib].qs[j] & 0x0F) | xh_0) - 16), x1 = (int8_t) (((x[ib].qs[j] >> 4) | xh_1) - 16);
            sumi0 += x0 * y[ib].qs[j];
            sumi1 += x1 * y[ib].qs[j + 16];
        }
        sumf += (F16(x[ib].d) * F16(y[ib].d)) * (sumi0 + sumi1);
    }
    return sumf;
}
static float vec_dot_q5_1_q8_1(int64_t n, const block_q5_1 * x, const block_q8_1 * y) {
    float sumf = 0;
    for (int64_t ib = 0; ib < n / 32; ++ib) {
        uint32_t qh;
        memcpy(&qh, x[ib].qh, sizeof(qh));
        int sumi0 = 0, sumi1 = 0;
        for (int j = 0; j < 16; ++j) {
            const uint8_t xh_0 = ((qh >> (j + 0)) << 4) & 0x10, xh_1 = ((qh >> (j + 12))) & 0x10;
            const int32_t x0 = (x[ib].qs[j] & 0xF) | xh_0, x1 = (x[ib].qs[j] >> 4) | xh_1;
            sumi0 += x0 * y[ib].qs[j];
            sumi1 += x1 * y[ib].qs[j + 16];
        }
        sumf += (F16(x[ib].d) * F16(y[ib].d)) * (sumi0 + sumi1) + F16(x[ib].m) * F16(y[ib].s);
    }
    return sumf;
}
static float vec_dot_iq4_nl_q8_0(int64_t n, const block_iq4_nl * x, const block_q8_0 * y) {
    float sumf = 0;
    for (int64_t ib = 0; ib < n / 32; ++ib) {
        const float d = F16(y[ib].d) * F16(x[ib].d);
        int sumi1 = 0, sumi2 = 0;
        for (int j = 0; j < 16; ++j) {
            sumi1 += y[ib].qs[j + 0] * kvalues_iq4nl[x[ib].qs[j] & 0xf];
            sumi2 += y[ib].qs[j + 16] * kvalues_iq4nl[x[ib].qs[j] >> 4];
        }
        sumf += d * (sumi1 + sumi2);
    }
    return sumf;
}
/* type_traits[type].from_float / type_traits_cpu[type].from_float for the types a cache row may be stored in; 0 when the type has none here */
int oracle_quantize_row(enum ggml_type type, const float * x, void * y, int64_t k) {
    switch (type) {
        case GGML_TYPE_F32: memcpy(y, x, (size_t) k * 4); return 1;
        case GGML_TYPE_F16: for (int64_t i = 0; i < k; ++i) ((ggml_fp16_t *) y)[i] = oracle_fp32_to_fp16(x[i]); return 1;
        case GGML_TYPE_BF16: for (int64_t i = 0; i < k; ++i) ((uint16_t *) y)[i] = fp32_to_bf16(x[i]); return 1;
        case GGML_TYPE_Q8_0: oracle_quantize_row_q8_0(x, (block_q8_0 *) y, k); return 1;
        case GGML_TYPE_Q8_1: quantize_row_q8_1_ref(x, (block_q8_1 *) y, k); return 1;
        case GGML_TYPE_Q4_0: quantize_row_q4_0_ref(x, (block_q4_0 *) y, k); return 1;
        case GGML_TYPE_Q4_1: quantize_row_q4_1_ref(x, (block_q4_1 *) y, k); return 1;
        case GGML_TYPE_Q5_0: quantize_row_q5_0_ref(x, (block_q5_0 *) y, k); return 1;
        case GGML_TYPE_Q5_1: quantize_row_q5_1_ref(x, (block_q5_1 *) y, k); return 1;
        case GGML_TYPE_IQ4_NL: quantize_row_iq4_nl_ref(x, (block_iq4_nl *) y, k); return 1;
        default: return 0;
    }
}
/* one K.q logit as FLASH_ATTN_EXT forms it for a block-format K row (tests: against the float64 golden value): the query through the
   from_float of K's vec_dot_type, then K's vec_dot */
float oracle_kq_dot(enum ggml_type kt, int64_t n, const void * krow, const float * q) {
    block_q8_1 * a = (block_q8_1 *) malloc((size_t) (n / 32 + 1) * sizeof(block_q8_1));
    if (!a) return NAN;
    float s = NAN;
    const int q81 = kt == GGML_TYPE_Q4_1 || kt == GGML_TYPE_Q5_1;
    (void) oracle_quantize_row(q81 ? GGML_TYPE_Q8_1 : GGML_TYPE_Q8_0, q, a, n);
    switch (kt) {
        case GGML_TYPE_Q8_0: s = oracle_vec_dot_q8_0_q8_0(n, (const block_q8_0 *) krow, (const block_q8_0 *) a); break;
        case GGML_TYPE_Q4_0: s = vec_dot_q4_0_q8_0(n, (const block_q4_0 *) krow, (const block_q8_0 *) a); break;
        case GGML_TYPE_Q5_0: s = vec_dot_q5_0_q8_0(n, (const block_q5_0 *) krow, (const block_q8_0 *) a); break;
        case GGML_TYPE_IQ4_NL: s = vec_dot_iq4_nl_q8_0(n, (const block_iq4_nl *) krow, (const block_q8_0 *) a); break;
        case GGML_TYPE_Q4_1: s = vec_dot_q4_1_q8_1(n, (const block_q4_1 *) krow, a); break;
        case GGML_TYPE_Q5_1: s = vec_dot_q5_1_q8_1(n, (const block_q5_1 *) krow, a); break;
        default: break;
    }
    free(a);
    return s;
}
static int is_cache_block_type(enum ggml_type t) {
    return t == GGML_TYPE_Q8_0 || t == GGML_TYPE_Q4_0 || t == GGML_TYPE_Q4_1 || t == GGML_TYPE_Q5_0 || t == GGML_TYPE_Q5_1 || t == GGML_TYPE_IQ4_NL;
}

void oracle_dequantize_row(enum ggml_type type, const void * x, float * y, int64_t k) {
    switch (type) {
        case GGML_TYPE_F32: memcpy(y, x, (size_t) k * 4); break;
        case GGML_TYPE_F16: for (int64_t i = 0; i < k; ++i) y[i] = F16(((const ggml_fp16_t *) x)[i]); break;
        case GGML_TYPE_Q8_0: dequantize_row_q8_0((const block_q8_0 *) x, y, k); break;
        case GGML_TYPE_BF16: for (int64_t i = 0; i < k; ++i) y[i] = bf16_to_fp32(((const uint16_t *) x)[i]); break;
        case GGML_TYPE_Q4_0: dequantize_row_q4_0((const block_q4_0 *) x, y, k); break;
        case GGML_TYPE_Q4_1: dequantize_row_q4_1((const block_q4_1 *) x, y, k); break;
        case GGML_TYPE_Q5_0: dequantize_row_q5_0((const block_q5_0 *) x, y, k); break;
        case GGML_TYPE_Q5_1: dequantize_row_q5_1((const block_q5_1 *) x, y, k); break;
        case GGML_TYPE_IQ4_NL: dequantize_row_iq4_nl((const block_iq4_nl *) x, y, k); break;
        case GGML_TYPE_Q4_K: dequantize_row_q4_K((const block_q4_K *) x, y, k); break;
        case GGML_TYPE_Q5_K: dequantize_row_q5_K((const block_q5_K *) x, y, k); break;
        case GGML_TYPE_Q6_K: dequantize_row_q6_K((const block_q6_K *) x, y, k); break;
        default: abort();
    }
}

/* ------------------------------------------------------------------------------------------ */
/* activation quantisation: quantize_row_q8_0_ref / quantize_row_q8_K_ref (Appendix A.3)        */
/* ------------------------------------------------------------------------------------------ */
void oracle_quantize_row_q8_0(const float * x, block_q8_0 * y, int64_t k) {
    const int64_t nb = k / QK8_0;
    for (int64_t i = 0; i < nb; i++) {
        float amax = 0.0f;
        for (int j = 0; j < QK8_0; j++) {
            const float v = fabsf(x[i * QK8_0 + j]);
            if (v > amax) amax = v;
        }
        const float d = amax / ((1 << 7) - 1);
        const float id = d ? 1.0f / d : 0.0f;
        y[i].d = oracle_fp32_to_fp16(d);
        for (int j = 0; j < QK8_0; ++j) {
            const float x0 = x[i * QK8_0 + j] * id;
            y[i].qs[j] = (int8_t) roundf(x0);
        }
    }
}

/* round-half-even through the float magic-number add (ggml's nearest_int) */
static inline int nearest_int(float fval) {
    float val = fval + 12582912.f;
    int i;
    memcpy(&i, &val, sizeof(int));
    return (i & 0x007fffff) - 0x00400000;
}

void oracle_quantize_row_q8_K(const float * x, block_q8_K * y, int64_t k) {
    const int64_t nb = k / QK_K;
    for (int64_t i = 0; i < nb; i++) {
        float max = 0;
        float amax = 0;
        for (int j = 0; j < QK_K; ++j) {
            float ax = fabsf(x[j]);
            if (ax > amax) {
                amax = ax;
                max = x[j];
            }
        }
        if (!amax) {
            y[i].d = 0;
            memset(y[i].qs, 0, QK_K);
            memset(y[i].bsums, 0, sizeof(y[i].bsums)); /* upstream leaves these unset; they are multiplied by d=0 */
            x += QK_K;
            continue;
        }
        const float iscale = -127.f / max;
        for (int j = 0; j < QK_K; ++j) {
            int v = nearest_int(iscale * x[j]);
            y[i].qs[j] = (int8_t) (v < 127 ? v : 127);
        }
        for (int j = 0; j < QK_K / 16; ++j) {
            int sum = 0;
            for (int ii = 0; ii < 16; ++ii) sum += y[i].qs[j * 16 + ii];
            y[i].bsums[j] = (int16_t) sum;
        }
        y[i].d = 1 / iscale;
        x += QK_K;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* block dot products: ggml_vec_dot_*_generic (ggml-cpu/quants.c; Appendix A.3)                 */
/* ------------------------------------------------------------------------------------------ */
float oracle_vec_dot_q8_0_q8_0(int64_t n, const block_q8_0 * x, const block_q8_0 * y) {
    const int64_t nb = n / QK8_0;
    float sumf = 0;
    for (int64_t ib_ = 0; ib_ < nb; ++ib_) {
        const int64_t ib = BLK_IDX(ib_, nb);
        int sumi = 0;
        for (int j = 0; j < QK8_0; j++) sumi += x[ib].qs[j] * y[ib].qs[j];
        sumf += sumi * (F16(x[ib].d) * F16(y[ib].d));
    }
    return sumf;
}

static float vec_dot_k45(int64_t n, const uint8_t * xb, size_t xstride, int is5, const block_q8_K * y) {
    const int64_t nb = n / QK_K;
    int8_t aux8[QK_K];
    int16_t aux16[8];
    float sums[8];
    int32_t aux32[8];
    memset(sums, 0, sizeof(sums));
    float sumf = 0;
    for (int64_t i_ = 0; i_ < nb; ++i_) {
        const int64_t i = BLK_IDX(i_, nb);
        const uint8_t * blk = xb + (size_t) i * xstride;
        ggml_fp16_t hd, hdmin;
        memcpy(&hd, blk, 2);
        memcpy(&hdmin, blk + 2, 2);
        const uint8_t * scales12 = blk + 4;
        const uint8_t * hm = is5 ? blk + 16 : NULL;
        const uint8_t * q4 = is5 ? blk + 48 : blk + 16;
        const int8_t * q8 = y[i].qs;
        memset(aux32, 0, sizeof(aux32));
        int8_t * a = aux8;
        uint8_t m = 1;
        for (int j = 0; j < QK_K / 64; ++j) {
            for (int l = 0; l < 32; ++l) a[l] = (int8_t) ((q4[l] & 0xF) + ((is5 && (hm[l] & m)) ? 16 : 0));
            a += 32;
            m <<= 1;
            for (int l = 0; l < 32; ++l) a[l] = (int8_t) ((q4[l] >> 4) + ((is5 && (hm[l] & m)) ? 16 : 0));
            a += 32;
            m <<= 1;
            q4 += 32;
        }
        uint8_t scales[8], mins[8];
        for (int j = 0; j < 8; ++j) get_scale_min_k4(j, scales12, &scales[j], &mins[j]);
        int sumi = 0;
        for (int j = 0; j < QK_K / 16; ++j) sumi += y[i].bsums[j] * mins[j / 2];
        a = aux8;
        int is = 0;
        for (int j = 0; j < QK_K / 32; ++j) {
            int32_t scale = scales[is++];
            for (int g = 0; g < 4; ++g) {
                for (int l = 0; l < 8; ++l) aux16[l] = (int16_t) (q8[l] * a[l]);
                for (int l = 0; l < 8; ++l) aux32[l] += scale * aux16[l];
                q8 += 8;
                a += 8;
            }
        }
        const float d = F16(hd) * y[i].d;
        for (int l = 0; l < 8; ++l) sums[l] += d * aux32[l];
        const float dmin = F16(hdmin) * y[i].d;
        sumf -= dmin * sumi;
    }
    for (int l = 0; l < 8; ++l) sumf += sums[l];
    return sumf;
}

float oracle_vec_dot_q4_K_q8_K(int64_t n, const block_q4_K * x, const block_q8_K * y) {
    return vec_dot_k45(n, (const uint8_t *) x, sizeof(block_q4_K), 0, y);
}
float oracle_vec_dot_q5_K_q8_K(int64_t n, const block_q5_K * x, const block_q8_K * y) {
    return vec_dot_k45(n, (const uint8_t *) x, sizeof(block_q5_K), 1, y);
}

float oracle_vec_dot_q6_K_q8_K(int64_t n, const block_q6_K * x, const block_q8_K * y) {
    const int64_t nb = n / QK_K;
    int8_t aux8[QK_K];
    int16_t aux16[8];
    float sums[8];
    int32_t aux32[8];
    memset(sums, 0, sizeof(sums));
    float sumf = 0;
    for (int64_t i_ = 0; i_ < nb; ++i_) {
        const int64_t i = BLK_IDX(i_, nb);
        const uint8_t * q4 = x[i].ql;
        const uint8_t * qh = x[i].qh;
        const int8_t * q8 = y[i].qs;
        memset(aux32, 0, sizeof(aux32));
        int8_t * a = aux8;
        for (int j = 0; j < QK_K; j += 128) {
            for (int l = 0; l < 32; ++l) {
                a[l + 0] = (int8_t) ((q4[l + 0] & 0xF) | (((qh[l] >> 0) & 3) << 4)) - 32;
                a[l + 32] = (int8_t) ((q4[l + 32] & 0xF) | (((qh[l] >> 2) & 3) << 4)) - 32;
                a[l + 64] = (int8_t) ((q4[l + 0] >> 4) | (((qh[l] >> 4) & 3) << 4)) - 32;
                a[l + 96] = (int8_t) ((q4[l + 32] >> 4) | (((qh[l] >> 6) & 3) << 4)) - 32;
            }
            a += 128;
            q4 += 64;
            qh += 32;
        }
        a = aux8;
        int is = 0;
        for (int j = 0; j < QK_K / 16; ++j) {
            int scale = x[i].scales[is++];
            for (int g = 0; g < 2; ++g) {
                for (int l = 0; l < 8; ++l) aux16[l] = (int16_t) (q8[l] * a[l]);
                for (int l = 0; l < 8; ++l) aux32[l] += scale * aux16[l];
                q8 += 8;
                a += 8;
            }
        }
        const float d = F16(x[i].d) * y[i].d;
        for (int l = 0; l < 8; ++l) sums[l] += d * aux32[l];
    }
    for (int l = 0; l < 8; ++l) sumf += sums[l];
    return sumf;
}

/* ------------------------------------------------------------------------------------------ */
/* FAST block dots — bench.py's cpu_baseline leg ONLY (oracle_set_fast(1)); never the checker.  */
/* The generic routines above are what ggml-cpu computes on a machine without SIMD; a host of   */
/* llama-box runs ggml-cpu's x86 kernels (arch/x86/quants.c), whose published AVX2 algorithm is  */
/* restated here so that the CPU number beside the GPU's is a credible one: vpmaddubsw/vpmaddwd  */
/* integer dots (the same integers: tests check the sums bit for bit on unit scales), eight       */
/* float lanes accumulated with FMA and summed at the end (a different f32 order than generic).   */
/* ------------------------------------------------------------------------------------------ */
#if defined(__AVX2__) && defined(__FMA__)
#include <immintrin.h>
static inline float hsum8(__m256 v) {
    __m128 r = _mm_add_ps(_mm256_castps256_ps128(v), _mm256_extractf128_ps(v, 1));
    r = _mm_add_ps(r, _mm_movehl_ps(r, r));
    r = _mm_add_ss(r, _mm_movehdup_ps(r));
    return _mm_cvtss_f32(r);
}
static inline __m256i bcast16(int i) { return _mm256_set1_epi16((short) (((2 * i + 1) << 8) | (2 * i))); } /* pshufb mask: word i everywhere */
static float fast_vec_dot_k45(int64_t n, const uint8_t * xb, size_t xstride, int is5, const block_q8_K * y) {
    const int64_t nb = n / QK_K;
    const __m256i m4 = _mm256_set1_epi8(0xF);
    __m256 acc = _mm256_setzero_ps();
    __m128 acc_m = _mm_setzero_ps();
    for (int64_t i = 0; i < nb; ++i) {
        const uint8_t * blk = xb + (size_t) i * xstride;
        ggml_fp16_t hd, hdmin;
        memcpy(&hd, blk, 2);
        memcpy(&hdmin, blk + 2, 2);
        const float d = y[i].d * F16(hd), dmin = -y[i].d * F16(hdmin);
        uint32_t ut[4];
        memcpy(ut, blk + 4, 12);
        ut[3] = ((ut[2] >> 4) & 0x0f0f0f0fu) | (((ut[1] >> 6) & 0x03030303u) << 4);
        const uint32_t uaux = ut[1] & 0x3f3f3f3fu;
        ut[1] = (ut[2] & 0x0f0f0f0fu) | (((ut[0] >> 6) & 0x03030303u) << 4);
        ut[2] = uaux;
        ut[0] &= 0x3f3f3f3fu; /* bytes 0..7: the eight scales, 8..15: the eight mins */
        const __m256i ms = _mm256_cvtepu8_epi16(_mm_set_epi32((int) ut[3], (int) ut[2], (int) ut[1], (int) ut[0]));
        const __m256i q8sums = _mm256_loadu_si256((const __m256i *) y[i].bsums);
        const __m128i q8s = _mm_hadd_epi16(_mm256_extracti128_si256(q8sums, 0), _mm256_extracti128_si256(q8sums, 1));
        const __m128i prod = _mm_madd_epi16(_mm256_extracti128_si256(ms, 1), q8s);
        acc_m = _mm_fmadd_ps(_mm_set1_ps(dmin), _mm_cvtepi32_ps(prod), acc_m);
        const __m128i sc128 = _mm256_extracti128_si256(ms, 0);
        const __m256i scales = _mm256_set_m128i(sc128, sc128);
        const uint8_t * q4 = is5 ? blk + 48 : blk + 16;
        const __m256i hbits = is5 ? _mm256_loadu_si256((const __m256i *) (blk + 16)) : _mm256_setzero_si256();
        const int8_t * q8 = y[i].qs;
        __m256i sumi = _mm256_setzero_si256();
        for (int j = 0; j < QK_K / 64; ++j) {
            const __m256i scale_l = _mm256_shuffle_epi8(scales, bcast16(2 * j)), scale_h = _mm256_shuffle_epi8(scales, bcast16(2 * j + 1));
            const __m256i q4bits = _mm256_loadu_si256((const __m256i *) q4);
            q4 += 32;
            __m256i ql = _mm256_and_si256(q4bits, m4), qh = _mm256_and_si256(_mm256_srli_epi16(q4bits, 4), m4);
            if (is5) {
                const __m256i one = _mm256_set1_epi8(1);
                ql = _mm256_or_si256(ql, _mm256_slli_epi16(_mm256_and_si256(_mm256_srli_epi16(hbits, 2 * j), one), 4));
                qh = _mm256_or_si256(qh, _mm256_slli_epi16(_mm256_and_si256(_mm256_srli_epi16(hbits, 2 * j + 1), one), 4));
            }
            const __m256i q8l = _mm256_loadu_si256((const __m256i *) q8), q8h = _mm256_loadu_si256((const __m256i *) (q8 + 32));
            q8 += 64;
            const __m256i pl = _mm256_madd_epi16(scale_l, _mm256_maddubs_epi16(ql, q8l));
            const __m256i ph = _mm256_madd_epi16(scale_h, _mm256_maddubs_epi16(qh, q8h));
            sumi = _mm256_add_epi32(sumi, _mm256_add_epi32(pl, ph));
        }
        acc = _mm256_fmadd_ps(_mm256_set1_ps(d), _mm256_cvtepi32_ps(sumi), acc);
    }
    acc_m = _mm_add_ps(acc_m, _mm_movehl_ps(acc_m, acc_m));
    acc_m = _mm_add_ss(acc_m, _mm_movehdup_ps(acc_m));
    return hsum8(acc) + _mm_cvtss_f32(acc_m);
}
static float fast_vec_dot_q6_K_q8_K(int64_t n, const block_q6_K * x, const block_q8_K * y) {
    const int64_t nb = n / QK_K;
    const __m256i m4 = _mm256_set1_epi8(0xF), m2 = _mm256_set1_epi8(3);
    __m256 acc = _mm256_setzero_ps();
    for (int64_t i = 0; i < nb; ++i) {
        const float d = y[i].d * F16(x[i].d);
        const uint8_t * q4 = x[i].ql;
        const uint8_t * qh = x[i].qh;
        const int8_t * q8 = y[i].qs;
        /* sum (q - 32) y = sum q y - 32 sum y: the second term from the 16-value sums of the activation block */
        const __m256i sc16 = _mm256_cvtepi8_epi16(_mm_loadu_si128((const __m128i *) x[i].scales));
        const __m256i off = _mm256_madd_epi16(sc16, _mm256_loadu_si256((const __m256i *) y[i].bsums));
        __m256i sumi = _mm256_setzero_si256();
        for (int j = 0; j < QK_K / 128; ++j) {
            const __m256i b1 = _mm256_loadu_si256((const __m256i *) q4), b2 = _mm256_loadu_si256((const __m256i *) (q4 + 32)), bh = _mm256_loadu_si256((const __m256i *) qh);
            q4 += 64;
            qh += 32;
            const __m256i q[4] = {
                _mm256_or_si256(_mm256_and_si256(b1, m4), _mm256_slli_epi16(_mm256_and_si256(bh, m2), 4)),
                _mm256_or_si256(_mm256_and_si256(b2, m4), _mm256_slli_epi16(_mm256_and_si256(_mm256_srli_epi16(bh, 2), m2), 4)),
                _mm256_or_si256(_mm256_and_si256(_mm256_srli_epi16(b1, 4), m4), _mm256_slli_epi16(_mm256_and_si256(_mm256_srli_epi16(bh, 4), m2), 4)),
                _mm256_or_si256(_mm256_and_si256(_mm256_srli_epi16(b2, 4), m4), _mm256_slli_epi16(_mm256_and_si256(_mm256_srli_epi16(bh, 6), m2), 4)),
            };
            for (int k = 0; k < 4; ++k) {
                /* 32 values: scale 8j + 2k for the first 16, 8j + 2k + 1 for the next 16 (words 0..7 / 8..15 after vpmaddubsw) */
                const int is = 8 * j + 2 * k;
                const __m256i sc = _mm256_set_m128i(_mm_set1_epi16(x[i].scales[is + 1]), _mm_set1_epi16(x[i].scales[is]));
                const __m256i p = _mm256_maddubs_epi16(q[k], _mm256_loadu_si256((const __m256i *) q8));
                q8 += 32;
                sumi = _mm256_add_epi32(sumi, _mm256_madd_epi16(sc, p));
            }
        }
        sumi = _mm256_sub_epi32(sumi, _mm256_slli_epi32(off, 5));
        acc = _mm256_fmadd_ps(_mm256_set1_ps(d), _mm256_cvtepi32_ps(sumi), acc);
    }
    return hsum8(acc);
}
static float fast_vec_dot_q8_0_q8_0(int64_t n, const block_q8_0 * x, const block_q8_0 * y) {
    const int64_t nb = n / QK8_0;
    __m256 acc = _mm256_setzero_ps();
    const __m256i ones = _mm256_set1_epi16(1);
    for (int64_t i = 0; i < nb; ++i) {
        const __m256i qx = _mm256_loadu_si256((const __m256i *) x[i].qs), qy = _mm256_loadu_si256((const __m256i *) y[i].qs);
        const __m256i p = _mm256_madd_epi16(ones, _mm256_maddubs_epi16(_mm256_sign_epi8(qx, qx), _mm256_sign_epi8(qy, qx)));
        acc = _mm256_fmadd_ps(_mm256_set1_ps(F16(x[i].d) * F16(y[i].d)), _mm256_cvtepi32_ps(p), acc);
    }
    return hsum8(acc);
}
#define ORACLE_HAVE_FAST 1
#else
#define ORACLE_HAVE_FAST 0
#endif
static int g_fast = 0;
int oracle_set_fast(int on) { g_fast = (on && ORACLE_HAVE_FAST) ? 1 : 0; return g_fast; }
float oracle_fast_vec_dot(enum ggml_type type, int64_t n, const void * x, const void * y) {
#if ORACLE_HAVE_FAST
    switch (type) {
        case GGML_TYPE_Q4_K: return fast_vec_dot_k45(n, (const uint8_t *) x, sizeof(block_q4_K), 0, (const block_q8_K *) y);
        case GGML_TYPE_Q5_K: return fast_vec_dot_k45(n, (const uint8_t *) x, sizeof(block_q5_K), 1, (const block_q8_K *) y);
        case GGML_TYPE_Q6_K: return fast_vec_dot_q6_K_q8_K(n, (const block_q6_K *) x, (const block_q8_K *) y);
        case GGML_TYPE_Q8_0: return fast_vec_dot_q8_0_q8_0(n, (const block_q8_0 *) x, (const block_q8_0 *) y);
        default: break;
    }
#endif
    return NAN;
}

/* ------------------------------------------------------------------------------------------ */
/* helpers                                                                                      */
/* ------------------------------------------------------------------------------------------ */
#define TDATA(t) ((char *) (t)->data)

static inline float op_f32(const struct ggml_tensor * t, int i) {
    float v;
    memcpy(&v, &t->op_params[i], 4);
    return v;
}

static int g_max_threads = 0;
int oracle_max_threads(void) {
#ifdef _OPENMP
    if (!g_max_threads) g_max_threads = omp_get_max_threads();
    return g_max_threads;
#else
    return 1;
#endif
}

static enum ggml_type vec_dot_type(enum ggml_type t) {
    switch (t) {
        case GGML_TYPE_F32: return GGML_TYPE_F32;
        case GGML_TYPE_F16: return GGML_TYPE_F16;
        case GGML_TYPE_Q8_0: return GGML_TYPE_Q8_0;
        case GGML_TYPE_Q4_K: case GGML_TYPE_Q5_K: case GGML_TYPE_Q6_K: return GGML_TYPE_Q8_K;
        case GGML_TYPE_BF16: return GGML_TYPE_BF16;
        case GGML_TYPE_Q4_0: case GGML_TYPE_Q5_0: case GGML_TYPE_IQ4_NL: return GGML_TYPE_Q8_0; /* (a K cache in these types, -fa off: K.q is a MUL_MAT over cache blocks) */
        case GGML_TYPE_Q4_1: case GGML_TYPE_Q5_1: return GGML_TYPE_Q8_1;
        default: return GGML_TYPE_COUNT;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* MUL_MAT: ggml_compute_forward_mul_mat (ggml-cpu.c) — src1 rows converted to vec_dot_type,     */
/* then one vec_dot per (src0 row, src1 row); dims 2/3 of src0 broadcast over src1.              */
/* ------------------------------------------------------------------------------------------ */
static enum ggml_status op_mul_mat(struct ggml_tensor * dst, int nth) {
    const struct ggml_tensor * src0 = dst->src[0];
    const struct ggml_tensor * src1 = dst->src[1];
    const int64_t ne00 = src0->ne[0], ne01 = src0->ne[1], ne02 = src0->ne[2], ne03 = src0->ne[3];
    const int64_t ne10 = src1->ne[0], ne11 = src1->ne[1], ne12 = src1->ne[2], ne13 = src1->ne[3];
    if (ne00 != ne10 || src1->type != GGML_TYPE_F32 || dst->type != GGML_TYPE_F32) return GGML_STATUS_FAILED;
    const enum ggml_type vdt = vec_dot_type(src0->type);
    if (vdt == GGML_TYPE_COUNT) return GGML_STATUS_FAILED;
    const size_t row_size = ggml_abi_row_size(vdt, ne10);
    const int64_t nrows1 = ne11 * ne12 * ne13;
    char * wdata = (char *) malloc(row_size * (size_t) nrows1 + 64);
    if (!wdata) return GGML_STATUS_ALLOC_FAILED;

    /* convert src1 to vec_dot_type, one contiguous row per (i11,i12,i13) */
#pragma omp parallel for num_threads(nth) schedule(static)
    for (int64_t r = 0; r < nrows1; ++r) {
        const int64_t i11 = r % ne11, i12 = (r / ne11) % ne12, i13 = r / (ne11 * ne12);
        const char * s = TDATA(src1) + i11 * src1->nb[1] + i12 * src1->nb[2] + i13 * src1->nb[3];
        char * w = wdata + (size_t) r * row_size;
        if (src1->nb[0] != sizeof(float)) abort();
        switch (vdt) {
            case GGML_TYPE_F32: memcpy(w, s, (size_t) ne10 * 4); break;
            case GGML_TYPE_F16:
                for (int64_t i = 0; i < ne10; ++i) ((ggml_fp16_t *) w)[i] = oracle_fp32_to_fp16(((const float *) s)[i]);
                break;
            case GGML_TYPE_Q8_0: oracle_quantize_row_q8_0((const float *) s, (block_q8_0 *) w, ne10); break;
            case GGML_TYPE_Q8_K: oracle_quantize_row_q8_K((const float *) s, (block_q8_K *) w, ne10); break;
            case GGML_TYPE_Q8_1: case GGML_TYPE_BF16: (void) oracle_quantize_row(vdt, (const float *) s, w, ne10); break;
            default: abort();
        }
    }

    const int64_t r2 = ne12 / ne02, r3 = ne13 / ne03;
    const int64_t total = ne01 * nrows1;
#pragma omp parallel for num_threads(nth) schedule(static)
    for (int64_t idx = 0; idx < total; ++idx) {
        /* src0 row fastest so that consecutive iterations stream the weight matrix */
        const int64_t i01 = idx % ne01;
        const int64_t r = idx / ne01;
        const int64_t i11 = r % ne11, i12 = (r / ne11) % ne12, i13 = r / (ne11 * ne12);
        const int64_t i02 = i12 / r2, i03 = i13 / r3;
        const char * a = TDATA(src0) + i01 * src0->nb[1] + i02 * src0->nb[2] + i03 * src0->nb[3];
        const char * w = wdata + (size_t) r * row_size;
        float * d = (float *) (TDATA(dst) + i01 * dst->nb[0] + i11 * dst->nb[1] + i12 * dst->nb[2] + i13 * dst->nb[3]);
        switch (src0->type) {
            case GGML_TYPE_F32: { /* ggml_vec_dot_f32 generic: double accumulation of float products */
                ggml_float s = 0.0;
                const float * x = (const float *) a;
                const float * y = (const float *) w;
                for (int64_t i = 0; i < ne00; ++i) s += (ggml_float) (x[i] * y[i]);
                *d = (float) s;
            } break;
            case GGML_TYPE_F16: { /* ggml_vec_dot_f16 generic */
                ggml_float s = 0.0;
                const ggml_fp16_t * x = (const ggml_fp16_t *) a;
                const ggml_fp16_t * y = (const ggml_fp16_t *) w;
                for (int64_t i = 0; i < ne00; ++i) s += (ggml_float) (F16(x[i]) * F16(y[i]));
                *d = (float) s;
            } break;
            case GGML_TYPE_Q8_0: case GGML_TYPE_Q4_K: case GGML_TYPE_Q5_K: case GGML_TYPE_Q6_K:
                if (g_fast) { *d = oracle_fast_vec_dot(src0->type, ne00, a, w); break; }  /* (cpu_baseline timing leg only) */
                if (src0->type == GGML_TYPE_Q8_0) *d = oracle_vec_dot_q8_0_q8_0(ne00, (const block_q8_0 *) a, (const block_q8_0 *) w);
                else if (src0->type == GGML_TYPE_Q4_K) *d = oracle_vec_dot_q4_K_q8_K(ne00, (const block_q4_K *) a, (const block_q8_K *) w);
                else if (src0->type == GGML_TYPE_Q5_K) *d = oracle_vec_dot_q5_K_q8_K(ne00, (const block_q5_K *) a, (const block_q8_K *) w);
                else *d = oracle_vec_dot_q6_K_q8_K(ne00, (const block_q6_K *) a, (const block_q8_K *) w);
                break;
            case GGML_TYPE_Q4_0: *d = vec_dot_q4_0_q8_0(ne00, (const block_q4_0 *) a, (const block_q8_0 *) w); break;
            case GGML_TYPE_Q5_0: *d = vec_dot_q5_0_q8_0(ne00, (const block_q5_0 *) a, (const block_q8_0 *) w); break;
            case GGML_TYPE_IQ4_NL: *d = vec_dot_iq4_nl_q8_0(ne00, (const block_iq4_nl *) a, (const block_q8_0 *) w); break;
            case GGML_TYPE_Q4_1: *d = vec_dot_q4_1_q8_1(ne00, (const block_q4_1 *) a, (const block_q8_1 *) w); break;
            case GGML_TYPE_Q5_1: *d = vec_dot_q5_1_q8_1(ne00, (const block_q5_1 *) a, (const block_q8_1 *) w); break;
            case GGML_TYPE_BF16: { /* ggml_vec_dot_bf16 generic */
                ggml_float s = 0.0;
                for (int64_t i = 0; i < ne00; ++i) s += (ggml_float) (bf16_to_fp32(((const uint16_t *) a)[i]) * bf16_to_fp32(((const uint16_t *) w)[i]));
                *d = (float) s;
            } break;
            default: abort();
        }
    }
    free(wdata);
    return GGML_STATUS_SUCCESS;
}

/* ------------------------------------------------------------------------------------------ */
/* element-wise binary ops with ggml broadcasting (ggml_compute_forward_{add,sub,mul,div}_f32)  */
/* ------------------------------------------------------------------------------------------ */
static enum ggml_status op_binary(struct ggml_tensor * dst, int nth) {
    const struct ggml_tensor * a = dst->src[0];
    const struct ggml_tensor * b = dst->src[1];
    if (a->type != GGML_TYPE_F32 || b->type != GGML_TYPE_F32 || dst->type != GGML_TYPE_F32) return GGML_STATUS_FAILED;
    const int64_t nr = ggml_abi_nrows(a);
    const enum ggml_op op = dst->op;
#pragma omp parallel for num_threads(nth) schedule(static)
    for (int64_t ir = 0; ir < nr; ++ir) {
        const int64_t i01 = ir % a->ne[1], i02 = (ir / a->ne[1]) % a->ne[2], i03 = ir / (a->ne[1] * a->ne[2]);
        const int64_t i11 = i01 % b->ne[1], i12 = i02 % b->ne[2], i13 = i03 % b->ne[3];
        const char * pa = TDATA(a) + i01 * a->nb[1] + i02 * a->nb[2] + i03 * a->nb[3];
        const char * pb = TDATA(b) + i11 * b->nb[1] + i12 * b->nb[2] + i13 * b->nb[3];
        char * pd = TDATA(dst) + i01 * dst->nb[1] + i02 * dst->nb[2] + i03 * dst->nb[3];
        for (int64_t i0 = 0; i0 < a->ne[0]; ++i0) {
            const float x = *(const float *) (pa + i0 * a->nb[0]);
            const float y = *(const float *) (pb + (i0 % b->ne[0]) * b->nb[0]);
            float r;
            switch (op) {
                case GGML_OP_ADD: r = x + y; break;
                case GGML_OP_SUB: r = x - y; break;
                case GGML_OP_MUL: r = x * y; break;
                default: r = x / y; break;
            }
            *(float *) (pd + i0 * dst->nb[0]) = r;
        }
    }
    return GGML_STATUS_SUCCESS;
}

/* SCALE: dst = scale*x + bias (llama-box/patches/llama.cpp/ggml-cuda.patch:8-15) */
static enum ggml_status op_scale(struct ggml_tensor * dst) {
    const struct ggml_tensor * a = dst->src[0];
    if (a->type != GGML_TYPE_F32 || !ggml_abi_is_contiguous(a) || !ggml_abi_is_contiguous(dst)) return GGML_STATUS_FAILED;
    const float s = op_f32(dst, 0), b = op_f32(dst, 1);
    const int64_t n = ggml_abi_nelements(a);
    const float * x = (const float *) a->data;
    float * y = (float *) dst->data;
    for (int64_t i = 0; i < n; ++i) {
        const float p = x[i] * s;
        y[i] = p + b;
    }
    return GGML_STATUS_SUCCESS;
}

/* RMS_NORM: ggml_compute_forward_rms_norm_f32 — sum of squares in double, scale = 1/sqrtf(mean+eps) */
static enum ggml_status op_rms_norm(struct ggml_tensor * dst, int nth) {
    const struct ggml_tensor * a = dst->src[0];
    if (a->type != GGML_TYPE_F32 || a->nb[0] != 4 || dst->nb[0] != 4) return GGML_STATUS_FAILED;
    const float eps = op_f32(dst, 0);
    const int64_t ne00 = a->ne[0];
    const int64_t nr = ggml_abi_nrows(a);
#pragma omp parallel for num_threads(nth) schedule(static)
    for (int64_t ir = 0; ir < nr; ++ir) {
        const int64_t i01 = ir % a->ne[1], i02 = (ir / a->ne[1]) % a->ne[2], i03 = ir / (a->ne[1] * a->ne[2]);
        const float * x = (const float *) (TDATA(a) + i01 * a->nb[1] + i02 * a->nb[2] + i03 * a->nb[3]);
        float * y = (float *) (TDATA(dst) + i01 * dst->nb[1] + i02 * dst->nb[2] + i03 * dst->nb[3]);
        ggml_float sum = 0.0;
        if (g_variant == 2) {
            float sumf = 0.0f;
            for (int64_t i = 0; i < ne00; ++i) sumf += x[i] * x[i];
            sum = sumf;
        } else
        for (int64_t i = 0; i < ne00; ++i) sum += (ggml_float) (x[i] * x[i]);
        const float mean = (float) (sum / ne00);
        const float scale = 1.0f / sqrtf(mean + eps);
        for (int64_t i = 0; i < ne00; ++i) y[i] = x[i] * scale;
    }
    return GGML_STATUS_SUCCESS;
}

/* UNARY: silu(x) = x/(1+oracle_expf(-x)) etc. (ggml-cpu/vec.h scalar forms) */
static inline float silu_f32(float x) { return x / (1.0f + oracle_expf(-x)); }
static enum ggml_status op_unary(struct ggml_tensor * dst) {
    const struct ggml_tensor * a = dst->src[0];
    if (a->type != GGML_TYPE_F32 || !ggml_abi_is_contiguous(a) || !ggml_abi_is_contiguous(dst)) return GGML_STATUS_FAILED;
    const int64_t n = ggml_abi_nelements(a);
    const float * x = (const float *) a->data;
    float * y = (float *) dst->data;
    const int uop = dst->op_params[0];
    for (int64_t i = 0; i < n; ++i) {
        switch (uop) {
            case GGML_UNARY_OP_SILU: y[i] = silu_f32(x[i]); break;
            case GGML_UNARY_OP_RELU: y[i] = x[i] > 0.f ? x[i] : 0.f; break;
            case GGML_UNARY_OP_NEG: y[i] = -x[i]; break;
            case GGML_UNARY_OP_EXP: y[i] = oracle_expf(x[i]); break;
            case GGML_UNARY_OP_TANH: y[i] = tanhf(x[i]); break;
            case GGML_UNARY_OP_SIGMOID: y[i] = 1.f / (1.f + oracle_expf(-x[i])); break;
            default: return GGML_STATUS_FAILED;
        }
    }
    return GGML_STATUS_SUCCESS;
}

/* GLU (SWIGLU): ggml_compute_forward_swiglu_f32 — y = silu(x) * g; split (src1) or fused halves */
static enum ggml_status op_glu(struct ggml_tensor * dst) {
    const struct ggml_tensor * a = dst->src[0];
    const struct ggml_tensor * b = dst->src[1];
    if (dst->op_params[0] != GGML_GLU_OP_SWIGLU || a->type != GGML_TYPE_F32) return GGML_STATUS_FAILED;
    const int swapped = dst->op_params[1];
    const int64_t nc = b ? a->ne[0] : a->ne[0] / 2;
    const int64_t nr = ggml_abi_nrows(a);
    if (dst->ne[0] != nc) return GGML_STATUS_FAILED;
    for (int64_t ir = 0; ir < nr; ++ir) {
        const float * pa = (const float *) (TDATA(a) + ir * a->nb[1]);
        const float * pb = b ? (const float *) (TDATA(b) + ir * b->nb[1]) : pa;
        if (!b) {
            pa += swapped ? nc : 0;
            pb += swapped ? 0 : nc;
        }
        float * y = (float *) (TDATA(dst) + ir * dst->nb[1]);
        for (int64_t i = 0; i < nc; ++i) y[i] = silu_f32(pa[i]) * pb[i];
    }
    return GGML_STATUS_SUCCESS;
}

/* GET_ROWS: ggml_compute_forward_get_rows — dst row (i10,i11,i12) = dequant(src0 row (idx,i11,i12)) */
static enum ggml_status op_get_rows(struct ggml_tensor * dst) {
    const struct ggml_tensor * a = dst->src[0];
    const struct ggml_tensor * idx = dst->src[1];
    if (idx->type != GGML_TYPE_I32 || dst->type != GGML_TYPE_F32) return GGML_STATUS_FAILED;
    const int64_t nc = a->ne[0];
    for (int64_t i12 = 0; i12 < idx->ne[2]; ++i12)
        for (int64_t i11 = 0; i11 < idx->ne[1]; ++i11)
            for (int64_t i10 = 0; i10 < idx->ne[0]; ++i10) {
                const int32_t i01 = *(const int32_t *) (TDATA(idx) + i10 * idx->nb[0] + i11 * idx->nb[1] + i12 * idx->nb[2]);
                if (i01 < 0 || i01 >= a->ne[1]) return GGML_STATUS_FAILED;
                oracle_dequantize_row(a->type, TDATA(a) + i01 * a->nb[1] + i11 * a->nb[2] + i12 * a->nb[3],
                                      (float *) (TDATA(dst) + i10 * dst->nb[1] + i11 * dst->nb[2] + i12 * dst->nb[3]), nc);
            }
    return GGML_STATUS_SUCCESS;
}

/* SET_ROWS: ggml_compute_forward_set_rows_f32 — dst row idx[i] = convert(src0 row i), I64 indices */
static enum ggml_status op_set_rows(struct ggml_tensor * dst) {
    const struct ggml_tensor * a = dst->src[0];
    const struct ggml_tensor * idx = dst->src[1];
    if (a->type != GGML_TYPE_F32 || idx->type != GGML_TYPE_I64) return GGML_STATUS_FAILED;
    if (dst->type != GGML_TYPE_F32 && dst->type != GGML_TYPE_F16 && dst->type != GGML_TYPE_BF16 && !is_cache_block_type(dst->type)) return GGML_STATUS_FAILED;
    const int64_t nc = a->ne[0];
    if (is_cache_block_type(dst->type) && (nc % 32) != 0) return GGML_STATUS_FAILED;
    for (int64_t i03 = 0; i03 < a->ne[3]; ++i03)
        for (int64_t i02 = 0; i02 < a->ne[2]; ++i02)
            for (int64_t i01 = 0; i01 < a->ne[1]; ++i01) {
                const int64_t i12 = i03 % idx->ne[2], i11 = i02 % idx->ne[1], i10 = i01;
                const int64_t i1 = *(const int64_t *) (TDATA(idx) + i10 * idx->nb[0] + i11 * idx->nb[1] + i12 * idx->nb[2]);
                if (i1 < 0 || i1 >= dst->ne[1]) return GGML_STATUS_FAILED;
                const float * s = (const float *) (TDATA(a) + i01 * a->nb[1] + i02 * a->nb[2] + i03 * a->nb[3]);
                char * d = TDATA(dst) + i1 * dst->nb[1] + i02 * dst->nb[2] + i03 * dst->nb[3];
                if (dst->type == GGML_TYPE_F32) memcpy(d, s, (size_t) nc * 4);
                else (void) oracle_quantize_row(dst->type, s, d, nc); /* from_float of the destination type (f16 / bf16 / quantised KV cache) */
            }
    return GGML_STATUS_SUCCESS;
}

/* CPY / DUP / CONT: ggml_compute_forward_dup — logical element order, f32/f16 conversion */
static enum ggml_status op_cpy(struct ggml_tensor * dst) {
    const struct ggml_tensor * a = dst->src[0];
    const int64_t n = ggml_abi_nelements(a);
    if (n != ggml_abi_nelements(dst)) return GGML_STATUS_FAILED;
    if (a->type == dst->type && ggml_abi_is_contiguous(a) && ggml_abi_is_contiguous(dst)) {
        memcpy(dst->data, a->data, ggml_abi_nbytes(a));
        return GGML_STATUS_SUCCESS;
    }
    /* quantised KV cache, K-shift (llama.cpp build_rope_shift: cast to f32 -> rope -> cpy back): contiguous Q8_0 <-> F32
       — ggml_compute_forward_dup_from_q dequantises whole rows, ggml_compute_forward_dup_f32 with a contiguous quantised
       destination runs from_float (quantize_row_q8_0) over rows of ne00 values */
    if ((is_cache_block_type(a->type) || a->type == GGML_TYPE_BF16) && dst->type == GGML_TYPE_F32 && ggml_abi_is_contiguous(a) && ggml_abi_is_contiguous(dst)) {
        oracle_dequantize_row(a->type, TDATA(a), (float *) TDATA(dst), n);
        return GGML_STATUS_SUCCESS;
    }
    if (a->type == GGML_TYPE_F32 && (is_cache_block_type(dst->type) || dst->type == GGML_TYPE_BF16) && ggml_abi_is_contiguous(a) && ggml_abi_is_contiguous(dst) &&
        (a->ne[0] % ggml_abi_blck_size(dst->type)) == 0) {
        (void) oracle_quantize_row(dst->type, (const float *) TDATA(a), TDATA(dst), n);
        return GGML_STATUS_SUCCESS;
    }
    const int sf32 = a->type == GGML_TYPE_F32, sf16 = a->type == GGML_TYPE_F16;
    const int df32 = dst->type == GGML_TYPE_F32, df16 = dst->type == GGML_TYPE_F16;
    const int si32 = a->type == GGML_TYPE_I32 && dst->type == GGML_TYPE_I32;
    if (!si32 && (!(sf32 || sf16) || !(df32 || df16))) return GGML_STATUS_FAILED;
    for (int64_t e = 0; e < n; ++e) {
        int64_t r = e;
        const int64_t s0 = r % a->ne[0]; r /= a->ne[0];
        const int64_t s1 = r % a->ne[1]; r /= a->ne[1];
        const int64_t s2 = r % a->ne[2]; r /= a->ne[2];
        const int64_t s3 = r;
        r = e;
        const int64_t d0 = r % dst->ne[0]; r /= dst->ne[0];
        const int64_t d1 = r % dst->ne[1]; r /= dst->ne[1];
        const int64_t d2 = r % dst->ne[2]; r /= dst->ne[2];
        const int64_t d3 = r;
        const char * ps = TDATA(a) + s0 * a->nb[0] + s1 * a->nb[1] + s2 * a->nb[2] + s3 * a->nb[3];
        char * pd = TDATA(dst) + d0 * dst->nb[0] + d1 * dst->nb[1] + d2 * dst->nb[2] + d3 * dst->nb[3];
        if (si32) { memcpy(pd, ps, 4); continue; }
        if (sf32 && df32) memcpy(pd, ps, 4);
        else if (sf16 && df16) memcpy(pd, ps, 2);
        else if (sf32 && df16) { float v; memcpy(&v, ps, 4); ggml_fp16_t h = oracle_fp32_to_fp16(v); memcpy(pd, &h, 2); }
        else { ggml_fp16_t h; memcpy(&h, ps, 2); float v = F16(h); memcpy(pd, &v, 4); }
    }
    return GGML_STATUS_SUCCESS;
}

/* ------------------------------------------------------------------------------------------ */
/* SOFT_MAX: ggml_compute_forward_soft_max_f32, with llama-box's zero-sum guard                  */
/* (llama-box/patches/llama.cpp/ggml-cpu.patch:5-15: assert(sum>0) -> sum = -INFINITY)           */
/* ------------------------------------------------------------------------------------------ */
static enum ggml_status op_soft_max(struct ggml_tensor * dst, int nth) {
    const struct ggml_tensor * a = dst->src[0];
    const struct ggml_tensor * mask = dst->src[1];
    const struct ggml_tensor * sinks = dst->src[2];
    if (a->type != GGML_TYPE_F32 || !ggml_abi_is_contiguous(a) || !ggml_abi_is_contiguous(dst)) return GGML_STATUS_FAILED;
    if (mask && mask->type != GGML_TYPE_F16 && mask->type != GGML_TYPE_F32) return GGML_STATUS_FAILED;
    const float scale = op_f32(dst, 0), max_bias = op_f32(dst, 1);
    const int64_t nc = a->ne[0], ne01 = a->ne[1], ne02 = a->ne[2], ne03 = a->ne[3];
    const uint32_t n_head = (uint32_t) ne02;
    const uint32_t n_head_log2 = 1u << (uint32_t) floor(log2(n_head));
    const float m0 = powf(2.0f, -(max_bias) / n_head_log2);
    const float m1 = powf(2.0f, -(max_bias / 2.0f) / n_head_log2);
    const int64_t nr = ne01 * ne02 * ne03;
    int fail = 0;
#pragma omp parallel for num_threads(nth) schedule(static)
    for (int64_t ir = 0; ir < nr; ++ir) {
        const int64_t i01 = ir % ne01, i02 = (ir / ne01) % ne02, i03 = ir / (ne01 * ne02);
        const uint32_t h = (uint32_t) i02;
        const float slope = (max_bias > 0.0f) ? (h < n_head_log2 ? powf(m0, h + 1) : powf(m1, 2 * (h - n_head_log2) + 1)) : 1.0f;
        const float * sp = (const float *) (TDATA(a) + i01 * a->nb[1] + i02 * a->nb[2] + i03 * a->nb[3]);
        float * dp = (float *) (TDATA(dst) + i01 * dst->nb[1] + i02 * dst->nb[2] + i03 * dst->nb[3]);
        float * wp = (float *) malloc((size_t) nc * 4);
        if (!wp) { fail = 1; continue; }
        for (int64_t i = 0; i < nc; ++i) wp[i] = sp[i] * scale;
        if (mask) {
            const char * mp = TDATA(mask) + i01 * mask->nb[1] + (i02 % mask->ne[2]) * mask->nb[2] + (i03 % mask->ne[3]) * mask->nb[3];
            if (mask->type == GGML_TYPE_F16) for (int64_t i = 0; i < nc; ++i) wp[i] += slope * F16(((const ggml_fp16_t *) mp)[i]);
            else for (int64_t i = 0; i < nc; ++i) wp[i] += slope * ((const float *) mp)[i];
        }
        float max = -INFINITY;
        for (int64_t i = 0; i < nc; ++i) if (wp[i] > max) max = wp[i];
        const float * sk = sinks ? (const float *) sinks->data : NULL;
        if (sk && sk[i02] > max) max = sk[i02];
        ggml_float sum = 0.0;
        for (int64_t i = 0; i < nc; ++i) {
            const float val = oracle_expf(wp[i] - max);
            sum += (ggml_float) val;
            dp[i] = val;
        }
        if (sk) sum += (ggml_float) oracle_expf(sk[i02] - max);
        if (isnan(sum) || sum == 0) sum = -INFINITY; /* llama-box patch */
        sum = 1.0 / sum;
        const float fs = (float) sum;
        for (int64_t i = 0; i < nc; ++i) dp[i] *= fs;
        free(wp);
    }
    return fail ? GGML_STATUS_ALLOC_FAILED : GGML_STATUS_SUCCESS;
}

/* ------------------------------------------------------------------------------------------ */
/* ROPE: ggml_compute_forward_rope_f32/_f16 (patch->ggml-cpu/ops.cpp:6204,:6390; mrope.patch:5-26) */
/* ------------------------------------------------------------------------------------------ */
static float rope_yarn_ramp(const float low, const float high, const int i0) {
    const float y = (i0 / 2 - low) / fmaxf(0.001f, high - low);
    return 1 - fminf(1, fmaxf(0, y));
}
static void rope_yarn(float theta_extrap, float freq_scale, const float corr_dims[2], int64_t i0, float ext_factor,
                      float mscale, float * cos_theta, float * sin_theta) {
    float theta_interp = freq_scale * theta_extrap;
    float theta = theta_interp;
    if (ext_factor != 0.0f) {
        float ramp_mix = rope_yarn_ramp(corr_dims[0], corr_dims[1], (int) i0) * ext_factor;
        theta = theta_interp * (1 - ramp_mix) + theta_extrap * ramp_mix;
        mscale *= 1.0f + 0.1f * logf(1.0f / freq_scale);
    }
    *cos_theta = cosf(theta) * mscale;
    *sin_theta = sinf(theta) * mscale;
}
static float rope_yarn_corr_dim(int n_dims, int n_ctx_orig, float n_rot, float base) {
    return n_dims * logf(n_ctx_orig / (n_rot * 2 * (float) M_PI)) / (2 * logf(base));
}
static void rope_yarn_corr_dims(int n_dims, int n_ctx_orig, float freq_base, float beta_fast, float beta_slow, float dims[2]) {
    float start = floorf(rope_yarn_corr_dim(n_dims, n_ctx_orig, beta_fast, freq_base));
    float end = ceilf(rope_yarn_corr_dim(n_dims, n_ctx_orig, beta_slow, freq_base));
    dims[0] = fmaxf(0, start);
    dims[1] = fminf((float) (n_dims - 1), end);
}

/* ggml_mrope_cache_init (ggml_rope_multi: Qwen2-VL multimodal sections / vision towers): four position streams
   (time, height, width, extra), one per `section` of the rotation pairs; all four angles advance by theta_scale every pair;
   with independent sections (vision mode) an angle restarts from its position when its section begins.
   llama-box drops the "some section > 0" assertion (mrope.patch:5-26); with all sections zero the upstream loop divides by
   zero, so that case has no defined result and is refused here (and by the backend's supports_op). */
static void mrope_cache_init(float theta_base_t, float theta_base_h, float theta_base_w, float theta_base_e, const int sections[4], int indep_sects,
                             float freq_scale, const float * freq_factors, const float corr_dims[2], int64_t ne0, float ext_factor, float mscale,
                             float * cache, float theta_scale) {
    float theta_t = theta_base_t, theta_h = theta_base_h, theta_w = theta_base_w, theta_e = theta_base_e;
    const int sect_dims = sections[0] + sections[1] + sections[2] + sections[3];
    const int sec_w = sections[1] + sections[0];
    const int sec_e = sections[2] + sec_w;
    for (int64_t i0 = 0; i0 < ne0; i0 += 2) {
        const float ff = freq_factors ? freq_factors[i0 / 2] : 1.0f;
        const int sector = (int) ((i0 / 2) % sect_dims);
        if (indep_sects) {
            if (sector == 0) theta_t = theta_base_t;
            else if (sector == sections[0]) theta_h = theta_base_h;
            else if (sector == sec_w) theta_w = theta_base_w;
            else if (sector == sec_e) theta_e = theta_base_e;
        }
        float theta = theta_t;
        if (sector >= sections[0] && sector < sec_w) theta = theta_h;
        else if (sector >= sec_w && sector < sec_w + sections[2]) theta = theta_w;
        else if (sector >= sec_w + sections[2]) theta = theta_e;
        rope_yarn(theta / ff, freq_scale, corr_dims, i0, ext_factor, mscale, &cache[i0 + 0], &cache[i0 + 1]);
        theta_t *= theta_scale;
        theta_w *= theta_scale;
        theta_h *= theta_scale;
        theta_e *= theta_scale;
    }
}

static enum ggml_status op_rope(struct ggml_tensor * dst) {
    const struct ggml_tensor * a = dst->src[0];
    const struct ggml_tensor * pos_t = dst->src[1];
    const struct ggml_tensor * ff_t = dst->src[2];
    const int n_dims = dst->op_params[1];
    const int mode = dst->op_params[2];
    const int n_ctx_orig = dst->op_params[4];
    const float freq_base = op_f32(dst, 5), freq_scale = op_f32(dst, 6), ext_factor = op_f32(dst, 7);
    const float attn_factor = op_f32(dst, 8), beta_fast = op_f32(dst, 9), beta_slow = op_f32(dst, 10);
    int sections[4];
    memcpy(sections, dst->op_params + 11, sizeof(sections));
    const int is_neox = mode & GGML_ROPE_TYPE_NEOX;
    const int is_mrope = mode & GGML_ROPE_TYPE_MROPE;
    const int is_vision = mode == GGML_ROPE_TYPE_VISION;
    if (a->type != dst->type || (a->type != GGML_TYPE_F32 && a->type != GGML_TYPE_F16)) return GGML_STATUS_FAILED;
    if (pos_t->type != GGML_TYPE_I32) return GGML_STATUS_FAILED;
    const int is16 = a->type == GGML_TYPE_F16;
    const int64_t ne0 = a->ne[0], ne1 = a->ne[1], ne2 = a->ne[2], ne3 = a->ne[3];
    if (is_mrope) {
        const int sect_dims = sections[0] + sections[1] + sections[2] + sections[3];
        if (sect_dims <= 0 || sect_dims > ne0 || ggml_abi_nelements(pos_t) < 4 * ne2) return GGML_STATUS_FAILED;
    }
    if (is_vision && n_dims != ne0 / 2) return GGML_STATUS_FAILED;
    const float theta_scale = powf(freq_base, -2.0f / n_dims);
    float corr_dims[2];
    rope_yarn_corr_dims(n_dims, n_ctx_orig, freq_base, beta_fast, beta_slow, corr_dims);
    const float * freq_factors = ff_t ? (const float *) ff_t->data : NULL;
    const int32_t * pos = (const int32_t *) pos_t->data;
    float * cache = (float *) malloc((size_t) ne0 * 4 + 16);
#define ROPE_LD(p, i) (is16 ? F16(((const ggml_fp16_t *) (p))[i]) : ((const float *) (p))[i])
#define ROPE_ST(p, i, v) do { if (is16) ((ggml_fp16_t *) (p))[i] = oracle_fp32_to_fp16(v); else ((float *) (p))[i] = (v); } while (0)
    for (int64_t i3 = 0; i3 < ne3; i3++) {
        for (int64_t i2 = 0; i2 < ne2; i2++) {
            if (!is_mrope) {
                /* ggml_rope_cache_init: theta advanced by repeated multiplication */
                float theta = (float) pos[i2];
                for (int64_t i0 = 0; i0 < ne0; i0 += 2) {
                    const float ff = freq_factors ? freq_factors[i0 / 2] : 1.0f;
                    rope_yarn(theta / ff, freq_scale, corr_dims, i0, ext_factor, attn_factor, &cache[i0 + 0], &cache[i0 + 1]);
                    theta *= theta_scale;
                }
            } else {
                mrope_cache_init((float) pos[i2], (float) pos[i2 + ne2], (float) pos[i2 + ne2 * 2], (float) pos[i2 + ne2 * 3], sections, is_vision,
                                 freq_scale, freq_factors, corr_dims, ne0, ext_factor, attn_factor, cache, theta_scale);
            }
            for (int64_t i1 = 0; i1 < ne1; i1++) {
                const char * src = TDATA(a) + i3 * a->nb[3] + i2 * a->nb[2] + i1 * a->nb[1];
                char * dp = TDATA(dst) + i3 * dst->nb[3] + i2 * dst->nb[2] + i1 * dst->nb[1];
                if (is_vision) {
                    /* pairs (ic, ic + n_dims) over the WHOLE row (n_dims == ne0/2): the upstream routine's two loops */
                    for (int64_t i0 = 0; i0 < ne0; i0 += 2) {
                        const float cos_theta = cache[i0 + 0], sin_theta = cache[i0 + 1];
                        const int64_t ic = i0 / 2;
                        const float x0 = ROPE_LD(src, ic), x1 = ROPE_LD(src, ic + n_dims);
                        ROPE_ST(dp, ic, x0 * cos_theta - x1 * sin_theta);
                        ROPE_ST(dp, ic + n_dims, x0 * sin_theta + x1 * cos_theta);
                    }
                    continue;
                }
                for (int64_t i0 = 0; i0 < n_dims; i0 += 2) {
                    const float cos_theta = cache[i0 + 0], sin_theta = cache[i0 + 1];
                    const int half_pairs = is_neox || is_mrope;
                    const int64_t ia = half_pairs ? i0 / 2 : i0;
                    const int64_t ib = half_pairs ? i0 / 2 + n_dims / 2 : i0 + 1;
                    const float x0 = ROPE_LD(src, ia), x1 = ROPE_LD(src, ib);
                    ROPE_ST(dp, ia, x0 * cos_theta - x1 * sin_theta);
                    ROPE_ST(dp, ib, x0 * sin_theta + x1 * cos_theta);
                }
                for (int64_t i0 = n_dims; i0 < ne0; ++i0) { /* pass-through tail */
                    if (is16) ((ggml_fp16_t *) dp)[i0] = ((const ggml_fp16_t *) src)[i0];
                    else ((float *) dp)[i0] = ((const float *) src)[i0];
                }
            }
        }
    }
#undef ROPE_LD
#undef ROPE_ST
    free(cache);
    return GGML_STATUS_SUCCESS;
}

/* ------------------------------------------------------------------------------------------ */
/* FLASH_ATTN_EXT: ggml_compute_forward_flash_attn_ext_f16 — online softmax per query row,       */
/* Q rounded to f16, K·Q dot in double, V accumulated IN F16 when V is f16 (Appendix A.3).       */
/* Quantised cache (-ctk/-ctv q8_0): Q goes through K's vec_dot_type (quantize_row_q8_0) and the  */
/* score is ggml_vec_dot_q8_0_q8_0; a non-f16 V row is dequantised (to_float) and accumulated in  */
/* f32 (ggml_vec_mad_f32) — the upstream routine's second branch.                                 */
/* ------------------------------------------------------------------------------------------ */
static enum ggml_status op_flash_attn_ext(struct ggml_tensor * dst, int nth) {
    const struct ggml_tensor * q = dst->src[0];
    const struct ggml_tensor * k = dst->src[1];
    const struct ggml_tensor * v = dst->src[2];
    const struct ggml_tensor * mask = dst->src[3];
    const struct ggml_tensor * sinks = dst->src[4];
    /* K: vec_dot of its type against the query converted to that type's vec_dot_type (f16 -> f16, bf16 -> bf16, f32 -> f32, q8_0 / q4_0 / q5_0 /
       iq4_nl -> q8_0, q4_1 / q5_1 -> q8_1); V: f16 rows accumulate in f16, every other type through to_float into an f32 accumulator */
    const enum ggml_type kt = k->type, vt = v->type;
    const int k_ok = kt == GGML_TYPE_F16 || kt == GGML_TYPE_BF16 || kt == GGML_TYPE_F32 || is_cache_block_type(kt);
    const int v_ok = vt == GGML_TYPE_F16 || vt == GGML_TYPE_BF16 || vt == GGML_TYPE_F32 || is_cache_block_type(vt);
    if (q->type != GGML_TYPE_F32 || !k_ok || !v_ok) return GGML_STATUS_FAILED;
    const int kq8 = is_cache_block_type(kt), vq8 = vt != GGML_TYPE_F16;
    const int kq81 = kt == GGML_TYPE_Q4_1 || kt == GGML_TYPE_Q5_1;
    if ((kq8 && (k->ne[0] % 32) != 0) || (is_cache_block_type(vt) && (v->ne[0] % 32) != 0)) return GGML_STATUS_FAILED;
    if (mask && mask->type != GGML_TYPE_F16) return GGML_STATUS_FAILED;
    const int64_t DK = k->ne[0], DV = v->ne[0];
    const int64_t neq1 = q->ne[1], neq2 = q->ne[2], neq3 = q->ne[3];
    const int64_t nek1 = k->ne[1];
    const int64_t rk2 = neq2 / k->ne[2], rk3 = neq3 / k->ne[3];
    const int64_t rv2 = neq2 / v->ne[2], rv3 = neq3 / v->ne[3];
    float scale = op_f32(dst, 0);
    const float max_bias = op_f32(dst, 1);
    const float logit_softcap = op_f32(dst, 2);
    if (logit_softcap != 0) scale /= logit_softcap;
    const uint32_t n_head = (uint32_t) neq2;
    const uint32_t n_head_log2 = 1u << (uint32_t) floor(log2(n_head));
    const float m0 = powf(2.0f, -(max_bias) / n_head_log2);
    const float m1 = powf(2.0f, -(max_bias / 2.0f) / n_head_log2);
    const int64_t nr = neq1 * neq2 * neq3;
    int fail = 0;
#pragma omp parallel for num_threads(nth) schedule(dynamic, 4)
    for (int64_t ir = 0; ir < nr; ++ir) {
        const int64_t iq3 = ir / (neq2 * neq1), iq2 = (ir - iq3 * neq2 * neq1) / neq1, iq1 = ir - iq3 * neq2 * neq1 - iq2 * neq1;
        const uint32_t h = (uint32_t) iq2;
        const float slope = (max_bias > 0.0f) ? (h < n_head_log2 ? powf(m0, h + 1) : powf(m1, 2 * (h - n_head_log2) + 1)) : 1.0f;
        float S = 0.0f, M = -INFINITY;
        float * VKQ32 = (float *) calloc((size_t) DV, 4);
        ggml_fp16_t * VKQ16 = (ggml_fp16_t *) calloc((size_t) DV, 2);
        ggml_fp16_t * Q16 = (ggml_fp16_t *) malloc((size_t) DK * 2);
        block_q8_0 * Qq = (block_q8_0 *) malloc((size_t) (DK / 32 + 1) * sizeof(block_q8_1) + (size_t) DK * 4); /* (room for any vec_dot_type: q8_0, q8_1, bf16, f32) */
        float * V32 = (float *) malloc((size_t) (DV + DK) * 4);
        if (!VKQ32 || !VKQ16 || !Q16 || !Qq || !V32) { fail = 1; free(VKQ32); free(VKQ16); free(Q16); free(Qq); free(V32); continue; }
        float * V32k = V32 + DV;
        const ggml_fp16_t * mp = mask ? (const ggml_fp16_t *) (TDATA(mask) + iq1 * mask->nb[1] + (iq2 % mask->ne[2]) * mask->nb[2] + (iq3 % mask->ne[3]) * mask->nb[3]) : NULL;
        const int64_t ik3 = iq3 / rk3, ik2 = iq2 / rk2, iv3 = iq3 / rv3, iv2 = iq2 / rv2;
        const float * pq = (const float *) (TDATA(q) + iq1 * q->nb[1] + iq2 * q->nb[2] + iq3 * q->nb[3]);
        if (kq8) (void) oracle_quantize_row(kq81 ? GGML_TYPE_Q8_1 : GGML_TYPE_Q8_0, pq, Qq, DK);
        else if (kt == GGML_TYPE_BF16) for (int64_t i = 0; i < DK; ++i) ((uint16_t *) Qq)[i] = fp32_to_bf16(pq[i]);
        else if (kt == GGML_TYPE_F32) memcpy(Qq, pq, (size_t) DK * 4);
        else for (int64_t i = 0; i < DK; ++i) Q16[i] = oracle_fp32_to_fp16(pq[i]);
        for (int64_t ic = 0; ic < nek1; ++ic) {
            const float mv = mp ? slope * F16(mp[ic]) : 0.0f;
            if (mv == -INFINITY) continue;
            const ggml_fp16_t * kd = (const ggml_fp16_t *) (TDATA(k) + ic * k->nb[1] + ik2 * k->nb[2] + ik3 * k->nb[3]);
            float s;
            if (kq8 && g_variant == 4) { /* yardstick: the block-format K row against the UNQUANTISED query (see oracle_set_variant) */
                oracle_dequantize_row(kt, kd, V32k, DK);
                ggml_float acc = 0.0;
                for (int64_t i = 0; i < DK; ++i) acc += (ggml_float) V32k[i] * (ggml_float) pq[i];
                s = (float) acc;
            } else if (kq8) {
                switch (kt) {
                    case GGML_TYPE_Q8_0: s = oracle_vec_dot_q8_0_q8_0(DK, (const block_q8_0 *) kd, Qq); break;
                    case GGML_TYPE_Q4_0: s = vec_dot_q4_0_q8_0(DK, (const block_q4_0 *) kd, Qq); break;
                    case GGML_TYPE_Q5_0: s = vec_dot_q5_0_q8_0(DK, (const block_q5_0 *) kd, Qq); break;
                    case GGML_TYPE_IQ4_NL: s = vec_dot_iq4_nl_q8_0(DK, (const block_iq4_nl *) kd, Qq); break;
                    case GGML_TYPE_Q4_1: s = vec_dot_q4_1_q8_1(DK, (const block_q4_1 *) kd, (const block_q8_1 *) Qq); break;
                    default: s = vec_dot_q5_1_q8_1(DK, (const block_q5_1 *) kd, (const block_q8_1 *) Qq); break;
                }
            } else if (kt == GGML_TYPE_BF16) { /* ggml_vec_dot_bf16 */
                ggml_float acc = 0.0;
                for (int64_t i = 0; i < DK; ++i) acc += (ggml_float) (bf16_to_fp32(((const uint16_t *) kd)[i]) * bf16_to_fp32(((const uint16_t *) Qq)[i]));
                s = (float) acc;
            } else if (kt == GGML_TYPE_F32) { /* ggml_vec_dot_f32 */
                ggml_float acc = 0.0;
                for (int64_t i = 0; i < DK; ++i) acc += (ggml_float) (((const float *) kd)[i] * ((const float *) Qq)[i]);
                s = (float) acc;
            } else {
                ggml_float acc = 0.0;
                for (int64_t i = 0; i < DK; ++i) acc += (ggml_float) (F16(kd[i]) * F16(Q16[i]));
                s = (float) acc;
            }
            s = s * scale;
            if (logit_softcap != 0.0f) s = logit_softcap * tanhf(s);
            s += mv;
            const float Mold = M;
            float ms = 1.0f, vs = 1.0f;
            const ggml_fp16_t * vd = (const ggml_fp16_t *) (TDATA(v) + ic * v->nb[1] + iv2 * v->nb[2] + iv3 * v->nb[3]);
            if (s > M) {
                M = s;
                ms = oracle_expf(Mold - M);
                if (vq8) for (int64_t i = 0; i < DV; ++i) VKQ32[i] *= ms;                                   /* ggml_vec_scale_f32 */
                else for (int64_t i = 0; i < DV; ++i) VKQ16[i] = oracle_fp32_to_fp16(F16(VKQ16[i]) * ms); /* ggml_vec_scale_f16 */
            } else {
                vs = oracle_expf(s - M);
            }
            if (vq8) { /* v_to_float + ggml_vec_mad_f32 (an f32 V is read in place) */
                oracle_dequantize_row(vt, vd, V32, DV);
                for (int64_t i = 0; i < DV; ++i) VKQ32[i] += V32[i] * vs;
            } else {
                for (int64_t i = 0; i < DV; ++i) { /* ggml_vec_mad_f16 */
                    const float p = F16(vd[i]) * vs;
                    VKQ16[i] = oracle_fp32_to_fp16(F16(VKQ16[i]) + p);
                }
            }
            S = S * ms + vs;
        }
        if (!vq8) for (int64_t i = 0; i < DV; ++i) VKQ32[i] = F16(VKQ16[i]);
        if (sinks) {
            const float s = ((const float *) sinks->data)[h];
            float ms = 1.0f, vs = 1.0f;
            if (s > M) {
                ms = oracle_expf(M - s);
                for (int64_t i = 0; i < DV; ++i) VKQ32[i] *= ms;
            } else {
                vs = oracle_expf(s - M);
            }
            S = S * ms + vs;
        }
        const float S_inv = 1.0f / S;
        for (int64_t i = 0; i < DV; ++i) VKQ32[i] *= S_inv;
        /* dst layout [DV, n_head, n_q, batch] (permuted) */
        memcpy(TDATA(dst) + (iq3 * dst->ne[2] * dst->ne[1] + iq2 + iq1 * dst->ne[1]) * dst->nb[1], VKQ32, (size_t) DV * 4);
        free(VKQ32);
        free(VKQ16);
        free(Q16);
        free(Qq);
        free(V32);
    }
    return fail ? GGML_STATUS_ALLOC_FAILED : GGML_STATUS_SUCCESS;
}

/* ARGMAX: ggml_compute_forward_argmax_f32 — first index of the row maximum */
static enum ggml_status op_argmax(struct ggml_tensor * dst) {
    const struct ggml_tensor * a = dst->src[0];
    if (a->type != GGML_TYPE_F32 || dst->type != GGML_TYPE_I32) return GGML_STATUS_FAILED;
    for (int64_t i1 = 0; i1 < a->ne[1]; ++i1) {
        const float * x = (const float *) (TDATA(a) + i1 * a->nb[1]);
        float mx = -INFINITY;
        int32_t idx = 0;
        for (int64_t i = 0; i < a->ne[0]; ++i) if (x[i] > mx) { mx = x[i]; idx = (int32_t) i; }
        ((int32_t *) dst->data)[i1] = idx;
    }
    return GGML_STATUS_SUCCESS;
}

/* ------------------------------------------------------------------------------------------ */
int oracle_supports_op(const struct ggml_tensor * node) {
    switch (node->op) {
        case GGML_OP_NONE: case GGML_OP_VIEW: case GGML_OP_RESHAPE: case GGML_OP_PERMUTE: case GGML_OP_TRANSPOSE:
        case GGML_OP_MUL_MAT: case GGML_OP_ADD: case GGML_OP_SUB: case GGML_OP_MUL: case GGML_OP_DIV:
        case GGML_OP_SCALE: case GGML_OP_RMS_NORM: case GGML_OP_UNARY: case GGML_OP_GLU: case GGML_OP_GET_ROWS:
        case GGML_OP_SET_ROWS: case GGML_OP_CPY: case GGML_OP_DUP: case GGML_OP_CONT: case GGML_OP_SOFT_MAX:
        case GGML_OP_ROPE: case GGML_OP_FLASH_ATTN_EXT: case GGML_OP_ARGMAX:
            return 1;
        default: return 0;
    }
}

enum ggml_status oracle_compute_node(struct ggml_tensor * node, int n_threads) {
    int nth = n_threads > 0 ? n_threads : oracle_max_threads();
    /* ggml-cpu skips tensors without elements (ggml_compute_forward: `if (tensor->op == GGML_OP_NONE || ggml_is_empty(tensor)) return;`):
       llama.cpp relies on it for micro-batches nobody wants logits from — out_ids is empty and everything behind the last layer's get_rows has 0 rows */
    if (node->ne[0] == 0 || node->ne[1] == 0 || node->ne[2] == 0 || node->ne[3] == 0) return GGML_STATUS_SUCCESS;
    switch (node->op) {
        case GGML_OP_NONE: case GGML_OP_VIEW: case GGML_OP_RESHAPE: case GGML_OP_PERMUTE: case GGML_OP_TRANSPOSE:
            return GGML_STATUS_SUCCESS;
        case GGML_OP_MUL_MAT: return op_mul_mat(node, nth);
        case GGML_OP_ADD: case GGML_OP_SUB: case GGML_OP_MUL: case GGML_OP_DIV: return op_binary(node, nth);
        case GGML_OP_SCALE: return op_scale(node);
        case GGML_OP_RMS_NORM: return op_rms_norm(node, nth);
        case GGML_OP_UNARY: return op_unary(node);
        case GGML_OP_GLU: return op_glu(node);
        case GGML_OP_GET_ROWS: return op_get_rows(node);
        case GGML_OP_SET_ROWS: return op_set_rows(node);
        case GGML_OP_CPY: case GGML_OP_DUP: case GGML_OP_CONT: return op_cpy(node);
        case GGML_OP_SOFT_MAX: return op_soft_max(node, nth);
        case GGML_OP_ROPE: return op_rope(node);
        case GGML_OP_FLASH_ATTN_EXT: return op_flash_attn_ext(node, nth);
        case GGML_OP_ARGMAX: return op_argmax(node);
        default: return GGML_STATUS_FAILED;
    }
}

/* FAULT INJECTION for the tests of the parity gates themselves (tests/test_oracle_golden.py: does a gate trip when it should?): variant 5 scales the result of the
   node named "result_output" (the logits) by 1 + 4e-3, variant 6 the result of every node whose name starts with "ffn_out" (a layer's FFN branch) by 1 + 1e-2 */
static void inject_fault(struct ggml_tensor * t) {
    float f = 1.0f;
    if (g_variant == 5 && strcmp(t->name, "result_output") == 0) f = 1.004f;
    else if (g_variant == 6 && strncmp(t->name, "ffn_out", 7) == 0) f = 1.01f;
    if (f == 1.0f || t->type != GGML_TYPE_F32) return;
    const int64_t n = t->ne[0] * t->ne[1] * t->ne[2] * t->ne[3];
    float * d = (float *) t->data;
    for (int64_t i = 0; i < n; ++i) d[i] *= f;
}
enum ggml_status oracle_graph_compute(struct ggml_cgraph * graph, int n_threads) {
    for (int i = 0; i < graph->n_nodes; ++i) {
        enum ggml_status st = oracle_compute_node(graph->nodes[i], n_threads);
        if (st != GGML_STATUS_SUCCESS) return st;
        if (g_variant >= 5) inject_fault(graph->nodes[i]);
    }
    return GGML_STATUS_SUCCESS;
}
