// kv_types.hip — a KV cache kept in one of the other types llama-box lets the user pick (-ctk / -ctv, llama-box/engine_param.hpp:51-54: f32, f16,
// bf16, q8_0, q4_0, q4_1, iq4_nl, q5_0, q5_1).  f16 and q8_0 are the fast paths of qkv.hip / fattn.hip; everything here serves the rest so that
// such a cache stays on the device instead of sending its SET_ROWS / FLASH_ATTN_EXT / K-shift nodes back to the CPU backend:
//
//   * SET_ROWS f32 -> type: the destination type's from_float, exactly as ggml's quantize_row_*_ref evaluates it (one f32 rounding per operation,
//     the FIRST element of largest magnitude sets the sign of the scale, truncating conversions) — byte-identical blocks;
//   * CPY type <-> f32 on contiguous tensors: the K-shift of such a cache (llama.cpp build_rope_shift: cast to f32 -> rope -> copy back);
//   * FLASH_ATTN_EXT: the K / V views are expanded to an f16 image in scratch memory once per node (level * d in f32 — exact — then ONE rounding to
//     f16) and the f16 attention kernels run on the image.  ggml-cpu instead quantises every query row to 8 bits (Q8_0 / Q8_1) and takes integer block
//     dots: this path keeps the query in f16, which is closer to exact attention than the reference (the tests measure both distances).
//
// A block is 32 values; one thread handles one block (a cache row is 32 blocks per 1024 values: launch-bound work, not bandwidth-bound).
#include "common.h"
#include "dev_util.h"
#include "kernels.h"
#include "kv_dequant.h"
#include "kv_quant.h"

namespace mi355x {

// one block -> 32 f32 values (dequantize_row_*): level * d (+ m), one rounding per operation
template <int TYPE> __device__ __forceinline__ void dequantize_block(const char * blk, float (&y)[32]) {
    if constexpr (TYPE == GGML_TYPE_F32) {
#pragma unroll
        for (int j = 0; j < 32; ++j) y[j] = ((const float *) blk)[j];
    } else if constexpr (TYPE == GGML_TYPE_BF16) {
#pragma unroll
        for (int j = 0; j < 32; ++j) y[j] = __uint_as_float((uint32_t) ((const uint16_t *) blk)[j] << 16);
    } else if constexpr (TYPE == GGML_TYPE_F16) {
#pragma unroll
        for (int j = 0; j < 32; ++j) y[j] = h2f(((const uint16_t *) blk)[j]);
    } else if constexpr (TYPE == GGML_TYPE_Q8_0) {
        const float d = h2f(ld16(blk));
#pragma unroll
        for (int j = 0; j < 32; j += 4) {
            const uint32_t w = ld32_a2(blk + 2 + j);
#pragma unroll
            for (int e = 0; e < 4; ++e) y[j + e] = d * (float) (int8_t) (w >> (8 * e));
        }
    } else {
        constexpr bool OFFSET = TYPE == GGML_TYPE_Q4_1 || TYPE == GGML_TYPE_Q5_1;
        constexpr bool FIVE = TYPE == GGML_TYPE_Q5_0 || TYPE == GGML_TYPE_Q5_1;
        const float d = h2f(ld16(blk));
        const float m = OFFSET ? h2f(ld16(blk + 2)) : 0.0f;
        const int o = (OFFSET ? 4 : 2) + (FIVE ? 4 : 0);
        uint32_t qh = 0;
        if (FIVE) qh = ld32_a2(blk + o - 4);
        // (the 16 nibble bytes as four dwords: blocks are 2-byte aligned, the hardware takes the unaligned dword load)
        const uint32_t qw[4] = {ld32_a2(blk + o), ld32_a2(blk + o + 4), ld32_a2(blk + o + 8), ld32_a2(blk + o + 12)};
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
            const uint32_t w = (qw[j >> 2] >> (8 * (j & 3))) & 0xFFFFu;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int b = (w >> (8 * e)) & 0xFF;
                int q0 = b & 0x0F, q1 = b >> 4;
                if (FIVE) {
                    q0 |= (int) ((qh >> (j + e)) & 1u) << 4;
                    q1 |= (int) ((qh >> (j + e + 16)) & 1u) << 4;
                }
                if constexpr (TYPE == GGML_TYPE_IQ4_NL) {
                    y[j + e] = d * (float) k_iq4nl_values[q0];
                    y[j + e + 16] = d * (float) k_iq4nl_values[q1];
                } else if constexpr (OFFSET) {
                    y[j + e] = (float) q0 * d + m;
                    y[j + e + 16] = (float) q1 * d + m;
                } else {
                    y[j + e] = (float) (q0 - (FIVE ? 16 : 8)) * d;
                    y[j + e + 16] = (float) (q1 - (FIVE ? 16 : 8)) * d;
                }
            }
        }
    }
}

template <int TYPE> __device__ __forceinline__ constexpr int kv_block_bytes() {
    return TYPE == GGML_TYPE_F32 ? 128 : (TYPE == GGML_TYPE_BF16 || TYPE == GGML_TYPE_F16) ? 64 : TYPE == GGML_TYPE_Q8_0 ? 34 : (TYPE == GGML_TYPE_Q4_0 || TYPE == GGML_TYPE_IQ4_NL) ? 18 :
           TYPE == GGML_TYPE_Q4_1 ? 20 : TYPE == GGML_TYPE_Q5_0 ? 22 : 24;
}

__device__ __forceinline__ void load_block_f32(const char * src, float (&x)[32]) {  // 32 f32 of a row (rows are 16-byte aligned: supports_op)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float4 v = ((const float4 *) src)[j];
        x[4 * j] = v.x;
        x[4 * j + 1] = v.y;
        x[4 * j + 2] = v.z;
        x[4 * j + 3] = v.w;
    }
}

// ---- SET_ROWS: f32 rows [ne0, ne1, ne2, ne3] -> rows idx[i1] of a tensor of blocks (or bf16).  A thread per block of 32 values.
template <int TYPE> __global__ void __launch_bounds__(64) k_set_rows_kv(const tdesc a, const tdesc idx, const tdesc d) {
    const int64_t per_row = a.ne[0] / 32;
    const int64_t gid = (int64_t) blockIdx.x * 64 + threadIdx.x;
    if (gid >= per_row * a.ne[1] * a.ne[2] * a.ne[3]) return;
    const int64_t row = gid / per_row, blk = gid - row * per_row;
    const int64_t i01 = row % a.ne[1], i02 = (row / a.ne[1]) % a.ne[2], i03 = row / (a.ne[1] * a.ne[2]);
    const int64_t i12 = i03 % idx.ne[2], i11 = i02 % idx.ne[1];
    const int64_t r = *(const int64_t *) (idx.data + i01 * idx.nb[0] + i11 * idx.nb[1] + i12 * idx.nb[2]);
    float x[32];
    load_block_f32(a.data + i01 * a.nb[1] + i02 * a.nb[2] + i03 * a.nb[3] + blk * 128, x);
    char * out = d.data + r * d.nb[1] + i02 * d.nb[2] + i03 * d.nb[3] + blk * kv_block_bytes<TYPE>();
    if constexpr (TYPE == GGML_TYPE_BF16) {
#pragma unroll
        for (int j = 0; j < 32; ++j) st16(out + 2 * j, f2bf(x[j]));
    } else {
        quantize_block<TYPE>(x, out);
    }
}

// ... two SET_ROWS in one launch (the K and the V rows of a step): grid.y picks the pair member, the destination type is a run-time switch
__device__ __forceinline__ void set_rows_block_any(const tdesc & a, const tdesc & idx, const tdesc & d, const int64_t gid) {
    const int64_t per_row = a.ne[0] / 32;
    if (gid >= per_row * a.ne[1] * a.ne[2] * a.ne[3]) return;
    const int64_t row = gid / per_row, blk = gid - row * per_row;
    const int64_t i01 = row % a.ne[1], i02 = (row / a.ne[1]) % a.ne[2], i03 = row / (a.ne[1] * a.ne[2]);
    const int64_t i12 = i03 % idx.ne[2], i11 = i02 % idx.ne[1];
    const int64_t r = *(const int64_t *) (idx.data + i01 * idx.nb[0] + i11 * idx.nb[1] + i12 * idx.nb[2]);
    float x[32];
    load_block_f32(a.data + i01 * a.nb[1] + i02 * a.nb[2] + i03 * a.nb[3] + blk * 128, x);
    char * row_out = d.data + r * d.nb[1] + i02 * d.nb[2] + i03 * d.nb[3];
    switch (d.type) {
        case GGML_TYPE_Q4_0: quantize_block<GGML_TYPE_Q4_0>(x, row_out + blk * 18); break;
        case GGML_TYPE_Q4_1: quantize_block<GGML_TYPE_Q4_1>(x, row_out + blk * 20); break;
        case GGML_TYPE_Q5_0: quantize_block<GGML_TYPE_Q5_0>(x, row_out + blk * 22); break;
        case GGML_TYPE_Q5_1: quantize_block<GGML_TYPE_Q5_1>(x, row_out + blk * 24); break;
        case GGML_TYPE_IQ4_NL: quantize_block<GGML_TYPE_IQ4_NL>(x, row_out + blk * 18); break;
        default:  // BF16
#pragma unroll
            for (int j = 0; j < 32; ++j) st16(row_out + blk * 64 + 2 * j, f2bf(x[j]));
            break;
    }
}
__global__ void __launch_bounds__(64) k_set_rows_kv_pair(const tdesc a0, const tdesc i0, const tdesc d0, const tdesc a1, const tdesc i1, const tdesc d1) {
    const int64_t gid = (int64_t) blockIdx.x * 64 + threadIdx.x;
    if (blockIdx.y == 0) set_rows_block_any(a0, i0, d0, gid);
    else set_rows_block_any(a1, i1, d1, gid);
}

// ---- SET_ROWS into IQ4_NL, lane-parallel: the level search is ~90 VALU per value, 3 000 for a block on one thread (25 us per layer for the two rows of a decode
// step).  One block per 32 lanes, a value per lane: the block's extreme by an index-aware butterfly (the FIRST largest magnitude, as the reference's scan finds
// it), the level per lane, and the two weighted sums added in the reference's order — every lane walks j = 0 .. 31 over its block's products (the same 64
// dependent adds the reference does, once per block instead of being the tail of 3 000 instructions): the bytes stay identical.
__device__ __forceinline__ void iq4nl_set_block_lanes(const tdesc & a, const tdesc & idx, const tdesc & d, const int64_t blk_id, const int lane) {
    const int64_t per_row = a.ne[0] / 32;
    if (blk_id >= per_row * a.ne[1] * a.ne[2] * a.ne[3]) return;  // (whole half-waves: a block is 32 lanes)
    const int l32 = lane & 31;
    const int64_t row = blk_id / per_row, blk = blk_id - row * per_row;
    const int64_t i01 = row % a.ne[1], i02 = (row / a.ne[1]) % a.ne[2], i03 = row / (a.ne[1] * a.ne[2]);
    const int64_t i12 = i03 % idx.ne[2], i11 = i02 % idx.ne[1];
    const int64_t r = *(const int64_t *) (idx.data + i01 * idx.nb[0] + i11 * idx.nb[1] + i12 * idx.nb[2]);
    const float xv = ((const float *) (a.data + i01 * a.nb[1] + i02 * a.nb[2] + i03 * a.nb[3]))[blk * 32 + l32];
    quantize_block_lanes<GGML_TYPE_IQ4_NL>(xv, lane, d.data + r * d.nb[1] + i02 * d.nb[2] + i03 * d.nb[3] + blk * 18, true);
}
__global__ void __launch_bounds__(256) k_set_rows_iq4nl(const tdesc a0, const tdesc i0, const tdesc d0, const tdesc a1, const tdesc i1, const tdesc d1) {
    const int64_t blk_id = ((int64_t) blockIdx.x * 256 + threadIdx.x) >> 5;
    if (blockIdx.y == 0) iq4nl_set_block_lanes(a0, i0, d0, blk_id, (int) (threadIdx.x & 63));
    else iq4nl_set_block_lanes(a1, i1, d1, blk_id, (int) (threadIdx.x & 63));
}

// ---- CPY between contiguous tensors: blocks -> f32 and f32 -> blocks (K-shift)
template <int TYPE> __global__ void __launch_bounds__(64) k_cpy_kv_to_f32(const char * __restrict__ src, float * __restrict__ dst, const int64_t n_blocks) {
    const int64_t i = (int64_t) blockIdx.x * 64 + threadIdx.x;
    if (i >= n_blocks) return;
    float y[32];
    dequantize_block<TYPE>(src + i * kv_block_bytes<TYPE>(), y);
#pragma unroll
    for (int j = 0; j < 8; ++j) ((float4 *) (dst + i * 32))[j] = make_float4(y[4 * j], y[4 * j + 1], y[4 * j + 2], y[4 * j + 3]);
}
template <int TYPE> __global__ void __launch_bounds__(64) k_cpy_f32_to_kv(const float * __restrict__ src, char * __restrict__ dst, const int64_t n_blocks) {
    const int64_t i = (int64_t) blockIdx.x * 64 + threadIdx.x;
    if (i >= n_blocks) return;
    float x[32];
    load_block_f32((const char *) (src + i * 32), x);
    char * out = dst + i * kv_block_bytes<TYPE>();
    if constexpr (TYPE == GGML_TYPE_BF16) {
#pragma unroll
        for (int j = 0; j < 32; ++j) st16(out + 2 * j, f2bf(x[j]));
    } else {
        quantize_block<TYPE>(x, out);
    }
}

// ---- the f16 image of a K and / or V view [D, n_kv, n_kv_head] (any row / head strides) -> rows [n_kv][n_kv_head * D].  ONE launch per attention node:
// grid.y picks the tensor, the type is a run-time switch (uniform over the launch's half)
__global__ void __launch_bounds__(256) k_kv_image_f16(const tdesc k, const tdesc v, uint16_t * __restrict__ ko, uint16_t * __restrict__ vo, const int first) {
    const bool second = (int) blockIdx.y + first == 1;
    const tdesc & t = second ? v : k;
    uint16_t * out = second ? vo : ko;
    const int64_t per_head = t.ne[0] / 32, per_cell = per_head * t.ne[2];
    const int64_t gid = (int64_t) blockIdx.x * 256 + threadIdx.x;
    const int64_t i = gid >> 2;
    const int o = (int) (gid & 3);
    if (i >= per_cell * t.ne[1]) return;
    const int64_t cell = i / per_cell, rem = i - cell * per_cell, h = rem / per_head, b = rem - h * per_head;
    const char * blk = t.data + cell * t.nb[1] + h * t.nb[2] + b * kv_block_bytes_any(t.type);
    uint32_t w[4];
    if (t.type == GGML_TYPE_F32) {  // (eight dwords: not an octet the shared helpers carry)
        const float4 lo = ((const float4 *) blk)[2 * o], hi = ((const float4 *) blk)[2 * o + 1];
        w[0] = (uint32_t) f2h(lo.x) | ((uint32_t) f2h(lo.y) << 16);
        w[1] = (uint32_t) f2h(lo.z) | ((uint32_t) f2h(lo.w) << 16);
        w[2] = (uint32_t) f2h(hi.x) | ((uint32_t) f2h(hi.y) << 16);
        w[3] = (uint32_t) f2h(hi.z) | ((uint32_t) f2h(hi.w) << 16);
    } else {
        kv_octet_f16(t.type, kv_load_octet_raw(t.type, blk, o), o, w);
    }
    *(uint4 *) (out + (cell * t.ne[2] + h) * t.ne[0] + b * 32 + 8 * o) = make_uint4(w[0], w[1], w[2], w[3]);
}

bool kv_type_is_block(int type) {
    return type == GGML_TYPE_Q4_0 || type == GGML_TYPE_Q4_1 || type == GGML_TYPE_Q5_0 || type == GGML_TYPE_Q5_1 || type == GGML_TYPE_IQ4_NL;
}
// types whose cache rows this file stores (SET_ROWS / CPY from f32)
bool kv_store_type(int type) { return kv_type_is_block(type) || type == GGML_TYPE_BF16; }
// types FLASH_ATTN_EXT reads through the f16 image (f16 itself is read in place)
bool kv_image_type(int type) { return kv_type_is_block(type) || type == GGML_TYPE_BF16 || type == GGML_TYPE_F32 || type == GGML_TYPE_Q8_0; }

#define KV_DISPATCH_STORE(TYPE_, CALL)                                  \
    switch (TYPE_) {                                                    \
        case GGML_TYPE_Q4_0: { constexpr int T = GGML_TYPE_Q4_0; CALL; break; }   \
        case GGML_TYPE_Q4_1: { constexpr int T = GGML_TYPE_Q4_1; CALL; break; }   \
        case GGML_TYPE_Q5_0: { constexpr int T = GGML_TYPE_Q5_0; CALL; break; }   \
        case GGML_TYPE_Q5_1: { constexpr int T = GGML_TYPE_Q5_1; CALL; break; }   \
        case GGML_TYPE_IQ4_NL: { constexpr int T = GGML_TYPE_IQ4_NL; CALL; break; } \
        case GGML_TYPE_BF16: { constexpr int T = GGML_TYPE_BF16; CALL; break; }   \
        default: MI_ERR("kv_types: type %d has no store kernel", (int) (TYPE_)); abort(); \
    }

void launch_set_rows_kv(hipStream_t s, const tdesc & a, const tdesc & idx, const tdesc & d) {
    const int64_t n = (a.ne[0] / 32) * a.ne[1] * a.ne[2] * a.ne[3];
    if (n <= 0) return;
    if (d.type == GGML_TYPE_IQ4_NL) {  // a lane per value
        hipLaunchKernelGGL(k_set_rows_iq4nl, dim3((unsigned) ((n * 32 + 255) / 256), 1), dim3(256), 0, s, a, idx, d, a, idx, d);
        return;
    }
    const dim3 grid((unsigned) ((n + 63) / 64));
    KV_DISPATCH_STORE(d.type, hipLaunchKernelGGL((k_set_rows_kv<T>), grid, dim3(64), 0, s, a, idx, d))
}
void launch_set_rows_kv_pair(hipStream_t s, const tdesc & a0, const tdesc & i0, const tdesc & d0, const tdesc & a1, const tdesc & i1, const tdesc & d1) {
    const int64_t n0 = (a0.ne[0] / 32) * a0.ne[1] * a0.ne[2] * a0.ne[3], n1 = (a1.ne[0] / 32) * a1.ne[1] * a1.ne[2] * a1.ne[3];
    if (std::max(n0, n1) <= 0) return;
    if (d0.type == GGML_TYPE_IQ4_NL && d1.type == GGML_TYPE_IQ4_NL) {
        hipLaunchKernelGGL(k_set_rows_iq4nl, dim3((unsigned) ((std::max(n0, n1) * 32 + 255) / 256), 2), dim3(256), 0, s, a0, i0, d0, a1, i1, d1);
        return;
    }
    hipLaunchKernelGGL(k_set_rows_kv_pair, dim3((unsigned) ((std::max(n0, n1) + 63) / 64), 2), dim3(64), 0, s, a0, i0, d0, a1, i1, d1);
}
void launch_cpy_kv(hipStream_t s, int type, const void * src, void * dst, int64_t n_values, bool to_type) {
    const int64_t n = n_values / 32;
    if (n <= 0) return;
    const dim3 grid((unsigned) ((n + 63) / 64));
    if (to_type) {
        KV_DISPATCH_STORE(type, hipLaunchKernelGGL((k_cpy_f32_to_kv<T>), grid, dim3(64), 0, s, (const float *) src, (char *) dst, n))
    } else {
        KV_DISPATCH_STORE(type, hipLaunchKernelGGL((k_cpy_kv_to_f32<T>), grid, dim3(64), 0, s, (const char *) src, (float *) dst, n))
    }
}
size_t kv_image_bytes(const tdesc & t) { return (((size_t) (t.ne[0] * t.ne[1] * t.ne[2]) * sizeof(uint16_t)) + 255) & ~(size_t) 255; }
static tdesc image_desc(const tdesc & t, void * image) {
    tdesc o = t;
    o.data = (char *) image;
    o.type = GGML_TYPE_F16;
    o.nb[0] = 2;
    o.nb[2] = t.ne[0] * 2;
    o.nb[1] = t.ne[2] * o.nb[2];
    o.nb[3] = t.ne[1] * o.nb[1];
    return o;
}
tdesc kv_image_desc(const tdesc & t, void * image) { return image_desc(t, image); }
// expands whichever of K / V is not f16 into `image` (K's image first) in ONE launch and rewrites the descriptors to the f16 rows [n_kv][n_kv_head * D]
void launch_kv_images_f16(hipStream_t s, tdesc & k, tdesc & v, void * image) {
    const bool do_k = k.type != GGML_TYPE_F16, do_v = v.type != GGML_TYPE_F16;
    if (!do_k && !do_v) return;
    uint16_t * ko = (uint16_t *) image;
    uint16_t * vo = (uint16_t *) ((char *) image + (do_k ? kv_image_bytes(k) : 0));
    const int64_t nk = do_k ? (k.ne[0] / 32) * k.ne[1] * k.ne[2] : 0, nv = do_v ? (v.ne[0] / 32) * v.ne[1] * v.ne[2] : 0;
    const dim3 grid((unsigned) ((4 * std::max(nk, nv) + 255) / 256), (do_k && do_v) ? 2 : 1);  // four lanes per block of 32 values
    hipLaunchKernelGGL(k_kv_image_f16, grid, dim3(256), 0, s, k, v, ko, vo, do_k ? 0 : 1);
    if (do_k) k = image_desc(k, ko);
    if (do_v) v = image_desc(v, vo);
}

MI_TU_TOUCH(kv_types)

}  // namespace mi355x
