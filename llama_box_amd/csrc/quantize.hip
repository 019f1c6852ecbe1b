// quantize.hip — on-device activation quantisation, bit-identical to ggml-cpu's quantize_row_q8_K_ref /
// quantize_row_q8_0_ref (oracle/ggml_cpu_ref.c restates the same two routines).
// Matching the CPU's activation rounding exactly is what makes the integer block dot products downstream
// exact, leaving only f32 summation order as a source of difference (SURVEY.md §7 "hard parts" #2).
//
// One wave (64 lanes x float4) per 256-element chunk: HBM/L2 traffic is 1 KiB in, 304 B out per wave.
#include "dev_util.h"
#include "kernels.h"

namespace mi355x {

__global__ void __launch_bounds__(64) k_quantize_q8_K(const char * __restrict__ src, int64_t ne1, int64_t ne2, int64_t nb1, int64_t nb2,
                                                      int64_t nb3, int nb_per_row, q8k_dev * __restrict__ dst) {
    const int64_t gid = blockIdx.x;
    const int64_t row = gid / nb_per_row;
    const int ib = (int) (gid - row * nb_per_row);
    const int64_t i1 = row % ne1, i2 = (row / ne1) % ne2, i3 = row / (ne1 * ne2);
    const float * x = (const float *) (src + i1 * nb1 + i2 * nb2 + i3 * nb3) + (int64_t) ib * 256;
    const int lane = threadIdx.x;
    float v[4];
    if ((((uintptr_t) x) & 15) == 0) {
        const float4 t = ((const float4 *) x)[lane];
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = x[lane * 4 + k];
    }
    wave_quantize_q8_K(v, lane, dst + gid);
}

// Q8_0 activations: 8 lanes per 32-element block; d = amax/127 stored through fp16; q = roundf(x * (1/d))
template <bool PANEL>
__global__ void __launch_bounds__(64) k_quantize_q8_0(const char * __restrict__ src, int64_t ne1, int64_t ne2, int64_t nb1, int64_t nb2,
                                                      int64_t nb3, int K, int chunks_per_row, q80_dev * __restrict__ dst) {
    const int64_t gid = blockIdx.x;
    const int64_t row = gid / chunks_per_row;
    const int ic = (int) (gid - row * chunks_per_row);
    const int64_t i1 = row % ne1, i2 = (row / ne1) % ne2, i3 = row / (ne1 * ne2);
    const float * xrow = (const float *) (src + i1 * nb1 + i2 * nb2 + i3 * nb3);
    const int lane = threadIdx.x;
    const int e0 = ic * 256 + lane * 4;
    const bool live = e0 < K;  // K is a multiple of 32, so a block is either entirely live or entirely dead
    float v[4] = {0, 0, 0, 0};
    if (live) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = xrow[e0 + k];
    }
    float amax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
    const float d = amax / 127.0f;
    const float id = d != 0.0f ? 1.0f / d : 0.0f;
    if (!live) return;
    uint32_t packed = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = (int) roundf(v[k] * id);
        packed |= (uint32_t) (r & 0xFF) << (8 * k);
    }
    if constexpr (PANEL) {  // the block's tile: [K half][column][16 quants], then the 32 columns' scales
        char * tile = (char *) dst + ((size_t) (row >> 5) * (size_t) (K / 32) + (size_t) (e0 >> 5)) * 1152;  // (columns in groups of 32, a group's tiles together)
        const int w8 = lane & 7, cr = (int) (row & 31);
        *(uint32_t *) (tile + (w8 >> 2) * 512 + cr * 16 + (w8 & 3) * 4) = packed;
        if (w8 == 0) *(float *) (tile + 1024 + cr * 4) = h2f(f2h(d));
    } else {
        q80_dev * y = dst + row * (int64_t) (K / 32) + (e0 >> 5);
        ((uint32_t *) y->qs)[lane & 7] = packed;
        if ((lane & 7) == 0) y->d = h2f(f2h(d));
    }
}

// ---- producers fused with the quantisation (batches: the f32 intermediate is never written).  Same f32 arithmetic as the
// stand-alone kernels (ops.hip k_rms_norm + k_binary MUL, k_swiglu), so the Q8_K blocks are the ones the unfused path builds.
// RMS_NORM(x) * w -> Q8_K: one 16-wave workgroup per row, wave w owns blocks w, w+16, w+32, w+48 (K <= 16384)
// SK: the row does not exist yet — it is the sum of the split-K partial products of the mat-mul before (in split order, then the
// epilogue add: exactly what k_splitk_reduce computes), written out as f32 on the way because the residual stream reads it later
template <bool SK>
__global__ void __launch_bounds__(1024) k_rms_norm_mul_q8_K(const char * __restrict__ src, const int64_t nb1, const float * __restrict__ w, const float eps, const int K,
                                                            q8k_dev * __restrict__ dst, const splitk_src sk) {
    __shared__ double red[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nblk = K / 256;
    const float4 * x4 = (const float4 *) (src + (int64_t) blockIdx.x * nb1);
    const float4 * w4 = (const float4 *) w;
    float4 v[4], g[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int b = wave + 16 * u;
        if (b < nblk) {
            if constexpr (SK) {
                const int64_t e = (int64_t) blockIdx.x * K + (b * 64 + lane) * 4;
                float4 acc = *(const float4 *) (sk.part + e);
                for (int k = 1; k < sk.ks; ++k) {
                    const float4 p = *(const float4 *) (sk.part + (int64_t) k * sk.mn + e);
                    acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
                }
                if (sk.add) {
                    const float4 ad = *(const float4 *) (sk.add + (int64_t) blockIdx.x * sk.add_stride + (b * 64 + lane) * 4);
                    acc.x += ad.x; acc.y += ad.y; acc.z += ad.z; acc.w += ad.w;
                }
                *(float4 *) (sk.out + (int64_t) blockIdx.x * sk.out_stride + (b * 64 + lane) * 4) = acc;
                v[u] = acc;
            } else {
                v[u] = x4[b * 64 + lane];
            }
            g[u] = w4[b * 64 + lane];
        } else {
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            g[u] = v[u];
        }
    }
    double ss = 0.0;
#pragma unroll
    for (int u = 0; u < 4; ++u) ss += (double) (v[u].x * v[u].x) + (double) (v[u].y * v[u].y) + (double) (v[u].z * v[u].z) + (double) (v[u].w * v[u].w);
    ss = wave_sum_d(ss);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) tot += red[i];
    const float mean = (float) (tot / (double) K);
    const float scale = 1.0f / sqrtf(mean + eps);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int b = wave + 16 * u;
        if (b < nblk) {
            float t[4] = {(v[u].x * scale) * g[u].x, (v[u].y * scale) * g[u].y, (v[u].z * scale) * g[u].z, (v[u].w * scale) * g[u].w};
            wave_quantize_q8_K(t, lane, dst + (int64_t) blockIdx.x * nblk + b);
        }
    }
}
void launch_rms_norm_mul_quantize(hipStream_t s, const tdesc & x, const float * w, float eps, void * dst) {
    const int64_t rows = x.ne[1] * x.ne[2] * x.ne[3];
    hipLaunchKernelGGL(k_rms_norm_mul_q8_K<false>, dim3((unsigned) rows), dim3(1024), 0, s, x.data, x.nb[1], w, eps, (int) x.ne[0], (q8k_dev *) dst, splitk_src{});
}
void launch_splitk_rms_norm_mul_quantize(hipStream_t s, const splitk_src & sk, int rows, int K, const float * w, float eps, void * dst) {
    hipLaunchKernelGGL(k_rms_norm_mul_q8_K<true>, dim3((unsigned) rows), dim3(1024), 0, s, (const char *) nullptr, (int64_t) 0, w, eps, K, (q8k_dev *) dst, sk);
}

// silu(a) * b -> Q8_K: one wave per 256-value block, four blocks per workgroup
__global__ void __launch_bounds__(256) k_swiglu_q8_K(const char * __restrict__ pa, const char * __restrict__ pb, const int64_t nba1, const int64_t nbb1, const int nb_per_row,
                                                     const int64_t total, q8k_dev * __restrict__ dst) {
    const int lane = threadIdx.x & 63;
    const int64_t gid = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gid >= total) return;
    const int64_t row = gid / nb_per_row;
    const int ib = (int) (gid - row * nb_per_row);
    const float4 a = ((const float4 *) (pa + row * nba1))[ib * 64 + lane], b = ((const float4 *) (pb + row * nbb1))[ib * 64 + lane];
    float t[4] = {silu_f(a.x) * b.x, silu_f(a.y) * b.y, silu_f(a.z) * b.z, silu_f(a.w) * b.w};
    wave_quantize_q8_K(t, lane, dst + gid);
}
void launch_swiglu_quantize(hipStream_t s, const tdesc & a, const tdesc * b, int64_t nc, int swapped, void * dst) {
    const int64_t rows = a.ne[1] * a.ne[2] * a.ne[3];
    const char * pa = a.data;
    const char * pb = b ? b->data : a.data;
    if (!b) {
        pa += swapped ? nc * 4 : 0;
        pb += swapped ? 0 : nc * 4;
    }
    const int nb = (int) (nc / 256);
    const int64_t total = rows * nb;
    hipLaunchKernelGGL(k_swiglu_q8_K, dim3((unsigned) ((total + 3) / 4)), dim3(256), 0, s, pa, pb, a.nb[1], b ? b->nb[1] : a.nb[1], nb, total, (q8k_dev *) dst);
}

size_t quantized_act_bytes(int kind, int64_t K, int64_t rows) {
    if (kind == GGML_TYPE_Q8_K) return (size_t) rows * (size_t) (K / 256) * sizeof(q8k_dev);
    return (size_t) rows * (size_t) (K / 32) * sizeof(q80_dev);
}

void launch_quantize_act(hipStream_t s, int kind, const tdesc & src, void * dst) {
    const int64_t K = src.ne[0];
    const int64_t rows = src.ne[1] * src.ne[2] * src.ne[3];
    if (kind == GGML_TYPE_Q8_K) {
        const int nb = (int) (K / 256);
        hipLaunchKernelGGL(k_quantize_q8_K, dim3((unsigned) (rows * nb)), dim3(64), 0, s, src.data, src.ne[1], src.ne[2], src.nb[1], src.nb[2], src.nb[3], nb, (q8k_dev *) dst);
    } else {
        const int chunks = (int) ((K + 255) / 256);
        hipLaunchKernelGGL(k_quantize_q8_0<false>, dim3((unsigned) (rows * chunks)), dim3(64), 0, s, src.data, src.ne[1], src.ne[2], src.nb[1], src.nb[2], src.nb[3], (int) K, chunks, (q80_dev *) dst);
    }
}
void launch_quantize_q80_panel(hipStream_t s, const tdesc & src, void * dst) {
    const int64_t K = src.ne[0];
    const int64_t rows = src.ne[1] * src.ne[2] * src.ne[3];
    if (rows < 1 || (K % 32) != 0) { MI_ERR("launch_quantize_q80_panel: %lld columns / K = %lld", (long long) rows, (long long) K); abort(); }
    const int chunks = (int) ((K + 255) / 256);
    hipLaunchKernelGGL(k_quantize_q8_0<true>, dim3((unsigned) (rows * chunks)), dim3(64), 0, s, src.data, src.ne[1], src.ne[2], src.nb[1], src.nb[2], src.nb[3], (int) K, chunks, (q80_dev *) dst);
}

MI_TU_TOUCH(quantize)

}  // namespace mi355x
