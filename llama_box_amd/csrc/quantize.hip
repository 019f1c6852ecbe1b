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
    q80_dev * y = dst + row * (int64_t) (K / 32) + (e0 >> 5);
    uint32_t packed = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = (int) roundf(v[k] * id);
        packed |= (uint32_t) (r & 0xFF) << (8 * k);
    }
    ((uint32_t *) y->qs)[lane & 7] = packed;
    if ((lane & 7) == 0) y->d = h2f(f2h(d));
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
        hipLaunchKernelGGL(k_quantize_q8_0, dim3((unsigned) (rows * chunks)), dim3(64), 0, s, src.data, src.ne[1], src.ne[2], src.nb[1], src.nb[2], src.nb[3], (int) K, chunks, (q80_dev *) dst);
    }
}

}  // namespace mi355x
