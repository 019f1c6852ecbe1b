// mmf.hip — mat-mul with f16 / f32 src0 (K·Q and V·softmax on the non-flash path, small dense tensors).
//
// Restates ggml_compute_forward_mul_mat for vec_dot_type F16 / F32: when src0 is f16 the
// f32 activations are FIRST rounded to f16 (that is what ggml-cpu's from_float does to src1) and the products are
// accumulated in f32.  ggml broadcasting: src0 dims 2/3 repeat over src1's.
//
// Layout: a sub-wave group of LPR lanes owns one src0 row (LPR = 8..64 chosen so that 8-element vectors cover K),
// 64/LPR rows per wave, 4 waves per workgroup; rows are read with 16-byte loads when alignment allows.
#include <algorithm>

#include "dev_util.h"
#include "kernels.h"

namespace mi355x {

template <bool W16>
__global__ void __launch_bounds__(256) k_mul_mat_f(const tdesc a, const tdesc b, const tdesc d, const int lpr, const int vec_ok) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rpw = 64 / lpr;  // rows per wave
    const int sub = lane / lpr, sl = lane % lpr;
    const int64_t i01 = ((int64_t) blockIdx.x * 4 + wave) * rpw + sub;
    const int64_t i11 = blockIdx.y;
    const int64_t i12 = blockIdx.z % b.ne[2], i13 = blockIdx.z / b.ne[2];
    const int64_t i02 = i12 / (b.ne[2] / a.ne[2]), i03 = i13 / (b.ne[3] / a.ne[3]);
    const int64_t K = a.ne[0];
    const bool live = i01 < a.ne[1];
    const int64_t r = live ? i01 : a.ne[1] - 1;
    const char * wrow = a.data + r * a.nb[1] + i02 * a.nb[2] + i03 * a.nb[3];
    const char * xcol = b.data + i11 * b.nb[1] + i12 * b.nb[2] + i13 * b.nb[3];
    float acc = 0.0f;
    if (vec_ok) {
        for (int64_t k = (int64_t) sl * 8; k < K; k += (int64_t) lpr * 8) {
            float w[8];
            if (W16) {
                const uint4 t = *(const uint4 *) (wrow + k * 2);
                const uint32_t u[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    w[2 * i] = h2f((uint16_t) (u[i] & 0xFFFF));
                    w[2 * i + 1] = h2f((uint16_t) (u[i] >> 16));
                }
            } else {
                const float4 t0 = *(const float4 *) (wrow + k * 4), t1 = *(const float4 *) (wrow + k * 4 + 16);
                w[0] = t0.x; w[1] = t0.y; w[2] = t0.z; w[3] = t0.w; w[4] = t1.x; w[5] = t1.y; w[6] = t1.z; w[7] = t1.w;
            }
            const float4 x0 = *(const float4 *) (xcol + k * 4), x1 = *(const float4 *) (xcol + k * 4 + 16);
            float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float xv = W16 ? h2f(f2h(x[i])) : x[i];
                acc = fmaf(w[i], xv, acc);
            }
        }
    } else {
        for (int64_t k = sl; k < K; k += lpr) {
            const float wv = W16 ? h2f(*(const uint16_t *) (wrow + k * a.nb[0])) : *(const float *) (wrow + k * a.nb[0]);
            float xv = *(const float *) (xcol + k * b.nb[0]);
            if (W16) xv = h2f(f2h(xv));
            acc = fmaf(wv, xv, acc);
        }
    }
    for (int o = lpr >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (live && sl == 0) *(float *) (d.data + i01 * d.nb[0] + i11 * d.nb[1] + i12 * d.nb[2] + i13 * d.nb[3]) = acc;
}

// ---- batches (K.Q and V.softmax of a prompt chunk on the non-flash path — llama-box's default, engine_param.hpp:772-779): the same
// contract on v_mfma_f32_32x32x16_f16.  The dot kernel above reads a src0 row once per src1 column: 512 tokens against 512 cells took
// 0.64 ms per call, 41 of the 61 ms of a 512-token prefill.  Here a wave owns a 32 (src0 rows) x 32 (src1 columns) tile of one head
// and walks K in steps of 16: src0 is f16 already — 16 bytes per lane straight from memory into the B operand (n = row) —, src1 is
// rounded to f16 on the way into the A operand (m = column), exactly the rounding ggml-cpu's from_float applies; products of two
// f16 are exact in f32 and the accumulation is f32 (order differs from the CPU's, as between any two of its SIMD builds).  No LDS:
// the operand loads are row-strided (65 clocks per wave-instruction, scripts/ubench/ta_probe.hip) and that is the bound — 3 loads
// per MFMA —, which is still ~30x the dot kernel on these shapes.
typedef _Float16 mmf_half8 __attribute__((ext_vector_type(8)));
typedef float mmf_float16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256) k_mul_mat_f16_mma(const tdesc a, const tdesc b, const tdesc d) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r32 = lane & 31, g = lane >> 5;
    const int64_t row0 = ((int64_t) blockIdx.x * 2 + (wave & 1)) * 32, col0 = ((int64_t) blockIdx.y * 2 + (wave >> 1)) * 32;
    if (row0 >= a.ne[1] || col0 >= b.ne[1]) return;
    const int64_t i12 = blockIdx.z % b.ne[2], i13 = blockIdx.z / b.ne[2];
    const int64_t i02 = i12 / (b.ne[2] / a.ne[2]), i03 = i13 / (b.ne[3] / a.ne[3]);
    const int64_t K = a.ne[0];
    const char * wrow = a.data + std::min<int64_t>(row0 + r32, a.ne[1] - 1) * a.nb[1] + i02 * a.nb[2] + i03 * a.nb[3] + g * 16;  // this lane's src0 row, its k-group
    const char * xcol = b.data + std::min<int64_t>(col0 + r32, b.ne[1] - 1) * b.nb[1] + i12 * b.nb[2] + i13 * b.nb[3] + g * 32;  // this lane's src1 column
    mmf_float16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t k = 0; k < K; k += 16) {
        // k-group g of the step: elements k + 8g .. k + 8g + 7 (the last step of a K that is 8 mod 16 has no second group)
        mmf_half8 w = {0, 0, 0, 0, 0, 0, 0, 0}, x = {0, 0, 0, 0, 0, 0, 0, 0};
        if (k + 8 * g < K) {
            w = *(const mmf_half8 *) (wrow + k * 2);
            const float4 x0 = *(const float4 *) (xcol + k * 4), x1 = *(const float4 *) (xcol + k * 4 + 16);
            x = (mmf_half8){(_Float16) x0.x, (_Float16) x0.y, (_Float16) x0.z, (_Float16) x0.w, (_Float16) x1.x, (_Float16) x1.y, (_Float16) x1.z, (_Float16) x1.w};
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, w, acc, 0, 0, 0);
    }
    // lane: src0 row row0 + r32; register i: src1 column col0 + (i & 3) + 8 (i >> 2) + 4 g
    const int64_t row = row0 + r32;
    if (row >= a.ne[1]) return;
    char * out = d.data + row * d.nb[0] + i12 * d.nb[2] + i13 * d.nb[3];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int64_t col = col0 + (i & 3) + 8 * (i >> 2) + 4 * g;
        if (col < b.ne[1]) *(float *) (out + col * d.nb[1]) = acc[i];
    }
}

void launch_mul_mat_f(hipStream_t s, const tdesc & a, const tdesc & b, const tdesc & d) {
    const int64_t K = a.ne[0];
    const bool w16 = a.type == GGML_TYPE_F16;
    const int esz = w16 ? 2 : 4;
    // 16-byte vector path: contiguous dim 0 on both sides, K multiple of 8, every row start 16-byte aligned
    bool vec_ok = a.nb[0] == esz && b.nb[0] == 4 && (K % 8) == 0 && (((uintptr_t) a.data) & 15) == 0 && (((uintptr_t) b.data) & 15) == 0;
    for (int i = 1; i < 4; ++i) vec_ok = vec_ok && (a.nb[i] % 16) == 0 && (b.nb[i] % 16) == 0;
    if (w16 && vec_ok && b.ne[1] >= 16) {  // a batch of columns: the matrix cores
        dim3 grid((unsigned) ((a.ne[1] + 63) / 64), (unsigned) ((b.ne[1] + 63) / 64), (unsigned) (b.ne[2] * b.ne[3]));
        hipLaunchKernelGGL(k_mul_mat_f16_mma, grid, dim3(256), 0, s, a, b, d);
        return;
    }
    int lpr = 64;
    if (vec_ok) {
        while (lpr > 8 && (int64_t) (lpr / 2) * 8 >= K) lpr >>= 1;
    } else {
        while (lpr > 8 && (int64_t) (lpr / 2) >= K) lpr >>= 1;
    }
    const int rows_per_block = 4 * (64 / lpr);
    dim3 grid((unsigned) ((a.ne[1] + rows_per_block - 1) / rows_per_block), (unsigned) b.ne[1], (unsigned) (b.ne[2] * b.ne[3]));
    if (w16) hipLaunchKernelGGL(k_mul_mat_f<true>, grid, dim3(256), 0, s, a, b, d, lpr, vec_ok ? 1 : 0);
    else hipLaunchKernelGGL(k_mul_mat_f<false>, grid, dim3(256), 0, s, a, b, d, lpr, vec_ok ? 1 : 0);
}

}  // namespace mi355x
