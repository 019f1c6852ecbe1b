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
    for (int64_t k = 0; k < K; k += 64) {  // 4 steps of 16 per trip, their loads in flight together
        // k-group g of a step: elements kk + 8g .. kk + 8g + 7; past the end (the last step of a K that is 8 mod 16 has no second group): the
        // row's first group is fetched instead — no branch between the loads — and masked to zeros
        uint4 w[4], x0[4], x1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t kk = k + 16 * u;
            const bool in = kk + 8 * g < K;
            const int64_t kc = in ? kk : -8 * g;
            const uint32_t keep = in ? 0xFFFFFFFFu : 0u;
            w[u] = *(const uint4 *) (wrow + kc * 2);
            x0[u] = *(const uint4 *) (xcol + kc * 4);
            x1[u] = *(const uint4 *) (xcol + kc * 4 + 16);
            w[u].x &= keep; w[u].y &= keep; w[u].z &= keep; w[u].w &= keep;
            x0[u].x &= keep; x0[u].y &= keep; x0[u].z &= keep; x0[u].w &= keep;
            x1[u].x &= keep; x1[u].y &= keep; x1[u].z &= keep; x1[u].w &= keep;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const mmf_half8 x = {(_Float16) __builtin_bit_cast(float, x0[u].x), (_Float16) __builtin_bit_cast(float, x0[u].y), (_Float16) __builtin_bit_cast(float, x0[u].z),
                                 (_Float16) __builtin_bit_cast(float, x0[u].w), (_Float16) __builtin_bit_cast(float, x1[u].x), (_Float16) __builtin_bit_cast(float, x1[u].y),
                                 (_Float16) __builtin_bit_cast(float, x1[u].z), (_Float16) __builtin_bit_cast(float, x1[u].w)};
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, __builtin_bit_cast(mmf_half8, w[u]), acc, 0, 0, 0);
        }
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

// ---- short rows against a batch (K.Q on the non-flash path: K = head size, rows = cells): the tile kernel above fetches 16 KB of src1
// and 8 KB of src0 per 32 x 32 tile — 205 MB through the L2 for the 32 x 32 x 7300 logits of a -np 32 step, 39 us.  Here a wave
// keeps its 32 columns of GH heads that share the src0 rows (grouped queries) in registers as f16 A operands — the whole K, at most
// 16 NK — and walks RT row tiles: 8 KB of src0 per tile feed GH x NK MFMAs, the next tile's rows are requested before the current
// one is multiplied.
template <int GH, int NK>
__global__ void __launch_bounds__(256, 2) k_mul_mat_f16_mma_xres(const tdesc a, const tdesc b, const tdesc d, const int RT) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r32 = lane & 31, g = lane >> 5;
    const int64_t K = a.ne[0], col0 = (int64_t) blockIdx.y * 32;
    const int64_t hg = blockIdx.z % (b.ne[2] / GH), i13 = blockIdx.z / (b.ne[2] / GH);  // group of GH src1 heads with one src0 head
    const int64_t i02 = hg * GH / (b.ne[2] / a.ne[2]), i03 = i13 / (b.ne[3] / a.ne[3]);
    const int64_t tile0 = ((int64_t) blockIdx.x * 4 + wave) * RT, n_tiles = (a.ne[1] + 31) / 32;
    if (tile0 >= n_tiles) return;
    // k-group g of step s: elements 16 s + 8 g ..; past K: the first group is fetched instead and masked to zeros
    mmf_half8 xs[GH][NK];
#pragma unroll
    for (int h = 0; h < GH; ++h) {
        const char * xcol = b.data + std::min<int64_t>(col0 + r32, b.ne[1] - 1) * b.nb[1] + (hg * GH + h) * b.nb[2] + i13 * b.nb[3];
#pragma unroll
        for (int st = 0; st < NK; ++st) {
            const int64_t kk = 16 * st + 8 * g;
            const bool in = kk < K;
            const uint32_t keep = in ? 0xFFFFFFFFu : 0u;
            uint4 p0 = *(const uint4 *) (xcol + (in ? kk : 0) * 4), p1 = *(const uint4 *) (xcol + (in ? kk : 0) * 4 + 16);
            p0.x &= keep; p0.y &= keep; p0.z &= keep; p0.w &= keep; p1.x &= keep; p1.y &= keep; p1.z &= keep; p1.w &= keep;
            xs[h][st] = (mmf_half8){(_Float16) __builtin_bit_cast(float, p0.x), (_Float16) __builtin_bit_cast(float, p0.y), (_Float16) __builtin_bit_cast(float, p0.z),
                                    (_Float16) __builtin_bit_cast(float, p0.w), (_Float16) __builtin_bit_cast(float, p1.x), (_Float16) __builtin_bit_cast(float, p1.y),
                                    (_Float16) __builtin_bit_cast(float, p1.z), (_Float16) __builtin_bit_cast(float, p1.w)};
        }
    }
    const char * abase = a.data + i02 * a.nb[2] + i03 * a.nb[3];
    auto fetch = [&](const int64_t tile, uint4 (&w)[NK]) {
        const char * wrow = abase + std::min<int64_t>(tile * 32 + r32, a.ne[1] - 1) * a.nb[1];
#pragma unroll
        for (int st = 0; st < NK; ++st) {
            const int64_t kk = 16 * st + 8 * g;
            const bool in = kk < K;
            const uint32_t keep = in ? 0xFFFFFFFFu : 0u;
            w[st] = *(const uint4 *) (wrow + (in ? kk : 0) * 2);
            w[st].x &= keep; w[st].y &= keep; w[st].z &= keep; w[st].w &= keep;
        }
    };
    uint4 w[NK], wn[NK];
    fetch(tile0, w);
    const int64_t tile_end = std::min<int64_t>(n_tiles, tile0 + RT);
    for (int64_t tile = tile0; tile < tile_end; ++tile) {
        if (tile + 1 < tile_end) fetch(tile + 1, wn);
        const int64_t row = tile * 32 + r32;
#pragma unroll
        for (int h = 0; h < GH; ++h) {
            mmf_float16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int st = 0; st < NK; ++st) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xs[h][st], __builtin_bit_cast(mmf_half8, w[st]), acc, 0, 0, 0);
            if (row < a.ne[1]) {
                char * out = d.data + row * d.nb[0] + (hg * GH + h) * d.nb[2] + i13 * d.nb[3];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int64_t col = col0 + (i & 3) + 8 * (i >> 2) + 4 * g;
                    if (col < b.ne[1]) *(float *) (out + col * d.nb[1]) = acc[i];
                }
            }
        }
#pragma unroll
        for (int st = 0; st < NK; ++st) w[st] = wn[st];
    }
}
template <int GH> static void launch_xres_t(hipStream_t s, const tdesc & a, const tdesc & b, const tdesc & d) {
    const int64_t n_tiles = (a.ne[1] + 31) / 32, ct = (b.ne[1] + 31) / 32, nz = (b.ne[2] / GH) * b.ne[3];
    const int RT = (int) std::max<int64_t>(1, std::min<int64_t>(8, n_tiles * ct * nz / 512));
    dim3 grid((unsigned) ((n_tiles + 4 * RT - 1) / (4 * RT)), (unsigned) ct, (unsigned) nz);
    if (a.ne[0] <= 64) hipLaunchKernelGGL((k_mul_mat_f16_mma_xres<GH, 4>), grid, dim3(256), 0, s, a, b, d, RT);
    else hipLaunchKernelGGL((k_mul_mat_f16_mma_xres<GH, 8>), grid, dim3(256), 0, s, a, b, d, RT);
}

// ---- the same contract for few output tiles and long rows (V^T.p of a -np decode batch on the non-flash path: 128 x 32 results per head
// over thousands of cells — 128 waves of the kernel above, each walking all of K, took 170 us), and for 2..15 columns: 16 x 16 tiles
// on v_mfma_f32_16x16x32_f16 (A: src1 column m = lane & 15, B: src0 row n = lane & 15, k-group lane >> 4), the NWK waves of a
// workgroup take the K steps of ONE tile in turn and add up through LDS in wave order (deterministic).
typedef float mmf_float4 __attribute__((ext_vector_type(4)));
template <int NWK>
__global__ void __launch_bounds__(64 * NWK) k_mul_mat_f16_mma16(const tdesc a, const tdesc b, const tdesc d) {
    __shared__ mmf_float4 red[NWK][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
    const int64_t row0 = (int64_t) blockIdx.x * 16, col0 = (int64_t) blockIdx.y * 16;
    const int64_t i12 = blockIdx.z % b.ne[2], i13 = blockIdx.z / b.ne[2];
    const int64_t i02 = i12 / (b.ne[2] / a.ne[2]), i03 = i13 / (b.ne[3] / a.ne[3]);
    const int64_t K = a.ne[0];
    const char * wrow = a.data + std::min<int64_t>(row0 + r16, a.ne[1] - 1) * a.nb[1] + i02 * a.nb[2] + i03 * a.nb[3] + kg * 16;
    const char * xcol = b.data + std::min<int64_t>(col0 + r16, b.ne[1] - 1) * b.nb[1] + i12 * b.nb[2] + i13 * b.nb[3] + kg * 32;
    mmf_float4 acc = {0, 0, 0, 0};
    for (int64_t k = (int64_t) wave * 32; k < K; k += 32 * NWK * 4) {  // 4 steps' loads in flight
        uint4 w[4], x0[4], x1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t kk = k + (int64_t) u * 32 * NWK;
            const bool in = kk + 8 * kg < K;  // past the end: fetch the row's first group instead (no branch between the loads) and mask it to zeros
            const int64_t kc = in ? kk : -8 * kg;
            const uint32_t keep = in ? 0xFFFFFFFFu : 0u;
            w[u] = *(const uint4 *) (wrow + kc * 2);
            x0[u] = *(const uint4 *) (xcol + kc * 4);
            x1[u] = *(const uint4 *) (xcol + kc * 4 + 16);
            w[u].x &= keep; w[u].y &= keep; w[u].z &= keep; w[u].w &= keep;
            x0[u].x &= keep; x0[u].y &= keep; x0[u].z &= keep; x0[u].w &= keep;
            x1[u].x &= keep; x1[u].y &= keep; x1[u].z &= keep; x1[u].w &= keep;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const mmf_half8 x = {(_Float16) __builtin_bit_cast(float, x0[u].x), (_Float16) __builtin_bit_cast(float, x0[u].y), (_Float16) __builtin_bit_cast(float, x0[u].z),
                                 (_Float16) __builtin_bit_cast(float, x0[u].w), (_Float16) __builtin_bit_cast(float, x1[u].x), (_Float16) __builtin_bit_cast(float, x1[u].y),
                                 (_Float16) __builtin_bit_cast(float, x1[u].z), (_Float16) __builtin_bit_cast(float, x1[u].w)};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, __builtin_bit_cast(mmf_half8, w[u]), acc, 0, 0, 0);
        }
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 1; w < NWK; ++w) acc += red[w][lane];
    // lane: src0 row row0 + r16; register i: src1 column col0 + 4 kg + i
    const int64_t row = row0 + r16;
    if (row >= a.ne[1]) return;
    char * out = d.data + row * d.nb[0] + i12 * d.nb[2] + i13 * d.nb[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t col = col0 + 4 * kg + i;
        if (col < b.ne[1]) *(float *) (out + col * d.nb[1]) = acc[i];
    }
}

// ---- ... and when scratch is at hand (few columns, at most 128 rows, long rows — V^T.p of a -np decode batch): 16 x 16 tiles fetch
// every probability 8 times and every V^T element twice — 360 MB through the L2 for a 15 MB cache, 45 us.  Here a workgroup's four
// waves cover ALL rows (32 each) of one head over a slice of K, so the probabilities of the slice cross the L2 once (the waves read
// the same lines together), K is cut over workgroups, the partial tiles go to scratch ([slice][batch][column][row]) and
// launch_splitk_reduce adds them in slice order.
template <int NCT>
__global__ void __launch_bounds__(256) k_mul_mat_f16_mma_ksplit(const tdesc a, const tdesc b, float * __restrict__ part, const int64_t Kc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r32 = lane & 31, g = lane >> 5;
    const int64_t row0 = (int64_t) wave * 32;
    if (row0 >= a.ne[1]) return;
    const int64_t i12 = blockIdx.y % b.ne[2], i13 = blockIdx.y / b.ne[2];
    const int64_t i02 = i12 / (b.ne[2] / a.ne[2]), i03 = i13 / (b.ne[3] / a.ne[3]);
    const int64_t kb = (int64_t) blockIdx.x * Kc, ke = std::min<int64_t>(a.ne[0], kb + Kc);
    const char * wrow = a.data + std::min<int64_t>(row0 + r32, a.ne[1] - 1) * a.nb[1] + i02 * a.nb[2] + i03 * a.nb[3] + g * 16;
    const char * xcol[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) xcol[ct] = b.data + std::min<int64_t>(ct * 32 + r32, b.ne[1] - 1) * b.nb[1] + i12 * b.nb[2] + i13 * b.nb[3] + g * 32;
    mmf_float16 acc[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) acc[ct] = (mmf_float16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t k = kb; k < ke; k += 64) {
        uint4 w[4], x0[4][NCT], x1[4][NCT];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t kk = k + 16 * u;
            const bool in = kk + 8 * g < ke;
            const int64_t kc = in ? kk : -8 * g;
            const uint32_t keep = in ? 0xFFFFFFFFu : 0u;
            w[u] = *(const uint4 *) (wrow + kc * 2);
            w[u].x &= keep; w[u].y &= keep; w[u].z &= keep; w[u].w &= keep;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                x0[u][ct] = *(const uint4 *) (xcol[ct] + kc * 4);
                x1[u][ct] = *(const uint4 *) (xcol[ct] + kc * 4 + 16);
                x0[u][ct].x &= keep; x0[u][ct].y &= keep; x0[u][ct].z &= keep; x0[u][ct].w &= keep;
                x1[u][ct].x &= keep; x1[u][ct].y &= keep; x1[u][ct].z &= keep; x1[u][ct].w &= keep;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const uint4 p0 = x0[u][ct], p1 = x1[u][ct];
                const mmf_half8 x = {(_Float16) __builtin_bit_cast(float, p0.x), (_Float16) __builtin_bit_cast(float, p0.y), (_Float16) __builtin_bit_cast(float, p0.z),
                                     (_Float16) __builtin_bit_cast(float, p0.w), (_Float16) __builtin_bit_cast(float, p1.x), (_Float16) __builtin_bit_cast(float, p1.y),
                                     (_Float16) __builtin_bit_cast(float, p1.z), (_Float16) __builtin_bit_cast(float, p1.w)};
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, __builtin_bit_cast(mmf_half8, w[u]), acc[ct], 0, 0, 0);
            }
    }
    const int64_t row = row0 + r32;
    if (row >= a.ne[1]) return;
    float * out = part + ((int64_t) blockIdx.x * gridDim.y + blockIdx.y) * b.ne[1] * a.ne[1] + row;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int64_t col = ct * 32 + (i & 3) + 8 * (i >> 2) + 4 * g;
            if (col < b.ne[1]) out[col * a.ne[1]] = acc[ct][i];
        }
}
// slices of K for the form above (0: the form does not apply)
static int mmf_ksplit_slices(const tdesc & a, const tdesc & b, int64_t * Kc_out) {
    const int64_t K = a.ne[0], nbatch = b.ne[2] * b.ne[3];
    if (a.type != GGML_TYPE_F16 || b.ne[1] < 2 || b.ne[1] > 64 || a.ne[1] > 128 || (a.ne[1] % 4) != 0 || K < 2048 || b.ne[2] % a.ne[2] != 0 || b.ne[3] % a.ne[3] != 0) return 0;
    const int64_t want = std::max<int64_t>(1, std::min<int64_t>(K / 256, 512 / std::max<int64_t>(1, nbatch)));
    const int64_t Kc = ((K + want - 1) / want + 63) / 64 * 64;
    if (Kc_out) *Kc_out = Kc;
    return (int) ((K + Kc - 1) / Kc);
}
size_t mul_mat_f_workspace_bytes(const tdesc & a, const tdesc & b) {
    const int ks = mmf_ksplit_slices(a, b, nullptr);
    return ks <= 1 ? 0 : (size_t) ks * (size_t) (b.ne[2] * b.ne[3] * b.ne[1] * a.ne[1]) * sizeof(float);
}

// ---- one column against an f16 src0 that G heads of src1 share (K.q and V^T.p of a decode step on the non-flash path with grouped
// queries: a.ne[2] KV heads, b.ne[2] = G a.ne[2] query heads).  The dot kernel above runs one workgroup per (rows, query head): every
// src0 row is fetched G times and each wave waits out one or two dependent loads — 10 us per call at 2k cells, for 4.7 MB.  Here a
// row owner fetches its slice of the row ONCE and feeds G accumulators (src1 is rounded to f16 first, as from_float does), with
// several rows' loads in flight:
//   short rows (K <= 512, K.q: rows = cells): LPR lanes own a row, 4 rows per owner, 64 x 4 / LPR rows per wave;
//   long rows (V^T.p: K = cells): a wave owns 4 rows, the 4 waves of the workgroup take K in interleaved 512-element slices and
//   add up through LDS.
// f32 accumulation in a different order than the dot kernel (and the CPU); same NMSE gate (tests/test_gpu_ops.py::test_mul_mat_f).
__device__ __forceinline__ void mmf_w8(const char * p, float (&w)[8]) {
    const uint4 t = *(const uint4 *) p;
    const uint32_t u[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        w[2 * i] = h2f((uint16_t) (u[i] & 0xFFFF));
        w[2 * i + 1] = h2f((uint16_t) (u[i] >> 16));
    }
}
__device__ __forceinline__ void mmf_x8(const char * p, float (&x)[8]) {
    const float4 x0 = *(const float4 *) p, x1 = *(const float4 *) (p + 16);
    const float t[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = h2f(f2h(t[i]));
}
// sum over each group of LPR consecutive lanes, on the DPP path (the 64 ds_bpermute a shuffle butterfly needs here — 16 sums, 4
// dependent steps each — were 3.7 of the kernel's 7.8 us: scripts/ubench/attn_probe.hip)
template <int LPR> __device__ __forceinline__ float group_sum(float v) {
    if (LPR == 64) return wave_sum(v);
    if (LPR >= 16) {
        v = row16_sum(v);
        if (LPR == 32) v += __shfl_xor(v, 16, 64);
        return v;
    }
    v += dpp_f32<MI_DPP_QUAD_XOR1>(v);
    v += dpp_f32<MI_DPP_QUAD_XOR2>(v);
    v += dpp_f32<MI_DPP_HALF_MIRROR>(v);
    return v;
}
template <int G, int LPR>
__global__ void __launch_bounds__(256) k_mul_mat_f16_gqa_short(const tdesc a, const tdesc b, const tdesc d) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int lpr = LPR;
    const int rpw = 64 / lpr, sub = lane / lpr, sl = lane % lpr;
    const int64_t i02 = blockIdx.y, K = a.ne[0];
    const int64_t base = (int64_t) blockIdx.x * 16 * rpw + wave * rpw + sub;  // owner's rows: base + 4 rpw r
    const bool kin = (int64_t) sl * 8 < K;
    float x[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (kin) mmf_x8(b.data + (i02 * G + g) * b.nb[2] + (int64_t) sl * 32, x[g]);
        else
#pragma unroll
            for (int i = 0; i < 8; ++i) x[g][i] = 0.0f;
    }
    float w[4][8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t row = std::min<int64_t>(base + 4 * rpw * r, a.ne[1] - 1);
        if (kin) mmf_w8(a.data + row * a.nb[1] + i02 * a.nb[2] + (int64_t) sl * 16, w[r]);
        else
#pragma unroll
            for (int i = 0; i < 8; ++i) w[r][i] = 0.0f;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t row = base + 4 * rpw * r;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc = fmaf(w[r][i], x[g][i], acc);
            acc = group_sum<LPR>(acc);
            if (sl == 0 && row < a.ne[1]) *(float *) (d.data + row * d.nb[0] + (i02 * G + g) * d.nb[2]) = acc;
        }
    }
}
template <int G>
__global__ void __launch_bounds__(256) k_mul_mat_f16_gqa_long(const tdesc a, const tdesc b, const tdesc d) {
    __shared__ float part[4][4 * G];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i02 = blockIdx.y, K = a.ne[0], row0 = (int64_t) blockIdx.x * 4;
    const char * wr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) wr[r] = a.data + std::min<int64_t>(row0 + r, a.ne[1] - 1) * a.nb[1] + i02 * a.nb[2];
    float acc[4][G];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int g = 0; g < G; ++g) acc[r][g] = 0.0f;
    for (int64_t k = (int64_t) wave * 512 + lane * 8; k < K; k += 2048) {
        float w[4][8], x[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) mmf_w8(wr[r] + k * 2, w[r]);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            mmf_x8(b.data + (i02 * G + g) * b.nb[2] + k * 4, x);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[r][g] = fmaf(w[r][i], x[i], acc[r][g]);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float v = wave_sum(acc[r][g]);
            if (lane == 0) part[wave][r * G + g] = v;
        }
    __syncthreads();
    if (threadIdx.x < 4 * G) {
        const int r = threadIdx.x / G, g = threadIdx.x % G;
        const float v = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x];
        if (row0 + r < a.ne[1]) *(float *) (d.data + (row0 + r) * d.nb[0] + (i02 * G + g) * d.nb[2]) = v;
    }
}
// ---- SOFT_MAX(K.q) folded into V^T.p (a decode step on the non-flash path): the long-row layout above with 8 waves, whose 512 threads
// x 8 cells tile the whole row.  The workgroup first turns its slices of the G logit rows into probabilities —
// ggml_compute_forward_soft_max_f32 as ops.hip restates it: w = x scale + mask, max, e = expf(w - max), sum in double, llama-box's
// zero-sum guard, p = e (1 / sum) — and feeds them (rounded to f16, as from_float would) to the products.  One launch and one round
// trip of the probabilities less; every row group of a KV head repeats the G rows' exponentials (32 x for 128 dims / 4), so the work
// per thread is kept to one or two 8-cell slices, the logits / exponentials stay in registers between the passes when the row is
// short enough (NIT slices of 4096 cells; NIT = 0: re-read and recomputed), and the V^T slices are requested before the statistics
// are made.  The per-thread partial sums cover other cells than soft_max's own kernel; they are doubles, the float reciprocal is
// the same in all but freak cases.  scripts/ubench/attn_probe.hip: 6.7 + 5.6 us as two launches at 2304 cells.
#define SMM_NW 8
template <int G, int NIT>
__global__ void __launch_bounds__(64 * SMM_NW) k_soft_max_mul_mat_f16(const tdesc a, const tdesc kq, const tdesc m, const int has_mask, const tdesc d, const float scale) {
    __shared__ float shm[SMM_NW][G];
    __shared__ double shs[SMM_NW][G];
    __shared__ float part[SMM_NW][4 * G];
    constexpr int64_t STEP = 512 * SMM_NW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i02 = blockIdx.y, K = a.ne[0], row0 = (int64_t) blockIdx.x * 4;
    const bool m16 = m.type == GGML_TYPE_F16;
    const int64_t k0 = (int64_t) wave * 512 + lane * 8;
    auto logits = [&](const int g, const int64_t k, float (&w)[8]) {
        const int64_t h = i02 * G + g;
        const char * xp = kq.data + h * kq.nb[2] + k * 4;
        const float4 x0 = *(const float4 *) xp, x1 = *(const float4 *) (xp + 16);
        const float t[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        if (has_mask) {
            const char * mp = m.data + (h % m.ne[2]) * m.nb[2];
            float mv[8];
            if (m16) mmf_w8(mp + k * 2, mv);
            else {
                const float4 m0 = *(const float4 *) (mp + k * 4), m1 = *(const float4 *) (mp + k * 4 + 16);
                mv[0] = m0.x; mv[1] = m0.y; mv[2] = m0.z; mv[3] = m0.w; mv[4] = m1.x; mv[5] = m1.y; mv[6] = m1.z; mv[7] = m1.w;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float v = t[i] * scale;
                v += mv[i];
                w[i] = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) w[i] = t[i] * scale;
        }
    };
    const char * wr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) wr[r] = a.data + std::min<int64_t>(row0 + r, a.ne[1] - 1) * a.nb[1] + i02 * a.nb[2];
    constexpr int NC = NIT > 0 ? NIT : 1;
    uint4 vq[NC][4];  // this thread's slices of the 4 V^T rows (NIT > 0)
    if (NIT > 0) {
#pragma unroll
        for (int it = 0; it < NC; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) vq[it][r] = k0 + STEP * it < K ? *(const uint4 *) (wr[r] + (k0 + STEP * it) * 2) : make_uint4(0, 0, 0, 0);
    }
    float e[NC][G][8];  // logits, then exponentials (NIT > 0)
    float mx[G];
#pragma unroll
    for (int g = 0; g < G; ++g) mx[g] = -INFINITY;
    if (NIT > 0) {
#pragma unroll
        for (int it = 0; it < NC; ++it)
            if (k0 + STEP * it < K)
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    logits(g, k0 + STEP * it, e[it][g]);
#pragma unroll
                    for (int i = 0; i < 8; ++i) mx[g] = fmaxf(mx[g], e[it][g][i]);
                }
    } else {
        for (int64_t k = k0; k < K; k += STEP)
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float w[8];
                logits(g, k, w);
#pragma unroll
                for (int i = 0; i < 8; ++i) mx[g] = fmaxf(mx[g], w[i]);
            }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const float v = wave_max(mx[g]);
        if (lane == 0) shm[wave][g] = v;
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float v = shm[0][g];
#pragma unroll
        for (int w = 1; w < SMM_NW; ++w) v = fmaxf(v, shm[w][g]);
        mx[g] = v;
    }
    double sum[G];
#pragma unroll
    for (int g = 0; g < G; ++g) sum[g] = 0.0;
    if (NIT > 0) {
#pragma unroll
        for (int it = 0; it < NC; ++it)
            if (k0 + STEP * it < K)
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        e[it][g][i] = expf(e[it][g][i] - mx[g]);
                        sum[g] += (double) e[it][g][i];
                    }
    } else {
        for (int64_t k = k0; k < K; k += STEP)
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float w[8];
                logits(g, k, w);
#pragma unroll
                for (int i = 0; i < 8; ++i) sum[g] += (double) expf(w[i] - mx[g]);
            }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const double v = wave_sum_d(sum[g]);
        if (lane == 0) shs[wave][g] = v;
    }
    __syncthreads();
    float inv[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        double t = shs[0][g];
#pragma unroll
        for (int w = 1; w < SMM_NW; ++w) t += shs[w][g];
        if (isnan(t) || t == 0.0) t = -INFINITY;
        inv[g] = (float) (1.0 / t);
    }
    float acc[4][G];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int g = 0; g < G; ++g) acc[r][g] = 0.0f;
    auto product = [&](const int64_t k, const int it) {
        float w[4][8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (NIT > 0) {
                const uint32_t u[4] = {vq[it][r].x, vq[it][r].y, vq[it][r].z, vq[it][r].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    w[r][2 * i] = h2f((uint16_t) (u[i] & 0xFFFF));
                    w[r][2 * i + 1] = h2f((uint16_t) (u[i] >> 16));
                }
            } else mmf_w8(wr[r] + k * 2, w[r]);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float x[8];
            if (NIT > 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = h2f(f2h(e[it][g][i] * inv[g]));
            } else {
                logits(g, k, x);
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = h2f(f2h(expf(x[i] - mx[g]) * inv[g]));
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[r][g] = fmaf(w[r][i], x[i], acc[r][g]);
        }
    };
    if (NIT > 0) {
#pragma unroll
        for (int it = 0; it < NC; ++it)
            if (k0 + STEP * it < K) product(k0 + STEP * it, it);
    } else {
        for (int64_t k = k0; k < K; k += STEP) product(k, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float v = wave_sum(acc[r][g]);
            if (lane == 0) part[wave][r * G + g] = v;
        }
    __syncthreads();
    if (threadIdx.x < 4 * G) {
        const int r = threadIdx.x / G, g = threadIdx.x % G;
        float v = part[0][threadIdx.x];
#pragma unroll
        for (int w = 1; w < SMM_NW; ++w) v += part[w][threadIdx.x];
        if (row0 + r < a.ne[1]) *(float *) (d.data + (row0 + r) * d.nb[0] + (i02 * G + g) * d.nb[2]) = v;
    }
}
template <int G, int MAXC>
static void launch_smm_t(hipStream_t s, dim3 grid, int nit, const tdesc & a, const tdesc & kq, const tdesc & md, int has_mask, const tdesc & d, float scale) {
#define SMM(NIT_) hipLaunchKernelGGL((k_soft_max_mul_mat_f16<G, NIT_>), grid, dim3(64 * SMM_NW), 0, s, a, kq, md, has_mask, d, scale)
    if (nit == 1) { SMM(1); return; }
    if constexpr (MAXC >= 2) if (nit == 2) { SMM(2); return; }
    if constexpr (MAXC >= 4) if (nit <= 4) { SMM(4); return; }
    SMM(0);
#undef SMM
}
static bool mmf_vec_ok(const tdesc & a, const tdesc & b) {
    const int64_t K = a.ne[0];
    bool ok = a.type == GGML_TYPE_F16 && a.nb[0] == 2 && b.nb[0] == 4 && (K % 8) == 0 && (((uintptr_t) a.data) & 15) == 0 && (((uintptr_t) b.data) & 15) == 0;
    for (int i = 1; i < 4; ++i) ok = ok && (a.nb[i] % 16) == 0 && (b.nb[i] % 16) == 0;
    return ok;
}
// a = the f16 matrix (V^T view), kq = the logits SOFT_MAX reads, d = the product; false: shapes this kernel does not serve (the caller
// runs SOFT_MAX and MUL_MAT on their own)
bool launch_soft_max_mul_mat_f16(hipStream_t s, const tdesc & a, const tdesc & kq, const tdesc * mask, const tdesc & d, const float scale) {
    const int64_t K = a.ne[0];
    if (!mmf_vec_ok(a, kq) || kq.ne[0] != K || kq.ne[1] != 1 || kq.ne[3] != 1 || a.ne[3] != 1 || kq.ne[2] % a.ne[2] != 0 || d.nb[0] != 4) return false;
    if (mask) {
        const int esz = mask->type == GGML_TYPE_F16 ? 2 : 4;
        if ((mask->type != GGML_TYPE_F16 && mask->type != GGML_TYPE_F32) || mask->nb[0] != esz || mask->ne[0] < K || mask->ne[3] != 1 || (((uintptr_t) mask->data) & 15) || (mask->nb[2] % 16)) return false;
    }
    const int G = (int) (kq.ne[2] / a.ne[2]);
    const int nit = (int) ((K + 512 * SMM_NW - 1) / (512 * SMM_NW));
    const tdesc md = mask ? *mask : kq;
    dim3 grid((unsigned) ((a.ne[1] + 3) / 4), (unsigned) a.ne[2]);
    switch (G) {  // cached logits / exponentials: at most 64 registers
        case 1: launch_smm_t<1, 4>(s, grid, nit, a, kq, md, mask ? 1 : 0, d, scale); return true;
        case 2: launch_smm_t<2, 4>(s, grid, nit, a, kq, md, mask ? 1 : 0, d, scale); return true;
        case 3: launch_smm_t<3, 2>(s, grid, nit, a, kq, md, mask ? 1 : 0, d, scale); return true;
        case 4: launch_smm_t<4, 2>(s, grid, nit, a, kq, md, mask ? 1 : 0, d, scale); return true;
        case 5: launch_smm_t<5, 1>(s, grid, nit, a, kq, md, mask ? 1 : 0, d, scale); return true;
        case 6: launch_smm_t<6, 1>(s, grid, nit, a, kq, md, mask ? 1 : 0, d, scale); return true;
        case 7: launch_smm_t<7, 1>(s, grid, nit, a, kq, md, mask ? 1 : 0, d, scale); return true;
        case 8: launch_smm_t<8, 1>(s, grid, nit, a, kq, md, mask ? 1 : 0, d, scale); return true;
        default: break;
    }
    return false;
}

template <int G> static void launch_gqa_t(hipStream_t s, const tdesc & a, const tdesc & b, const tdesc & d) {
    const int64_t K = a.ne[0];
    if (K <= 512) {
        int lpr = 64;
        while (lpr > 8 && (int64_t) (lpr / 2) * 8 >= K) lpr >>= 1;
        const int rows_per_block = 16 * (64 / lpr);
        dim3 grid((unsigned) ((a.ne[1] + rows_per_block - 1) / rows_per_block), (unsigned) a.ne[2]);
        switch (lpr) {
            case 8: hipLaunchKernelGGL((k_mul_mat_f16_gqa_short<G, 8>), grid, dim3(256), 0, s, a, b, d); break;
            case 16: hipLaunchKernelGGL((k_mul_mat_f16_gqa_short<G, 16>), grid, dim3(256), 0, s, a, b, d); break;
            case 32: hipLaunchKernelGGL((k_mul_mat_f16_gqa_short<G, 32>), grid, dim3(256), 0, s, a, b, d); break;
            default: hipLaunchKernelGGL((k_mul_mat_f16_gqa_short<G, 64>), grid, dim3(256), 0, s, a, b, d); break;
        }
    } else {
        dim3 grid((unsigned) ((a.ne[1] + 3) / 4), (unsigned) a.ne[2]);
        hipLaunchKernelGGL(k_mul_mat_f16_gqa_long<G>, grid, dim3(256), 0, s, a, b, d);
    }
}

void launch_mul_mat_f(hipStream_t s, const tdesc & a, const tdesc & b, const tdesc & d, float * ws, size_t ws_bytes) {
    const int64_t K = a.ne[0];
    const bool w16 = a.type == GGML_TYPE_F16;
    const int esz = w16 ? 2 : 4;
    // 16-byte vector path: contiguous dim 0 on both sides, K multiple of 8, every row start 16-byte aligned
    bool vec_ok = a.nb[0] == esz && b.nb[0] == 4 && (K % 8) == 0 && (((uintptr_t) a.data) & 15) == 0 && (((uintptr_t) b.data) & 15) == 0;
    for (int i = 1; i < 4; ++i) vec_ok = vec_ok && (a.nb[i] % 16) == 0 && (b.nb[i] % 16) == 0;
    if (w16 && vec_ok && ws && d.nb[0] == 4 && d.nb[1] == d.ne[0] * 4 && d.nb[2] == d.nb[1] * d.ne[1] && d.nb[3] == d.nb[2] * d.ne[2]) {
        int64_t Kc = 0;
        const int ks = mmf_ksplit_slices(a, b, &Kc);
        if (ks > 1 && mul_mat_f_workspace_bytes(a, b) <= ws_bytes) {
            const int64_t nbatch = b.ne[2] * b.ne[3];
            dim3 grid((unsigned) ks, (unsigned) nbatch);
            if (b.ne[1] <= 32) hipLaunchKernelGGL(k_mul_mat_f16_mma_ksplit<1>, grid, dim3(256), 0, s, a, b, ws, Kc);
            else hipLaunchKernelGGL(k_mul_mat_f16_mma_ksplit<2>, grid, dim3(256), 0, s, a, b, ws, Kc);
            launch_splitk_reduce(s, ws, ks, (int) (nbatch * b.ne[1]), (int) a.ne[1], (float *) d.data, a.ne[1], nullptr, 0);
            return;
        }
    }
    if (w16 && vec_ok && b.ne[1] >= 2) {
        const int64_t waves32 = ((a.ne[1] + 31) / 32) * ((b.ne[1] + 31) / 32) * b.ne[2] * b.ne[3];
        if (b.ne[1] < 16 || (K >= 1024 && waves32 < 2048)) {  // few tiles, long rows (or a handful of columns): split K inside the workgroup
            dim3 grid((unsigned) ((a.ne[1] + 15) / 16), (unsigned) ((b.ne[1] + 15) / 16), (unsigned) (b.ne[2] * b.ne[3]));
            if (K >= 4096) hipLaunchKernelGGL(k_mul_mat_f16_mma16<16>, grid, dim3(1024), 0, s, a, b, d);
            else hipLaunchKernelGGL(k_mul_mat_f16_mma16<8>, grid, dim3(512), 0, s, a, b, d);
            return;
        }
    }
    if (w16 && vec_ok && b.ne[1] >= 16 && K <= 128 && a.ne[1] >= 256 && b.ne[2] % a.ne[2] == 0) {  // short rows, many of them (K.Q): columns resident
        const int64_t gq = b.ne[2] / a.ne[2];
        if (gq % 2 == 0) launch_xres_t<2>(s, a, b, d);  // (4 heads' columns = 128 registers: with the row buffers past 256, one wave per SIMD, or spills)
        else launch_xres_t<1>(s, a, b, d);
        return;
    }
    if (w16 && vec_ok && b.ne[1] >= 16) {  // a batch of columns: the matrix cores
        dim3 grid((unsigned) ((a.ne[1] + 63) / 64), (unsigned) ((b.ne[1] + 63) / 64), (unsigned) (b.ne[2] * b.ne[3]));
        hipLaunchKernelGGL(k_mul_mat_f16_mma, grid, dim3(256), 0, s, a, b, d);
        return;
    }
    if (w16 && vec_ok && b.ne[1] == 1 && a.ne[3] == 1 && b.ne[3] == 1 && b.ne[2] % a.ne[2] == 0 && a.ne[1] >= 64) {  // a decode step's K.q / V^T.p
        switch (b.ne[2] / a.ne[2]) {
            case 1: launch_gqa_t<1>(s, a, b, d); return;
            case 2: launch_gqa_t<2>(s, a, b, d); return;
            case 3: launch_gqa_t<3>(s, a, b, d); return;
            case 4: launch_gqa_t<4>(s, a, b, d); return;
            case 5: launch_gqa_t<5>(s, a, b, d); return;
            case 6: launch_gqa_t<6>(s, a, b, d); return;
            case 7: launch_gqa_t<7>(s, a, b, d); return;
            case 8: launch_gqa_t<8>(s, a, b, d); return;
            default: break;
        }
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

MI_TU_TOUCH(mmf)

}  // namespace mi355x
