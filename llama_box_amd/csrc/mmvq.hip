// mmvq.hip — decode mat-vec for GGUF-quantised weights: y[N] (x up to 8 columns) = W[N x K] · x[K].
//
// This is THE bandwidth-bound kernel of the hot path (~95 % of decode time is weight streaming, SURVEY.md §8a
// row a4).  It replaces ggml-cpu's ggml_vec_dot_{q4_K,q5_K,q6_K}_q8_K / q8_0_q8_0 and ggml-hip's mul_mat_vec_q
// (patch->ggml/src/ggml-cuda/mmvq.cu:65-71, llama-box/patches/llama.cpp/ggml-hip.patch:34-46) with a design
// made for CDNA4 rather than recompiled from the CUDA tiling:
//
//   * weights are streamed exactly once, super-block header + quants as 16-byte coalesced loads where the format
//     is 16-byte aligned (Q4_K 144 B = 9x16, Q5_K 176 B = 11x16); Q6_K (210 B) / Q8_0 (34 B) use loads typed with
//     their true 2-byte alignment;
//   * activations arrive already Q8_K / Q8_0-quantised (quantize.hip, CPU-identical rounding) and are staged in
//     LDS once per workgroup (304 B per 256 values) so the inner loop is ds_read_b128 + v_dot4_i32_i8;
//   * integer sub-block sums are exact (same integers as the CPU); only the f32 scale-accumulate order differs;
//   * one wave64 owns R rows; lanes split a row's (super-block, 16-byte chunk) pairs; a 6-step wave64 butterfly
//     finishes the row; optional fused epilogues (bias / residual add, SwiGLU over a second matrix) remove the
//     element-wise launches that would otherwise sit between the mat-vecs of a layer.
//
// Algorithmic bytes per launch = N * K/blk * bytes_per_block (+ K*1.19 activations, negligible).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>

#include <type_traits>

#include "mmvq_types.h"

namespace mi355x {

thread_local launch_probe g_launch_probe;


// ------------------------------------------------------------------------------------------------ kernel
// PRO selects how the workgroup obtains its Q8 activations:
//   0  already quantised in global memory (quantize.hip) -> copied to LDS
//   3  already quantised, read straight from global/L2 (batches whose activations exceed the LDS budget)
//   1  f32 activations: every workgroup quantises the row itself into LDS (no separate quantize launch)
//   2  f32 residual stream + norm weight: RMS_NORM, * w and the quantisation all happen in the prologue, so
//      [RMS_NORM -> MUL -> quantise -> MUL_MAT] is ONE launch; arithmetic per element is identical to the unfused
//      kernels (sum of squares in double, (x*scale)*w with two roundings, CPU-identical Q8_K rounding)
template <typename T, int NC, int R, bool GLU, int PRO, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) k_mmvq(const mmvq_args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename T::act act;
    constexpr int NT = WAVES * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nblk = a.K / T::BLK;
    const int npairs = nblk * T::PPB;
    const int row0 = (blockIdx.x * WAVES + wave) * R;

    // issue the first weight loads before touching the activations so HBM latency overlaps the prologue
    const uint8_t * rows[R];
    const uint8_t * rows2[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int rr = min(row0 + r, a.N - 1);
        rows[r] = a.W + (size_t) rr * a.w_nb1;
        rows2[r] = GLU ? a.W2 + (size_t) rr * a.w_nb1 : nullptr;
    }
    typename T::raw w[R], w2[R];
    int p = lane;
    if (p < npairs) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            w[r] = T::load(rows[r], p, nblk);
            if (GLU) w2[r] = T::load(rows2[r], p, nblk);
        }
    }

    const act * y;
    if constexpr (PRO == 0) {
        const int nwords = (int) ((size_t) a.ncols * nblk * sizeof(act) / 4);
        const uint32_t * src = (const uint32_t *) a.act;
        uint32_t * dst = (uint32_t *) smem;
        if ((sizeof(act) & 15) == 0) {
            const int nvec = nwords >> 2;
            for (int i = tid; i < nvec; i += NT) ((uint4 *) dst)[i] = ((const uint4 *) src)[i];
        } else {
            for (int i = tid; i < nwords; i += NT) dst[i] = src[i];
        }
        __syncthreads();
        y = (const act *) smem;
    } else if constexpr (PRO == 3) {
        y = (const act *) a.act;
    } else if constexpr (T::BLK == 256) {
        q8k_dev * yl = (q8k_dev *) smem;
        const float4 * x4 = (const float4 *) a.x;
        const float4 * w4 = (const float4 *) a.norm_w;
        float scale = 1.0f;
        if constexpr (PRO == 2) {
            double * red = (double *) (smem + (size_t) nblk * sizeof(q8k_dev));
            const int n4 = a.K >> 2;
            double ss = 0.0;
            for (int i0 = tid; i0 < n4; i0 += 4 * NT) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = i0 + u * NT;
                    v[u] = idx < n4 ? x4[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) ss += (double) (v[u].x * v[u].x) + (double) (v[u].y * v[u].y) + (double) (v[u].z * v[u].z) + (double) (v[u].w * v[u].w);
            }
            ss = wave_sum_d(ss);
            if (lane == 0) red[wave] = ss;
            __syncthreads();
            double tot = 0.0;
#pragma unroll
            for (int i = 0; i < WAVES; ++i) tot += red[i];
            const float mean = (float) (tot / (double) a.K);
            scale = 1.0f / sqrtf(mean + a.eps);
        }
        for (int b0 = wave; b0 < nblk; b0 += 4 * WAVES) {
            float v[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int b = b0 + u * WAVES;
                if (b < nblk) {
                    float4 t = x4[b * 64 + lane];
                    if constexpr (PRO == 2) {
                        const float4 g = w4[b * 64 + lane];
                        t.x = (t.x * scale) * g.x;
                        t.y = (t.y * scale) * g.y;
                        t.z = (t.z * scale) * g.z;
                        t.w = (t.w * scale) * g.w;
                        if (a.norm_out && blockIdx.x == 0) ((float4 *) a.norm_out)[b * 64 + lane] = t;
                    }
                    v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = t.w;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int b = b0 + u * WAVES;
                if (b < nblk) wave_quantize_q8_K(v[u], lane, yl + b);
            }
        }
        __syncthreads();
        y = (const act *) smem;
    } else {
        y = (const act *) a.act;  // unreachable: the launcher never pairs PRO 1/2 with Q8_0 weights
    }
    if (row0 >= a.N) return;

    float acc[R][NC], acc2[R][NC];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < NC; ++c) { acc[r][c] = 0.0f; acc2[r][c] = 0.0f; }

    while (p < npairs) {
        // prefetch the next pair of every row before consuming the current one (two loads per row in flight)
        typename T::raw nw[R], nw2[R];
        const int pn = p + 64;
        if (pn < npairs) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                nw[r] = T::load(rows[r], pn, nblk);
                if (GLU) nw2[r] = T::load(rows2[r], pn, nblk);
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            T::template dot<NC>(w[r], p, y, nblk, acc[r]);
            if (GLU) T::template dot<NC>(w2[r], p, y, nblk, acc2[r]);
        }
        if (pn < npairs) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                w[r] = nw[r];
                if (GLU) w2[r] = nw2[r];
            }
        }
        p = pn;
    }

#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = row0 + r;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            float v = wave_sum(acc[r][c]);
            if (GLU) {
                const float u = wave_sum(acc2[r][c]);
                v = silu_f(v) * u;
            }
            if (lane == 0 && row < a.N && c < a.ncols) {
                if (a.add) v += a.add[(size_t) c * a.add_stride + row];
                if (a.add2) v += a.add2[(size_t) c * a.add2_stride + row];
                a.dst[(size_t) c * a.dst_stride + row] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ streaming kernel
// Single-column (batch-1 decode) variant built around what the profile showed: with one short row per wave the
// launch is bound by a CHAIN of memory round trips (prologue -> first load -> second load), not by bandwidth.  Here
//   * the grid is one 16-wave workgroup per CU (<= 256), all resident at once: no second scheduling round;
//   * every wave walks its rows (row = wave id + t * total waves: consecutive waves read consecutive rows, so the
//     chip sweeps the matrix front to back) as a flat sequence of items (row, chunk of U lane-pairs) and always has
//     the NEXT item's loads in flight while it computes the current one;
//   * the first item is requested before the activation prologue, so HBM latency hides under it;
//   * the prologue (PRO as above) runs once per workgroup with 1024 threads: one L2 round trip for x (and w), the
//     RMS-norm reduction, CPU-identical Q8_K quantisation straight into LDS.
template <typename T, bool GLU, int PRO>
__global__ void __launch_bounds__(1024) k_mmvq_stream(const mmvq_args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename T::act act;
    // pairs per item: ~16 dwords of weights per lane per matrix (8 for the two-matrix GLU form).  Measured on MI355X:
    // doubling this (128 KiB per CU in flight) made every variant 5-25 % SLOWER — the launch is not in-flight-bound
    constexpr int WAVES = 16, NT = 1024, UB = (GLU ? 8 : 16) / T::DW, U = UB < 1 ? 1 : (UB > 4 ? 4 : UB);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nblk = a.K / T::BLK;
    const int npairs = nblk * T::PPB;
    const int nchunks = (npairs + 64 * U - 1) / (64 * U);
    const int GW = gridDim.x * WAVES;
    // rows are dealt one per wave and pass (row = pass*GW + block*16 + wave); the LAST, partial pass is spread evenly over
    // the workgroups instead (rem_per rows each), so that every CU streams until the end: with N = 14336 (3.5 passes)
    // the natural order left half of the CUs idle for the final eighth of the launch
    const int full = a.balance_tail ? (a.N / GW) * GW : ((a.N + GW - 1) / GW) * GW, rem = a.N - (a.balance_tail ? full : a.N);
    const int rem_per = (rem + (int) gridDim.x - 1) / (int) gridDim.x;
    auto next_row = [&](const int r) {
        const int nr = r + GW;
        if (nr < full) return nr;
        if (r >= full) return a.N;  // the remainder pass was this wave's last
        const int rr = full + (int) blockIdx.x * rem_per + wave;
        return (wave < rem_per && rr < a.N) ? rr : a.N;
    };
    int row = blockIdx.x * WAVES + wave, ch = 0;
    if (full == 0) row = (wave < rem_per && (int) blockIdx.x * rem_per + wave < a.N) ? (int) blockIdx.x * rem_per + wave : a.N;
    bool have = row < a.N;

    struct item {
        typename T::raw w[U];
        typename T::raw w2[U];
    };
    auto load_item = [&](const int r, const int c, item & it) {
        const uint8_t * rp = a.W + (size_t) r * a.w_nb1;
        const uint8_t * rp2 = GLU ? a.W2 + (size_t) r * a.w_nb1 : nullptr;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = (c * U + u) * 64 + lane;
            if (p < npairs) {
                it.w[u] = T::load(rp, p, nblk);
                if (GLU) it.w2[u] = T::load(rp2, p, nblk);
            }
        }
    };
    // Order of the first requests (scripts/ubench/decode_lab.hip, stamp_lab.hip, prologue_probe.hip — DESIGN.md §4):
    //   f32 prologue (PRO 1: wo, ffn_down): the activation row FIRST, then the first weights, both unconditionally (clamped
    //     addresses): wo 6.0 -> 5.4 us, ffn_down Q4_K 11.7 -> 10.3, Q6_K 15.8 -> 14.7.  A load under `if (...)` makes hipcc's
    //     wait-count pass assume it may not have been issued, and the first use of x then waits for the weights as well.
    //   norm prologue (PRO 2: gate/up, lm_head): weights first, as in round 1 — with x AND the norm weights ahead of them the 66 MB
    //     gate/up launch lost 1.2 us and the lm_head 10 us.  (What delays x is not its place in the wave's own queue but the other
    //     waves' weight requests in the CU's memory pipeline: x returns 0.6 us after entry on an idle pipeline, 3-6 us behind
    //     ~100 KB of weight requests per CU; holding the weights back until x has arrived leaves HBM idle for as long as it gains.)
    constexpr int BPC = PRO == 0 ? 1 : 256 / T::BLK;  // activation blocks per 256-value chunk
    const int nchk = a.K / 256;
    float4 v[4], g[4];
    double ssp[4] = {0.0, 0.0, 0.0, 0.0};
    item cur;
    // PRO 3 (wo of a decode step): the activation row is the attention result, still in the form of its split partials
    // (fattn.hip, 8-wave form: [head][split] records of 128 values + (max, sum), FA_REC floats apart) — this prologue is the
    // combine pass, so that pass's launch (~4.5 us of a dependent launch for 16 KB of work) disappears.  Lane l of chunk b owns
    // dims 4(l & 31) .. +3 of head 2b + (l >> 5).
    float2 fml[PRO == 3 ? 16 : 1];
    float4 fr[PRO == 3 ? 12 : 1];
    if constexpr (PRO == 3) {
        const int S = a.fa_splits;
        const float * rec0 = a.fa_part + (size_t) (2 * min(wave, nchk - 1) + (lane >> 5)) * S * FA_REC;
#pragma unroll
        for (int sidx = 0; sidx < 16; ++sidx) fml[sidx] = *(const float2 *) (rec0 + (size_t) min(sidx, S - 1) * FA_REC + 128);
#pragma unroll
        for (int sidx = 0; sidx < 12; ++sidx) fr[sidx] = *(const float4 *) (rec0 + (size_t) min(sidx, S - 1) * FA_REC + 4 * (lane & 31));
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (PRO == 1 || PRO == 3) {
        if constexpr (PRO == 1) {
        const float4 * x4 = (const float4 *) a.x;
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = x4[min(wave + u * WAVES, nchk - 1) * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);  // keep the weight loads below behind them
        }
        const int rc = min(row, a.N - 1);
        const uint8_t * rp = a.W + (size_t) rc * a.w_nb1;
        const uint8_t * rp2 = GLU ? a.W2 + (size_t) rc * a.w_nb1 : nullptr;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = min(u * 64 + lane, npairs - 1);
            cur.w[u] = T::load(rp, p, nblk);
            if (GLU) cur.w2[u] = T::load(rp2, p, nblk);
        }
    } else {
        if (have) load_item(row, 0, cur);
        if constexpr (PRO == 2) {
            if (a.ss_in) {  // (uniform) the producer of x left its sum of squares as partial sums: one more load in this round trip
#pragma unroll
                for (int u = 0; u < 4; ++u) ssp[u] = lane + 64 * u < a.ss_n ? a.ss_in[lane + 64 * u] : 0.0;
            }
            const float4 * x4 = (const float4 *) a.x;
            const float4 * w4 = (const float4 *) a.norm_w;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int b = wave + u * WAVES;
                if (b < nchk) {
                    v[u] = x4[b * 64 + lane];
                    g[u] = w4[b * 64 + lane];
                } else {
                    v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    g[u] = v[u];
                }
            }
        }
    }

    // ---- activation prologue
    if constexpr (PRO == 0) {
        const int nwords = (int) ((size_t) nblk * sizeof(act) / 4);
        const uint32_t * src = (const uint32_t *) a.act;
        uint32_t * dst = (uint32_t *) smem;
        if ((sizeof(act) & 15) == 0) {
            const int nvec = nwords >> 2;
            for (int i = tid; i < nvec; i += NT) ((uint4 *) dst)[i] = ((const uint4 *) src)[i];
        } else {
            for (int i = tid; i < nwords; i += NT) dst[i] = src[i];
        }
    } else if constexpr (PRO == 3) {
        act * yl = (act *) smem;
        const int S = a.fa_splits, hsel = lane >> 5, d0 = 4 * (lane & 31);
        constexpr float LOG2E = 1.4426950408889634f;
        for (int b0 = wave; b0 < nchk; b0 += WAVES) {
            const float * rec0 = a.fa_part + (size_t) (2 * b0 + hsel) * S * FA_REC;
            if (b0 != wave) {
#pragma unroll
                for (int sidx = 0; sidx < 16; ++sidx) fml[sidx] = *(const float2 *) (rec0 + (size_t) min(sidx, S - 1) * FA_REC + 128);
#pragma unroll
                for (int sidx = 0; sidx < 12; ++sidx) fr[sidx] = *(const float4 *) (rec0 + (size_t) min(sidx, S - 1) * FA_REC + d0);
            }
            // softmax merge of the splits, in split order (k_fattn_combine's arithmetic; records are in the natural-log domain)
            float mn = -INFINITY;
#pragma unroll
            for (int sidx = 0; sidx < 16; ++sidx) if (sidx < S) mn = fmaxf(mn, fml[sidx].x);
            float cs[16], lt = 0.0f;
#pragma unroll
            for (int sidx = 0; sidx < 16; ++sidx) {
                cs[sidx] = (sidx < S && fml[sidx].x != -INFINITY) ? __builtin_amdgcn_exp2f((fml[sidx].x - mn) * LOG2E) : 0.0f;
                lt += sidx < S ? fml[sidx].y * cs[sidx] : 0.0f;
            }
            float4 acc4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int sidx = 0; sidx < 12; ++sidx)
                if (sidx < S && cs[sidx] != 0.0f) { acc4.x += fr[sidx].x * cs[sidx]; acc4.y += fr[sidx].y * cs[sidx]; acc4.z += fr[sidx].z * cs[sidx]; acc4.w += fr[sidx].w * cs[sidx]; }
            if (S > 12) {  // (fattn_fat_splits() never asks for more than 12; kept for callers that pass their own count)
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx) fr[sidx] = *(const float4 *) (rec0 + (size_t) min(12 + sidx, S - 1) * FA_REC + d0);
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx)
                    if (12 + sidx < S && cs[12 + sidx] != 0.0f) { acc4.x += fr[sidx].x * cs[12 + sidx]; acc4.y += fr[sidx].y * cs[12 + sidx]; acc4.z += fr[sidx].z * cs[12 + sidx]; acc4.w += fr[sidx].w * cs[12 + sidx]; }
            }
            const float il = 1.0f / lt;
            float t[4] = {acc4.x * il, acc4.y * il, acc4.z * il, acc4.w * il};
            // the graph's FLASH_ATTN_EXT result itself, for any other reader (graph.cpp passes null when that tensor is provably dead)
            if (a.x_out && blockIdx.x == 0) ((float4 *) a.x_out)[b0 * 64 + lane] = make_float4(t[0], t[1], t[2], t[3]);
            if constexpr (BPC == 1) wave_quantize_q8_K(t, lane, yl + b0);
            else wave_quantize_q8_0(t, lane, yl + (size_t) b0 * BPC);
        }
    } else {
        // one wave per 256-value chunk: a Q8_K block, or eight Q8_0 blocks (launcher guarantees K % 256 == 0)
        act * yl = (act *) smem;
        const float4 * x4 = (const float4 *) a.x;
        // batch 0 (blocks wave, wave+16, wave+32, wave+48: loaded above) is special: with the norm it must hold the WHOLE row
        // (launcher guarantees nblk <= 64) because the scale needs the full sum of squares; the barrier sits outside
        // any wave-dependent control flow.  Further batches (f32 prologue only, K > 16384) load inside the loop.
        for (int b0 = wave; b0 < nchk || b0 == wave; b0 += 4 * WAVES) {
            if (b0 != wave) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int b = b0 + u * WAVES;
                    v[u] = b < nchk ? x4[b * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            float scale = 1.0f;
            if constexpr (PRO == 2) {
              double tot = 0.0;
              if (a.ss_in) {
                // partial sums in a fixed order (lane-strided, then the wave tree): every wave of every workgroup gets the same bits,
                // and nobody waits for anybody
                tot = wave_sum_d(((ssp[0] + ssp[1]) + ssp[2]) + ssp[3]);
              } else {
                double * red = (double *) (smem + (size_t) nblk * sizeof(act));
                double ss = 0.0;
#pragma unroll
                for (int u = 0; u < 4; ++u) ss += (double) (v[u].x * v[u].x) + (double) (v[u].y * v[u].y) + (double) (v[u].z * v[u].z) + (double) (v[u].w * v[u].w);
                ss = wave_sum_d(ss);
                if (lane == 0) red[wave] = ss;
                __syncthreads();  // reached exactly once by every wave: the loop condition admits b0 == wave
#pragma unroll
                for (int i = 0; i < WAVES; ++i) tot += red[i];
              }
                const float mean = (float) (tot / (double) a.K);
                scale = 1.0f / sqrtf(mean + a.eps);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int b = b0 + u * WAVES;
                if (b < nchk) {
                    float t[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                    if constexpr (PRO == 2) {
                        t[0] = (t[0] * scale) * g[u].x;
                        t[1] = (t[1] * scale) * g[u].y;
                        t[2] = (t[2] * scale) * g[u].z;
                        t[3] = (t[3] * scale) * g[u].w;
                        if (a.norm_out && blockIdx.x == 0) ((float4 *) a.norm_out)[b * 64 + lane] = make_float4(t[0], t[1], t[2], t[3]);
                    }
                    if constexpr (BPC == 1) wave_quantize_q8_K(t, lane, yl + b);
                    else wave_quantize_q8_0(t, lane, yl + (size_t) b * BPC);
                }
            }
            if constexpr (PRO == 2) break;  // single batch by construction
        }
    }
    __syncthreads();
    const act * y = (const act *) smem;

    float acc = 0.0f, acc2 = 0.0f;
    double ssw = 0.0;  // lane 0: sum of squares of the rows this wave produced
    while (have) {
        int nrow = row, nch = ch + 1;
        if (nch == nchunks) { nch = 0; nrow = next_row(row); }
        const bool nhave = nrow < a.N;
        item nxt;
        if (nhave) load_item(nrow, nch, nxt);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = (ch * U + u) * 64 + lane;
            if (p < npairs) {
                T::template dot<1>(cur.w[u], p, y, nblk, &acc);
                if (GLU) T::template dot<1>(cur.w2[u], p, y, nblk, &acc2);
            }
        }
        if (ch == nchunks - 1) {
            float v = wave_sum(acc);
            if (GLU) {
                const float g = wave_sum(acc2);
                v = silu_f(v) * g;
            }
            if (lane == 0) {
                if (a.add) v += a.add[row];
                if (a.add2) v += a.add2[row];
                a.dst[row] = v;
                ssw += (double) (v * v);  // (as ggml-cpu's rms_norm: the f32 product, summed in double)
            }
            acc = 0.0f;
            acc2 = 0.0f;
        }
        cur = nxt;
        row = nrow;
        ch = nch;
        have = nhave;
    }
    if (a.ss_out) {  // (uniform) this launch writes a residual stream an RMS_NORM prologue reads next: leave its sum of squares, one partial per workgroup
        double * red = (double *) (smem + (size_t) nblk * sizeof(act));
        if constexpr (PRO == 2) __syncthreads();  // (the norm prologue's own exchange used this area)
        if (lane == 0) red[wave] = ssw;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
#pragma unroll
            for (int i = 0; i < WAVES; ++i) t += red[i];
            a.ss_out[blockIdx.x] = t;
        }
    }
}

template <typename T, bool GLU, int PRO> static void launch_stream(hipStream_t s, const mmvq_args & a) {
    const int nblk = a.K / T::BLK;
    const size_t lds = (size_t) nblk * sizeof(typename T::act) + 16 * sizeof(double) + 16;
    const unsigned grid = (unsigned) std::min<int64_t>(256, ((int64_t) a.N + 15) / 16);
    static const int bal = getenv("GGML_MI355X_BALANCE_TAIL") ? atoi(getenv("GGML_MI355X_BALANCE_TAIL")) : 1;
    mmvq_args a2 = a;
    a2.balance_tail = bal;
    if (g_launch_probe.armed && !g_launch_probe.used) {
        hipExtLaunchKernelGGL((k_mmvq_stream<T, GLU, PRO>), dim3(grid), dim3(1024), lds, s, g_launch_probe.e0, g_launch_probe.e1, 0, a2);
        g_launch_probe.used = true;
    } else {
        hipLaunchKernelGGL((k_mmvq_stream<T, GLU, PRO>), dim3(grid), dim3(1024), lds, s, a2);
    }
}

template <typename T, int NC, int R, bool GLU, int PRO, int WAVES> static void launch_one(hipStream_t s, const mmvq_args & a, size_t lds) {
    const int rows_per_block = WAVES * R;
    const unsigned grid = (unsigned) ((a.N + rows_per_block - 1) / rows_per_block);
    hipLaunchKernelGGL((k_mmvq<T, NC, R, GLU, PRO, WAVES>), dim3(grid), dim3(WAVES * 64), lds, s, a);
}

template <typename TP> static void launch_planes(hipStream_t s, const mmvq_args & ap, const bool glu) {
    if (ap.fa_part) launch_stream<TP, false, 3>(s, ap);
    else if (ap.x && ap.norm_w) { if (glu) launch_stream<TP, true, 2>(s, ap); else launch_stream<TP, false, 2>(s, ap); }
    else if (ap.x) { if (glu) launch_stream<TP, true, 1>(s, ap); else launch_stream<TP, false, 1>(s, ap); }
    else { if (glu) launch_stream<TP, true, 0>(s, ap); else launch_stream<TP, false, 0>(s, ap); }
}
template <typename T> static void launch_type(hipStream_t s, const mmvq_args & a0, int rows_per_wave) {
    mmvq_args a = a0;
    const int nblk = a.K / T::BLK;
    const bool glu = a.W2 != nullptr;
    // the streaming forms over the decode copy (plane layout, non-temporal loads): same registers per lane, same dot products
    typedef typename plane_of<T>::type TP;
    if constexpr (!std::is_same<T, TP>::value) {
        const bool planes = a0.Wp != nullptr && (!glu || a0.W2p != nullptr) && (a0.K % 256) == 0;
        const bool stream = a0.x != nullptr || a0.fa_part != nullptr || (a0.ncols == 1 && (size_t) nblk * sizeof(typename T::act) <= 60 * 1024);
        if (planes && stream) {
            mmvq_args ap = a0;
            ap.W = a0.Wp;
            ap.W2 = glu ? a0.W2p : nullptr;
            if ((nblk & 7) == 0) launch_planes<TP>(s, ap, glu);
            else launch_planes<typename plane_of<T>::short_type>(s, ap, glu);
            return;
        }
    }
    if (a0.x != nullptr || a0.fa_part != nullptr) {
        if ((a0.K % 256) != 0) {
            MI_ERR("launch_mmvq: the f32 prologue needs K %% 256 == 0 (K = %d)", a0.K);
            abort();
        }
        if (a0.fa_part) {
            if (glu || a0.norm_w || a0.fa_splits < 1 || a0.fa_splits > 16) { MI_ERR("launch_mmvq: bad attention-partials prologue request"); abort(); }
            launch_stream<T, false, 3>(s, a0);
            return;
        }
        if (a0.norm_w) { if (glu) launch_stream<T, true, 2>(s, a0); else launch_stream<T, false, 2>(s, a0); }
        else           { if (glu) launch_stream<T, true, 1>(s, a0); else launch_stream<T, false, 1>(s, a0); }
        return;
    }
    if (a0.ncols == 1 && (size_t) nblk * sizeof(typename T::act) <= 60 * 1024) {
        if (glu) launch_stream<T, true, 0>(s, a0); else launch_stream<T, false, 0>(s, a0);
        return;
    }
    // activations are laid out [ncols][nblk]; the column loop runs templates of exactly 8/4/2/1 columns
    // Q8_0 weights (no weight-streaming matrix-core kernel for 2 .. 32 columns: mmq_skinny.hip serves the K-quants): 16 or 32 columns in ONE pass when their Q8_0
    // blocks fit the LDS (up to 144 KB, on 16-wave workgroups so that a CU still holds 16 waves) — a -np 32 step used to stream every Q8_0 matrix four times
    // (TinyLlama: 55 % of the step in k_mmvq<T_Q80, 8>, profiles/r06_ab_fa_dec64.txt); same dot products per column, same f32 order
    static const int wide_cols = getenv("GGML_MI355X_Q80_WIDE_COLS") ? atoi(getenv("GGML_MI355X_Q80_WIDE_COLS")) : 1;
    int done = 0;
    while (done < a0.ncols) {
        const int left = a0.ncols - done;
        if constexpr (std::is_same<T, T_Q80>::value) {
            if (wide_cols && !glu && left >= 12) {
                const int ncw = (left >= 24 && (size_t) 32 * nblk * sizeof(typename T::act) <= 144 * 1024) ? 32 : ((size_t) 16 * nblk * sizeof(typename T::act) <= 144 * 1024 ? 16 : 0);
                if (ncw) {
                    const int take = std::min(left, ncw);  // (the kernel computes ncw columns; a.ncols says how many exist)
                    a.ncols = take;
                    a.act = (const char *) a0.act + (size_t) done * nblk * sizeof(typename T::act);
                    a.dst = a0.dst + (size_t) done * a0.dst_stride;
                    a.add = a0.add ? a0.add + (size_t) done * a0.add_stride : nullptr;
                    a.add2 = a0.add2 ? a0.add2 + (size_t) done * a0.add2_stride : nullptr;
                    const size_t lds = (size_t) ncw * nblk * sizeof(typename T::act);  // (the dot products walk all ncw columns; only `take` of them are copied in and stored)
                    static std::atomic<uint32_t> raised16{0}, raised32{0};
                    if (ncw == 32) {
                        (void) ensure_dyn_lds((const void *) k_mmvq<T, 32, 1, false, 0, 16>, lds, raised32);
                        launch_one<T, 32, 1, false, 0, 16>(s, a, lds);
                    } else {
                        (void) ensure_dyn_lds((const void *) k_mmvq<T, 16, 1, false, 0, 16>, lds, raised16);
                        launch_one<T, 16, 1, false, 0, 16>(s, a, lds);
                    }
                    done += take;
                    continue;
                }
            }
        }
        const int nc = left >= 8 ? 8 : left >= 4 ? 4 : left >= 2 ? 2 : 1;
        a.ncols = nc;
        a.act = (const char *) a0.act + (size_t) done * nblk * sizeof(typename T::act);
        a.dst = a0.dst + (size_t) done * a0.dst_stride;
        a.add = a0.add ? a0.add + (size_t) done * a0.add_stride : nullptr;
        a.add2 = a0.add2 ? a0.add2 + (size_t) done * a0.add2_stride : nullptr;
        const size_t lds = (size_t) nc * nblk * sizeof(typename T::act);
        const bool fits = lds <= 64 * 1024;
        if (nc == 1) {
            const bool r2 = rows_per_wave >= 2;
            if (glu) { if (r2) launch_one<T, 1, 2, true, 0, 4>(s, a, lds); else launch_one<T, 1, 1, true, 0, 4>(s, a, lds); }
            else     { if (r2) launch_one<T, 1, 2, false, 0, 4>(s, a, lds); else launch_one<T, 1, 1, false, 0, 4>(s, a, lds); }
        } else if (nc == 2) {
            if (glu) { if (fits) launch_one<T, 2, 1, true, 0, 4>(s, a, lds); else launch_one<T, 2, 1, true, 3, 4>(s, a, 0); }
            else     { if (fits) launch_one<T, 2, 1, false, 0, 4>(s, a, lds); else launch_one<T, 2, 1, false, 3, 4>(s, a, 0); }
        } else if (nc == 4) {
            if (glu) { if (fits) launch_one<T, 4, 1, true, 0, 4>(s, a, lds); else launch_one<T, 4, 1, true, 3, 4>(s, a, 0); }
            else     { if (fits) launch_one<T, 4, 1, false, 0, 4>(s, a, lds); else launch_one<T, 4, 1, false, 3, 4>(s, a, 0); }
        } else {
            if (glu) { if (fits) launch_one<T, 8, 1, true, 0, 4>(s, a, lds); else launch_one<T, 8, 1, true, 3, 4>(s, a, 0); }
            else     { if (fits) launch_one<T, 8, 1, false, 0, 4>(s, a, lds); else launch_one<T, 8, 1, false, 3, 4>(s, a, 0); }
        }
        done += nc;
    }
}

int launch_mmvq_ss_count(const mmvq_args & a) {
    // the streaming kernel with an f32 / norm prologue (launch_type): one column, K-quant or Q8_0 rows of whole 256-value chunks, no SwiGLU
    if (a.ncols != 1 || a.W2 != nullptr || a.fa_part != nullptr || a.x == nullptr || (a.K % 256) != 0) return 0;
    return (int) std::min<int64_t>(256, ((int64_t) a.N + 15) / 16);
}

void launch_mmvq(hipStream_t s, const mmvq_args & a, int rows_per_wave) {
    switch (a.type) {
        case GGML_TYPE_Q4_K: launch_type<T_Q4K>(s, a, rows_per_wave); break;
        case GGML_TYPE_Q5_K: launch_type<T_Q5K>(s, a, rows_per_wave); break;
        case GGML_TYPE_Q6_K: launch_type<T_Q6K>(s, a, rows_per_wave); break;
        case GGML_TYPE_Q8_0: launch_type<T_Q80>(s, a, rows_per_wave); break;
        default: MI_ERR("launch_mmvq: unsupported weight type %d", a.type); abort();
    }
}

MI_TU_TOUCH(mmvq)

}  // namespace mi355x
