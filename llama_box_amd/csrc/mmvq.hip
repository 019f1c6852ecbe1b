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
#include <algorithm>

#include "dev_util.h"
#include "kernels.h"

namespace mi355x {

// sc/m pair extraction for K-quants' 12 packed bytes, for sub-blocks (2j, 2j+1); hy/hz/hw = bytes 0-3 / 4-7 / 8-11
__device__ __forceinline__ void k4_scale_pair(uint32_t hy, uint32_t hz, uint32_t hw, int j, int & sc0, int & sc1, int & m0, int & m1) {
    const int sh = 16 * (j & 1);
    const uint32_t a = (hy >> sh) & 0xFFFFu, b = (hz >> sh) & 0xFFFFu, w = (hw >> sh) & 0xFFFFu;
    uint32_t scp, mp;
    if (j < 2) {
        scp = a & 0x3F3Fu;
        mp = b & 0x3F3Fu;
    } else {
        scp = (w & 0x0F0Fu) | ((a & 0xC0C0u) >> 2);
        mp = ((w >> 4) & 0x0F0Fu) | ((b & 0xC0C0u) >> 2);
    }
    sc0 = (int) (scp & 0xFF);
    sc1 = (int) (scp >> 8);
    m0 = (int) (mp & 0xFF);
    m1 = (int) (mp >> 8);
}

// ------------------------------------------------------------------------------------------------ Q4_K
struct T_Q4K {
    typedef q8k_dev act;
    static constexpr int BLK = 256, BYTES = 144, PPB = 8;  // pairs (lanes) per block
    struct raw { uint4 hdr, q; };
    static __device__ __forceinline__ raw load(const uint8_t * __restrict__ row, int p) {
        const uint8_t * blk = row + (size_t) (p >> 3) * BYTES;
        raw r;
        r.hdr = *(const uint4 *) blk;
        r.q = *(const uint4 *) (blk + 16 + 16 * (p & 7));
        return r;
    }
    template <int NC> static __device__ __forceinline__ void dot(const raw & r, int p, const act * __restrict__ y, int nblk, float * acc) {
        const int b = p >> 3, c = p & 7, j = c >> 1;
        const float d = h2f((uint16_t) (r.hdr.x & 0xFFFF)), dmin = h2f((uint16_t) (r.hdr.x >> 16));
        int sc0, sc1, m0, m1;
        k4_scale_pair(r.hdr.y, r.hdr.z, r.hdr.w, j, sc0, sc1, m0, m1);
        const int e0 = 64 * j + 16 * (c & 1);
        const uint32_t qv[4] = {r.q.x, r.q.y, r.q.z, r.q.w};
#pragma unroll
        for (int col = 0; col < NC; ++col) {
            const act * yb = y + (size_t) col * nblk + b;
            const uint4 ylo = *(const uint4 *) (yb->qs + e0);
            const uint4 yhi = *(const uint4 *) (yb->qs + e0 + 32);
            const uint32_t yl[4] = {ylo.x, ylo.y, ylo.z, ylo.w}, yh[4] = {yhi.x, yhi.y, yhi.z, yhi.w};
            int s_lo = 0, s_hi = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s_lo = dot4((int) (qv[k] & 0x0F0F0F0Fu), (int) yl[k], s_lo);
                s_hi = dot4((int) ((qv[k] >> 4) & 0x0F0F0F0Fu), (int) yh[k], s_hi);
            }
            const int bs_lo = yb->bsums[4 * j + (c & 1)], bs_hi = yb->bsums[4 * j + 2 + (c & 1)];
            const int isum = sc0 * s_lo + sc1 * s_hi;
            const int msum = m0 * bs_lo + m1 * bs_hi;
            acc[col] += yb->d * (d * (float) isum - dmin * (float) msum);
        }
    }
};

// ------------------------------------------------------------------------------------------------ Q5_K
struct T_Q5K {
    typedef q8k_dev act;
    static constexpr int BLK = 256, BYTES = 176, PPB = 8;
    struct raw { uint4 hdr, qh, q; };
    static __device__ __forceinline__ raw load(const uint8_t * __restrict__ row, int p) {
        const uint8_t * blk = row + (size_t) (p >> 3) * BYTES;
        raw r;
        r.hdr = *(const uint4 *) blk;
        r.qh = *(const uint4 *) (blk + 16 + 16 * (p & 1));
        r.q = *(const uint4 *) (blk + 48 + 16 * (p & 7));
        return r;
    }
    template <int NC> static __device__ __forceinline__ void dot(const raw & r, int p, const act * __restrict__ y, int nblk, float * acc) {
        const int b = p >> 3, c = p & 7, j = c >> 1;
        const float d = h2f((uint16_t) (r.hdr.x & 0xFFFF)), dmin = h2f((uint16_t) (r.hdr.x >> 16));
        int sc0, sc1, m0, m1;
        k4_scale_pair(r.hdr.y, r.hdr.z, r.hdr.w, j, sc0, sc1, m0, m1);
        const int e0 = 64 * j + 16 * (c & 1);
        const uint32_t qv[4] = {r.q.x, r.q.y, r.q.z, r.q.w}, qh[4] = {r.qh.x, r.qh.y, r.qh.z, r.qh.w};
        uint32_t lo[4], hi[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            lo[k] = (qv[k] & 0x0F0F0F0Fu) | (((qh[k] >> (2 * j)) & 0x01010101u) << 4);
            hi[k] = ((qv[k] >> 4) & 0x0F0F0F0Fu) | (((qh[k] >> (2 * j + 1)) & 0x01010101u) << 4);
        }
#pragma unroll
        for (int col = 0; col < NC; ++col) {
            const act * yb = y + (size_t) col * nblk + b;
            const uint4 ylo = *(const uint4 *) (yb->qs + e0);
            const uint4 yhi = *(const uint4 *) (yb->qs + e0 + 32);
            const uint32_t yl[4] = {ylo.x, ylo.y, ylo.z, ylo.w}, yh[4] = {yhi.x, yhi.y, yhi.z, yhi.w};
            int s_lo = 0, s_hi = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s_lo = dot4((int) lo[k], (int) yl[k], s_lo);
                s_hi = dot4((int) hi[k], (int) yh[k], s_hi);
            }
            const int bs_lo = yb->bsums[4 * j + (c & 1)], bs_hi = yb->bsums[4 * j + 2 + (c & 1)];
            const int isum = sc0 * s_lo + sc1 * s_hi;
            const int msum = m0 * bs_lo + m1 * bs_hi;
            acc[col] += yb->d * (d * (float) isum - dmin * (float) msum);
        }
    }
};

// ------------------------------------------------------------------------------------------------ Q6_K
// 16 lanes per super-block: lane (h, t) owns l = 4t..4t+3 of half h, i.e. 16 of the 256 values
struct T_Q6K {
    typedef q8k_dev act;
    static constexpr int BLK = 256, BYTES = 210, PPB = 16;
    struct raw { uint32_t ql0, ql1, qh, s0, s1; uint16_t d; };
    static __device__ __forceinline__ raw load(const uint8_t * __restrict__ row, int p) {
        const uint8_t * blk = row + (size_t) (p >> 4) * BYTES;
        const int h = (p >> 3) & 1, t = p & 7;
        raw r;
        r.ql0 = ld32_a2(blk + 64 * h + 4 * t);
        r.ql1 = ld32_a2(blk + 64 * h + 32 + 4 * t);
        r.qh = ld32_a2(blk + 128 + 32 * h + 4 * t);
        r.s0 = ld32_a2(blk + 192 + 8 * h);
        r.s1 = ld32_a2(blk + 196 + 8 * h);
        r.d = ld16(blk + 208);
        return r;
    }
    template <int NC> static __device__ __forceinline__ void dot(const raw & r, int p, const act * __restrict__ y, int nblk, float * acc) {
        const int b = p >> 4, h = (p >> 3) & 1, t = p & 7, is = t >> 2;
        const float d = h2f(r.d);
        const uint32_t v[4] = {
            (r.ql0 & 0x0F0F0F0Fu) | ((r.qh & 0x03030303u) << 4),
            (r.ql1 & 0x0F0F0F0Fu) | (((r.qh >> 2) & 0x03030303u) << 4),
            ((r.ql0 >> 4) & 0x0F0F0F0Fu) | (((r.qh >> 4) & 0x03030303u) << 4),
            ((r.ql1 >> 4) & 0x0F0F0F0Fu) | (((r.qh >> 6) & 0x03030303u) << 4),
        };
        const int sc[4] = {
            (int) (int8_t) (r.s0 >> (8 * is)), (int) (int8_t) (r.s0 >> (8 * (is + 2))),
            (int) (int8_t) (r.s1 >> (8 * is)), (int) (int8_t) (r.s1 >> (8 * (is + 2))),
        };
#pragma unroll
        for (int col = 0; col < NC; ++col) {
            const act * yb = y + (size_t) col * nblk + b;
            int isum = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int yk = *(const int *) (yb->qs + 128 * h + 32 * k + 4 * t);
                // sum (q - 32) * y = dot(q, y) - 32 * sum(y)
                const int s = dot4((int) v[k], yk, 0) - 32 * dot4(0x01010101, yk, 0);
                isum += sc[k] * s;
            }
            acc[col] += yb->d * d * (float) isum;
        }
    }
};

// ------------------------------------------------------------------------------------------------ Q8_0
struct T_Q80 {
    typedef q80_dev act;
    static constexpr int BLK = 32, BYTES = 34, PPB = 1;
    struct raw { uint32_t q[8]; uint16_t d; };
    static __device__ __forceinline__ raw load(const uint8_t * __restrict__ row, int p) {
        const uint8_t * blk = row + (size_t) p * BYTES;
        raw r;
        r.d = ld16(blk);
#pragma unroll
        for (int k = 0; k < 8; ++k) r.q[k] = ld32_a2(blk + 2 + 4 * k);
        return r;
    }
    template <int NC> static __device__ __forceinline__ void dot(const raw & r, int p, const act * __restrict__ y, int nblk, float * acc) {
        const float d = h2f(r.d);
#pragma unroll
        for (int col = 0; col < NC; ++col) {
            const act * yb = y + (size_t) col * nblk + p;
            int s = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) s = dot4((int) r.q[k], ((const int *) yb->qs)[k], s);
            acc[col] += (float) s * (d * yb->d);
        }
    }
};

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
            w[r] = T::load(rows[r], p);
            if (GLU) w2[r] = T::load(rows2[r], p);
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
                nw[r] = T::load(rows[r], pn);
                if (GLU) nw2[r] = T::load(rows2[r], pn);
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
    constexpr int WAVES = 16, NT = 1024, U = GLU ? 1 : 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nblk = a.K / T::BLK;
    const int npairs = nblk * T::PPB;
    const int nchunks = (npairs + 64 * U - 1) / (64 * U);
    const int GW = gridDim.x * WAVES;
    int row = blockIdx.x * WAVES + wave, ch = 0;
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
                it.w[u] = T::load(rp, p);
                if (GLU) it.w2[u] = T::load(rp2, p);
            }
        }
    };
    item cur;
    if (have) load_item(row, 0, cur);

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
    } else if constexpr (T::BLK == 256) {
        q8k_dev * yl = (q8k_dev *) smem;
        const float4 * x4 = (const float4 *) a.x;
        const float4 * w4 = (const float4 *) a.norm_w;
        // batch 0 (blocks wave, wave+16, wave+32, wave+48) is special: with the norm it must hold the WHOLE row
        // (launcher guarantees nblk <= 64) because the scale needs the full sum of squares; the barrier sits outside
        // any wave-dependent control flow
        for (int b0 = wave; b0 < nblk || b0 == wave; b0 += 4 * WAVES) {
            float4 v[4], g[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int b = b0 + u * WAVES;
                if (b < nblk) {
                    v[u] = x4[b * 64 + lane];
                    if (PRO == 2) g[u] = w4[b * 64 + lane];
                } else {
                    v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    g[u] = v[u];
                }
            }
            float scale = 1.0f;
            if constexpr (PRO == 2) {
                double * red = (double *) (smem + (size_t) nblk * sizeof(q8k_dev));
                double ss = 0.0;
#pragma unroll
                for (int u = 0; u < 4; ++u) ss += (double) (v[u].x * v[u].x) + (double) (v[u].y * v[u].y) + (double) (v[u].z * v[u].z) + (double) (v[u].w * v[u].w);
                ss = wave_sum_d(ss);
                if (lane == 0) red[wave] = ss;
                __syncthreads();  // reached exactly once by every wave: the loop condition admits b0 == wave
                double tot = 0.0;
#pragma unroll
                for (int i = 0; i < WAVES; ++i) tot += red[i];
                const float mean = (float) (tot / (double) a.K);
                scale = 1.0f / sqrtf(mean + a.eps);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int b = b0 + u * WAVES;
                if (b < nblk) {
                    float t[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                    if constexpr (PRO == 2) {
                        t[0] = (t[0] * scale) * g[u].x;
                        t[1] = (t[1] * scale) * g[u].y;
                        t[2] = (t[2] * scale) * g[u].z;
                        t[3] = (t[3] * scale) * g[u].w;
                    }
                    wave_quantize_q8_K(t, lane, yl + b);
                }
            }
            if constexpr (PRO == 2) break;  // single batch by construction
        }
    }
    __syncthreads();
    const act * y = (const act *) smem;

    float acc = 0.0f, acc2 = 0.0f;
    while (have) {
        int nrow = row, nch = ch + 1;
        if (nch == nchunks) { nch = 0; nrow = row + GW; }
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
            }
            acc = 0.0f;
            acc2 = 0.0f;
        }
        cur = nxt;
        row = nrow;
        ch = nch;
        have = nhave;
    }
}

template <typename T, bool GLU, int PRO> static void launch_stream(hipStream_t s, const mmvq_args & a) {
    const int nblk = a.K / T::BLK;
    const size_t lds = (size_t) nblk * sizeof(typename T::act) + 16 * sizeof(double) + 16;
    const unsigned grid = (unsigned) std::min<int64_t>(256, ((int64_t) a.N + 15) / 16);
    hipLaunchKernelGGL((k_mmvq_stream<T, GLU, PRO>), dim3(grid), dim3(1024), lds, s, a);
}

template <typename T, int NC, int R, bool GLU, int PRO, int WAVES> static void launch_one(hipStream_t s, const mmvq_args & a, size_t lds) {
    const int rows_per_block = WAVES * R;
    const unsigned grid = (unsigned) ((a.N + rows_per_block - 1) / rows_per_block);
    hipLaunchKernelGGL((k_mmvq<T, NC, R, GLU, PRO, WAVES>), dim3(grid), dim3(WAVES * 64), lds, s, a);
}

template <typename T> static void launch_type(hipStream_t s, const mmvq_args & a0, int rows_per_wave) {
    mmvq_args a = a0;
    const int nblk = a.K / T::BLK;
    const bool glu = a.W2 != nullptr;
    if (a0.x != nullptr) {
        if constexpr (T::BLK == 256) {
            if (a0.norm_w) { if (glu) launch_stream<T, true, 2>(s, a0); else launch_stream<T, false, 2>(s, a0); }
            else           { if (glu) launch_stream<T, true, 1>(s, a0); else launch_stream<T, false, 1>(s, a0); }
            return;
        } else {
            MI_ERR("launch_mmvq: f32 prologue requested for a non K-quant type");
            abort();
        }
    }
    if (a0.ncols == 1 && (size_t) nblk * sizeof(typename T::act) <= 60 * 1024) {
        if (glu) launch_stream<T, true, 0>(s, a0); else launch_stream<T, false, 0>(s, a0);
        return;
    }
    // activations are laid out [ncols][nblk]; the column loop runs templates of exactly 8/4/2/1 columns
    int done = 0;
    while (done < a0.ncols) {
        const int left = a0.ncols - done;
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

void launch_mmvq(hipStream_t s, const mmvq_args & a, int rows_per_wave) {
    switch (a.type) {
        case GGML_TYPE_Q4_K: launch_type<T_Q4K>(s, a, rows_per_wave); break;
        case GGML_TYPE_Q5_K: launch_type<T_Q5K>(s, a, rows_per_wave); break;
        case GGML_TYPE_Q6_K: launch_type<T_Q6K>(s, a, rows_per_wave); break;
        case GGML_TYPE_Q8_0: launch_type<T_Q80>(s, a, rows_per_wave); break;
        default: MI_ERR("launch_mmvq: unsupported weight type %d", a.type); abort();
    }
}

}  // namespace mi355x
