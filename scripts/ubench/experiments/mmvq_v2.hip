// mmvq_v2.hip — LAB: batch-1 mat-vec with the waves of a workgroup on BLOCK COLUMNS instead of rows (round 4).
//
// k_mmvq_stream (csrc/mmvq.hip) gives a wave whole rows, so every wave needs the whole quantised activation row: the prologue is a
// workgroup-wide affair (every wave quantises a share into LDS, barrier, then everybody reads everything) and the second weight item
// cannot be requested before that barrier has been passed (stamps: prologue-done 2.2 us into wo, 3.8 us into ffn_down, 5.4 us into
// gate/up; HBM idles between the arrival of the first items and that moment).
//
// Here wave w owns super-block columns w, w + CW, ... of EVERY row the workgroup walks: 64 lanes = 16 rows x 4 lanes of one super-block
// (the lane map of T::load / T::dot: p = 4 b + j).  The wave needs only ITS columns of the activation row, quantises them itself into
// its own LDS blocks (same CPU-identical arithmetic, wave_quantize_q8_K) and starts — no barrier for the f32 prologue, one barrier (the
// sum of squares) for the norm prologue.  Row sums meet once, at the end: quad sums by DPP, one float per (row, wave) in LDS, one
// barrier, a fixed-order sum over the waves (deterministic) + epilogue by one thread per row.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>

#include "../../../llama_box_amd/csrc/mmvq_types.h"

namespace mi355x {

#ifndef V2_VARIANT
#define V2_VARIANT 0
#endif
// lab variants: 0 = one item requested before the prologue (+ one more inside the loop, as k_mmvq_stream); 1 = two before the prologue
#define V2_DEPTH (V2_VARIANT == 1 ? 2 : 1)

template <typename T, bool GLU, int PRO>
__global__ void __launch_bounds__(1024) k_mmvq_cols(const mmvq_args a, const int rg_shift) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(T::BLK == 256 && T::PPB == 4, "K-quants: four lanes per super-block");
    typedef typename T::act act;
    constexpr int WAVES = 16, NM = GLU ? 2 : 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nblk = a.K >> 8;
    const int RG = 1 << rg_shift, CW = WAVES >> rg_shift;  // row groups a workgroup walks side by side x waves over the columns
    const int wcol = wave & (CW - 1), wrg = wave >> (4 - rg_shift);
    const int r = lane >> 2, j = lane & 3;
    const int RW = 16 * RG;                     // rows per workgroup and pass
    const int RP = RW * (int) gridDim.x;        // rows per pass of the grid
    const int n_full = a.N / RP;                // full passes
    const int rem = a.N - n_full * RP;
    const int rem_per = (rem + (int) gridDim.x - 1) / (int) gridDim.x;  // rows of the balanced tail per workgroup (<= RW)
    const int n_pass = n_full + (rem_per > 0 ? 1 : 0);
    const int ncol = wcol < nblk ? (nblk - wcol + CW - 1) / CW : 0;

    // row of this lane in pass t, and whether it exists
    auto row_of = [&](const int t, bool & valid) {
        if (t < n_full) { valid = true; return t * RP + (int) blockIdx.x * RW + wrg * 16 + r; }
        const int lr = wrg * 16 + r;
        const int rr = n_full * RP + (int) blockIdx.x * rem_per + lr;
        valid = lr < rem_per && rr < a.N;
        return rr;
    };
    // does this wave have anything to do in pass t? (the tail may cover fewer than 16 * RG rows)
    auto wave_active = [&](const int t) { return t < n_full || (wrg * 16 < rem_per && n_full * RP + (int) blockIdx.x * rem_per + wrg * 16 < a.N); };

    struct item { typename T::raw w, w2; };
    auto load_item = [&](const int t, const int ci, item & it) {
        bool valid;
        const int row = min(row_of(t, valid), a.N - 1);
        const int p = min(wcol + ci * CW, nblk - 1) * 4 + j;  // (clamped: a wave beyond the last column still issues its first, unused request)
        it.w = T::load(a.W + (size_t) row * a.w_nb1, p);
        if (GLU) it.w2 = T::load(a.W2 + (size_t) row * a.w_nb1, p);
    };

    act * yl = (act *) smem;
    double * red_ss = (double *) (smem + (size_t) nblk * sizeof(act));
    float * red = (float *) (red_ss + WAVES);  // [pass][matrix][row of the workgroup pass (RW)][CW]

    // ---- first requests: this wave's activation columns, then its first weight item
    const float4 * x4 = (const float4 *) a.x;
    const float4 * g4 = (const float4 *) a.norm_w;
    float4 v[4], g[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int b = min(wcol + u * CW, nblk - 1);
        v[u] = x4[b * 64 + lane];
        if constexpr (PRO == 2) g[u] = g4[b * 64 + lane];
    }
    __builtin_amdgcn_sched_barrier(0);
    // item iterator: (pass, column index) in the order the wave walks them
    struct pos { int t, ci; bool have; };
    auto advance = [&](const pos q) {
        pos n{q.t, q.ci + 1, false};
        if (n.ci >= ncol) { n.ci = 0; n.t = q.t + 1; }
        n.have = q.have && n.t < n_pass && wave_active(n.t);
        return n;
    };
    pos at{0, 0, ncol > 0 && n_pass > 0 && wave_active(0)};
    // (a wave without work in pass 0 has none in the tail either when n_full == 0; with n_full > 0 pass 0 is a full one)
    constexpr int D = V2_DEPTH;  // items requested before the prologue
    item ring[D];
    pos rp[D];
    {
        pos q = at;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            rp[d] = q;
            if (d == 0 || q.have) load_item(q.have ? q.t : 0, q.have ? q.ci : 0, ring[d]);
            q = advance(q);
        }
    }

    // ---- activation prologue: this wave's columns only
    float scale = 1.0f;
    if constexpr (PRO == 2) {
        // the norm's sum of squares is the one thing every wave needs from every other: the row is shared out over ALL 16 waves here
        // (wave w sums 256-value chunks w, w + 16, ...), whatever the column map
        double ss = 0.0;
        float4 sq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // (K <= 16384: four chunks per wave at most; with CW == 16 they ARE the wave's columns, already requested)
            const int b0 = wave + u * WAVES;
            if (CW == WAVES) sq[u] = v[u];
            else sq[u] = x4[min(b0, nblk - 1) * 64 + lane];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4 q = sq[u];
            if (wave + u * WAVES < nblk) ss += (double) (q.x * q.x) + (double) (q.y * q.y) + (double) (q.z * q.z) + (double) (q.w * q.w);
        }
        ss = wave_sum_d(ss);
        if (lane == 0) red_ss[wave] = ss;
        __syncthreads();
        double tot = 0.0;
#pragma unroll
        for (int i = 0; i < WAVES; ++i) tot += red_ss[i];
        const float mean = (float) (tot / (double) a.K);
        scale = 1.0f / sqrtf(mean + a.eps);
    }
    for (int c0 = 0; c0 < ncol; c0 += 4) {
        if (c0 > 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int b = min(wcol + (c0 + u) * CW, nblk - 1);
                v[u] = x4[b * 64 + lane];
                if constexpr (PRO == 2) g[u] = g4[b * 64 + lane];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int b = wcol + (c0 + u) * CW;
            if (c0 + u < ncol) {
                float q[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                if constexpr (PRO == 2) {
                    q[0] = (q[0] * scale) * g[u].x;
                    q[1] = (q[1] * scale) * g[u].y;
                    q[2] = (q[2] * scale) * g[u].z;
                    q[3] = (q[3] * scale) * g[u].w;
                    if (a.norm_out && blockIdx.x == 0 && wrg == 0) ((float4 *) a.norm_out)[b * 64 + lane] = make_float4(q[0], q[1], q[2], q[3]);
                }
                wave_quantize_q8_K(q, lane, yl + b);
            }
        }
    }
    // (no barrier: a wave reads back only the blocks it wrote itself, and a wave's LDS operations complete in order)

    float acc = 0.0f, acc2 = 0.0f;
    pos nq = advance(rp[D - 1]);  // the next item to request
    while (rp[0].have) {
        item nxt;
        if (nq.have) load_item(nq.t, nq.ci, nxt);
        const int t = rp[0].t, ci = rp[0].ci;
        const int p = (wcol + ci * CW) * 4 + j;
        T::template dot<1>(ring[0].w, p, yl, nblk, &acc);
        if (GLU) T::template dot<1>(ring[0].w2, p, yl, nblk, &acc2);
        if (ci == ncol - 1) {
            // the quad's four lanes hold one row's partial over this wave's columns
            float s = acc;
            s += dpp_f32<MI_DPP_QUAD_XOR1>(s);
            s += dpp_f32<MI_DPP_QUAD_XOR2>(s);
            float s2 = 0.0f;
            if (GLU) {
                s2 = acc2;
                s2 += dpp_f32<MI_DPP_QUAD_XOR1>(s2);
                s2 += dpp_f32<MI_DPP_QUAD_XOR2>(s2);
            }
            if (j == 0) {
                float * dst = red + ((size_t) (t * NM) * RW + wrg * 16 + r) * CW + wcol;
                dst[0] = s;
                if (GLU) dst[(size_t) RW * CW] = s2;
            }
            acc = 0.0f;
            acc2 = 0.0f;
        }
#pragma unroll
        for (int d = 0; d + 1 < D; ++d) { ring[d] = ring[d + 1]; rp[d] = rp[d + 1]; }
        ring[D - 1] = nxt;
        rp[D - 1] = nq;
        nq = advance(nq);
    }
    __syncthreads();
    // ---- rows of the workgroup: fixed-order sum over the column waves, epilogue
    const int ncw = min(CW, nblk);  // column waves that had work
    for (int i = tid; i < n_pass * RW; i += 1024) {
        const int tp = i / RW, lr = i - tp * RW;
        int row;
        bool valid;
        if (tp < n_full) { row = tp * RP + (int) blockIdx.x * RW + lr; valid = true; }
        else { row = n_full * RP + (int) blockIdx.x * rem_per + lr; valid = lr < rem_per && row < a.N; }
        if (!valid) continue;
        const float * src = red + ((size_t) (tp * NM) * RW + lr) * CW;
        float s = 0.0f;
        for (int w = 0; w < ncw; ++w) s += src[w];
        if (GLU) {
            const float * src2 = src + (size_t) RW * CW;
            float u = 0.0f;
            for (int w = 0; w < ncw; ++w) u += src2[w];
            s = silu_f(s) * u;
        }
        if (a.add) s += a.add[row];
        if (a.add2) s += a.add2[row];
        a.dst[row] = s;
    }
}

template <typename T, bool GLU, int PRO> static bool launch_cols(hipStream_t s, const mmvq_args & a) {
    const int nblk = a.K / 256;
    // 16 waves = CW column waves x RG row groups: short rows (K < 3072) walk 2 or 4 row groups side by side
    const int rg_shift = nblk >= 12 ? 0 : (nblk >= 6 ? 1 : 2);
    const int RW = 16 << rg_shift;
    const unsigned grid = (unsigned) std::min<int64_t>(256, ((int64_t) a.N + RW - 1) / RW);
    const int n_pass = (a.N + RW * (int) grid - 1) / (RW * (int) grid) + 1;
    const size_t lds = (size_t) nblk * sizeof(q8k_dev) + 16 * sizeof(double) + (size_t) n_pass * (GLU ? 2 : 1) * RW * (16 >> rg_shift) * sizeof(float);
    if (lds > 160 * 1024) return false;
    static bool raised = false;
    if (lds > 64 * 1024 && !raised) {
        if (hipFuncSetAttribute((const void *) k_mmvq_cols<T, GLU, PRO>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
        raised = true;
    }
    hipLaunchKernelGGL((k_mmvq_cols<T, GLU, PRO>), dim3(grid), dim3(1024), lds, s, a, rg_shift);
    return true;
}

template <typename T> static bool launch_cols_type(hipStream_t s, const mmvq_args & a) {
    const bool glu = a.W2 != nullptr;
    if (a.x == nullptr || a.fa_part != nullptr || a.ncols != 1 || (a.K % 256) != 0) return false;
    if (a.norm_w) {
        if (a.K > 16384) return false;  // (the norm prologue keeps the wave's columns in registers across the sum of squares: four at most)
        return glu ? launch_cols<T, true, 2>(s, a) : launch_cols<T, false, 2>(s, a);
    }
    return glu ? launch_cols<T, true, 1>(s, a) : launch_cols<T, false, 1>(s, a);
}

bool launch_mmvq_v2(hipStream_t s, const mmvq_args & a) {
    switch (a.type) {
        case GGML_TYPE_Q4_K: return launch_cols_type<T_Q4K>(s, a);
        case GGML_TYPE_Q5_K: return launch_cols_type<T_Q5K>(s, a);
        case GGML_TYPE_Q6_K: return launch_cols_type<T_Q6K>(s, a);
        default: return false;
    }
}

}  // namespace mi355x
