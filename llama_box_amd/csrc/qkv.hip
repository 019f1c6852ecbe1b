// qkv.hip — one launch for a token's Q/K/V projections: [RMS_NORM -> MUL ->] {MUL_MAT wq, wk, wv} [-> ADD bias]
// -> ROPE(q), ROPE(k) -> SET_ROWS(k cache), SET_ROWS(v cache).
//
// Why: at batch 1 these ten graph nodes move 14 MB of weights (Llama-3-8B) but, launched one by one, cost ten
// dependent kernel boundaries plus ten memory round trips — the profile of round 1 showed every small kernel
// sitting at a ~4.5 us floor (producer write-back + consumer miss), i.e. ~45 us of a layer's ~110 us.  Here the
// weight rows of all three matrices form one work list; a wave owns a PAIR of rows (the two rows a rotary pair
// mixes: (2i, 2i+1) for the normal layout, (i, i + d/2) for NeoX), streams both with all loads issued before the
// activation prologue, and lane 0 finishes the pair: + bias, rotation by the (cos, sin) of this token's position
// (table built once per workgroup, same f32 multiply chain as the CPU), then either f32 out (Q) or f16 straight
// into the KV cache row selected by the SET_ROWS index (K, V).  Arithmetic per element is identical to the
// unfused kernels, so fused and unfused graphs agree bit for bit.
#include <algorithm>

#include "mmvq_types.h"

namespace mi355x {

// pairs a lane holds per row and trip: sized so that K = 4096 is ONE trip per row for every format
template <typename T> struct qkv_u { static constexpr int U = T::PPB == 16 ? 4 : (T::DW > 8 ? 1 : 2); static constexpr int DW = U * T::DW; };

template <typename T> __device__ __forceinline__ void chunk_load(const uint8_t * __restrict__ row, const int c, const int lane, const int npairs, uint32_t * d) {
    constexpr int U = qkv_u<T>::U;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int p = (c * U + u) * 64 + lane;
        if (p < npairs) {
            const typename T::raw r = T::load(row, p);
            T::pack(r, d + u * T::DW);
        }
    }
}
template <typename T> __device__ __forceinline__ void chunk_dot(const uint32_t * d, const int c, const int lane, const int npairs, const q8k_dev * __restrict__ y, const int nblk, float & acc) {
    constexpr int U = qkv_u<T>::U;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int p = (c * U + u) * 64 + lane;
        if (p < npairs) {
            const typename T::raw r = T::unpack(d + u * T::DW);
            T::template dot<1>(r, p, y, nblk, &acc);
        }
    }
}

// TA: weight format of the segments flagged alt == 0, TB: of those flagged alt == 1 (Q4_K_M: wq/wk Q4_K, wv Q6_K in the
// "more bits" layers).  The branch on sg.alt is wave-uniform.
template <typename TA, typename TB>
__global__ void __launch_bounds__(1024) k_qkv_stream(const qkv_args a) {
    constexpr int BUF_DW = qkv_u<TA>::DW > qkv_u<TB>::DW ? qkv_u<TA>::DW : qkv_u<TB>::DW;
    auto load_any = [](const int alt, const uint8_t * row, const int c, const int lane, const int npairs, uint32_t * d) {
        if (alt) chunk_load<TB>(row, c, lane, npairs, d); else chunk_load<TA>(row, c, lane, npairs, d);
    };
    auto dot_any = [](const int alt, const uint32_t * d, const int c, const int lane, const int npairs, const q8k_dev * y, const int nblk, float & acc) {
        if (alt) chunk_dot<TB>(d, c, lane, npairs, y, nblk, acc); else chunk_dot<TA>(d, c, lane, npairs, y, nblk, acc);
    };
    auto type_ppb = [](const int alt) { return alt ? (int) TB::PPB : (int) TA::PPB; };
    auto type_pairs_per_chunk = [](const int alt) { return 64 * (alt ? (int) qkv_u<TB>::U : (int) qkv_u<TA>::U); };
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int WAVES = 16, QB = 2;  // QB activation blocks per wave per prologue trip (norm: nblk <= QB*WAVES)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nblk = a.K / 256;
    q8k_dev * yl = (q8k_dev *) smem;
    double * red = (double *) (smem + (size_t) nblk * sizeof(q8k_dev));
    float * cs_tab = (float *) (red + WAVES);  // [head_dim/2][2]
    const int GW = gridDim.x * WAVES;
    const int half = a.head_dim >> 1;
    const int u0 = a.seg[0].N >> 1;
    const int u1 = u0 + (a.nseg > 1 ? a.seg[1].N >> 1 : 0);
    const int UT = u1 + (a.nseg > 2 ? a.seg[2].N >> 1 : 0);

    int u = blockIdx.x * WAVES + wave;
    bool have = u < UT;
    int si = 0, r0 = 0, r1 = 0, pair_i = 0;
    qkv_seg sg = a.seg[0];
    auto locate = [&]() {
        si = u < u0 ? 0 : (u < u1 ? 1 : 2);
        sg = si == 0 ? a.seg[0] : (si == 1 ? a.seg[1] : a.seg[2]);
        const int lu = u - (si == 0 ? 0 : (si == 1 ? u0 : u1));
        if (sg.rope) {
            const int h = lu / half;
            pair_i = lu - h * half;
            r0 = h * a.head_dim + (a.neox ? pair_i : 2 * pair_i);
            r1 = a.neox ? r0 + half : r0 + 1;
        } else {
            r0 = 2 * lu;
            r1 = r0 + 1;
        }
    };
    uint32_t buf[2][BUF_DW];
    int npairs = 0;
    if (have) {
        locate();
        npairs = nblk * type_ppb(sg.alt);
        load_any(sg.alt, sg.W + (size_t) r0 * sg.w_nb1, 0, lane, npairs, buf[0]);
        load_any(sg.alt, sg.W + (size_t) r1 * sg.w_nb1, 0, lane, npairs, buf[1]);
    }

    // ---- prologue A: rotary table for this token (threads 0 .. head_dim/2-1)
    if (tid < half && a.pos != nullptr) {
        float cs, sn;
        rope_cos_sin(tid, (float) a.pos[0], a.freq_factors, rope_consts{a.theta_scale, a.freq_scale, a.ext_factor, a.attn_factor, a.corr0, a.corr1}, cs, sn);
        cs_tab[2 * tid] = cs;
        cs_tab[2 * tid + 1] = sn;
    }
    // ---- prologue B: activations -> Q8_K in LDS (see mmvq.hip: PRO 1 / PRO 2)
    {
        const float4 * x4 = (const float4 *) a.x;
        const float4 * w4 = (const float4 *) a.norm_w;
        const bool norm = a.norm_w != nullptr;
        for (int b0 = wave; b0 < nblk || b0 == wave; b0 += QB * WAVES) {
            float4 v[QB], g[QB];
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                const int b = b0 + q * WAVES;
                if (b < nblk) {
                    v[q] = x4[b * 64 + lane];
                    g[q] = norm ? w4[b * 64 + lane] : make_float4(1.f, 1.f, 1.f, 1.f);
                } else {
                    v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                    g[q] = v[q];
                }
            }
            float scale = 1.0f;
            if (norm) {  // caller guarantees nblk <= QB*WAVES (K <= 8192): the whole row is in this one batch
                double ss = 0.0;
#pragma unroll
                for (int q = 0; q < QB; ++q) ss += (double) (v[q].x * v[q].x) + (double) (v[q].y * v[q].y) + (double) (v[q].z * v[q].z) + (double) (v[q].w * v[q].w);
                ss = wave_sum_d(ss);
                if (lane == 0) red[wave] = ss;
                __syncthreads();
                double tot = 0.0;
#pragma unroll
                for (int i = 0; i < WAVES; ++i) tot += red[i];
                const float mean = (float) (tot / (double) a.K);
                scale = 1.0f / sqrtf(mean + a.eps);
            }
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                const int b = b0 + q * WAVES;
                if (b < nblk) {
                    float t[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
                    if (norm) {
                        t[0] = (t[0] * scale) * g[q].x;
                        t[1] = (t[1] * scale) * g[q].y;
                        t[2] = (t[2] * scale) * g[q].z;
                        t[3] = (t[3] * scale) * g[q].w;
                    }
                    wave_quantize_q8_K(t, lane, yl + b);
                }
            }
            if (norm) break;
        }
    }
    __syncthreads();

    const int64_t slot = a.slot ? a.slot[0] : 0;
    while (have) {
        float acc0 = 0.0f, acc1 = 0.0f;
        const int per_chunk = type_pairs_per_chunk(sg.alt);
        const int nch = (npairs + per_chunk - 1) / per_chunk;
        const uint8_t * row0 = sg.W + (size_t) r0 * sg.w_nb1;
        const uint8_t * row1 = sg.W + (size_t) r1 * sg.w_nb1;
        for (int c = 0; c < nch; ++c) {
            if (c > 0) {
                load_any(sg.alt, row0, c, lane, npairs, buf[0]);
                load_any(sg.alt, row1, c, lane, npairs, buf[1]);
            }
            dot_any(sg.alt, buf[0], c, lane, npairs, yl, nblk, acc0);
            dot_any(sg.alt, buf[1], c, lane, npairs, yl, nblk, acc1);
        }
        float v0 = wave_sum(acc0), v1 = wave_sum(acc1);
        if (lane == 0) {
            if (sg.bias) {
                v0 += sg.bias[r0];
                v1 += sg.bias[r1];
            }
            if (sg.rope) {
                const float cs = cs_tab[2 * pair_i], sn = cs_tab[2 * pair_i + 1];
                const float t0 = v0 * cs - v1 * sn;
                const float t1 = v0 * sn + v1 * cs;
                v0 = t0;
                v1 = t1;
            }
            if (sg.store_f16) {
                uint16_t * o = (uint16_t *) (sg.out + slot * sg.row_stride);
                o[r0] = f2h(v0);
                o[r1] = f2h(v1);
            } else {
                float * o = (float *) sg.out;
                o[r0] = v0;
                o[r1] = v1;
            }
        }
        u += GW;
        have = u < UT;
        if (have) {
            locate();
            npairs = nblk * type_ppb(sg.alt);
            load_any(sg.alt, sg.W + (size_t) r0 * sg.w_nb1, 0, lane, npairs, buf[0]);
            load_any(sg.alt, sg.W + (size_t) r1 * sg.w_nb1, 0, lane, npairs, buf[1]);
        }
    }
}

template <typename TA, typename TB> static void launch_qkv_t(hipStream_t s, const qkv_args & a, unsigned grid, size_t lds) {
    hipLaunchKernelGGL((k_qkv_stream<TA, TB>), dim3(grid), dim3(1024), lds, s, a);
}

bool qkv_types_supported(int ta, int tb) {
    auto ok = [](int t) { return t == GGML_TYPE_Q4_K || t == GGML_TYPE_Q5_K || t == GGML_TYPE_Q6_K; };
    return ok(ta) && ok(tb);
}

// a.seg[i].alt selects the second format; the caller guarantees at most two distinct formats among the segments
void launch_qkv(hipStream_t s, const qkv_args & a, int type_a, int type_b) {
    const int nblk = a.K / 256;
    int units = 0;
    for (int i = 0; i < a.nseg; ++i) units += a.seg[i].N / 2;
    const size_t lds = (size_t) nblk * sizeof(q8k_dev) + 16 * sizeof(double) + (size_t) a.head_dim * sizeof(float) + 16;
    const unsigned grid = (unsigned) std::min(256, (units + 15) / 16);
#define QKV_CASE(A, TA_, B, TB_) if (type_a == A && type_b == B) { launch_qkv_t<TA_, TB_>(s, a, grid, lds); return; }
    QKV_CASE(GGML_TYPE_Q4_K, T_Q4K, GGML_TYPE_Q4_K, T_Q4K) QKV_CASE(GGML_TYPE_Q4_K, T_Q4K, GGML_TYPE_Q5_K, T_Q5K) QKV_CASE(GGML_TYPE_Q4_K, T_Q4K, GGML_TYPE_Q6_K, T_Q6K)
    QKV_CASE(GGML_TYPE_Q5_K, T_Q5K, GGML_TYPE_Q4_K, T_Q4K) QKV_CASE(GGML_TYPE_Q5_K, T_Q5K, GGML_TYPE_Q5_K, T_Q5K) QKV_CASE(GGML_TYPE_Q5_K, T_Q5K, GGML_TYPE_Q6_K, T_Q6K)
    QKV_CASE(GGML_TYPE_Q6_K, T_Q6K, GGML_TYPE_Q4_K, T_Q4K) QKV_CASE(GGML_TYPE_Q6_K, T_Q6K, GGML_TYPE_Q5_K, T_Q5K) QKV_CASE(GGML_TYPE_Q6_K, T_Q6K, GGML_TYPE_Q6_K, T_Q6K)
#undef QKV_CASE
    MI_ERR("launch_qkv: unsupported weight formats %d/%d", type_a, type_b);
    abort();
}

}  // namespace mi355x
