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
#include "kv_quant.h"

namespace mi355x {

// pairs a lane holds per row and trip: sized so that K = 4096 is ONE trip per row for every format
template <typename T> struct qkv_u { static constexpr int U = T::DW > 8 ? 1 : 2; static constexpr int DW = U * T::DW; };

// the weight registers of one unit (two rows) of format T: statically indexed, so they stay in VGPRs
template <typename T> struct qkv_regs { typename T::raw w[2][qkv_u<T>::U]; };
template <typename T> __device__ __forceinline__ void qkv_load(const uint8_t * __restrict__ row0, const uint8_t * __restrict__ row1, const int c, const int lane,
                                                               const int npairs, qkv_regs<T> & r) {
    constexpr int U = qkv_u<T>::U;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int p = (c * U + u) * 64 + lane;
        if (p < npairs) {
            r.w[0][u] = T::load(row0, p, npairs / T::PPB);
            r.w[1][u] = T::load(row1, p, npairs / T::PPB);
        }
    }
}
template <typename T> __device__ __forceinline__ void qkv_dot(const qkv_regs<T> & r, const int c, const int lane, const int npairs, const typename T::act * __restrict__ y,
                                                              const int nblk, float & acc0, float & acc1) {
    constexpr int U = qkv_u<T>::U;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int p = (c * U + u) * 64 + lane;
        if (p < npairs) {
            T::template dot<1>(r.w[0][u], p, y, nblk, &acc0);
            T::template dot<1>(r.w[1][u], p, y, nblk, &acc1);
        }
    }
}
// all trips of one unit; trip 0 is already in `r`
template <typename T, bool PIPE> __device__ __forceinline__ void qkv_unit(qkv_regs<T> & r, const uint8_t * row0, const uint8_t * row1, const int lane, const int nblk,
                                                                          const typename T::act * __restrict__ y, float & acc0, float & acc1) {
    const int npairs = nblk * T::PPB;
    constexpr int per_chunk = 64 * qkv_u<T>::U;
    const int nch = (npairs + per_chunk - 1) / per_chunk;
    // K > 4096: the unit has several trips.  The next trip's weights are requested BEFORE the current one is multiplied (a second register set,
    // statically named: no copy, so hipcc keeps them in flight) — a trip used to be requested only after the previous one's dot products, one
    // dependent memory round trip per 4096 values of K (Llama-3-70B: K = 8192; a tensor-split rank's fused QKV launch 14-15 us for 5.9 MB)
    if constexpr (!PIPE) {
        for (int c = 0; c < nch; ++c) {
            if (c > 0) qkv_load<T>(row0, row1, c, lane, npairs, r);
            qkv_dot<T>(r, c, lane, npairs, y, nblk, acc0, acc1);
        }
        return;
    }
    // (PIPE: the K > 4096 instantiation, 8 waves per workgroup so that the second register set fits without scratch)
    qkv_regs<T> r2;
    for (int c = 0; c < nch; c += 2) {
        if (c + 1 < nch) qkv_load<T>(row0, row1, c + 1, lane, npairs, r2);
        qkv_dot<T>(r, c, lane, npairs, y, nblk, acc0, acc1);
        if (c + 1 < nch) {
            if (c + 2 < nch) qkv_load<T>(row0, row1, c + 2, lane, npairs, r);
            qkv_dot<T>(r2, c + 1, lane, npairs, y, nblk, acc0, acc1);
        }
    }
}

// The work of one workgroup for the segments stored in weight format T (seg.alt == alt); `wg` of `nwg` workgroups share
// those segments.  A launch with two formats (Q4_K_M keeps wv in Q6_K in its "more bits" layers) gives each format its
// own range of workgroups: a workgroup only ever executes one instantiation, so the register demand is the maximum of
// the two, not the sum — holding both register sets in one code path spilled weight registers to scratch memory.
// Q8S: the launch stores into a block_q8_0 KV cache (a separate instantiation: the block assembly costs registers the default
// f16-cache kernel does not have to spare)
// BND: the workgroup size the kernel is compiled for.  1024 caps a wave at 128 registers, which the Q5_K register sets exceed by a few (6 .. 16 dwords of scratch:
// a launch with a scratch frame pays for it at every dispatch); Qwen2-7B's launches use 9 waves, so the Q5_K forms also exist compiled for 640 threads (round 6)
template <typename TA, typename TB, bool Q8S, bool PIPE = false, int BND = 1024>
__global__ void __launch_bounds__(PIPE ? 512 : BND) k_qkv_stream2(const qkv_args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // (a generic lambda, not a device function: passing the kernel-argument struct to a function copies it to scratch)
    // (u_first / GW: this wave's first unit among the segments of its format, and the number of waves that share them)
    auto body = [&](auto tag, const int alt, const int u_first, const int GW) {
    using T = decltype(tag);
    constexpr int MAXW = 16, QB = PIPE ? 4 : 2;  // QB activation blocks per wave per prologue trip (norm: nblk <= QB*WAVES)
    const int WAVES = (int) blockDim.x >> 6;  // 8..16 waves: the launcher sizes the workgroup so that units/WAVES ~ 256 workgroups
    const int tid = threadIdx.x, lane = tid & 63;
    // wave-uniform, and the compiler must know it: the unit a wave works on — its segment descriptor, row numbers, pointers — then lives
    // in SGPRs instead of every lane's VGPRs (two-format launch: 128 VGPRs + 20 bytes of scratch -> 120 and none; a launch with a
    // scratch frame pays for it at every dispatch)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    typedef typename T::act act;            // Q8_K blocks for the K-quants, Q8_0 blocks (8 per 256-value chunk) for Q8_0 weights
    constexpr int BPC = 256 / T::BLK;
    const int nchk = a.K / 256, nblk = nchk * BPC;
    act * yl = (act *) smem;
    double * red = (double *) (smem + (size_t) nblk * sizeof(act));
    float * cs_tab = (float *) (red + MAXW);  // [head_dim/2][2]
    float * stash = cs_tab + a.head_dim;      // [32]: one Q8_0 block of cache-row values (store == 2)
    const int half = a.head_dim >> 1;
    const int u0 = a.seg[0].alt == alt ? a.seg[0].N >> 1 : 0;
    const int u1 = u0 + (a.nseg > 1 && a.seg[1].alt == alt ? a.seg[1].N >> 1 : 0);
    const int UT = u1 + (a.nseg > 2 && a.seg[2].alt == alt ? a.seg[2].N >> 1 : 0);

    int u = u_first;
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
    qkv_regs<T> ra;
    if (have) {
        locate();
        qkv_load<T>(sg.W + (size_t) r0 * sg.w_nb1, sg.W + (size_t) r1 * sg.w_nb1, 0, lane, nblk * T::PPB, ra);
    }

    const int64_t slot = a.slot ? a.slot[0] : 0;  // (requested here: after the prologue it was one more dependent round trip in front of the cache-row stores)
    // ---- prologue A: rotary table for this token (threads 0 .. head_dim/2-1)
    if (tid < half && a.pos != nullptr) {
        float cs, sn;
        if (a.rope_tab) {  // written once per graph run: the multiply chain + accurate cosf / sinf are ~1 us in the prologue of this workgroup's first wave
            const float2 t2 = ((const float2 *) a.rope_tab)[tid];
            cs = t2.x;
            sn = t2.y;
        } else
            rope_cos_sin(tid, (float) a.pos[0], a.freq_factors, rope_consts{a.theta_scale, a.freq_scale, a.ext_factor, a.attn_factor, a.corr0, a.corr1}, cs, sn);
        cs_tab[2 * tid] = cs;
        cs_tab[2 * tid + 1] = sn;
    }
    // ---- prologue B: activations -> Q8_K in LDS (see mmvq.hip: PRO 1 / PRO 2)
    {
        const float4 * x4 = (const float4 *) a.x;
        const float4 * w4 = (const float4 *) a.norm_w;
        const bool norm = a.norm_w != nullptr;
        for (int b0 = wave; b0 < nchk || b0 == wave; b0 += QB * WAVES) {
            float4 v[QB], g[QB];
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                const int b = b0 + q * WAVES;
                if (b < nchk) {
                    v[q] = x4[b * 64 + lane];
                    g[q] = norm ? w4[b * 64 + lane] : make_float4(1.f, 1.f, 1.f, 1.f);
                } else {
                    v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                    g[q] = v[q];
                }
            }
            float scale = 1.0f;
            if (norm) {  // caller guarantees nblk <= QB*WAVES (K <= 8192): the whole row is in this one batch
                double tot = 0.0;
                if (a.ss_in) {
                    // the mat-vec that wrote x left its sum of squares as partial sums (mmvq.hip, ss_out): no pass over x, no barrier
                    double p4[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) p4[q] = lane + 64 * q < a.ss_n ? a.ss_in[lane + 64 * q] : 0.0;
                    tot = wave_sum_d(((p4[0] + p4[1]) + p4[2]) + p4[3]);
                } else {
                    double ss = 0.0;
#pragma unroll
                    for (int q = 0; q < QB; ++q) ss += (double) (v[q].x * v[q].x) + (double) (v[q].y * v[q].y) + (double) (v[q].z * v[q].z) + (double) (v[q].w * v[q].w);
                    ss = wave_sum_d(ss);
                    if (lane == 0) red[wave] = ss;
                    __syncthreads();
#pragma unroll
                    for (int i = 0; i < MAXW; ++i) tot += i < WAVES ? red[i] : 0.0;
                }
                const float mean = (float) (tot / (double) a.K);
                scale = 1.0f / sqrtf(mean + a.eps);
            }
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                const int b = b0 + q * WAVES;
                if (b < nchk) {
                    float t[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
                    if (norm) {
                        t[0] = (t[0] * scale) * g[q].x;
                        t[1] = (t[1] * scale) * g[q].y;
                        t[2] = (t[2] * scale) * g[q].z;
                        t[3] = (t[3] * scale) * g[q].w;
                        if (a.norm_out && blockIdx.x == 0) ((float4 *) a.norm_out)[b * 64 + lane] = make_float4(t[0], t[1], t[2], t[3]);
                    }
                    if constexpr (BPC == 1) wave_quantize_q8_K(t, lane, yl + b);
                    else wave_quantize_q8_0(t, lane, yl + (size_t) b * BPC);
                }
            }
            if (norm) break;
        }
    }
    __syncthreads();

    while (have) {
        float acc0 = 0.0f, acc1 = 0.0f;
        {
            const uint8_t * row0 = sg.W + (size_t) r0 * sg.w_nb1;
            const uint8_t * row1 = sg.W + (size_t) r1 * sg.w_nb1;
            qkv_unit<T, PIPE>(ra, row0, row1, lane, nblk, yl, acc0, acc1);
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
            if (Q8S && sg.store == 2) {
                stash[2 * wave] = v0;
                stash[2 * wave + 1] = v1;
            } else if (sg.store == 1) {
                uint16_t * o = (uint16_t *) (sg.out + slot * sg.row_stride);
                o[r0] = f2h(v0);
                o[r1] = f2h(v1);
            } else if (sg.store == 4) {  // a bf16 cache (-ctk / -ctv bf16)
                uint16_t * o = (uint16_t *) (sg.out + slot * sg.row_stride);
                o[r0] = f2bf(v0);
                o[r1] = f2bf(v1);
            } else if (sg.store == 3) {
                uint16_t * o = (uint16_t *) sg.out;
                const int64_t * ix = (const int64_t *) (uintptr_t) sg.row_stride;
                o[ix[r0]] = f2h(v0);
                o[ix[r1]] = f2h(v1);
            } else {
                float * o = (float *) sg.out;
                o[r0] = v0;
                o[r1] = v1;
            }
        }
        if (Q8S && sg.store == 2) {
            // quantised KV cache: the 16 waves of this workgroup hold 32 consecutive values of the cache row (the launcher
            // aligns the units; segment and trip count are workgroup-uniform) = one block_q8_0, quantised as
            // quantize_row_q8_0_ref does: d = amax / 127 (stored f16), q = roundf(x * (1 / d))
            __syncthreads();
            if (wave == 0 && sg.kvt != GGML_TYPE_Q8_0) {
                // the other block formats of -ctk / -ctv (round 6: their rows used to leave as f32 and a second launch stored them — 7 us per layer): the
                // type's quantize_row_*_ref with a lane per value (kv_quant.h), byte for byte; the type is uniform over the segment
                const float x = stash[lane & 31];
                char * row = sg.out + slot * sg.row_stride;
                const size_t b = (size_t) ((r0 & ~31) >> 5);
                switch (sg.kvt) {
                    case GGML_TYPE_Q4_0: quantize_block_lanes<GGML_TYPE_Q4_0>(x, lane, row + b * 18, lane < 32); break;
                    case GGML_TYPE_Q4_1: quantize_block_lanes<GGML_TYPE_Q4_1>(x, lane, row + b * 20, lane < 32); break;
                    case GGML_TYPE_Q5_0: quantize_block_lanes<GGML_TYPE_Q5_0>(x, lane, row + b * 22, lane < 32); break;
                    case GGML_TYPE_Q5_1: quantize_block_lanes<GGML_TYPE_Q5_1>(x, lane, row + b * 24, lane < 32); break;
                    default: quantize_block_lanes<GGML_TYPE_IQ4_NL>(x, lane, row + b * 18, lane < 32); break;
                }
            } else if (wave == 0) {
                const float x = stash[lane & 31];
                float amax = fabsf(x);
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
                const float d = amax / 127.0f;
                const float id = d != 0.0f ? 1.0f / d : 0.0f;
                const int q = (int) roundf(x * id) & 0xFF;
                const int qn = __shfl_down(q, 1);
                char * blk = sg.out + slot * sg.row_stride + (size_t) ((r0 & ~31) >> 5) * 34;  // wave 0 owns rows base, base + 1
                if (lane < 32 && (lane & 1) == 0) *(uint16_t *) (blk + 2 + lane) = (uint16_t) (q | (qn << 8));
                if (lane == 0) *(uint16_t *) blk = f2h(d);
            }
            __syncthreads();
        }
        u += GW;
        have = u < UT;
        if (have) {
            locate();
            qkv_load<T>(sg.W + (size_t) r0 * sg.w_nb1, sg.W + (size_t) r1 * sg.w_nb1, 0, lane, nblk * T::PPB, ra);
        }
    }
    };
    const int nwv = (int) blockDim.x >> 6, wv = __builtin_amdgcn_readfirstlane((int) threadIdx.x >> 6);
    if constexpr (std::is_same<TA, TB>::value) {
        body(TA{}, 0, (int) blockIdx.x * nwv + wv, (int) gridDim.x * nwv);
    } else {
        // the two formats share the launch wave by wave (wv_a > 0, round 6): a workgroup's waves may run different instantiations — each executes the same prologue
        // and the same number of workgroup barriers — or workgroup by workgroup (the block-format cache stores add barriers per unit).  ONE call site per
        // instantiation: a second one stops the compiler from inlining the body and sends the argument block through scratch memory
        bool is_a;
        int uf, gwn;
        if (!Q8S && a.wv_a > 0) {
            const int gw = (int) blockIdx.x * nwv + wv, total = (int) gridDim.x * nwv;
            is_a = gw < a.wv_a;
            uf = is_a ? gw : gw - a.wv_a;
            gwn = is_a ? a.wv_a : total - a.wv_a;
        } else {
            is_a = (int) blockIdx.x < a.wg_a;
            uf = (is_a ? (int) blockIdx.x : (int) blockIdx.x - a.wg_a) * nwv + wv;
            gwn = (is_a ? a.wg_a : (int) gridDim.x - a.wg_a) * nwv;
        }
        if (is_a) body(TA{}, 0, uf, gwn);
        else body(TB{}, 1, uf, gwn);
    }
}


bool qkv_types_supported(int ta, int tb) {
    auto ok = [](int t) { return t == GGML_TYPE_Q4_K || t == GGML_TYPE_Q5_K || t == GGML_TYPE_Q6_K; };
    if (ta == tb) return ok(ta) || ta == GGML_TYPE_Q8_0;
    return (ta == GGML_TYPE_Q4_K && tb == GGML_TYPE_Q6_K) || (ta == GGML_TYPE_Q5_K && tb == GGML_TYPE_Q6_K) || (ta == GGML_TYPE_Q4_K && tb == GGML_TYPE_Q5_K);
}

// segments with alt == 0 are stored in type_a, those with alt == 1 in type_b (type_b == type_a: one format)
void launch_qkv(hipStream_t s, const qkv_args & a0, int type_a, int type_b) {
    qkv_args a = a0;
    const int nblk = a.K / 256;  // 256-value chunks
    int units[2] = {0, 0};
    double bytes[2] = {0, 0};
    for (int i = 0; i < a.nseg; ++i) {
        units[a.seg[i].alt ? 1 : 0] += a.seg[i].N / 2;
        bytes[a.seg[i].alt ? 1 : 0] += (double) a.seg[i].N * (double) a.seg[i].w_nb1;
    }
    const size_t lds = (size_t) nblk * (type_a == GGML_TYPE_Q8_0 ? 8 * sizeof(q80_dev) : sizeof(q8k_dev)) + 16 * sizeof(double) + (size_t) a.head_dim * sizeof(float) + 32 * sizeof(float) + 16;
    bool q8_store = false;
    for (int i = 0; i < a.nseg; ++i) q8_store = q8_store || a.seg[i].store == 2;
    // waves per workgroup: 8..16, chosen so that the row-pair units spread over ~256 workgroups (Llama-3-8B: 3072 units ->
    // 12 waves x 256 workgroups; with fixed 16-wave workgroups a quarter of the CUs had nothing to do); the norm prologue
    // needs the whole activation row in one batch of 2 blocks per wave
    static const int force_nw = getenv("GGML_MI355X_QKV_WAVES") ? atoi(getenv("GGML_MI355X_QKV_WAVES")) : 0;
    // K > 4096 (a unit has several trips) on an f16 cache: the pipelined instantiation — the next trip in flight under the current one's dot
    // products, four activation blocks per wave in the prologue, 8 waves (256 registers each: the Q5_K / Q6_K register sets spill at 12)
    static const bool pipe_on = !getenv("GGML_MI355X_QKV_PIPE") || atoi(getenv("GGML_MI355X_QKV_PIPE")) != 0;
    // ... where 16-wave workgroups would leave half the chip without one (a tensor-split rank's 640 row pairs: 40 workgroups -> 80; measured
    // 13.9 -> 11.9 and 15.8 -> 12.9 us per launch, profiles/r04_ab_qkv_pipe.txt); the unsharded 70B launch (5120 row pairs) measures the same either way
    const bool pipe = pipe_on && nblk > 16 && nblk <= 32 && !q8_store && type_a != GGML_TYPE_Q8_0 && (units[0] + units[1] + 15) / 16 < 128;
    int nw = pipe ? 8 : std::min(16, std::max((nblk + 1) / 2, 8));
    const int nw_cap = pipe ? 8 : 16;
    // smallest workgroup that still gives every wave one unit — counted over both formats together: two formats at 12 waves need 257
    // workgroups for Llama-3-8B (the 256 are then shared by bytes, a few waves take two units), which measures 12.5 us against 13.2 us for
    // 13-wave workgroups that fit (profiles/r03_decode_lab.txt #6)
    while (nw < nw_cap && (units[0] + units[1] + nw - 1) / nw > 256) ++nw;
    if (force_nw) nw = force_nw;
    if (q8_store) nw = 16;  // a workgroup trip = 16 row pairs = one block_q8_0 of the cache row (caller checked the alignment)
    const dim3 block((unsigned) nw * 64);
    const bool has_q5 = type_a == GGML_TYPE_Q5_K || type_b == GGML_TYPE_Q5_K;
    const bool small_wg = has_q5 && !q8_store && !pipe && nw <= 10;  // (the 640-thread build of the Q5_K forms: no scratch)
#define QKV_LAUNCH(TA, TB, GRID)                                                                              \
    do {                                                                                                      \
        if (q8_store) hipLaunchKernelGGL((k_qkv_stream2<TA, TB, true>), GRID, block, lds, s, a);              \
        else if (pipe) hipLaunchKernelGGL((k_qkv_stream2<TA, TB, false, true>), GRID, block, lds, s, a);      \
        else if constexpr (std::is_same<TA, T_Q5K>::value || std::is_same<TB, T_Q5K>::value || std::is_same<TA, T_Q5KP>::value || std::is_same<TB, T_Q5KP>::value || std::is_same<TA, T_Q5KS>::value || std::is_same<TB, T_Q5KS>::value) { \
            if (small_wg) hipLaunchKernelGGL((k_qkv_stream2<TA, TB, false, false, 640>), GRID, block, lds, s, a); \
            else hipLaunchKernelGGL((k_qkv_stream2<TA, TB, false>), GRID, block, lds, s, a);                  \
        } else if constexpr (!std::is_same<TA, TB>::value) {  /* Q4_K + Q6_K: Llama-3-8B's launches use 12 waves (the 768-thread build: 170 registers, no scratch) */ \
            if (nw <= 12) hipLaunchKernelGGL((k_qkv_stream2<TA, TB, false, false, 768>), GRID, block, lds, s, a); \
            else hipLaunchKernelGGL((k_qkv_stream2<TA, TB, false>), GRID, block, lds, s, a);                  \
        } else hipLaunchKernelGGL((k_qkv_stream2<TA, TB, false>), GRID, block, lds, s, a);                    \
    } while (0)
    const bool planes = a.planes != 0 && (a.K % 2048) == 0 && type_a != GGML_TYPE_Q8_0;
    const bool planes_s = a.planes != 0 && !planes && (a.K % 256) == 0 && type_a != GGML_TYPE_Q8_0;  // rows with a short last group: the T_Q*KS forms  // every segment's W is its decode copy (graph.cpp made sure): the plane forms
    if (type_a == type_b || units[1] == 0) {
        const unsigned grid = (unsigned) std::min(256, (units[0] + nw - 1) / nw);
        a.wg_a = (int) grid;
        if (planes && type_a == GGML_TYPE_Q4_K) QKV_LAUNCH(T_Q4KP, T_Q4KP, dim3(grid));
        else if (planes_s && type_a == GGML_TYPE_Q4_K) QKV_LAUNCH(T_Q4KS, T_Q4KS, dim3(grid));
        else if (planes && type_a == GGML_TYPE_Q5_K) QKV_LAUNCH(T_Q5KP, T_Q5KP, dim3(grid));
        else if (planes_s && type_a == GGML_TYPE_Q5_K) QKV_LAUNCH(T_Q5KS, T_Q5KS, dim3(grid));
        else if (planes && type_a == GGML_TYPE_Q6_K) QKV_LAUNCH(T_Q6KP, T_Q6KP, dim3(grid));
        else if (planes_s && type_a == GGML_TYPE_Q6_K) QKV_LAUNCH(T_Q6KS, T_Q6KS, dim3(grid));
        else if (type_a == GGML_TYPE_Q4_K) QKV_LAUNCH(T_Q4K, T_Q4K, dim3(grid));
        else if (type_a == GGML_TYPE_Q5_K) QKV_LAUNCH(T_Q5K, T_Q5K, dim3(grid));
        else if (type_a == GGML_TYPE_Q6_K) QKV_LAUNCH(T_Q6K, T_Q6K, dim3(grid));
        else if (type_a == GGML_TYPE_Q8_0) QKV_LAUNCH(T_Q80, T_Q80, dim3(grid));
        else { MI_ERR("launch_qkv: unsupported weight format %d", type_a); abort(); }
        return;
    }
    static const bool wave_split = !getenv("GGML_MI355X_QKV_WAVE_SPLIT") || atoi(getenv("GGML_MI355X_QKV_WAVE_SPLIT")) != 0;
    if (wave_split && !q8_store && !pipe) {
        // split by waves: the smallest grid of nw-wave workgroups that holds one unit per wave (at most 256), format A's units on the first waves
        const int total_units = units[0] + units[1];
        const int g = std::min(256, (total_units + nw - 1) / nw), waves = g * nw;
        a.wg_a = g;
        a.wv_a = total_units <= waves ? units[0] : std::max(1, std::min(waves - 1, (int) ((double) waves * bytes[0] / (bytes[0] + bytes[1]) + 0.5)));
        const dim3 grid_w((unsigned) g);
        if (planes && type_a == GGML_TYPE_Q4_K && type_b == GGML_TYPE_Q6_K) QKV_LAUNCH(T_Q4KP, T_Q6KP, grid_w);
        else if (planes_s && type_a == GGML_TYPE_Q4_K && type_b == GGML_TYPE_Q6_K) QKV_LAUNCH(T_Q4KS, T_Q6KS, grid_w);
        else if (planes && type_a == GGML_TYPE_Q5_K && type_b == GGML_TYPE_Q6_K) QKV_LAUNCH(T_Q5KP, T_Q6KP, grid_w);
        else if (planes_s && type_a == GGML_TYPE_Q5_K && type_b == GGML_TYPE_Q6_K) QKV_LAUNCH(T_Q5KS, T_Q6KS, grid_w);
        else if (planes && type_a == GGML_TYPE_Q4_K && type_b == GGML_TYPE_Q5_K) QKV_LAUNCH(T_Q4KP, T_Q5KP, grid_w);
        else if (planes_s && type_a == GGML_TYPE_Q4_K && type_b == GGML_TYPE_Q5_K) QKV_LAUNCH(T_Q4KS, T_Q5KS, grid_w);
        else if (type_a == GGML_TYPE_Q4_K && type_b == GGML_TYPE_Q6_K) QKV_LAUNCH(T_Q4K, T_Q6K, grid_w);
        else if (type_a == GGML_TYPE_Q5_K && type_b == GGML_TYPE_Q6_K) QKV_LAUNCH(T_Q5K, T_Q6K, grid_w);
        else if (type_a == GGML_TYPE_Q4_K && type_b == GGML_TYPE_Q5_K) QKV_LAUNCH(T_Q4K, T_Q5K, grid_w);
        else { MI_ERR("launch_qkv: unsupported weight format pair %d/%d", type_a, type_b); abort(); }
        return;
    }
    // one workgroup per CU at most; when the units would need more, share the 256 slots by weight bytes
    int ga = (units[0] + nw - 1) / nw, gb = (units[1] + nw - 1) / nw;
    if (ga + gb > 256) {
        gb = std::max(1, std::min(255, (int) (256.0 * bytes[1] / (bytes[0] + bytes[1]) + 0.5)));
        ga = 256 - gb;
    }
    a.wg_a = ga;
    const dim3 grid((unsigned) (ga + gb));
    if (planes && type_a == GGML_TYPE_Q4_K && type_b == GGML_TYPE_Q6_K) QKV_LAUNCH(T_Q4KP, T_Q6KP, grid);
    else if (planes_s && type_a == GGML_TYPE_Q4_K && type_b == GGML_TYPE_Q6_K) QKV_LAUNCH(T_Q4KS, T_Q6KS, grid);
    else if (planes && type_a == GGML_TYPE_Q5_K && type_b == GGML_TYPE_Q6_K) QKV_LAUNCH(T_Q5KP, T_Q6KP, grid);
    else if (planes_s && type_a == GGML_TYPE_Q5_K && type_b == GGML_TYPE_Q6_K) QKV_LAUNCH(T_Q5KS, T_Q6KS, grid);
    else if (planes && type_a == GGML_TYPE_Q4_K && type_b == GGML_TYPE_Q5_K) QKV_LAUNCH(T_Q4KP, T_Q5KP, grid);
    else if (planes_s && type_a == GGML_TYPE_Q4_K && type_b == GGML_TYPE_Q5_K) QKV_LAUNCH(T_Q4KS, T_Q5KS, grid);
    else if (type_a == GGML_TYPE_Q4_K && type_b == GGML_TYPE_Q6_K) QKV_LAUNCH(T_Q4K, T_Q6K, grid);
    else if (type_a == GGML_TYPE_Q5_K && type_b == GGML_TYPE_Q6_K) QKV_LAUNCH(T_Q5K, T_Q6K, grid);
    else if (type_a == GGML_TYPE_Q4_K && type_b == GGML_TYPE_Q5_K) QKV_LAUNCH(T_Q4K, T_Q5K, grid);
    else { MI_ERR("launch_qkv: unsupported weight format pair %d/%d", type_a, type_b); abort(); }
}

MI_TU_TOUCH(qkv)

}  // namespace mi355x
