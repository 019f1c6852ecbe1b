// fattn.hip — FLASH_ATTN_EXT for f16 K/V with GQA: online-softmax attention, split over the KV range.
//
// Restates ggml_compute_forward_flash_attn_ext_f16: Q is rounded to f16,
// s = (K·Q)*scale (+softcap) + slope*mask, masked (-inf) positions are skipped, running (max, sum) softmax.
// Deliberate difference: the CPU accumulates V in f16 (ggml_vec_mad_f16); this kernel accumulates in f32, so its
// result is closer to exact arithmetic than the CPU's and agrees with it to f16-accumulation noise (tests bound it
// with the NMSE gate upstream's test-backend-ops uses for this op).
//
// Decode-first layout (batch-1 / small-batch attention is bound by the KV bytes, 131 072 B x n_past per token for
// Llama-3-8B): one workgroup per (KV split, KV head, query token) serves ALL G query heads that share the KV head,
// so K/V are read once per group.  A K/V row (D halves) is covered by D/8 lanes with 16-byte loads: a wave64 load
// instruction moves 8 (D=64) or 4 (D=128) whole rows = 1 KiB, fully coalesced.  Scores are finished with a
// sub-wave butterfly; there is no LDS traffic in the KV loop.  Partial (m, l, acc) triples are merged across
// sub-rows (shuffles), waves (LDS) and splits (second tiny kernel).
#include <algorithm>

#include "dev_util.h"
#include "kernels.h"
#include <vector>
#include <algorithm>
#include "kv_dequant.h"

namespace mi355x {

// query heads per KV head the lane-parallel kernel serves: 1 .. 8, in the template form of the next power of two from 2 on (G = 3 runs as 4, G = 5 / 6 / 7 as 8,
// multi-head attention — G = 1 — as 2: the surplus head slots read the group's last head with a zero query and are never stored — Llama-3.2-3B is 24 / 8 heads,
// Qwen2.5-1.5B 12 / 2, Qwen2-7B 28 / 4, Llama-2-7B 32 / 32: 467 -> 523 tok/s against the generic kernel, profiles/r06_other_families.txt)
static inline bool fa_g_ok(const int64_t G) {
    static const int g_min = getenv("GGML_MI355X_FA_G_MIN") ? atoi(getenv("GGML_MI355X_FA_G_MIN")) : 1;  // (2: multi-head attention back on the generic kernel)
    return G >= g_min && G <= 8;
}
static inline int fa_gg(const int64_t G) { return G <= 2 ? 2 : (G <= 4 ? 4 : 8); }

struct fa_geom {
    int n_q, n_head, n_kv_head, n_kv, n_splits, has_mask;
    int rec_stride;  // floats between the partial records of consecutive splits (D + 2; 132 for the records the wo prologue reads)
    float scale, softcap, max_bias, m0, m1;
    uint32_t n_head_log2;
    unsigned * arrive;  // != null (lane-parallel decode kernel, n_splits > 1): one counter per (batch, token, kv head); the LAST split workgroup to
                        // arrive merges the partial records itself — no combine launch (the counter is left at zero again)
    void * q8;          // with `arrive`: leave the merged result as Q8_K blocks here instead of f32 in dst (see fattn_params::q8_out)
    int one_batch;      // q.ne[3] == 1 (every graph llama.cpp builds): the batch index is 0 and the kernels skip its integer divisions — the lane-parallel decode kernel ran
                        // 974 instructions, four emulated 64-bit divisions among them, before it issued its first load (round 6)
    int per;            // cells per split, as the kernels would compute it (launcher: MODE-dependent rounding)
    int merge2;         // with `arrive` (round 4): records [kv head][split][head of the group][132] written with 8-byte agent-scope stores, merged by
                        // the last workgroup to arrive in ONE round trip of 8-byte agent-scope loads (MODE 0, f32 output, <= 32 splits, any wave count)
};

// element-wise all-reduce over the four 16-lane DPP rows of a wave (lane l ends with x[l&15] + x[16+(l&15)] + ...):
// gfx950's v_permlane{16,32}_swap exchange odd/even rows (halves) of two registers, so two swaps and two adds do it
// without touching LDS.  (Inline asm: with identical operands the builtin form folds to x + x in this compiler.)
__device__ __forceinline__ float xrow_allsum(const float x) {
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    float y = a + b;
    a = y;
    b = y;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}

template <int D, int G>
__global__ void __launch_bounds__(256) k_fattn_split(const tdesc q, const tdesc k, const tdesc v, const tdesc mask, const float * __restrict__ sinks,
                                                     const tdesc dst, const fa_geom geo, float * __restrict__ ws) {
    constexpr int LPR = D / 8;     // lanes per K/V row
    constexpr int RPW = 64 / LPR;  // rows per wave-instruction
    constexpr int NG = 4;          // row groups in flight per wave and trip: 4 waves x NG x RPW positions per trip
    __shared__ float sh[4][G][D + 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = lane / LPR, sl = lane % LPR;
    const int split = blockIdx.x, kvh = blockIdx.y;
    const int tok = blockIdx.z % geo.n_q, bat = blockIdx.z / geo.n_q;
    const int per = (geo.n_kv + geo.n_splits - 1) / geo.n_splits;
    const int kv0 = split * per, kv1 = min(geo.n_kv, kv0 + per);
    const int64_t kb = bat / (q.ne[3] / k.ne[3]), vb = bat / (q.ne[3] / v.ne[3]);
    const uint16_t * mp = geo.has_mask ? (const uint16_t *) (mask.data + (int64_t) tok * mask.nb[1] + (int64_t) (bat % mask.ne[3]) * mask.nb[3]) : nullptr;
    const char * kbase = k.data + (int64_t) kvh * k.nb[2] + kb * k.nb[3] + sl * 16;
    const char * vbase = v.data + (int64_t) kvh * v.nb[2] + vb * v.nb[3] + sl * 16;

    // K/V of the first trip are requested before anything else: for decode-sized splits that is the only trip, and
    // the whole kernel is then ONE memory round trip + arithmetic + the cross-wave merge
    uint4 kraw[NG], vraw[NG];
    float mv[NG];
    bool inr[NG];
    int p0 = kv0 + wave * RPW;
#define FA_LOAD_TRIP()                                                        \
    _Pragma("unroll") for (int u = 0; u < NG; ++u) {                         \
        const int p = p0 + u * 4 * RPW + sub;                                 \
        inr[u] = p < kv1;                                                     \
        const int pc = min(p, kv1 - 1);                                       \
        mv[u] = mp ? h2f(mp[pc]) : 0.0f;                                      \
        kraw[u] = *(const uint4 *) (kbase + (int64_t) pc * k.nb[1]);          \
        vraw[u] = *(const uint4 *) (vbase + (int64_t) pc * v.nb[1]);          \
    }
    if (kv0 < kv1) { FA_LOAD_TRIP() }

    float qr[G][8], acc[G][8], m[G], l[G], slope[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int h = kvh * G + g;
        const float4 * qp = (const float4 *) ((const float *) (q.data + (int64_t) tok * q.nb[1] + (int64_t) h * q.nb[2] + (int64_t) bat * q.nb[3]) + sl * 8);
        const float4 qa = qp[0], qb = qp[1];
        const float qv[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            qr[g][i] = h2f(f2h(qv[i]));  // q_to_vec_dot: Q is converted to K's vec_dot_type (f16)
            acc[g][i] = 0.0f;
        }
        m[g] = -INFINITY;
        l[g] = 0.0f;
        slope[g] = geo.max_bias > 0.0f ? ((uint32_t) h < geo.n_head_log2 ? powf(geo.m0, (float) (h + 1)) : powf(geo.m1, (float) (2 * (h - geo.n_head_log2) + 1))) : 1.0f;
    }

    // One softmax state per WAVE (not per row): a trip computes all its scores first, takes the wave-wide maximum on the
    // DPP path, rescales once, and only then exponentiates — one expf per (position, head), no data-dependent branches,
    // and the rows of a wave later merge by plain addition.
    for (; p0 < kv1; p0 += NG * 4 * RPW) {
        float sc[NG][G];
        float vf[NG][8];
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            const uint32_t ku[4] = {kraw[u].x, kraw[u].y, kraw[u].z, kraw[u].w}, vu[4] = {vraw[u].x, vraw[u].y, vraw[u].z, vraw[u].w};
            float kf[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                kf[2 * i] = h2f((uint16_t) (ku[i] & 0xFFFF));
                kf[2 * i + 1] = h2f((uint16_t) (ku[i] >> 16));
                vf[u][2 * i] = h2f((uint16_t) (vu[i] & 0xFFFF));
                vf[u][2 * i + 1] = h2f((uint16_t) (vu[i] >> 16));
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float t = 0.0f;
#pragma unroll
                for (int i = 0; i < 8; ++i) t = fmaf(kf[i], qr[g][i], t);
                // finish the D-wide dot inside the LPR lanes of the row on the DPP path
                if constexpr (LPR == 16) {
                    t = row16_sum(t);
                } else {
                    t += dpp_f32<MI_DPP_QUAD_XOR1>(t);
                    t += dpp_f32<MI_DPP_QUAD_XOR2>(t);
                    t += dpp_f32<MI_DPP_HALF_MIRROR>(t);
                }
                float sv = t * geo.scale;
                if (geo.softcap != 0.0f) sv = geo.softcap * tanhf(sv);
                const float mvs = slope[g] * mv[u];
                sv += mvs;
                sc[u][g] = (inr[u] && !(mvs == -INFINITY)) ? sv : -INFINITY;
            }
        }
        const bool more = p0 + NG * 4 * RPW < kv1;
        if (more) {  // long splits (prefill-sized KV per workgroup): next trip's loads go out before the exponentials
            p0 += NG * 4 * RPW;
            FA_LOAD_TRIP()
            p0 -= NG * 4 * RPW;
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float mx = sc[0][g];
#pragma unroll
            for (int u = 1; u < NG; ++u) mx = fmaxf(mx, sc[u][g]);
            mx = wave_max(mx);
            const float mn = fmaxf(m[g], mx);
            if (mn == -INFINITY) continue;  // wave-uniform: nothing visible yet
            const float alpha = m[g] == -INFINITY ? 0.0f : expf(m[g] - mn);
            float ps = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[g][i] *= alpha;
#pragma unroll
            for (int u = 0; u < NG; ++u) {
                const float pe = sc[u][g] == -INFINITY ? 0.0f : expf(sc[u][g] - mn);
                ps += pe;
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[g][i] = fmaf(pe, vf[u][i], acc[g][i]);
            }
            l[g] = l[g] * alpha + ps;
            m[g] = mn;
        }
    }
#undef FA_LOAD_TRIP
    // rows of the wave share (m); their partial sums add up element-wise across the sub-rows
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float lt = l[g];
        if constexpr (LPR == 8) lt += dpp_f32<MI_DPP_ROR8>(lt);
        lt = xrow_allsum(lt);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float a = acc[g][i];
            if constexpr (LPR == 8) a += dpp_f32<MI_DPP_ROR8>(a);
            acc[g][i] = xrow_allsum(a);
        }
        l[g] = lt;
    }
    if (sub == 0) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            *(float4 *) &sh[wave][g][sl * 8] = make_float4(acc[g][0], acc[g][1], acc[g][2], acc[g][3]);
            *(float4 *) &sh[wave][g][sl * 8 + 4] = make_float4(acc[g][4], acc[g][5], acc[g][6], acc[g][7]);
            if (sl == 0) {
                sh[wave][g][D] = m[g];
                sh[wave][g][D + 1] = l[g];
            }
        }
    }
    __syncthreads();
    // merge the four waves; one thread per (g, d)
    for (int e = tid; e < G * D; e += 256) {
        const int g = e / D, dd = e % D;
        const float m0 = sh[0][g][D], m1 = sh[1][g][D], m2 = sh[2][g][D], m3 = sh[3][g][D];
        const float mt = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
        const float c0 = m0 == -INFINITY ? 0.0f : expf(m0 - mt), c1 = m1 == -INFINITY ? 0.0f : expf(m1 - mt);
        const float c2 = m2 == -INFINITY ? 0.0f : expf(m2 - mt), c3 = m3 == -INFINITY ? 0.0f : expf(m3 - mt);
        float a = ((sh[0][g][dd] * c0 + sh[1][g][dd] * c1) + sh[2][g][dd] * c2) + sh[3][g][dd] * c3;
        float lt = ((sh[0][g][D + 1] * c0 + sh[1][g][D + 1] * c1) + sh[2][g][D + 1] * c2) + sh[3][g][D + 1] * c3;
        const int h = kvh * G + g;
        if (geo.n_splits == 1) {
            if (sinks) {
                const float sk = sinks[h];
                const float mn = fmaxf(mt, sk);
                const float c = mt == -INFINITY ? 0.0f : expf(mt - mn);
                a *= c;
                lt = lt * c + expf(sk - mn);
            }
            float * out = (float *) (dst.data + (int64_t) h * dst.nb[1] + (int64_t) tok * dst.nb[2] + (int64_t) bat * dst.nb[3]);
            out[dd] = a * (1.0f / lt);
        } else {
            float * rec = ws + ((((int64_t) bat * geo.n_q + tok) * geo.n_head + h) * geo.n_splits + split) * (D + 2);
            rec[dd] = a;
            if (dd == 0) {
                rec[D] = mt;
                rec[D + 1] = lt;
            }
        }
    }
}

// ---- decode kernel for head_dim 128 with the softmax done LANE-PARALLEL ------------------------------------------
// The generic kernel above finishes every (position, head) score on all 16 lanes of a row and then runs the same
// exponential 16 times in a row — ~1200 dependent VALU instructions per trip with one wave per SIMD, which is what
// bounded batch-1 attention (12 us per layer for 8.7 MB of KV).  Here a trip owns exactly 16 (row group u, head g)
// pairs per KV row, and a 4-level transpose-reduce leaves the finished score of pair j in lane j of the row: scale,
// mask, running max, ONE v_exp_f32 and the running sum are then done once per lane for all 16 pairs at the same
// time, and P.V takes each probability straight out of its lane with a DPP row broadcast on the fma.  K stays packed
// f16 (v_dot2_f32_f16 against the f16-rounded Q).  Rows of the wave merge with permlane swaps (two values per swap),
// waves through four LDS slots.  G = 2 / 4 / 8 query heads per KV head (7 runs as 8 with a zero head).
#define MI_DPP_NEWBCAST(n) (0x150 + (n))
typedef _Float16 fa_half2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float fa_exp2(const float x) { return __builtin_amdgcn_exp2f(x); }
// (x.lo+x.hi | y.lo+y.hi): lanes 0-31 get the two-half total of x, lanes 32-63 that of y
__device__ __forceinline__ float swap32_pairsum(float x, float y) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return x + y;
}
// rows (0,1,2,3) get (c.r0+c.r1, d.r0+d.r1, c.r2+c.r3, d.r2+d.r3)
__device__ __forceinline__ float swap16_pairsum(float c, float d) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(c), "+v"(d));
    return c + d;
}
__device__ __forceinline__ float xrow_allmax(const float x) {
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    const float y = fmaxf(a, b);
    a = y;
    b = y;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}
// records that another workgroup will read in the SAME launch: agent-scope (sc1, write-through / L1-bypassing) accesses on both sides
// instead of release / acquire fences — a fence per split workgroup writes the whole L2 back and made the launch 3-5x slower
__device__ __forceinline__ void st_agent(float * p, const float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float * p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// two floats as ONE naturally aligned 8-byte agent-scope access (global_store_dwordx2 / global_load_dwordx2 ... sc1): a 4-byte sc1 store is
// one fabric write each and costs ~6x the time per byte of a wide one (MI355X_MICROARCH.md, stores of each flavour)
__device__ __forceinline__ void st_agent2(float * p, const float a, const float b) {
    const unsigned long long v = (unsigned long long) __builtin_bit_cast(unsigned, a) | ((unsigned long long) __builtin_bit_cast(unsigned, b) << 32);
    __hip_atomic_store((unsigned long long *) p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float2 ld_agent2(const float * p) {
    const unsigned long long v = __hip_atomic_load((const unsigned long long *) p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__builtin_bit_cast(float, (unsigned) v), __builtin_bit_cast(float, (unsigned) (v >> 32)));
}
#define FA_M2_REC 132  // floats per (split, head) record of the merge2 layout: 128 values, max (natural-log domain), sum, 2 pad

// merge2 (round 4): the LAST workgroup of a (token, kv head) to arrive merges the n_splits x g_real records.  Two waves per head and trip
// (a thread per output dim, as k_fattn_combine); lane s of every wave owns split s for the (max, sum) pairs and ALL loads of a trip — the
// pair and up to 32 values per thread — go out together: one memory round trip per trip of WV / 2 heads, where round 2's merge walked the
// records in four dependent trips (16.5 us against 12.6 us for split + combine launches).  Kept SMALL on purpose: this code runs once per
// (token, kv head) on a cold instruction cache — the first version of this merge, 32 records x 3 share counts fully unrolled (~24 KB of
// straight-line code), cost 6 us more than the combine launch it replaced.  n_splits <= 32.
__device__ __forceinline__ void fa_merge2(const float * __restrict__ hb, const int h0, const int g_real, const int n_splits, const float * __restrict__ sinks, const tdesc & dst,
                                          const int tok, const int bat, const int n_threads) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t ss = (int64_t) g_real * FA_M2_REC;  // floats between the records of consecutive splits
#pragma unroll 1
    for (int g = tid >> 7; g < g_real; g += n_threads >> 7) {
        const int dd = tid & 127;
        const float * __restrict__ base = hb + (int64_t) g * FA_M2_REC;
        const float2 ml = ld_agent2(base + (int64_t) min(lane, n_splits - 1) * ss + 128);
        float r[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) r[u] = ld_agent(base + (int64_t) min(u, n_splits - 1) * ss + dd);
        const float ms = lane < n_splits ? ml.x : -INFINITY, ls = lane < n_splits ? ml.y : 0.0f;
        float mn = wave_max(ms);
        float sink_term = 0.0f;
        if (sinks) {
            mn = fmaxf(mn, sinks[h0 + g]);
            sink_term = expf(sinks[h0 + g] - mn);
        }
        const float cs = ms == -INFINITY ? 0.0f : expf(ms - mn);
        const float lt = wave_sum(ls * cs) + sink_term;
        // (a split that left an empty record — coefficient 0 — may hold anything in its values: select, not multiply)
        float a = 0.0f;
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const float c = readlane_f32(cs, u);
            a += c != 0.0f ? r[u] * c : 0.0f;
        }
        float * out = (float *) (dst.data + (int64_t) (h0 + g) * dst.nb[1] + (int64_t) tok * dst.nb[2] + (int64_t) bat * dst.nb[3]);
        out[dd] = a * (1.0f / lt);
    }
}

// Merge of the partial records of `g_real` consecutive heads (head_dim 128) by ONE 4-wave workgroup — what k_fattn_combine does
// per head in its own launch: wave w takes heads w, w + 4, ..; lane s owns split s for the (max, sum) pairs, every lane two of
// the 128 values; coefficients come out of their lanes through v_readlane.  q8 != null: the result leaves as Q8_K blocks (two
// heads each) through `vals` (LDS, g_real * 128 floats), else as f32 rows of dst.
__device__ __forceinline__ void fa_merge_heads(const float * __restrict__ base0, const int h0, const int g_real, const int n_splits, const float * __restrict__ sinks, const tdesc & dst,
                                               const int tok, const int bat, q8k_dev * __restrict__ q8, float * __restrict__ vals) {
    constexpr int D = 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (q8) __syncthreads();  // `vals` aliases the wave-merge buffer: everyone is done reading it
    for (int g = wave; g < g_real; g += 4) {
        const float * __restrict__ base = base0 + (int64_t) g * n_splits * (D + 2);
        const bool has = lane < n_splits;
        const float ms = has ? ld_agent(base + (int64_t) lane * (D + 2) + D) : -INFINITY;
        const float ls = has ? ld_agent(base + (int64_t) lane * (D + 2) + D + 1) : 0.0f;
        float mn = wave_max(ms);
        float sink_term = 0.0f;
        if (sinks) {
            mn = fmaxf(mn, sinks[h0 + g]);
            sink_term = expf(sinks[h0 + g] - mn);
        }
        const float cs = ms == -INFINITY ? 0.0f : expf(ms - mn);
        const float lt = wave_sum(ls * cs) + sink_term;
        float a0 = 0.0f, a1 = 0.0f;
        // (a split that left an empty record — coefficient 0 — may hold anything in its values: select, not multiply)
        for (int u0 = 0; u0 < n_splits; u0 += 8) {  // eight records in flight per lane
            float r0[8], r1[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int u = min(u0 + k, n_splits - 1);
                r0[k] = ld_agent(base + (int64_t) u * (D + 2) + lane);
                r1[k] = ld_agent(base + (int64_t) u * (D + 2) + 64 + lane);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float c = u0 + k < n_splits ? readlane_f32(cs, min(u0 + k, 63)) : 0.0f;
                a0 += c != 0.0f ? r0[k] * c : 0.0f;
                a1 += c != 0.0f ? r1[k] * c : 0.0f;
            }
        }
        const float inv = 1.0f / lt;
        if (q8) {
            vals[g * D + lane] = a0 * inv;
            vals[g * D + 64 + lane] = a1 * inv;
        } else {
            float * out = (float *) (dst.data + (int64_t) (h0 + g) * dst.nb[1] + (int64_t) tok * dst.nb[2] + (int64_t) bat * dst.nb[3]);
            out[lane] = a0 * inv;
            out[64 + lane] = a1 * inv;
        }
    }
    if (q8) {
        __syncthreads();
        if (wave < g_real / 2) {  // quantize_row_q8_K of the block (two heads), as k_quantize_q8_K does it
            const float4 t4 = ((const float4 *) (vals + wave * 256))[lane];
            const float t[4] = {t4.x, t4.y, t4.z, t4.w};
            wave_quantize_q8_K(t, lane, q8 + wave);
        }
    }
}

// LAB (scripts/lab/fa_stamps.py, -DFA_STAMP=1): where the time of one decode-attention workgroup goes — wave 0 / lane 0 of every workgroup of the 8-wave
// f16 form leaves s_memtime stamps at seven points; mi355x_fa_stamps_dump() prints their medians over the workgroups of the last launch.
#ifdef FA_STAMP
__device__ unsigned long long g_fa_stamps[1024][8];
#define FA_STAMP_AT(i) { if (WV == 8 && MODE == 0 && threadIdx.x == 0) g_fa_stamps[(blockIdx.y * gridDim.x + blockIdx.x) & 1023][(i)] = __builtin_readcyclecounter(); }
#else
#define FA_STAMP_AT(i)
#endif
// SKIP = true (several query tokens, e.g. -np 32 decode over a unified cache where each token sees ~1/32 of the cells): the mask
// of the whole split is read first — one round trip — and trips without a visible position for this wave load no K/V at all.
// Q8 = true: K and V rows are block_q8_0 (quantised KV cache, -ctk/-ctv q8_0).  As in ggml-cpu the query is quantised to Q8_0
// too (K's vec_dot_type), a score is sum over the four 32-value blocks of sumi * (d_k * d_q) with an integer block sum, and a
// V row is de-quantised (q * d) and accumulated in f32.  A lane still owns 8 dims: 8 int8 of one block (lanes 4b..4b+3 = block b).
// MODE 2 (LIST): the visible POSITIONS of every query token were listed once per graph by k_fattn_pos_scan (the mask is the same
// tensor in every layer); split s of token t walks its share of THAT list, a trip = the next NG*16 listed positions, so the work
// is proportional to what the token can see, not to the size of the unified cache nor to how its cells are scattered in it (the
// decode cells of `-np 32` sequences interleave: every 64-cell tile of that region holds two cells of each sequence, and a list
// of visible TILES — round 1 — made every token walk all of them: 10 us at the first step, 24 us sixty steps later).
// WV = waves per workgroup: 4 (many thin splits + combine pass) or 8 (few fat splits whose partials the wo mat-vec prologue combines:
// mmvq.hip PRO 3; 16 waves would cap the kernel at 128 VGPRs, and it needs ~200: the first 16-wave build spilled and ran 2x slower)
// KVT != 0 (round 5): K and V live in ANOTHER cache type, both the same one of q4_0, q4_1, q5_0, q5_1, iq4_nl (KVT = its ggml type): a lane fetches the raw bytes of
// its eight dims of the row's block (kv_dequant.h: 8 nibble bytes + scale [+ minimum, fifth bits] — four dwords that stay in registers while the
// loads fly) and expands them to packed f16 where the f16 cache's values are consumed (byte permutes + one packed subtract + one packed multiply);
// everything else is the f16 path.  Same values as the f16 image of kv_types.hip (q4_0 / q5_0: bit for bit; the offset formats within half an ulp),
// without its pass over the cache.  A first form with the type as a RUN-TIME switch (any pair of types) lost to the image: 29 us against 12.8 + 9.2.
// D = 64 (round 6: TinyLlama, Llama-3.2-1B ...; f16 cache, no Q8_K output, records + combine pass): a K / V row is EIGHT lanes, a wave-instruction fetches eight
// rows, the (row group, head) pairs of a row are 8 = NG * G; everything else is the head_dim-128 kernel — its 16-lane transpose-reduce minus the first stage,
// probabilities broadcast per 8-lane half of a DPP row, the two halves' accumulators added before the rows meet.
template <int G, int MODE, bool Q8, int WV = 4, int KVT = 0, int D = 128>
__global__ void __launch_bounds__(WV * 64) k_fattn_dec128(const tdesc q, const tdesc k, const tdesc v, const tdesc mask, const float * __restrict__ sinks,
                                                      const tdesc dst, const fa_geom geo, float * __restrict__ ws, const int g_real,
                                                      const int * __restrict__ lists, const int list_stride) {
    constexpr bool SKIP = MODE == 1, LIST = MODE == 2;
    constexpr int LPR = D / 8, RPW = 64 / LPR, NG = LPR / G;  // lanes per K / V row, rows per wave-instruction, row groups per trip
    static_assert(D == 128 || (D == 64 && !Q8 && KVT == 0 && G <= 8), "head_dim 64: f16 cache only");
    constexpr float LOG2E = 1.4426950408889634f;
    constexpr int TRIP = NG * WV * RPW;  // positions per trip
    // KVT == BF16 (round 6): a bf16 cache has the f16 cache's geometry (a lane's 16 bytes are its eight values), so only the arithmetic differs: as ggml-cpu's
    // ggml_vec_dot_bf16 the query is rounded to bf16 and multiplied with K's values in f32 (a value is its 16 bits shifted up), V accumulates in f32 (to_float)
    constexpr bool BF = KVT == GGML_TYPE_BF16;
    constexpr bool DQ = KVT != 0 && !BF;
    static_assert(WV == 4 || (!Q8 && !DQ && !BF), "the eight-wave forms serve an f16 cache only");
    static_assert(!(Q8 && DQ), "q8_0 has its own integer path");
    // FAT: one decode token, a few fat splits whose records the wo mat-vec's prologue merges (records also for ONE split, no Q8_K output)
    constexpr bool FAT = WV == 8 && MODE == 0;
    FA_STAMP_AT(0)
    __shared__ float sh[WV][G][D + 2];
    __shared__ float qv[FAT ? 1 : G * D];  // one pass, Q8_K output: the normalised heads of this kv group before quantisation
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = lane / LPR, sl = lane & (LPR - 1);
    const int ul = sl / G, gl = sl % G;  // the (row group, head) pair this lane owns in the lane-parallel part
    const int split = blockIdx.x, kvh = blockIdx.y;
    // SKIP (a few query tokens, each seeing its own part of a unified cache): the workgroup first asks the mask whether its
    // token can see anything in its split — one 8-byte load per lane covers 256 positions — and leaves an empty record
    // if not: of the 64 x 32 (split, token) pairs of a -np 32 decode step only ~1/16 touch K/V at all
    // (no batch index, no division: the launcher issues one launch per q.ne[3] slice with the descriptors, the records, the counters and the Q8_K area moved
    // to that slice, and passes the cells per split — the preamble used to fetch its kernel arguments in four dependent rounds around emulated divisions,
    // ~1.9 us between a workgroup's entry and its first K / V request: profiles/r06_fa_stamps.txt)
    const int tok = (int) blockIdx.z;
    constexpr int bat = 0;
    const int per = geo.per;
    const int kv0 = LIST ? 0 : min(split * per, geo.n_kv);
    int kv1 = LIST ? 0 : min(geo.n_kv, kv0 + per);  // LIST: positions are list ENTRIES, [0, cnt)
    const int * tl = nullptr;  // LIST: this token's visible positions, ascending; trips [ti, ti1) of TRIP entries each are ours
    int ti = 0, ti1 = 0, cnt = 0;
    bool empty = false;
    // LIST: the list entries of the NEXT trip to be loaded (one per row group u), requested a round trip before its K/V rows are — for one
    // split together with the count itself, so that neither the count nor the entries sit in the chain count -> entries -> rows that
    // every trip used to walk (entries past the count are stale: they are replaced by the first entry, a visible cell, before use)
    int nidx[NG];
    int first = 0;
#define FA_LIST_AHEAD(base)                                                                                         \
    _Pragma("unroll") for (int u = 0; u < NG; ++u) nidx[u] = tl[min((base) + u * (WV * RPW) + wave * RPW + sub, geo.n_kv - 1)];
    if constexpr (LIST) {
        const int * lt = lists + (int64_t) tok * list_stride;
        tl = lt + 1;
        cnt = lt[0];
        first = lt[1];
        if (geo.n_splits == 1) FA_LIST_AHEAD(0)
        const int trips = (cnt + TRIP - 1) / TRIP, share = (trips + geo.n_splits - 1) / geo.n_splits;
        ti = split * share;
        ti1 = min(trips, ti + share);
        if (ti >= ti1) {
            if (geo.n_splits == 1) {
                // a token that sees nothing and no combine pass to say so: the CPU's result for such a row is 0 * (1 / 0) = NaN
                for (int e = tid; e < g_real * D; e += WV * 64) {
                    float * out = (float *) (dst.data + (int64_t) (kvh * g_real + e / D) * dst.nb[1] + (int64_t) tok * dst.nb[2] + (int64_t) bat * dst.nb[3]);
                    out[e % D] = __builtin_nanf("");
                }
            } else if (tid < g_real) {
                float * rec = ws + ((((int64_t) bat * geo.n_q + tok) * geo.n_head + kvh * g_real + tid) * geo.n_splits + split) * (D + 2);
                if (geo.arrive) { st_agent(rec + D, -INFINITY); st_agent(rec + D + 1, 0.0f); }
                else { rec[D] = -INFINITY; rec[D + 1] = 0.0f; }
            }
            if (geo.n_splits == 1 || !geo.arrive) return;
            empty = true;  // (still has to arrive: it may be the workgroup that merges the records)
        }
        kv1 = cnt;
        if (geo.n_splits != 1 && !empty) FA_LIST_AHEAD(ti * TRIP)
    }
    constexpr int64_t kb = 0, vb = 0, mb = 0;
    if constexpr (SKIP) {
        const uint16_t * mrow = (const uint16_t *) (mask.data + (int64_t) tok * mask.nb[1] + mb * mask.nb[3]);
        bool any = false;
        for (int c0 = kv0; c0 < kv1; c0 += 256) {
            const int pp = c0 + 4 * lane;
            if (pp < kv1) {  // kv0 and n_kv are multiples of 4 (checked by the launcher): the 4 positions are in range together
                const uint2 w = *(const uint2 *) (mrow + pp);
                any = any || w.x != 0xFC00FC00u || w.y != 0xFC00FC00u;
            }
        }
        if (!__any(any)) {  // every wave reaches the same verdict: no barrier has been passed yet
            if (tid < g_real) {
                float * rec = ws + ((((int64_t) bat * geo.n_q + tok) * geo.n_head + kvh * g_real + tid) * geo.n_splits + split) * (D + 2);
                if (geo.arrive) { st_agent(rec + D, -INFINITY); st_agent(rec + D + 1, 0.0f); }
                else { rec[D] = -INFINITY; rec[D + 1] = 0.0f; }
            }
            if (!geo.arrive) return;
            empty = true;
        }
    }
    if (!empty) {
    const uint16_t * mp = geo.has_mask ? (const uint16_t *) (mask.data + (int64_t) tok * mask.nb[1] + mb * mask.nb[3]) : nullptr;
    const char * kbase = k.data + (int64_t) kvh * k.nb[2] + kb * k.nb[3] + (DQ ? (sl >> 2) * kv_block_bytes_t<KVT>() : Q8 ? (sl >> 2) * 34 + 2 + (sl & 3) * 8 : sl * 16);
    const char * vbase = v.data + (int64_t) kvh * v.nb[2] + vb * v.nb[3] + (DQ ? (sl >> 2) * kv_block_bytes_t<KVT>() : Q8 ? (sl >> 2) * 34 + 2 + (sl & 3) * 8 : sl * 16);

    // a trip covers NG*16 consecutive positions: position of (u, wave, sub) = p0 + u*16 + wave*4 + sub
    uint4 kraw[NG], vraw[NG];
    float mvl = 0.0f;   // mask value / validity of THIS lane's pair (ul, sub)
    bool okl = false;
    int p0 = LIST ? ti * TRIP : kv0;
#define FA_LOAD_TRIP()                                                                  \
    {                                                                                   \
        _Pragma("unroll") for (int u = 0; u < NG; ++u) {                               \
            const int pr_ = p0 + u * (WV * RPW) + wave * RPW + sub;                     \
            const int pc = LIST ? (pr_ < kv1 ? nidx[u] : first) : min(pr_, kv1 - 1);    \
            const char * kp_ = kbase + (int64_t) pc * k.nb[1];                          \
            const char * vp_ = vbase + (int64_t) pc * v.nb[1];                          \
            if constexpr (DQ) {  /* the raw octet of this lane's block, whatever the type */ \
                kraw[u] = kv_load_octet_raw_t<KVT>(kp_, sl & 3);                        \
                vraw[u] = kv_load_octet_raw_t<KVT>(vp_, sl & 3);                        \
            } else if constexpr (Q8) {  /* 8 quants (2-byte aligned) + the block's f16 scale */  \
                kraw[u] = make_uint4(ld32_a2(kp_), ld32_a2(kp_ + 4), (uint32_t) ld16(kp_ - 2 - (sl & 3) * 8), 0u);  \
                vraw[u] = make_uint4(ld32_a2(vp_), ld32_a2(vp_ + 4), (uint32_t) ld16(vp_ - 2 - (sl & 3) * 8), 0u);  \
            } else {                                                                    \
                kraw[u] = *(const uint4 *) kp_;                                         \
                vraw[u] = *(const uint4 *) vp_;                                         \
            }                                                                           \
        }                                                                               \
        const int pl = p0 + ul * (WV * RPW) + wave * RPW + sub;                         \
        okl = pl < kv1;                                                                 \
        int plc_ = min(pl, kv1 - 1);                                                    \
        if constexpr (LIST) {                                                           \
            plc_ = nidx[0];                                                             \
            _Pragma("unroll") for (int u = 1; u < NG; ++u) plc_ = ul == u ? nidx[u] : plc_; \
            plc_ = okl ? plc_ : first;                                                  \
        }                                                                               \
        mvl = mp ? h2f(mp[plc_]) : 0.0f;                                                \
        if constexpr (LIST) FA_LIST_AHEAD(p0 + TRIP)                                    \
    }
    uint32_t vis = 0xFFFFFFFFu;  // bit i: trip i has a position this wave can see
    if constexpr (SKIP) {
        vis = 0;
        const int ntrips = (kv1 - kv0 + TRIP - 1) / TRIP;  // <= 32 (launcher bounds the split length)
        for (int i0 = 0; i0 < ntrips; i0 += 8) {
            uint16_t mraw[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int pl = kv0 + (i0 + i) * TRIP + ul * (WV * RPW) + wave * RPW + sub;
                mraw[i] = (i0 + i < ntrips && pl < kv1) ? (mp ? mp[pl] : (uint16_t) 0) : (uint16_t) 0xFC00;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (__any(mraw[i] != 0xFC00)) vis |= 1u << (i0 + i);
        }
        if (vis & 1u) FA_LOAD_TRIP()
    } else {
        if (kv0 < kv1) FA_LOAD_TRIP()
    }
    FA_STAMP_AT(1)

    fa_half2 qh[G][4];    // f16 cache: the query as packed f16
    int qq[G][2];         // q8_0 cache: the query's 8 int8 of this lane's block ...
    float dq[G];          // ... and the block scale (through f16, as block_q8_0.d)
    int qsb[G];           // the other block formats: the sum of the block's 32 quants ...
    float sq[G];          // ... and block_q8_1.s (q4_1 / q5_1: multiplies K's minimum)
    float qf[BF ? G : 1][8];  // bf16 cache: the query rounded to bf16, as f32
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int h = kvh * g_real + min(g, g_real - 1);
        const float4 * qp = (const float4 *) ((const float *) (q.data + (int64_t) tok * q.nb[1] + (int64_t) h * q.nb[2] + (int64_t) bat * q.nb[3]) + sl * 8);
        const float4 qa = qp[0], qb = qp[1];
        const float z = g < g_real ? 1.0f : 0.0f;
        if constexpr (BF) {
            const float xv[8] = {qa.x * z, qa.y * z, qa.z * z, qa.w * z, qb.x * z, qb.y * z, qb.z * z, qb.w * z};
#pragma unroll
            for (int i = 0; i < 8; ++i) {  // ggml_compute_fp32_to_bf16: nearest even (a NaN query does not occur)
                const uint32_t u = __float_as_uint(xv[i]);
                qf[g][i] = __uint_as_float((u + (0x7fffu + ((u >> 16) & 1u))) & 0xffff0000u);
            }
        } else if constexpr (Q8 || DQ) {  // quantize_row_q8_0 / _q8_1 of the query (the vec_dot_type of K's format): a block = the 4 lanes of a quad
            const float xv[8] = {qa.x * z, qa.y * z, qa.z * z, qa.w * z, qb.x * z, qb.y * z, qb.z * z, qb.w * z};
            float amax = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(xv[i]));
            amax = fmaxf(amax, dpp_f32<MI_DPP_QUAD_XOR1>(amax));
            amax = fmaxf(amax, dpp_f32<MI_DPP_QUAD_XOR2>(amax));
            const float d = amax / 127.0f;
            const float id = d != 0.0f ? 1.0f / d : 0.0f;
            uint32_t w0 = 0, w1 = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                w0 |= (uint32_t) ((int) roundf(xv[i] * id) & 0xFF) << (8 * i);
                w1 |= (uint32_t) ((int) roundf(xv[4 + i] * id) & 0xFF) << (8 * i);
            }
            qq[g][0] = (int) w0;
            qq[g][1] = (int) w1;
            dq[g] = h2f(f2h(d));
            if constexpr (DQ) {  // the block's sum of quants: the "- 8" / "- 16" of q4_0 / q5_0 on the integer sum; block_q8_1.s = f16(sum * d) for q4_1 / q5_1
                int qs_ = dot4((int) w0, 0x01010101, dot4((int) w1, 0x01010101, 0));
                qs_ += dpp_i32<MI_DPP_QUAD_XOR1>(qs_);
                qs_ += dpp_i32<MI_DPP_QUAD_XOR2>(qs_);
                qsb[g] = qs_;
                sq[g] = h2f(f2h((float) qs_ * d));
            }
        } else {
            qh[g][0] = (fa_half2){(_Float16) (qa.x * z), (_Float16) (qa.y * z)};  // q_to_vec_dot: Q -> f16 (round to nearest even)
            qh[g][1] = (fa_half2){(_Float16) (qa.z * z), (_Float16) (qa.w * z)};
            qh[g][2] = (fa_half2){(_Float16) (qb.x * z), (_Float16) (qb.y * z)};
            qh[g][3] = (fa_half2){(_Float16) (qb.z * z), (_Float16) (qb.w * z)};
        }
    }
#ifdef FA_STAMP
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // (lab: the query's loads are the youngest NG * 2 ... the K / V trip is older: roughly "q has arrived")
    FA_STAMP_AT(2)
#endif
    float acc[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[g][i] = 0.0f;
    float m = -INFINITY, l = 0.0f;  // running max / sum of head gl (max is wave-uniform per head, the sum a per-lane partial)
    const bool b3 = (sl & 8) != 0, b2 = (sl & 4) != 0, b1 = (sl & 2) != 0, b0 = (sl & 1) != 0;
    const float sc2 = geo.scale * LOG2E;  // scores are kept in the log2 domain: p = 2^(s*scale*log2e + mask*log2e - m)

    for (int trip = 0; LIST ? ti < ti1 : p0 < kv1; ++trip) {
        // position of the next trip (LIST: next entry of the tile list; otherwise the next NG*16 positions of the split)
        int p_next = p0 + TRIP;
        if constexpr (LIST) {
            ++ti;
            p_next = ti < ti1 ? ti * TRIP : kv1;
        }
        if constexpr (SKIP) {
            if (!((vis >> trip) & 1u)) {  // nothing visible: only keep the pipeline primed for the next trip
                if (p_next < kv1 && ((vis >> (trip + 1)) & 1u)) {
                    p0 = p_next;
                    FA_LOAD_TRIP()
                }
                p0 = p_next;
                continue;
            }
        }
        // ---- partial dots of this lane's 8 dims for the 16 (u, g) pairs
        float t[LPR];
        float vf[NG][8];
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            uint32_t ku[4] = {kraw[u].x, kraw[u].y, kraw[u].z, kraw[u].w}, vu[4] = {vraw[u].x, vraw[u].y, vraw[u].z, vraw[u].w};
            if constexpr (DQ) {
                // ggml-cpu's arithmetic for a block-format cache (round 6): integer block dots of K's levels with the Q8_0 / Q8_1 query, one f32 term per
                // block in the reference's operation order (ggml_vec_dot_{q4_0,q5_0,iq4_nl}_q8_0, _{q4_1,q5_1}_q8_1); V de-quantised to f32 (to_float)
                constexpr bool OFFSET = KVT == GGML_TYPE_Q4_1 || KVT == GGML_TYPE_Q5_1;
                uint32_t k0, k1;
                kv_octet_levels_t<KVT, true>(kraw[u], sl & 3, k0, k1);
                const float dk = h2f((uint16_t) (ku[2] & 0xFFFFu));
                const float mk = OFFSET ? h2f((uint16_t) (ku[2] >> 16)) : 0.0f;
                kv_octet_f32_t<KVT>(vraw[u], sl & 3, vf[u]);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    int isum = dot4((int) k0, qq[g][0], dot4((int) k1, qq[g][1], 0));
                    isum += dpp_i32<MI_DPP_QUAD_XOR1>(isum);
                    isum += dpp_i32<MI_DPP_QUAD_XOR2>(isum);
                    float term;
                    if constexpr (KVT == GGML_TYPE_Q4_0) term = ((float) (isum - 8 * qsb[g]) * dk) * dq[g];
                    else if constexpr (KVT == GGML_TYPE_Q5_0) term = (dk * dq[g]) * (float) (isum - 16 * qsb[g]);
                    else if constexpr (KVT == GGML_TYPE_IQ4_NL) term = (dq[g] * dk) * (float) isum;
                    else term = (dk * dq[g]) * (float) isum + mk * sq[g];
                    t[u * G + g] = (sl & 3) == 0 ? term : 0.0f;
                }
            } else if constexpr (Q8) {
                const float dv = h2f((uint16_t) vu[2]), dk = h2f((uint16_t) ku[2]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {  // dequantize_row_q8_0: q * d
                    vf[u][i] = (float) (int) (int8_t) (vu[0] >> (8 * i)) * dv;
                    vf[u][4 + i] = (float) (int) (int8_t) (vu[1] >> (8 * i)) * dv;
                }
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    int isum = dot4((int) ku[0], qq[g][0], dot4((int) ku[1], qq[g][1], 0));
                    isum += dpp_i32<MI_DPP_QUAD_XOR1>(isum);
                    isum += dpp_i32<MI_DPP_QUAD_XOR2>(isum);  // the block's integer sum, in all 4 lanes of the quad
                    t[u * G + g] = (sl & 3) == 0 ? (float) isum * (dk * dq[g]) : 0.0f;  // one lane per block feeds the 16-lane sum below
                }
            } else if constexpr (BF) {
                float kf_[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    vf[u][2 * i] = __uint_as_float(vu[i] << 16);
                    vf[u][2 * i + 1] = __uint_as_float(vu[i] & 0xffff0000u);
                    kf_[2 * i] = __uint_as_float(ku[i] << 16);
                    kf_[2 * i + 1] = __uint_as_float(ku[i] & 0xffff0000u);
                }
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    float d = 0.0f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) d = __builtin_fmaf(kf_[i], qf[g][i], d);
                    t[u * G + g] = d;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    vf[u][2 * i] = h2f((uint16_t) (vu[i] & 0xFFFF));
                    vf[u][2 * i + 1] = h2f((uint16_t) (vu[i] >> 16));
                }
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    float d = 0.0f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) d = __builtin_amdgcn_fdot2(__builtin_bit_cast(fa_half2, ku[i]), qh[g][i], d, false);
                    t[u * G + g] = d;
                }
            }
        }
#ifdef FA_STAMP
        if (trip == 0) { asm volatile("" :: "v"(t[0]), "v"(t[15])); FA_STAMP_AT(3) }  // the first trip's K has arrived and its dot products are done
#endif
        const bool more = p_next < kv1 && (!SKIP || ((vis >> (trip + 1)) & 1u));
        const float mv_cur = mvl;
        const bool ok_cur = okl;
        p0 = p_next;  // (nothing below uses the current trip's position)
        if (more) FA_LOAD_TRIP()  // long splits: the next trip's loads go out before the reductions
        // ---- transpose-reduce over the 16 lanes of the row: lane j ends with the complete dot of pair j
        float w8[8], w4[4], w2[2];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if constexpr (LPR == 16) {
                const float keep = b3 ? t[j + 8] : t[j], send = b3 ? t[j] : t[j + 8];
                w8[j] = keep + dpp_f32<MI_DPP_ROR8>(send);
            } else {
                w8[j] = t[j];  // (a row is eight lanes: the reduce starts at its second stage)
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float keep = b2 ? w8[j + 4] : w8[j], send = b2 ? w8[j] : w8[j + 4];
            w4[j] = keep + dpp_f32<MI_DPP_HALF_MIRROR>(send);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float keep = b1 ? w4[j + 2] : w4[j], send = b1 ? w4[j] : w4[j + 2];
            w2[j] = keep + dpp_f32<MI_DPP_QUAD_XOR2>(send);
        }
        float sv;
        {
            const float keep = b0 ? w2[1] : w2[0], send = b0 ? w2[0] : w2[1];
            sv = keep + dpp_f32<MI_DPP_QUAD_XOR1>(send);
        }
        // ---- softmax bookkeeping, once per lane for its own pair
        sv = (ok_cur && !(mv_cur == -INFINITY)) ? fmaf(sv, sc2, mv_cur * LOG2E) : -INFINITY;
        float mx = sv;
        if constexpr (G <= 2) mx = fmaxf(mx, dpp_f32<MI_DPP_ROR2>(mx));
        if constexpr (G <= 4) mx = fmaxf(mx, dpp_f32<MI_DPP_ROR4>(mx));
        mx = fmaxf(mx, dpp_f32<MI_DPP_ROR8>(mx));
        mx = xrow_allmax(mx);
        const float mn = fmaxf(m, mx);
        const float mref = mn == -INFINITY ? 0.0f : mn;  // nothing visible yet: every exponent below is -inf -> 0
        const float alpha = fa_exp2(m - mref);
        const float pe = fa_exp2(sv - mref);
        l = l * alpha + pe;
        m = mn;
        // ---- rescale and accumulate: probabilities / factors come out of their lanes by DPP row broadcast
#define FA_HEAD(g)                                                                          \
    if constexpr ((g) < G) {                                                                \
        const float ag_ = dpp_f32<MI_DPP_NEWBCAST((g))>(alpha);                             \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) acc[(g)][i] *= ag_;                  \
    }
        FA_HEAD(0) FA_HEAD(1) FA_HEAD(2) FA_HEAD(3) FA_HEAD(4) FA_HEAD(5) FA_HEAD(6) FA_HEAD(7)
#undef FA_HEAD
#define FA_PAIR(j)                                                                          \
    {                                                                                       \
        const float pj_ = dpp_f32<MI_DPP_NEWBCAST((j))>(pe);                                \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) acc[(j) % G][i] = fmaf(pj_, vf[(j) / G][i], acc[(j) % G][i]); \
    }
        // (eight-lane rows: a DPP row of 16 lanes holds two K / V rows, pair j's probability sits in lane j of each HALF)
#define FA_PAIR8(j)                                                                         \
    {                                                                                       \
        const float pa_ = dpp_f32<MI_DPP_NEWBCAST((j))>(pe), pb_ = dpp_f32<MI_DPP_NEWBCAST((j) + 8)>(pe); \
        const float pj_ = (lane & 8) ? pb_ : pa_;                                           \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) acc[(j) % G][i] = fmaf(pj_, vf[(j) / G][i], acc[(j) % G][i]); \
    }
        if constexpr (LPR == 16) {
            FA_PAIR(0) FA_PAIR(1) FA_PAIR(2) FA_PAIR(3) FA_PAIR(4) FA_PAIR(5) FA_PAIR(6) FA_PAIR(7)
            FA_PAIR(8) FA_PAIR(9) FA_PAIR(10) FA_PAIR(11) FA_PAIR(12) FA_PAIR(13) FA_PAIR(14) FA_PAIR(15)
        } else {
            FA_PAIR8(0) FA_PAIR8(1) FA_PAIR8(2) FA_PAIR8(3) FA_PAIR8(4) FA_PAIR8(5) FA_PAIR8(6) FA_PAIR8(7)
        }
#undef FA_PAIR8
#undef FA_PAIR
    }
#undef FA_LOAD_TRIP
#undef FA_LIST_AHEAD
#ifdef FA_STAMP
    asm volatile("" :: "v"(acc[0][0]), "v"(acc[G - 1][7]));
    FA_STAMP_AT(4)
#endif
    // ---- sum of head gl over the row groups (lanes gl, gl+G, ...) and the four rows
    if constexpr (G <= 2) l += dpp_f32<MI_DPP_ROR2>(l);
    if constexpr (G <= 4) l += dpp_f32<MI_DPP_ROR4>(l);
    l += dpp_f32<MI_DPP_ROR8>(l);
    l = xrow_allsum(l);
    // ---- accumulators: the four rows add up; pair-wise swaps reduce four values per register, row r of register kk
    //      ends with the total of value 4*kk + r (value index = g*8 + i)
    if constexpr (LPR == 8) {  // the two K / V rows of a DPP row first (both halves end with the sum; they store the same values below)
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[g][i] += dpp_f32<MI_DPP_ROR8>(acc[g][i]);
    }
#pragma unroll
    for (int kk = 0; kk < 2 * G; ++kk) {
        const int g = (4 * kk) / 8, i0 = (4 * kk) % 8;
        const float y1 = swap32_pairsum(acc[g][i0], acc[g][i0 + 2]);
        const float y2 = swap32_pairsum(acc[g][i0 + 1], acc[g][i0 + 3]);
        const float z = swap16_pairsum(y1, y2);
        sh[wave][g][sl * 8 + i0 + (lane >> 4)] = z;
    }
    if (sub == 0 && ul == 0) {
        sh[wave][gl][D] = m;
        sh[wave][gl][D + 1] = l;
    }
    __syncthreads();
    FA_STAMP_AT(5)
    // ---- merge the waves; one thread per (g, d)
    for (int e = tid; e < g_real * D; e += WV * 64) {
        const int g = e / D, dd = e % D;
        float mt, a, lt;
        if constexpr (WV == 4) {
            const float m0 = sh[0][g][D], m1 = sh[1][g][D], m2 = sh[2][g][D], m3 = sh[3][g][D];
            mt = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
            const float mref = mt == -INFINITY ? 0.0f : mt;
            const float c0 = fa_exp2(m0 - mref), c1 = fa_exp2(m1 - mref), c2 = fa_exp2(m2 - mref), c3 = fa_exp2(m3 - mref);
            a = ((sh[0][g][dd] * c0 + sh[1][g][dd] * c1) + sh[2][g][dd] * c2) + sh[3][g][dd] * c3;
            lt = ((sh[0][g][D + 1] * c0 + sh[1][g][D + 1] * c1) + sh[2][g][D + 1] * c2) + sh[3][g][D + 1] * c3;
        } else {
            mt = sh[0][g][D];
#pragma unroll
            for (int w = 1; w < WV; ++w) mt = fmaxf(mt, sh[w][g][D]);
            const float mref = mt == -INFINITY ? 0.0f : mt;
            a = 0.0f;
            lt = 0.0f;
#pragma unroll
            for (int w = 0; w < WV; ++w) {
                const float c = fa_exp2(sh[w][g][D] - mref);
                a += sh[w][g][dd] * c;
                lt += sh[w][g][D + 1] * c;
            }
        }
        const int h = kvh * g_real + g;
        const float mt_e = mt * (1.0f / LOG2E);  // records and sinks use the natural-log domain of the generic kernel
        if (geo.n_splits == 1 && (!FAT || geo.rec_stride == D + 2)) {  // (the fat form always leaves records: its reader is the wo prologue, also for one split)
            if (sinks) {
                const float sk = sinks[h];
                const float mn = fmaxf(mt_e, sk);
                const float c = mt == -INFINITY ? 0.0f : expf(mt_e - mn);
                a *= c;
                lt = lt * c + expf(sk - mn);
            }
            if (!FAT && geo.q8) qv[g * D + dd] = a * (1.0f / lt);  // (quantised below, two heads per Q8_K block)
            else {
                float * out = (float *) (dst.data + (int64_t) h * dst.nb[1] + (int64_t) tok * dst.nb[2] + (int64_t) bat * dst.nb[3]);
                out[dd] = a * (1.0f / lt);
            }
        } else {
            float * rec = ws + ((((int64_t) bat * geo.n_q + tok) * geo.n_head + h) * geo.n_splits + split) * geo.rec_stride;
            if (geo.merge2) {
                // merge2 layout; the two dims of a column travel as one 8-byte agent-scope store: even dims pick up their odd neighbour's
                // value from the next lane (e = tid + k * WV * 64 and D are even-aligned: dd and dd + 1 sit in neighbouring lanes)
                float * r2 = ws + (((((int64_t) bat * geo.n_q + tok) * geo.n_kv_head + kvh) * geo.n_splits + split) * g_real + g) * FA_M2_REC;
                const float nb = dpp_f32<0xF5>(a);  // quad_perm [1, 1, 3, 3]: lane 2i reads lane 2i + 1
                if ((dd & 1) == 0) st_agent2(r2 + dd, a, nb);
                if (dd == 0) st_agent2(r2 + D, mt_e, lt);
            } else if (WV == 4 && geo.arrive) {
                st_agent(rec + dd, a);
                if (dd == 0) { st_agent(rec + D, mt_e); st_agent(rec + D + 1, lt); }
            } else {
                rec[dd] = a;
                if (dd == 0) {
                    rec[D] = mt_e;
                    rec[D + 1] = lt;
                }
            }
        }
    }
    FA_STAMP_AT(6)
    }  // !empty
    if constexpr (D == 128)
    if (geo.merge2) {
        // ---- merge2: our record went out with 8-byte agent-scope (write-through) stores; once they have completed (vmcnt 0 in every storing
        // wave, then the barrier) the arrival is counted, and the workgroup that finds n_splits - 1 arrivals before it merges all records
        // with agent-scope loads — sc1 stores AND sc1 loads need no fence (MI355X_MICROARCH.md "inter-workgroup visibility") and the result
        // does not depend on where the workgroups ran
        __shared__ int s_last2;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            unsigned * cnt_p = geo.arrive + ((int64_t) bat * geo.n_q + tok) * geo.n_kv_head + kvh;
            const unsigned old = __hip_atomic_fetch_add(cnt_p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last2 = old == (unsigned) geo.n_splits - 1u;
            if (s_last2) __hip_atomic_store(cnt_p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
        }
        __syncthreads();
        if (!s_last2) return;
        fa_merge2(ws + ((((int64_t) bat * geo.n_q + tok) * geo.n_kv_head + kvh) * geo.n_splits) * (int64_t) (g_real * FA_M2_REC), kvh * g_real, g_real, geo.n_splits, sinks, dst, tok, bat,
                  WV * 64);
        return;
    }
    if constexpr (!FAT && D == 128) {
        if (geo.n_splits == 1 && geo.q8) {
            // one pass and the readers are quantised mat-muls: quantize_row_q8_K of head pairs, as k_quantize_q8_K does it
            __syncthreads();
            if (wave < g_real / 2) {
                const float4 t4 = ((const float4 *) (qv + wave * 256))[lane];
                const float t[4] = {t4.x, t4.y, t4.z, t4.w};
                wave_quantize_q8_K(t, lane, (q8k_dev *) geo.q8 + ((int64_t) bat * geo.n_q + tok) * (geo.n_head * D / 256) + (kvh * g_real) / 2 + wave);
            }
        }
        if constexpr (WV == 4)
        if (geo.n_splits > 1 && geo.arrive) {
            // ---- the last split workgroup of this (token, kv head) to get here merges all records.  Our record went out with agent-scope
            // stores; once they have completed (vmcnt 0) the arrival is counted, and the workgroup that finds n_splits - 1 arrivals before
            // it reads the others' records with agent-scope loads (MI355X_MICROARCH.md "inter-workgroup visibility": sc1 stores AND
            // sc1 loads need no fence)
            __shared__ int s_last;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                unsigned * cnt_p = geo.arrive + ((int64_t) bat * geo.n_q + tok) * geo.n_kv_head + kvh;
                const unsigned old = __hip_atomic_fetch_add(cnt_p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_last = old == (unsigned) geo.n_splits - 1u;
                if (s_last) __hip_atomic_store(cnt_p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
            }
            __syncthreads();
            if (!s_last) return;
            fa_merge_heads((const float *) ws + (((int64_t) bat * geo.n_q + tok) * geo.n_head + kvh * g_real) * geo.n_splits * (D + 2), kvh * g_real, g_real, geo.n_splits, sinks, dst,
                           tok, bat, geo.q8 ? (q8k_dev *) geo.q8 + ((int64_t) bat * geo.n_q + tok) * (geo.n_head * D / 256) + (kvh * g_real) / 2 : nullptr, &sh[0][0][0]);
        }
    }
}

// Q8OUT (head_dim 128, 256 threads = two heads): the result is consumed only by quantised mat-muls (wo of a batch) — it leaves
// the kernel as Q8_K blocks (one block = the 256 values of the two heads of a (token, head pair)), not as f32
template <int D, bool Q8OUT = false> __global__ void __launch_bounds__(Q8OUT ? 256 : D) k_fattn_combine(const float * __restrict__ ws, const float * __restrict__ sinks, const tdesc dst, const fa_geom geo,
                                                                                                 q8k_dev * __restrict__ q8 = nullptr) {
    // one thread per output dimension; every wave recomputes the (<= 64) split coefficients lane-parallel (lane s owns
    // split s: ONE round trip fetches every (m, l) pair), then the coefficients come out of their lanes through SGPRs
    // (v_readlane with compile-time indices) while up to 32 partial values per thread are in flight at once
    const int h = Q8OUT ? (int) blockIdx.x * (256 / D) + (int) threadIdx.x / D : (int) blockIdx.x, tok = blockIdx.y, bat = blockIdx.z, dd = threadIdx.x % D, lane = threadIdx.x & 63;
    const float * __restrict__ base = ws + (((int64_t) bat * geo.n_q + tok) * geo.n_head + h) * geo.n_splits * (D + 2);
    const bool has = lane < geo.n_splits;
    const float ms = has ? base[(int64_t) lane * (D + 2) + D] : -INFINITY;
    const float ls = has ? base[(int64_t) lane * (D + 2) + D + 1] : 0.0f;
    // the partial values are requested together with the (m, l) pairs — ONE memory round trip for the whole kernel; a split
    // that left an empty record (coefficient 0) may hold anything there, so it is masked by a select, not by a multiply
    float r0[32], r1[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) r0[u] = base[(int64_t) min(u, geo.n_splits - 1) * (D + 2) + dd];  // (clamped, not predicated: 32 branch-free requests; a split beyond
    const bool second = geo.n_splits > 32;                                                        //  n_splits has coefficient 0 and is dropped by the select below)
    if (second) {
#pragma unroll
        for (int u = 0; u < 32; ++u) r1[u] = base[(int64_t) min(32 + u, geo.n_splits - 1) * (D + 2) + dd];
    }
    float mn = wave_max(ms);
    float sink_term = 0.0f;
    if (sinks) {
        mn = fmaxf(mn, sinks[h]);
        sink_term = expf(sinks[h] - mn);
    }
    const float cs = ms == -INFINITY ? 0.0f : expf(ms - mn);
    const float lt = wave_sum(ls * cs) + sink_term;
    float a = 0.0f;
#pragma unroll
    for (int u = 0; u < 32; ++u) {
        const float c = readlane_f32(cs, u);
        a += c != 0.0f ? r0[u] * c : 0.0f;
    }
    if (second) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const float c = readlane_f32(cs, 32 + u);
            a += c != 0.0f ? r1[u] * c : 0.0f;
        }
    }
    if constexpr (Q8OUT) {
        __shared__ float vals[256];
        vals[threadIdx.x] = a * (1.0f / lt);
        __syncthreads();
        if (threadIdx.x < 64) {  // quantize_row_q8_K of the block, as k_quantize_q8_K does it
            const float4 t4 = ((const float4 *) vals)[lane];
            const float t[4] = {t4.x, t4.y, t4.z, t4.w};
            wave_quantize_q8_K(t, lane, q8 + ((int64_t) bat * geo.n_q + tok) * (geo.n_head * D / 256) + blockIdx.x);
        }
    } else {
        float * out = (float *) (dst.data + (int64_t) h * dst.nb[1] + (int64_t) tok * dst.nb[2] + (int64_t) bat * dst.nb[3]);
        out[dd] = a * (1.0f / lt);
    }
}

// Few splits, many (token, head) rows — the combine pass behind the matrix-core kernel of a prompt micro-batch (512 tokens x 32 heads x 4 splits:
// 16 384 rows): a thread per 4-dim chunk of a row, 8 rows per workgroup, every load of the thread (S float4 + S (m, l) pairs) requested at once.
// The arithmetic is k_fattn_combine's: coefficients exp(m_s - max), values selected (not multiplied) where the coefficient is 0, sum in split order.
// Q8OUT: the rows' only readers are quantised mat-muls (wo): a wave holds two adjacent heads of a token = one Q8_K block, lane l its values
// 4 l .. 4 l + 3 — the block is quantised in registers (wave_quantize_q8_K, the arithmetic of k_quantize_q8_K) and no f32 row is written.
template <int D, int S, bool Q8OUT = false>
__global__ void __launch_bounds__(256) k_fattn_combine_rows(const float * __restrict__ ws, const tdesc dst, const fa_geom geo, const int n_rows, q8k_dev * __restrict__ q8 = nullptr) {
    constexpr int CPR = D / 4;            // chunks (threads) per row
    constexpr int RPB = 256 / CPR;        // rows per workgroup
    const int r = (int) blockIdx.x * RPB + (int) threadIdx.x / CPR, j = (int) threadIdx.x % CPR;
    if (r >= n_rows) return;
    // row r = ((bat * n_q + tok) * n_head + h): the record layout of the split kernels
    const int h = r % geo.n_head, tok = (r / geo.n_head) % geo.n_q, bat = r / (geo.n_head * geo.n_q);
    const float * __restrict__ base = ws + (int64_t) r * geo.n_splits * (D + 2);
    float4 v[S];
    float2 ml[S];
#pragma unroll
    for (int u = 0; u < S; ++u) {
        const int uu = min(u, geo.n_splits - 1);
        // (records are D + 2 floats apart: 8-byte aligned, not 16)
        const float2 v0 = *(const float2 *) (base + (int64_t) uu * (D + 2) + 4 * j), v1 = *(const float2 *) (base + (int64_t) uu * (D + 2) + 4 * j + 2);
        v[u] = make_float4(v0.x, v0.y, v1.x, v1.y);
        ml[u] = *(const float2 *) (base + (int64_t) uu * (D + 2) + D);
    }
    float mn = -INFINITY;
#pragma unroll
    for (int u = 0; u < S; ++u) mn = fmaxf(mn, u < geo.n_splits ? ml[u].x : -INFINITY);
    float lt = 0.0f;
    float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int u = 0; u < S; ++u) {
        const float c = (u < geo.n_splits && ml[u].x != -INFINITY) ? expf(ml[u].x - mn) : 0.0f;
        lt += (u < geo.n_splits ? ml[u].y : 0.0f) * c;
        a.x += c != 0.0f ? v[u].x * c : 0.0f;
        a.y += c != 0.0f ? v[u].y * c : 0.0f;
        a.z += c != 0.0f ? v[u].z * c : 0.0f;
        a.w += c != 0.0f ? v[u].w * c : 0.0f;
    }
    const float inv = 1.0f / lt;
    if constexpr (Q8OUT) {
        static_assert(D == 128, "two heads of 128 = one block");
        const float t[4] = {a.x * inv, a.y * inv, a.z * inv, a.w * inv};
        wave_quantize_q8_K(t, (int) threadIdx.x & 63, q8 + (r >> 1));  // (n_head and n_rows are even: the launcher checked)
    } else {
        float * out = (float *) (dst.data + (int64_t) h * dst.nb[1] + (int64_t) tok * dst.nb[2] + (int64_t) bat * dst.nb[3]);
        *(float4 *) (out + 4 * j) = make_float4(a.x * inv, a.y * inv, a.z * inv, a.w * inv);
    }
}

// is the combine pass of this shape served by the row-parallel kernel (which also quantises for free)?
bool fattn_combine_rows_applies(int D, int64_t n_q, int64_t n_head, int64_t n_batch, int n_splits, const float * sinks) {
    static const bool rows_on = !getenv("GGML_MI355X_FA_COMBINE_ROWS") || atoi(getenv("GGML_MI355X_FA_COMBINE_ROWS")) != 0;
    const int64_t n_rows = n_batch * n_q * n_head;
    return rows_on && !sinks && D == 128 && n_splits >= 2 && n_splits <= 8 && n_rows >= 2048 && n_rows < (1 << 30);
}
void launch_flash_attn_combine(hipStream_t s, int D, const float * ws, const float * sinks, const tdesc & dst, int n_q, int n_head, int n_batch, int n_splits, void * q8_out) {
    fa_geom geo{};
    geo.n_q = n_q;
    geo.n_head = n_head;
    geo.n_splits = n_splits;
    const int64_t n_rows = (int64_t) n_batch * n_q * n_head;
    if (fattn_combine_rows_applies(D, n_q, n_head, n_batch, n_splits, sinks) && (((uintptr_t) ws) & 7) == 0 &&
        (q8_out ? (n_head % 2) == 0 : ((dst.nb[1] % 16) == 0 && (dst.nb[2] % 16) == 0 && (dst.nb[3] % 16) == 0 && (((uintptr_t) dst.data) & 15) == 0))) {
        const unsigned blocks = (unsigned) ((n_rows + 7) / 8);
        if (q8_out) {
            if (n_splits <= 4) hipLaunchKernelGGL((k_fattn_combine_rows<128, 4, true>), dim3(blocks), dim3(256), 0, s, ws, dst, geo, (int) n_rows, (q8k_dev *) q8_out);
            else hipLaunchKernelGGL((k_fattn_combine_rows<128, 8, true>), dim3(blocks), dim3(256), 0, s, ws, dst, geo, (int) n_rows, (q8k_dev *) q8_out);
        } else {
            if (n_splits <= 4) hipLaunchKernelGGL((k_fattn_combine_rows<128, 4>), dim3(blocks), dim3(256), 0, s, ws, dst, geo, (int) n_rows, (q8k_dev *) nullptr);
            else hipLaunchKernelGGL((k_fattn_combine_rows<128, 8>), dim3(blocks), dim3(256), 0, s, ws, dst, geo, (int) n_rows, (q8k_dev *) nullptr);
        }
        return;
    }
    dim3 g2((unsigned) n_head, (unsigned) n_q, (unsigned) n_batch);
    if (q8_out && D == 128) {
        hipLaunchKernelGGL((k_fattn_combine<128, true>), dim3((unsigned) (n_head / 2), (unsigned) n_q, (unsigned) n_batch), dim3(256), 0, s, ws, sinks, dst, geo, (q8k_dev *) q8_out);
        return;
    }
    if (D == 128) hipLaunchKernelGGL((k_fattn_combine<128>), g2, dim3(128), 0, s, ws, sinks, dst, geo);
    else hipLaunchKernelGGL((k_fattn_combine<64>), g2, dim3(64), 0, s, ws, sinks, dst, geo);
}

bool fattn_prefers_lists(const tdesc & q, const tdesc * mask, int mask_sparse) {
    static const bool on = !getenv("GGML_MI355X_FA_SPARSE_LISTS") || atoi(getenv("GGML_MI355X_FA_SPARSE_LISTS")) != 0;
    return on && mask != nullptr && q.ne[1] >= fattn_mma_min_q() && q.ne[1] <= 256 && mask_sparse != 0;
}
int fattn_pick_splits(const tdesc & q, const tdesc & k, const tdesc * mask, int mask_sparse) {
    if (q.ne[1] >= fattn_mma_min_q() && (k.ne[0] == 64 || k.ne[0] == 128) && !fattn_prefers_lists(q, mask, mask_sparse)) return fattn_mma_pick_splits(q, k);  // (soft-capped / ALiBi batches fall back to the generic kernel with this count)
    const int64_t n_kv = k.ne[1];
    if (q.ne[1] > 1 && (k.ne[0] == 128 || (k.ne[0] == 64 && k.type == GGML_TYPE_F16 && k.ne[2] > 0 && q.ne[2] % k.ne[2] == 0 && fa_g_ok(q.ne[2] / k.ne[2]))) && q.ne[3] == 1) {
        // a few tokens at head_dim 128 (continuous-batching decode, speculative batches): the tile-list kernel — every split takes a
        // share of the token's VISIBLE tiles, so the count follows the number of (token, kv head) groups, not the cache size
        const int64_t groups = k.ne[2] * q.ne[1];
        // position lists make a token's work its own visible cells (~n_kv / n_q of a shared cache), 64 per trip: split only when a
        // workgroup would otherwise walk more than ~8 trips, or when the (token, kv head) groups alone leave most CUs idle.  -np 32 at a
        // few hundred cells per sequence: ONE pass, which also writes the Q8_K blocks the wo mat-mul reads (4.35 -> 4.30 ms per step)
        const int64_t trips = std::max<int64_t>(1, (n_kv / q.ne[1] + 63) / 64);
        const int64_t by_len = (trips + 7) / 8, by_occ = std::min<int64_t>(trips, (256 + groups - 1) / groups);
        return (int) std::max<int64_t>(1, std::min<int64_t>(16, std::max(by_len, by_occ)));
    }
    if (q.ne[1] > 1) {  // a few tokens, other head sizes: short splits
        const int64_t by_len = (n_kv + 255) / 256;
        return (int) std::max<int64_t>(1, std::min<int64_t>(64, std::min<int64_t>(by_len, std::max<int64_t>(1, n_kv / 64))));
    }
    const int64_t groups = k.ne[2] * q.ne[1] * q.ne[3];
    int64_t want = (768 + groups - 1) / groups;  // ~3 workgroups per CU
    const int64_t max_by_len = std::max<int64_t>(1, n_kv / 86);  // (~1.4 trips of 64 cells per split: 24 splits at 2 100 cells measure 0.8 % of a decode step ahead of 32 — fewer records for the combine pass)
    want = std::max<int64_t>(1, std::min<int64_t>(want, std::min<int64_t>(max_by_len, 64)));
    // an f16 cache is served by EIGHT-wave workgroups (154 VGPRs: one workgroup fills a CU's register file), so more workgroups than CUs run in rounds of
    // one-trip latency chains: at 8 k context 64 splits x 8 KV heads measured 20.1 us per layer (split + combine), 48 splits 23.8, 32 splits — one workgroup
    // per CU, two pipelined trips each — 18.2 (profiles/r06_fa_splits_by_context.txt).  The four-wave forms (q8_0 / block-format caches: three workgroups
    // per CU) keep the longer list: 64 splits 17.3 us against 18.0 at 32
    if (k.type == GGML_TYPE_F16 && q.ne[1] == 1 && want * groups > 256) want = std::max<int64_t>(1, 256 / groups);
    return (int) want;
}
static size_t fattn_partials_bytes(const tdesc & q, const tdesc & v, int n_splits) {
    if (n_splits <= 1) return 0;
    return ((size_t) (q.ne[1] * q.ne[3] * q.ne[2]) * (size_t) std::max(n_splits, 16) * (size_t) (v.ne[0] + 4) * sizeof(float) + 255) & ~(size_t) 255;  // (covers the fat-split records too)
}
// Single-token decode at head_dim 128 whose result goes straight into a quantised mat-vec (wo): number of fat splits (<= 16) for
// the 8-wave kernel, 0 if that form does not serve the case.  A workgroup trip covers 8 waves x 4 rows x NG positions; splits are
// sized in whole trips so that no workgroup runs a nearly empty one, and there are at most 12 of them: the wo prologue fetches up
// to 12 records per head in one round trip.
int fattn_fat_splits(const tdesc & q, const tdesc & k, const tdesc * mask, const float * sinks, const fattn_params & p) {
    if (q.ne[1] != 1 || q.ne[3] != 1 || k.ne[0] != 128 || k.ne[3] != 1 || p.kv_type != GGML_TYPE_F16 || sinks || p.logit_softcap != 0.0f || p.max_bias != 0.0f) return 0;
    if (q.ne[2] % k.ne[2] != 0 || (q.ne[2] % 2) != 0) return 0;
    const int64_t G = q.ne[2] / k.ne[2];
    if (!fa_g_ok(G) || (q.nb[1] % 16) != 0 || (q.nb[2] % 16) != 0 || ((uintptr_t) q.data & 15) != 0) return 0;
    if (mask && (mask->type != GGML_TYPE_F16)) return 0;
    const int64_t trip = 32 * (16 / fa_gg(G));
    const int64_t trips = (k.ne[1] + trip - 1) / trip;
    int64_t splits = std::min<int64_t>(12, trips);
    // whole trips per split: e.g. 9 trips -> 9 splits of 1; 32 trips -> 16 splits of 2; 33 trips -> 11 splits of 3
    const int64_t per = (trips + splits - 1) / splits;
    splits = (trips + per - 1) / per;
    return (int) splits;
}
// prompt batches over a block_q8_0 cache run the matrix-core kernel on an f16 image of the K and V views (below)
static bool fattn_q8_via_f16(const tdesc & q, int kv_type) { return kv_type == GGML_TYPE_Q8_0 && q.ne[1] >= fattn_mma_min_q(); }
size_t fattn_workspace_bytes(const tdesc & q, const tdesc & k, const tdesc & v, int n_splits, int kv_type) {
    size_t b = fattn_partials_bytes(q, v, n_splits);
    if (fattn_q8_via_f16(q, kv_type)) b += 2 * (size_t) (k.ne[0] * k.ne[1] * k.ne[2] * k.ne[3]) * sizeof(uint16_t) + 512;
    return b;
}

// block_q8_0 K / V views [D, n_kv, n_kv_head] -> f16 rows [n_kv][n_kv_head * D]: one workgroup per cache cell and tensor,
// a thread per 4 values (d * q is exact in f32; one rounding to f16)
__global__ void __launch_bounds__(256) k_q8_0_rows_to_f16(const tdesc k, const tdesc v, uint16_t * __restrict__ ko, uint16_t * __restrict__ vo) {
    const tdesc & t = blockIdx.y ? v : k;
    uint16_t * out = blockIdx.y ? vo : ko;
    const int64_t cell = blockIdx.x, per_head = t.ne[0] / 4, per_row = per_head * t.ne[2];
    for (int64_t i = threadIdx.x; i < per_row; i += blockDim.x) {
        const int64_t h = i / per_head, e = (i - h * per_head) * 4;
        const char * blk = t.data + cell * t.nb[1] + h * t.nb[2] + (e >> 5) * 34;
        const float d = h2f(ld16(blk));
        const uint32_t qs = ld32_a2(blk + 2 + (e & 31));
        uint16_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = f2h(d * (float) (int8_t) (qs >> (8 * j)));
        *(uint2 *) (out + (cell * t.ne[2] + h) * t.ne[0] + e) = make_uint2(o[0] | ((uint32_t) o[1] << 16), o[2] | ((uint32_t) o[3] << 16));
    }
}

template <int D, int G> static void launch_fa(hipStream_t s, const tdesc & q, const tdesc & k, const tdesc & v, const tdesc & mask, const float * sinks,
                                              const tdesc & dst, const fa_geom & geo, float * ws) {
    dim3 grid((unsigned) geo.n_splits, (unsigned) geo.n_kv_head, (unsigned) (geo.n_q * q.ne[3]));
    hipLaunchKernelGGL((k_fattn_split<D, G>), grid, dim3(256), 0, s, q, k, v, mask, sinks, dst, geo, ws);
    if (geo.n_splits > 1) {
        dim3 g2((unsigned) geo.n_head, (unsigned) geo.n_q, (unsigned) q.ne[3]);
        hipLaunchKernelGGL((k_fattn_combine<D>), g2, dim3(D), 0, s, ws, sinks, dst, geo);
    }
}

// ---- position lists: for every query token, the cache cells at which its mask row holds anything but -inf, in ascending order;
// lists[t * stride] = count, entries follow (stride = n_kv + 1).  One workgroup per token, four cells per thread and pass,
// ordered compaction by a wave prefix sum over the per-thread counts.
// (M32: an f32 mask — what llama.cpp builds when flash attention is off; rows 16-byte aligned)
template <bool M32>
__global__ void __launch_bounds__(256) k_fattn_pos_scan(const tdesc mask, const int n_kv, int * __restrict__ lists, const int stride) {
    __shared__ int wcnt[4];
    const int tok = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint16_t * mrow = (const uint16_t *) (mask.data + (int64_t) tok * mask.nb[1]);
    int * out = lists + (int64_t) tok * stride + 1;
    int base = 0;
    for (int c0 = 0; c0 < n_kv; c0 += 1024) {
        const int pp = c0 + 4 * tid;
        bool v0 = false, v1 = false, v2 = false, v3 = false;
        if (M32) {
            if (pp < n_kv) {
                const uint4 w = *(const uint4 *) ((const float *) mrow + pp);
                v0 = w.x != 0xFF800000u; v1 = w.y != 0xFF800000u; v2 = w.z != 0xFF800000u; v3 = w.w != 0xFF800000u;
            }
        } else {
            uint2 w = make_uint2(0xFC00FC00u, 0xFC00FC00u);
            if (pp < n_kv) w = *(const uint2 *) (mrow + pp);  // n_kv is a multiple of 4 (checked by fattn_list_tile): the four cells are in range together
            v0 = (w.x & 0xFFFFu) != 0xFC00u; v1 = (w.x >> 16) != 0xFC00u; v2 = (w.y & 0xFFFFu) != 0xFC00u; v3 = (w.y >> 16) != 0xFC00u;
        }
        const int mine = (int) v0 + (int) v1 + (int) v2 + (int) v3;
        int incl = mine;  // inclusive prefix sum over the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wcnt[wave] = incl;
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) {
            woff += wv < wave ? wcnt[wv] : 0;
            tot += wcnt[wv];
        }
        int o = base + woff + incl - mine;
        if (v0) out[o++] = pp;
        if (v1) out[o++] = pp + 1;
        if (v2) out[o++] = pp + 2;
        if (v3) out[o++] = pp + 3;
        base += tot;
        __syncthreads();
    }
    if (tid == 0) out[-1] = base;
}
// does the position-list kernel apply to this attention shape?  Returns 1 (the scan granule: single cells) or 0 (`lists` is not used)
int fattn_list_tile(const tdesc & q, const tdesc & k, const tdesc * mask, const fattn_params & p, size_t lists_bytes) {
    const int64_t n_q = q.ne[1];
    const int G = k.ne[2] > 0 ? (int) (q.ne[2] / k.ne[2]) : 0;
    static const bool on = !getenv("GGML_MI355X_FA_LIST") || atoi(getenv("GGML_MI355X_FA_LIST")) != 0;
    const bool d64 = k.ne[0] == 64 && k.type == GGML_TYPE_F16;  // (round 6: the eight-lane-row form of the kernel; fattn_q8_out_ok refuses head_dim 64, arrival counters are not used)
    if (!on || n_q < 2 || (n_q >= fattn_mma_min_q() && !fattn_prefers_lists(q, mask, p.mask_sparse)) || !mask || q.ne[3] != 1 || mask->ne[3] != 1 || (k.ne[0] != 128 && !d64)) return 0;
    if (p.logit_softcap != 0.0f || p.max_bias != 0.0f || !fa_g_ok(G) || p.n_splits < 1) return 0;
    if ((k.ne[1] % 4) != 0 || (mask->nb[1] % 8) != 0 || ((uintptr_t) mask->data & 7) != 0 || mask->type != GGML_TYPE_F16) return 0;
    if ((size_t) (n_q * (k.ne[1] + 1)) * sizeof(int) > lists_bytes) return 0;
    return 1;
}
void launch_fattn_tile_scan(hipStream_t s, const tdesc & mask, int n_q, int n_kv, int tile, int * lists) {
    (void) tile;
    if (mask.type == GGML_TYPE_F32) hipLaunchKernelGGL(k_fattn_pos_scan<true>, dim3((unsigned) n_q), dim3(256), 0, s, mask, n_kv, lists, n_kv + 1);
    else hipLaunchKernelGGL(k_fattn_pos_scan<false>, dim3((unsigned) n_q), dim3(256), 0, s, mask, n_kv, lists, n_kv + 1);
}

// K / V kept in another cache type: can the lane-parallel kernel read them in place (its KVT form)?  The shapes of that kernel (head_dim 128, 2 / 4 / 7 / 8
// query heads per KV head, no soft-capping / ALiBi, up to 32 query tokens — bigger batches go to the matrix cores over the f16 image), K and V in the same
// one of the integer-level formats, 2-byte-aligned strides; everything else (iq4_nl, bf16, f32, mixed pairs, head_dim 64) goes through the image
static bool dq_type_ok(int t) { return t == GGML_TYPE_Q4_0 || t == GGML_TYPE_Q4_1 || t == GGML_TYPE_Q5_0 || t == GGML_TYPE_Q5_1 || t == GGML_TYPE_IQ4_NL || t == GGML_TYPE_BF16; }
bool fattn_native_kv_ok(const tdesc & q, const tdesc & k, const tdesc & v, const fattn_params & p) {
    static const bool on = !getenv("GGML_MI355X_FA_NATIVE_KV") || atoi(getenv("GGML_MI355X_FA_NATIVE_KV")) != 0;
    if (!on || !dq_type_ok(k.type) || v.type != k.type || k.ne[0] != 128 || v.ne[0] != 128 || k.ne[2] <= 0 || q.ne[2] % k.ne[2] != 0) return false;
    const int G = (int) (q.ne[2] / k.ne[2]);
    if (!fa_g_ok(G) || p.logit_softcap != 0.0f || p.max_bias != 0.0f || q.ne[1] > 32 || p.n_splits < 1) return false;
    if ((q.nb[1] % 16) != 0 || (q.nb[2] % 16) != 0 || ((uintptr_t) q.data & 15) != 0) return false;
    const int al = k.type == GGML_TYPE_BF16 ? 16 : 2;  // (bf16 rows are read as the f16 cache's are: 16 bytes per lane)
    for (const tdesc * t : {&k, &v})
        if ((t->nb[1] % al) || (t->nb[2] % al) || (t->nb[3] % al) || ((uintptr_t) t->data % al)) return false;
    return true;
}

// will launch_flash_attn end in the quantising combine pass for these arguments? (mirrors its dispatch)
bool fattn_q8_out_ok(const tdesc & q, const tdesc & k, const tdesc * mask, const float * sinks, const tdesc & dst, const fattn_params & p) {
    if (k.ne[0] != 128 || (q.ne[2] % 2) != 0 || p.n_splits < 1 || q.ne[3] != 1 || k.ne[2] <= 0) return false;
    const bool q8 = p.kv_type == GGML_TYPE_Q8_0;
    if (p.n_splits >= 2 && (!q8 || (fattn_q8_via_f16(q, p.kv_type) && k.ne[3] == 1)) && flash_attn_mma_applies(q, k, mask, sinks, dst, p)) return true;
    const int G = (int) (q.ne[2] / k.ne[2]);
    if (p.n_splits == 1 && ((G & 1) || p.fat || flash_attn_mma_applies(q, k, mask, sinks, dst, p))) return false;  // one pass: the lane-parallel kernel quantises whole head pairs of a kv group
    return p.logit_softcap == 0.0f && p.max_bias == 0.0f && fa_g_ok(G) && (q.nb[1] % 16) == 0 && (q.nb[2] % 16) == 0 && ((uintptr_t) q.data & 15) == 0;
}
void launch_flash_attn(hipStream_t s, const tdesc & q, const tdesc & k, const tdesc & v, const tdesc * mask, const float * sinks, const tdesc & dst,
                       const fattn_params & p, void * workspace) {
    const bool q8 = p.kv_type == GGML_TYPE_Q8_0;
    if (!q8 && !p.dq && !p.lists && launch_flash_attn_mma(s, q, k, v, mask, sinks, dst, p, workspace)) return;  // (lists set: the caller chose the position-list form)
    if (fattn_q8_via_f16(q, p.kv_type) && k.ne[3] == 1 && v.ne[3] == 1 && flash_attn_mma_applies(q, k, mask, sinks, dst, p)) {
        // a prompt batch over the quantised cache: the CPU quantises each query row to Q8_0 and takes integer block dots; here
        // the cells are expanded to f16 once (exact up to the f16 rounding of d * q) and the f16 matrix-core kernel runs on
        // that image with the queries in f16 — closer to exact attention than the CPU's 8-bit queries (tests pin that)
        uint16_t * kf = (uint16_t *) ((char *) workspace + fattn_partials_bytes(q, v, p.n_splits));
        uint16_t * vf = kf + (((size_t) (k.ne[0] * k.ne[1] * k.ne[2]) + 127) & ~(size_t) 127);
        hipLaunchKernelGGL(k_q8_0_rows_to_f16, dim3((unsigned) k.ne[1], 2), dim3(256), 0, s, k, v, kf, vf);
        tdesc k16 = k, v16 = v;
        k16.data = (char *) kf;
        v16.data = (char *) vf;
        for (tdesc * t : {&k16, &v16}) {
            t->nb[0] = 2;
            t->nb[2] = t->ne[0] * 2;
            t->nb[1] = t->ne[2] * t->nb[2];
            t->nb[3] = t->ne[1] * t->nb[1];
        }
        fattn_params p16 = p;
        p16.kv_type = GGML_TYPE_F16;
        if (launch_flash_attn_mma(s, q, k16, v16, mask, sinks, dst, p16, workspace)) return;
        MI_ERR("launch_flash_attn: the matrix-core kernel refused a batch flash_attn_mma_applies accepted");
        abort();
    }
    fa_geom geo{};
    geo.rec_stride = (int) k.ne[0] + 2;
    geo.n_q = (int) q.ne[1];
    geo.n_head = (int) q.ne[2];
    geo.n_kv_head = (int) k.ne[2];
    geo.n_kv = (int) k.ne[1];
    geo.n_splits = p.n_splits;
    geo.has_mask = mask ? 1 : 0;
    geo.scale = p.logit_softcap != 0.0f ? p.scale / p.logit_softcap : p.scale;
    geo.softcap = p.logit_softcap;
    geo.max_bias = p.max_bias;
    geo.one_batch = q.ne[3] == 1 ? 1 : 0;
    geo.per = geo.n_splits > 0 ? (geo.n_kv + geo.n_splits - 1) / geo.n_splits : 0;
    geo.n_head_log2 = 1u << (uint32_t) floor(log2((double) geo.n_head));
    geo.m0 = powf(2.0f, -(p.max_bias) / (float) geo.n_head_log2);
    geo.m1 = powf(2.0f, -(p.max_bias / 2.0f) / (float) geo.n_head_log2);
    const int D = (int) k.ne[0], G = geo.n_head / geo.n_kv_head;
    const tdesc mk = mask ? *mask : q;
    float * ws = (float *) workspace;
    // batch-1 / small-batch decode at head_dim 128 without soft-capping or ALiBi: the lane-parallel kernel
    if (D == 128 && p.logit_softcap == 0.0f && p.max_bias == 0.0f && fa_g_ok(G) && (q.nb[1] % 16) == 0 && (q.nb[2] % 16) == 0 &&
        ((uintptr_t) q.data & 15) == 0) {
        dim3 grid((unsigned) geo.n_splits, (unsigned) geo.n_kv_head, (unsigned) (geo.n_q * q.ne[3]));
        if (p.fat) {
            // one decode token, few fat splits on 8-wave workgroups; the partial records stay in the workspace for the wo prologue
            if (q8 || p.dq || geo.n_q != 1 || q.ne[3] != 1 || sinks != nullptr) { MI_ERR("launch_flash_attn: fat-split form requested for a case it does not serve"); abort(); }
            geo.rec_stride = FA_REC;
            if (fa_gg(G) == 2) hipLaunchKernelGGL((k_fattn_dec128<2, 0, false, 8>), grid, dim3(512), 0, s, q, k, v, mk, sinks, dst, geo, ws, G, nullptr, 0);
            else if (fa_gg(G) == 4) hipLaunchKernelGGL((k_fattn_dec128<4, 0, false, 8>), grid, dim3(512), 0, s, q, k, v, mk, sinks, dst, geo, ws, G, nullptr, 0);
            else hipLaunchKernelGGL((k_fattn_dec128<8, 0, false, 8>), grid, dim3(512), 0, s, q, k, v, mk, sinks, dst, geo, ws, G, nullptr, 0);
            return;
        }
        // several query tokens with a mask: skip the KV trips a token cannot see — needs splits of at most 32 trips
        const int per = ((geo.n_kv + geo.n_splits - 1) / geo.n_splits + 63) / 64 * 64;
        static const bool skip_on = !getenv("GGML_MI355X_FA_SKIP") || atoi(getenv("GGML_MI355X_FA_SKIP")) != 0;
        // p.lists set: the caller has run (or re-used) k_fattn_pos_scan for this mask (fattn_list_tile() said the list form applies)
        const bool list = p.lists != nullptr;
        const int lstride = geo.n_kv + 1;
        const bool dq = p.dq != 0;
        const bool skip = !list && skip_on && geo.n_q > 1 && (geo.n_q <= 64 || q8) && mask != nullptr && per <= 32 * 16 * (16 / fa_gg(G)) && (geo.n_kv % 4) == 0 && (mask->nb[1] % 8) == 0 &&
                          (mask->nb[3] % 8) == 0 && ((uintptr_t) mask->data & 7) == 0 && geo.n_splits > 1;
        // one decode token over an f16 cache: eight waves per workgroup (a trip covers 128 cells: splits of up to 128 cells are ONE trip)
        // (measured, 2048-token context, 24 splits: 13.45 -> 12.6 us per layer for both launches, 480.7 -> 488.5 tok/s; profiles/r03_decode_ab_fa_wv8.txt)
        static const int wv8 = getenv("GGML_MI355X_FA_WV8") ? atoi(getenv("GGML_MI355X_FA_WV8")) : 1;
        static const int list_wv8 = getenv("GGML_MI355X_FA_LIST_WV8") ? atoi(getenv("GGML_MI355X_FA_LIST_WV8")) : 0;
        // n_splits > 1: the last split workgroup of a (token, kv head) merges the records itself when the caller gave arrival counters
        // (one launch instead of two); Q8_K output needs whole head pairs inside a kv group.  merge2 (round 4): the one-round-trip merge,
        // for the plain form (no lists, no trip skipping) with f32 output and at most 32 splits, on four or eight waves
        const bool self_merge = geo.n_splits > 1 && p.arrive != nullptr && (int64_t) geo.n_q * q.ne[3] * geo.n_kv_head <= (int64_t) p.arrive_slots && !(p.q8_out && (G & 1));
        const bool merge2 = self_merge && !list && !skip && !q8 && !dq && p.q8_out == nullptr && geo.n_splits <= 32;
        const bool list8 = list_wv8 && list && !q8 && !dq && !p.arrive;
        const bool wide8 = wv8 && !q8 && !dq && !list && !skip && geo.n_q == 1 && geo.n_splits > 1 && (!p.arrive || merge2);
#define FA_DEC_T(GG, TT)                                                                                                                    \
    {                                                                                                                                       \
        if (list) hipLaunchKernelGGL((k_fattn_dec128<GG, 2, false, 4, TT>), grid, dim3(256), 0, s, q, k, v, mk, sinks, dst, geo, ws, G, p.lists, lstride);  \
        else if (skip) hipLaunchKernelGGL((k_fattn_dec128<GG, 1, false, 4, TT>), grid, dim3(256), 0, s, q, k, v, mk, sinks, dst, geo, ws, G, nullptr, 0);   \
        else hipLaunchKernelGGL((k_fattn_dec128<GG, 0, false, 4, TT>), grid, dim3(256), 0, s, q, k, v, mk, sinks, dst, geo, ws, G, nullptr, 0);             \
    }
#define FA_DEC(GG)                                                                                                                          \
    {                                                                                                                                       \
        if (wide8) hipLaunchKernelGGL((k_fattn_dec128<GG, 0, false, 8>), grid, dim3(512), 0, s, q, k, v, mk, sinks, dst, geo, ws, G, nullptr, 0);           \
        else if (list8) hipLaunchKernelGGL((k_fattn_dec128<GG, 2, false, 8>), grid, dim3(512), 0, s, q, k, v, mk, sinks, dst, geo, ws, G, p.lists, lstride); \
        else if (dq) {                                                                                                                           \
            if (k.type == GGML_TYPE_Q4_0) FA_DEC_T(GG, GGML_TYPE_Q4_0) else if (k.type == GGML_TYPE_Q4_1) FA_DEC_T(GG, GGML_TYPE_Q4_1)               \
            else if (k.type == GGML_TYPE_Q5_0) FA_DEC_T(GG, GGML_TYPE_Q5_0) else if (k.type == GGML_TYPE_Q5_1) FA_DEC_T(GG, GGML_TYPE_Q5_1)           \
            else if (k.type == GGML_TYPE_BF16) FA_DEC_T(GG, GGML_TYPE_BF16)                                                                       \
            else FA_DEC_T(GG, GGML_TYPE_IQ4_NL)                                                                                                  \
        } else if (q8) {                                                                                                                         \
            if (list) hipLaunchKernelGGL((k_fattn_dec128<GG, 2, true>), grid, dim3(256), 0, s, q, k, v, mk, sinks, dst, geo, ws, G, p.lists, lstride);      \
            else if (skip) hipLaunchKernelGGL((k_fattn_dec128<GG, 1, true>), grid, dim3(256), 0, s, q, k, v, mk, sinks, dst, geo, ws, G, nullptr, 0);       \
            else hipLaunchKernelGGL((k_fattn_dec128<GG, 0, true>), grid, dim3(256), 0, s, q, k, v, mk, sinks, dst, geo, ws, G, nullptr, 0);                 \
        } else {                                                                                                                            \
            if (list) hipLaunchKernelGGL((k_fattn_dec128<GG, 2, false>), grid, dim3(256), 0, s, q, k, v, mk, sinks, dst, geo, ws, G, p.lists, lstride);     \
            else if (skip) hipLaunchKernelGGL((k_fattn_dec128<GG, 1, false>), grid, dim3(256), 0, s, q, k, v, mk, sinks, dst, geo, ws, G, nullptr, 0);      \
            else hipLaunchKernelGGL((k_fattn_dec128<GG, 0, false>), grid, dim3(256), 0, s, q, k, v, mk, sinks, dst, geo, ws, G, nullptr, 0);                \
        }                                                                                                                                   \
    }
        geo.arrive = self_merge ? p.arrive : nullptr;
        geo.merge2 = merge2 ? 1 : 0;
        geo.q8 = (self_merge || (geo.n_splits == 1 && !(G & 1))) ? p.q8_out : nullptr;
        geo.per = skip ? per : (geo.n_kv + geo.n_splits - 1) / geo.n_splits;
        {
            // one launch per batch slice (q.ne[3]; 1 in every graph llama.cpp builds): the kernel has no batch arithmetic
            const tdesc q0 = q, k0 = k, v0 = v, mk0 = mk, dst0 = dst;
            const fa_geom geo0 = geo;
            float * const ws0 = ws;
            const int64_t nb_ = q0.ne[3];
            const dim3 grid((unsigned) geo0.n_splits, (unsigned) geo0.n_kv_head, (unsigned) geo0.n_q);
            for (int64_t b = 0; b < nb_; ++b) {
                tdesc q = q0, k = k0, v = v0, mk = mk0, dst = dst0;
                fa_geom geo = geo0;
                q.data += b * q0.nb[3];
                k.data += (b / (q0.ne[3] / k0.ne[3])) * k0.nb[3];
                v.data += (b / (q0.ne[3] / v0.ne[3])) * v0.nb[3];
                if (mask) mk.data += (b % mk0.ne[3]) * mk0.nb[3];
                dst.data += b * dst0.nb[3];
                const int64_t rows = b * geo0.n_q;  // (batch, token) pairs in front of this slice
                float * ws = ws0 + (merge2 ? rows * geo0.n_kv_head * geo0.n_splits * G * FA_M2_REC : rows * geo0.n_head * geo0.n_splits * geo0.rec_stride);
                if (geo.arrive) geo.arrive += rows * geo0.n_kv_head;
                if (geo.q8) geo.q8 = (q8k_dev *) geo.q8 + rows * (geo0.n_head * 128 / 256);
                if (fa_gg(G) == 2) FA_DEC(2) else if (fa_gg(G) == 4) FA_DEC(4) else FA_DEC(8)
            }
        }
#undef FA_DEC
#undef FA_DEC_T
        if (geo.n_splits > 1 && !self_merge) launch_flash_attn_combine(s, 128, ws, sinks, dst, geo.n_q, geo.n_head, (int) q.ne[3], geo.n_splits, p.q8_out);
        return;
    }
    if (p.dq) {
        MI_ERR("launch_flash_attn: K / V in another cache type reached the generic kernel (head_dim %d, group %d) — fattn_native_kv_ok should have refused it", D, G);
        abort();
    }
    if (q8) {
        MI_ERR("launch_flash_attn: q8_0 K/V reached the generic kernel (head_dim %d, group %d) — supports_op should have refused it", D, G);
        abort();
    }
    // one decode token at head_dim 64 over an f16 cache (round 6): the lane-parallel kernel in its eight-lane-row form, records + combine pass
    // (generic kernel: 14.9 us per layer for TinyLlama at a 600-cell context; profiles/r06_secondary_kernel_stats.txt)
    static const bool dec64_on = !getenv("GGML_MI355X_FA_DEC64") || atoi(getenv("GGML_MI355X_FA_DEC64")) != 0;
    if (dec64_on && D == 64 && k.type == GGML_TYPE_F16 && v.type == GGML_TYPE_F16 && (geo.n_q == 1 || (p.lists && q.ne[3] == 1)) && p.logit_softcap == 0.0f && p.max_bias == 0.0f && fa_g_ok(G) &&
        (q.nb[1] % 16) == 0 && (q.nb[2] % 16) == 0 && ((uintptr_t) q.data & 15) == 0 && (k.nb[1] % 16) == 0 && (v.nb[1] % 16) == 0 && (k.nb[2] % 16) == 0 && (v.nb[2] % 16) == 0 &&
        ((uintptr_t) k.data & 15) == 0 && ((uintptr_t) v.data & 15) == 0 && p.q8_out == nullptr) {
        geo.arrive = nullptr;
        geo.merge2 = 0;
        geo.q8 = nullptr;
        geo.per = (geo.n_kv + geo.n_splits - 1) / geo.n_splits;
        const tdesc q0 = q, k0 = k, v0 = v, mk0 = mk, dst0 = dst;
        const dim3 grid((unsigned) geo.n_splits, (unsigned) geo.n_kv_head, (unsigned) geo.n_q);
        const bool list = p.lists != nullptr;  // (2 .. 32 tokens of a -np decode step: each walks the list of its own visible cells, as at head_dim 128)
        const int lstride = geo.n_kv + 1;
        const bool wide = !list && geo.n_splits > 1;  // (eight waves: a trip covers 64 / 128 cells)
        for (int64_t b = 0; b < q0.ne[3]; ++b) {  // one launch per batch slice: the kernel has no batch arithmetic
            tdesc qb = q0, kb = k0, vb = v0, mb = mk0, db = dst0;
            qb.data += b * q0.nb[3];
            kb.data += (b / (q0.ne[3] / k0.ne[3])) * k0.nb[3];
            vb.data += (b / (q0.ne[3] / v0.ne[3])) * v0.nb[3];
            if (mask) mb.data += (b % mk0.ne[3]) * mk0.nb[3];
            db.data += b * dst0.nb[3];
            float * wsb = ws + b * geo.n_head * geo.n_splits * geo.rec_stride;
            if (list) {
                if (fa_gg(G) == 2) hipLaunchKernelGGL((k_fattn_dec128<2, 2, false, 4, 0, 64>), grid, dim3(256), 0, s, qb, kb, vb, mb, sinks, db, geo, wsb, G, p.lists, lstride);
                else if (fa_gg(G) == 4) hipLaunchKernelGGL((k_fattn_dec128<4, 2, false, 4, 0, 64>), grid, dim3(256), 0, s, qb, kb, vb, mb, sinks, db, geo, wsb, G, p.lists, lstride);
                else hipLaunchKernelGGL((k_fattn_dec128<8, 2, false, 4, 0, 64>), grid, dim3(256), 0, s, qb, kb, vb, mb, sinks, db, geo, wsb, G, p.lists, lstride);
            } else if (fa_gg(G) == 2) {
                if (wide) hipLaunchKernelGGL((k_fattn_dec128<2, 0, false, 8, 0, 64>), grid, dim3(512), 0, s, qb, kb, vb, mb, sinks, db, geo, wsb, G, nullptr, 0);
                else hipLaunchKernelGGL((k_fattn_dec128<2, 0, false, 4, 0, 64>), grid, dim3(256), 0, s, qb, kb, vb, mb, sinks, db, geo, wsb, G, nullptr, 0);
            } else if (fa_gg(G) == 4) {
                if (wide) hipLaunchKernelGGL((k_fattn_dec128<4, 0, false, 8, 0, 64>), grid, dim3(512), 0, s, qb, kb, vb, mb, sinks, db, geo, wsb, G, nullptr, 0);
                else hipLaunchKernelGGL((k_fattn_dec128<4, 0, false, 4, 0, 64>), grid, dim3(256), 0, s, qb, kb, vb, mb, sinks, db, geo, wsb, G, nullptr, 0);
            } else {
                if (wide) hipLaunchKernelGGL((k_fattn_dec128<8, 0, false, 8, 0, 64>), grid, dim3(512), 0, s, qb, kb, vb, mb, sinks, db, geo, wsb, G, nullptr, 0);
                else hipLaunchKernelGGL((k_fattn_dec128<8, 0, false, 4, 0, 64>), grid, dim3(256), 0, s, qb, kb, vb, mb, sinks, db, geo, wsb, G, nullptr, 0);
            }
        }
        if (geo.n_splits > 1) launch_flash_attn_combine(s, 64, ws, sinks, dst, geo.n_q, geo.n_head, (int) q.ne[3], geo.n_splits, nullptr);
        return;
    }
#define FA_CASE(DD, GG) \
    if (D == DD && G == GG) { launch_fa<DD, GG>(s, q, k, v, mk, sinks, dst, geo, ws); return; }
    FA_CASE(64, 1) FA_CASE(64, 2) FA_CASE(64, 3) FA_CASE(64, 4) FA_CASE(64, 5) FA_CASE(64, 6) FA_CASE(64, 7) FA_CASE(64, 8)
    FA_CASE(128, 1) FA_CASE(128, 2) FA_CASE(128, 3) FA_CASE(128, 4) FA_CASE(128, 5) FA_CASE(128, 6) FA_CASE(128, 7) FA_CASE(128, 8)
#undef FA_CASE
    MI_ERR("launch_flash_attn: unsupported head_dim %d / group %d", D, G);
    abort();
}

#ifdef FA_STAMP
extern "C" __attribute__((visibility("default"))) void mi355x_fa_stamps_dump(int n_wg) {
    static unsigned long long h[1024][8];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_fa_stamps), sizeof(h)) != hipSuccess) { printf("fa_stamps: copy failed\n"); return; }
    n_wg = std::min(n_wg, 1024);
    static const char * what[7] = {"entry", "K/V (+ mask) loads issued", "query arrived (vmcnt 8)", "first trip: K arrived, dots done", "all trips: P.V done", "waves met (barrier)", "records stored (issued)"};
    unsigned long long t00 = ~0ull;
    for (int w = 0; w < n_wg; ++w) t00 = std::min(t00, h[w][0]);
    for (int i = 0; i < 7; ++i) {
        std::vector<double> d, d0;
        for (int w = 0; w < n_wg; ++w) { d.push_back((double) (h[w][i] - h[w][0])); d0.push_back((double) (h[w][i] - t00)); }
        std::sort(d.begin(), d.end()); std::sort(d0.begin(), d0.end());
        printf("fa_stamp %d %-34s: since own entry median %8.0f p90 %8.0f ticks | since the first workgroup's entry median %8.0f max %8.0f ticks\n", i, what[i], d[d.size() / 2], d[d.size() * 9 / 10], d0[d0.size() / 2], d0.back());
    }
}
#endif

MI_TU_TOUCH(fattn)

}  // namespace mi355x
