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
#include "dev_util.h"
#include "kernels.h"

namespace mi355x {

struct fa_geom {
    int n_q, n_head, n_kv_head, n_kv, n_splits, has_mask;
    float scale, softcap, max_bias, m0, m1;
    uint32_t n_head_log2;
};

template <int D, int G>
__global__ void __launch_bounds__(256) k_fattn_split(const tdesc q, const tdesc k, const tdesc v, const tdesc mask, const float * __restrict__ sinks,
                                                     const tdesc dst, const fa_geom geo, float * __restrict__ ws) {
    constexpr int LPR = D / 8;     // lanes per K/V row
    constexpr int RPW = 64 / LPR;  // rows per wave-instruction
    constexpr bool SLOTS_ALL = (size_t) 4 * RPW * G * (D + 2) * sizeof(float) <= 48 * 1024;
    constexpr int NSLOT = SLOTS_ALL ? 4 * RPW : 4;
    __shared__ float sh[NSLOT][G][D + 2];
    __shared__ float coef[NSLOT][G];
    __shared__ float mtot[G];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = lane / LPR, sl = lane % LPR;
    const int split = blockIdx.x, kvh = blockIdx.y;
    const int tok = blockIdx.z % geo.n_q, bat = blockIdx.z / geo.n_q;
    const int per = (geo.n_kv + geo.n_splits - 1) / geo.n_splits;
    const int kv0 = split * per, kv1 = min(geo.n_kv, kv0 + per);
    const int64_t kb = bat / (q.ne[3] / k.ne[3]), vb = bat / (q.ne[3] / v.ne[3]);

    float qr[G][8], acc[G][8], m[G], l[G], slope[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int h = kvh * G + g;
        const float * qp = (const float *) (q.data + (int64_t) tok * q.nb[1] + (int64_t) h * q.nb[2] + (int64_t) bat * q.nb[3]) + sl * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            qr[g][i] = h2f(f2h(qp[i]));  // q_to_vec_dot: Q is converted to K's vec_dot_type (f16)
            acc[g][i] = 0.0f;
        }
        m[g] = -INFINITY;
        l[g] = 0.0f;
        slope[g] = geo.max_bias > 0.0f ? ((uint32_t) h < geo.n_head_log2 ? powf(geo.m0, (float) (h + 1)) : powf(geo.m1, (float) (2 * (h - geo.n_head_log2) + 1))) : 1.0f;
    }
    const uint16_t * mp = geo.has_mask ? (const uint16_t *) (mask.data + (int64_t) tok * mask.nb[1] + (int64_t) (bat % mask.ne[3]) * mask.nb[3]) : nullptr;
    const char * kbase = k.data + (int64_t) kvh * k.nb[2] + kb * k.nb[3] + sl * 16;
    const char * vbase = v.data + (int64_t) kvh * v.nb[2] + vb * v.nb[3] + sl * 16;

    // NG independent row groups per trip; the NEXT trip's K/V loads are issued before the current trip is consumed, so
    // the memory round trips of successive trips overlap with the softmax arithmetic
    constexpr int NG = 2;
    struct trip_regs {
        uint4 kraw[NG], vraw[NG];
        float mv[NG];
        bool inr[NG];
    };
    auto load_trip = [&](const int p0, trip_regs & t) {
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            const int p = p0 + u * 4 * RPW + sub;
            t.inr[u] = p < kv1;
            const int pc = t.inr[u] ? p : kv1 - 1;
            t.mv[u] = mp ? h2f(mp[pc]) : 0.0f;
            t.kraw[u] = *(const uint4 *) (kbase + (int64_t) pc * k.nb[1]);
            t.vraw[u] = *(const uint4 *) (vbase + (int64_t) pc * v.nb[1]);
        }
    };
    trip_regs cur;
    int p0 = kv0 + wave * RPW;
    bool have = p0 < kv1;
    if (have) load_trip(p0, cur);
    while (have) {
        const int pn = p0 + NG * 4 * RPW;
        const bool nhave = pn < kv1;
        trip_regs nxt;
        if (nhave) load_trip(pn, nxt);
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            const uint32_t ku[4] = {cur.kraw[u].x, cur.kraw[u].y, cur.kraw[u].z, cur.kraw[u].w}, vu[4] = {cur.vraw[u].x, cur.vraw[u].y, cur.vraw[u].z, cur.vraw[u].w};
            float kf[8], vf[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                kf[2 * i] = h2f((uint16_t) (ku[i] & 0xFFFF));
                kf[2 * i + 1] = h2f((uint16_t) (ku[i] >> 16));
                vf[2 * i] = h2f((uint16_t) (vu[i] & 0xFFFF));
                vf[2 * i + 1] = h2f((uint16_t) (vu[i] >> 16));
            }
            float sc[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float t = 0.0f;
#pragma unroll
                for (int i = 0; i < 8; ++i) t = fmaf(kf[i], qr[g][i], t);
                sc[g] = t;
            }
            // finish the D-wide dot inside the LPR lanes of the row on the DPP path (LPR = 16: one DPP row; LPR = 8: half a row)
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if constexpr (LPR == 16) {
                    sc[g] = row16_sum(sc[g]);
                } else {
                    sc[g] += dpp_f32<MI_DPP_QUAD_XOR1>(sc[g]);
                    sc[g] += dpp_f32<MI_DPP_QUAD_XOR2>(sc[g]);
                    sc[g] += dpp_f32<MI_DPP_HALF_MIRROR>(sc[g]);
                }
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float sv = sc[g] * geo.scale;
                if (geo.softcap != 0.0f) sv = geo.softcap * tanhf(sv);
                const float mvs = slope[g] * cur.mv[u];
                sv += mvs;
                const bool use = cur.inr[u] && !(mvs == -INFINITY);
                if (use) {
                    float vs = 1.0f;
                    if (sv > m[g]) {
                        const float ms = expf(m[g] - sv);
                        l[g] *= ms;
#pragma unroll
                        for (int i = 0; i < 8; ++i) acc[g][i] *= ms;
                        m[g] = sv;
                    } else {
                        vs = expf(sv - m[g]);
                    }
                    l[g] += vs;
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[g][i] = fmaf(vs, vf[i], acc[g][i]);
                }
            }
        }
        cur = nxt;
        p0 = pn;
        have = nhave;
    }
    // every (wave, sub-row) partial goes to LDS as its own slot (no cross-lane shuffles: the round-1 profile showed the
    // 10 x G ds_bpermute per merge level dominating short splits); when the slots would not fit, sub-rows are first
    // merged in registers
    if constexpr (!SLOTS_ALL) {
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float mo = __shfl_xor(m[g], o, 64), lo = __shfl_xor(l[g], o, 64);
                const float mn = fmaxf(m[g], mo);
                const float ca = m[g] == -INFINITY ? 0.0f : expf(m[g] - mn);
                const float cb = mo == -INFINITY ? 0.0f : expf(mo - mn);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float ao = __shfl_xor(acc[g][i], o, 64);
                    acc[g][i] = acc[g][i] * ca + ao * cb;
                }
                l[g] = l[g] * ca + lo * cb;
                m[g] = mn;
            }
        }
    }
    if (SLOTS_ALL || sub == 0) {
        const int slot = SLOTS_ALL ? wave * RPW + sub : wave;
#pragma unroll
        for (int g = 0; g < G; ++g) {
#pragma unroll
            for (int i = 0; i < 8; ++i) sh[slot][g][sl * 8 + i] = acc[g][i];
            if (sl == 0) {
                sh[slot][g][D] = m[g];
                sh[slot][g][D + 1] = l[g];
            }
        }
    }
    __syncthreads();
    // per (slot, g): coefficient exp(m_slot - m_total), stored over the slot's m
    if (tid < NSLOT * G) {
        const int slot = tid / G, g = tid % G;
        float mt = -INFINITY;
        for (int w = 0; w < NSLOT; ++w) mt = fmaxf(mt, sh[w][g][D]);
        const float mw = sh[slot][g][D];
        coef[slot][g] = mw == -INFINITY ? 0.0f : expf(mw - mt);
        if (slot == 0) mtot[g] = mt;
    }
    __syncthreads();
    // merge the slots; one thread per (g, d)
    for (int e = tid; e < G * D; e += 256) {
        const int g = e / D, dd = e % D;
        const float mt = mtot[g];
        float a = 0.0f, lt = 0.0f;
#pragma unroll 4
        for (int w = 0; w < NSLOT; ++w) {
            const float c = coef[w][g];
            a += sh[w][g][dd] * c;
            lt += sh[w][g][D + 1] * c;
        }
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

template <int D> __global__ void __launch_bounds__(64) k_fattn_combine(const float * __restrict__ ws, const float * __restrict__ sinks, const tdesc dst, const fa_geom geo) {
    const int h = blockIdx.x, tok = blockIdx.y, bat = blockIdx.z, lane = threadIdx.x;
    const float * __restrict__ base = ws + (((int64_t) bat * geo.n_q + tok) * geo.n_head + h) * geo.n_splits * (D + 2);
    // lane s owns split s (n_splits <= 64): one round trip fetches every (m, l) pair
    const bool has = lane < geo.n_splits;
    const float ms = has ? base[(int64_t) lane * (D + 2) + D] : -INFINITY;
    const float ls = has ? base[(int64_t) lane * (D + 2) + D + 1] : 0.0f;
    float mn = wave_max(ms);
    float sink_term = 0.0f;
    if (sinks) {
        mn = fmaxf(mn, sinks[h]);
        sink_term = expf(sinks[h] - mn);
    }
    const float cs = ms == -INFINITY ? 0.0f : expf(ms - mn);
    const float lt = wave_sum(ls * cs) + sink_term;
    const float inv = 1.0f / lt;
    float * out = (float *) (dst.data + (int64_t) h * dst.nb[1] + (int64_t) tok * dst.nb[2] + (int64_t) bat * dst.nb[3]);
    for (int dd = lane; dd < D; dd += 64) {
        float a = 0.0f;
        for (int s0 = 0; s0 < geo.n_splits; s0 += 8) {
            float r[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) r[u] = (s0 + u) < geo.n_splits ? base[(int64_t) (s0 + u) * (D + 2) + dd] : 0.0f;
#pragma unroll
            for (int u = 0; u < 8; ++u) a += r[u] * __shfl(cs, (s0 + u) & 63, 64);
        }
        out[dd] = a * inv;
    }
}

int fattn_pick_splits(const tdesc & q, const tdesc & k) {
    const int64_t n_kv = k.ne[1];
    const int64_t groups = k.ne[2] * q.ne[1] * q.ne[3];
    int64_t want = (768 + groups - 1) / groups;  // ~3 workgroups per CU
    const int64_t max_by_len = std::max<int64_t>(1, n_kv / 64);
    want = std::max<int64_t>(1, std::min<int64_t>(want, std::min<int64_t>(max_by_len, 64)));
    return (int) want;
}
size_t fattn_workspace_bytes(const tdesc & q, const tdesc & v, int n_splits) {
    if (n_splits <= 1) return 0;
    return (size_t) (q.ne[1] * q.ne[3] * q.ne[2]) * (size_t) n_splits * (size_t) (v.ne[0] + 2) * sizeof(float);
}

template <int D, int G> static void launch_fa(hipStream_t s, const tdesc & q, const tdesc & k, const tdesc & v, const tdesc & mask, const float * sinks,
                                              const tdesc & dst, const fa_geom & geo, float * ws) {
    dim3 grid((unsigned) geo.n_splits, (unsigned) geo.n_kv_head, (unsigned) (geo.n_q * q.ne[3]));
    hipLaunchKernelGGL((k_fattn_split<D, G>), grid, dim3(256), 0, s, q, k, v, mask, sinks, dst, geo, ws);
    if (geo.n_splits > 1) {
        dim3 g2((unsigned) geo.n_head, (unsigned) geo.n_q, (unsigned) q.ne[3]);
        hipLaunchKernelGGL((k_fattn_combine<D>), g2, dim3(64), 0, s, ws, sinks, dst, geo);
    }
}

void launch_flash_attn(hipStream_t s, const tdesc & q, const tdesc & k, const tdesc & v, const tdesc * mask, const float * sinks, const tdesc & dst,
                       const fattn_params & p, void * workspace) {
    if (launch_flash_attn_mma(s, q, k, v, mask, sinks, dst, p)) return;
    fa_geom geo;
    geo.n_q = (int) q.ne[1];
    geo.n_head = (int) q.ne[2];
    geo.n_kv_head = (int) k.ne[2];
    geo.n_kv = (int) k.ne[1];
    geo.n_splits = p.n_splits;
    geo.has_mask = mask ? 1 : 0;
    geo.scale = p.logit_softcap != 0.0f ? p.scale / p.logit_softcap : p.scale;
    geo.softcap = p.logit_softcap;
    geo.max_bias = p.max_bias;
    geo.n_head_log2 = 1u << (uint32_t) floor(log2((double) geo.n_head));
    geo.m0 = powf(2.0f, -(p.max_bias) / (float) geo.n_head_log2);
    geo.m1 = powf(2.0f, -(p.max_bias / 2.0f) / (float) geo.n_head_log2);
    const int D = (int) k.ne[0], G = geo.n_head / geo.n_kv_head;
    const tdesc mk = mask ? *mask : q;
    float * ws = (float *) workspace;
#define FA_CASE(DD, GG) \
    if (D == DD && G == GG) { launch_fa<DD, GG>(s, q, k, v, mk, sinks, dst, geo, ws); return; }
    FA_CASE(64, 1) FA_CASE(64, 2) FA_CASE(64, 4) FA_CASE(64, 8)
    FA_CASE(128, 1) FA_CASE(128, 2) FA_CASE(128, 4) FA_CASE(128, 7) FA_CASE(128, 8)
#undef FA_CASE
    MI_ERR("launch_flash_attn: unsupported head_dim %d / group %d", D, G);
    abort();
}

}  // namespace mi355x
