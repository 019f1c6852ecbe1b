// fattn_mma.hip — FLASH_ATTN_EXT for batches of query tokens (prefill) on the gfx950 matrix cores.
//
// Same contract as fattn.hip (restates ggml_compute_forward_flash_attn_ext_f16: Q rounded to f16, s = (K·Q)*scale +
// mask, online softmax, f32 accumulation of P·V), but a workgroup owns 64 query tokens of one head and walks the KV
// range in 64-position tiles staged through LDS, so K/V are read once per 64 queries instead of once per query.
//
// Layout (wave64, v_mfma_f32_32x32x16_f16), chosen so that NO cross-lane data movement is needed between the two GEMMs:
//   * scores are computed TRANSPOSED, S^T = K·Q^T: A = K tile rows (LDS, 16-byte reads), B = Q^T fragments kept in
//     registers for the whole kernel.  In the accumulator a lane then holds one query (column = lane & 31) and 16 of the
//     32 KV positions of the tile — the row max / row sum of the online softmax are lane-local plus ONE exchange with
//     lane ^ 32, and the running max, sum and the rescale factor are per-lane scalars.
//   * O^T = V^T·P^T: the B operand must hold 8 KV positions of one query per lane — exactly what the lane already has
//     (registers 8s..8s+7 of the score tile), in the order {4h..4h+3, 8+4h..8+4h+3} (h = lane >> 5).  The contraction
//     index is only a label, so instead of permuting P the A operand reads V^T in that same order: V is transposed while
//     it is staged (V^T[d][kv] in LDS), and a lane reads two 8-byte runs of the row of its output dimension.
//   * O^T accumulators again have the query in the column, so alpha-rescaling and the final 1/l are per-lane scalars and
//     the result is written as float4 runs along d.
// P is rounded to f16 for the second GEMM (the CPU path rounds the whole accumulation to f16 at every step).
#include <algorithm>

#include "dev_util.h"
#include "kernels.h"

namespace mi355x {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));

struct fam_geom {
    int n_q, n_head, n_kv_head, n_kv, has_mask, n_splits;
    float scale;
};

// NW waves of 32 queries each share one staged K/V tile (NW = 4: 128 queries per workgroup, half the staging work per query)
template <int D, int NW>
__global__ void __launch_bounds__(NW * 64, 2) k_fattn_mma(const tdesc q, const tdesc k, const tdesc v, const tdesc mask, const tdesc dst, const fam_geom geo, float * __restrict__ ws,
                                                       const uint8_t * __restrict__ vis) {
    constexpr int BKV = 64;
    constexpr int KS = (D + 8) * 2;    // K tile row stride (bytes): odd multiple of 16 -> conflict-free ds_read_b128
    constexpr int VS = (BKV + 4) * 2;  // V^T tile row stride (bytes): 34 dwords -> conflict-free ds_read_b64 across 32 rows
    constexpr int NS = D / 16;         // MFMA k-steps over the head dimension
    constexpr int ND = D / 32;         // output d-tiles
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char * Kt = smem;                  // [BKV][D+8] f16
    char * Vt = smem + BKV * KS;       // [D][BKV+4] f16 (transposed)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, kg = lane >> 5;
    const int h = blockIdx.y, bat = blockIdx.z / geo.n_splits, split = blockIdx.z % geo.n_splits;
    // KV range of this split (whole 64-position tiles); few query tiles x heads do not fill 256 CUs on their own — e.g. 32
    // continuous-batching sequences decoding one token each over a 17k-cell unified cache — so the KV range is split as in
    // the decode kernel and k_fattn_combine merges the (m, l, O) records
    const int tiles = (geo.n_kv + BKV - 1) / BKV, tps = (tiles + geo.n_splits - 1) / geo.n_splits;
    const int kv_begin = split * tps * BKV, kv_end = min(geo.n_kv, kv_begin + tps * BKV);
    const int kvh = h / (geo.n_head / geo.n_kv_head);
    const int qi = blockIdx.x * (NW * 32) + wave * 32 + fr;  // this lane's query token
    const int qrow = min(qi, geo.n_q - 1);
    const int64_t kb = bat / (q.ne[3] / k.ne[3]), vb = bat / (q.ne[3] / v.ne[3]);

    // Q^T fragments: step s needs head dims 16s + 8kg .. +7 of this lane's query (Q -> f16, as the CPU's q_to_vec_dot)
    half8 qf[NS];
    {
        const float * qp = (const float *) (q.data + (int64_t) qrow * q.nb[1] + (int64_t) h * q.nb[2] + (int64_t) bat * q.nb[3]);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float4 a = *(const float4 *) (qp + 16 * s + 8 * kg), b = *(const float4 *) (qp + 16 * s + 8 * kg + 4);
            qf[s] = (half8){(_Float16) a.x, (_Float16) a.y, (_Float16) a.z, (_Float16) a.w, (_Float16) b.x, (_Float16) b.y, (_Float16) b.z, (_Float16) b.w};
        }
    }
    const uint16_t * mrow = geo.has_mask ? (const uint16_t *) (mask.data + (int64_t) qrow * mask.nb[1] + (int64_t) (bat % mask.ne[3]) * mask.nb[3]) : nullptr;
    const char * kbase = k.data + (int64_t) kvh * k.nb[2] + kb * k.nb[3];
    const char * vbase = v.data + (int64_t) kvh * v.nb[2] + vb * v.nb[3];

    constexpr float LOG2E = 1.4426950408889634f;
    const float sc2 = geo.scale * LOG2E;
    float16v O[ND];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[dt][r] = 0.0f;
    const float16v zero = O[0];
    float m = -INFINITY, l = 0.0f;

    // staging roles.  K: thread -> KV row and its slice of the head dimension (NW parts).  V (round 4): thread -> a block of 4 KV rows x 8 head
    // dims, transposed in registers (16 v_perm) and written as 8 ds_write_b64 of V^T — the first version wrote V^T with 32 ds_write_b16 per thread.
    constexpr int T = NW * 64;
    const int srow = tid / NW, spart = tid % NW;
    constexpr int CH = D / (8 * NW);  // 16-byte chunks per thread per tile
    constexpr int NBLK = 16 * (D / 8), VB = (NBLK + T - 1) / T;

    // vis (k_fattn_vis_scan, once per graph run — the mask is the same tensor in every layer): byte [q tile of 32][kv tile of 64]:
    // 0 = none of the 32 queries sees any of the 64 cells, 2 = every mask value of the tile is +0.0 (all visible, no bias: the kernel does
    // not read the mask at all), 1 = anything else.  A causal prompt chunk skips the tiles above its diagonal and reads the mask only ON the
    // diagonal; a continuous batch's prompt chunks (each sequence sees only its own cells of the unified cache) skip almost everything.
    const int n_qt = (geo.n_q + 31) / 32;
    const int kt_begin = kv_begin / BKV, kt_end = (kv_end + BKV - 1) / BKV;
    // the states of this split's tiles, all NW query tiles packed into one byte (2 bits each), copied to LDS once: the walk below decides
    // from LDS (a global byte load per decision put one L2 round trip per tile in front of the next tile's requests)
    uint8_t * const vt = (uint8_t *) (smem + BKV * KS + D * VS);
    if (vis) {
        const uint8_t * visrow = vis + (int64_t) (blockIdx.x * NW) * tiles;
        for (int i = kt_begin + tid; i < kt_end; i += T) {
            uint32_t b = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w)
                if ((int) blockIdx.x * NW + w < n_qt) b |= (uint32_t) (visrow[(int64_t) w * tiles + i] & 3) << (2 * w);
            vt[i - kt_begin] = (uint8_t) b;
        }
        __syncthreads();
    }
    auto next_tile = [&](int kt) __attribute__((always_inline)) {
        if (vis)
            while (kt < kt_end && __builtin_amdgcn_readfirstlane((int) vt[kt - kt_begin]) == 0) ++kt;  // (readfirstlane: the tile index stays in an SGPR)
        return kt;
    };
    // K / V of a tile travel global -> registers -> LDS; the NEXT visible tile's loads are issued as soon as this tile's registers have been
    // written to LDS, so they fly during the two GEMMs and the softmax (the first version loaded, waited and wrote inside the two barriers:
    // one exposed L2 / HBM round trip per tile, ~15 k clocks per tile and wave against 1 k of matrix pipe)
    typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));  // (a struct uint4 copied whole to LDS keeps the array in scratch)
    u32x4v kr[CH];
    uint4 vr[VB][4];
    auto load_tile = [&](const int kt) __attribute__((always_inline)) {
        const int kv0 = kt * BKV;
        {
            const int pos = min(kv0 + srow, geo.n_kv - 1);
            const u32x4v * kp = (const u32x4v *) (kbase + (int64_t) pos * k.nb[1]) + spart * CH;
#pragma unroll
            for (int i = 0; i < CH; ++i) kr[i] = kp[i];
        }
#pragma unroll
        for (int j = 0; j < VB; ++j) {
            const int b = min(tid + j * T, NBLK - 1), kvq = b / (D / 8), dch = b % (D / 8);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pos = min(kv0 + 4 * kvq + r, geo.n_kv - 1);
                vr[j][r] = *((const uint4 *) (vbase + (int64_t) pos * v.nb[1]) + dch);
            }
        }
    };
    auto store_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < CH; ++i) *(u32x4v *) (Kt + srow * KS + (spart * CH + i) * 16) = kr[i];
#pragma unroll
        for (int j = 0; j < VB; ++j) {
            const int b = tid + j * T;
            if (VB * T != NBLK && b >= NBLK) break;
            const int kvq = b / (D / 8), dch = b % (D / 8);
            const uint32_t r0[4] = {vr[j][0].x, vr[j][0].y, vr[j][0].z, vr[j][0].w}, r1[4] = {vr[j][1].x, vr[j][1].y, vr[j][1].z, vr[j][1].w};
            const uint32_t r2[4] = {vr[j][2].x, vr[j][2].y, vr[j][2].z, vr[j][2].w}, r3[4] = {vr[j][3].x, vr[j][3].y, vr[j][3].z, vr[j][3].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {  // dword e of a row holds head dims 2e, 2e + 1
                uint2 lo, hi;
                lo.x = __builtin_amdgcn_perm(r1[e], r0[e], 0x05040100u);  // {row0.lo16, row1.lo16}
                lo.y = __builtin_amdgcn_perm(r3[e], r2[e], 0x05040100u);
                hi.x = __builtin_amdgcn_perm(r1[e], r0[e], 0x07060302u);  // {row0.hi16, row1.hi16}
                hi.y = __builtin_amdgcn_perm(r3[e], r2[e], 0x07060302u);
                // (groups of 4 KV positions XOR-swizzled by the row's 32-dim tile: the 16 lanes that write one KV group — rows 8 dims apart,
                // 16 banks apart mod 64 — would hit 4 bank pairs; the readers' tile index dt is a compile-time constant)
                const int kvs = 8 * (kvq ^ (2 * (dch >> 2)));
                *(uint2 *) (Vt + (8 * dch + 2 * e) * VS + kvs) = lo;
                *(uint2 *) (Vt + (8 * dch + 2 * e + 1) * VS + kvs) = hi;
            }
        }
    };

    int kt = next_tile(kt_begin);
    if (kt < kt_end) load_tile(kt);
    while (kt < kt_end) {
        const int kv0 = kt * BKV;
        const int mine = !vis ? 1 : (int) __builtin_amdgcn_readfirstlane((vt[kt - kt_begin] >> (2 * wave)) & 3);
        const bool tail = kv0 + BKV > geo.n_kv;
        __syncthreads();  // nobody reads the previous tile any more
        store_tile();
        // the mask words of this tile (only where the tile is neither hidden nor plainly visible), requested BEFORE the next tile's K / V:
        // vmcnt retires in order, so waiting for them later leaves the younger K / V requests in flight
        uint2 mw[2][4] = {};
        const bool use_mask = geo.has_mask != 0 && mine == 1;  // (wave-uniform)
        if (use_mask) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) mw[t][g4] = *(const uint2 *) (mrow + min(kv0 + 32 * t + 8 * g4 + 4 * kg, geo.n_kv - 4));
        }
        const int kt_next = next_tile(kt + 1);
        if (kt_next < kt_end) load_tile(kt_next);
        // (a bare barrier behind the LDS writes: __syncthreads() would also wait vmcnt(0), i.e. for the loads just issued)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        kt = kt_next;
        if (mine == 0) continue;  // none of this wave's queries sees the tile (the barriers are at the loop top)

        // ---- S^T = K·Q^T for the two 32-position halves of the tile
        float16v S[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            S[t] = zero;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const half8 a = *(const half8 *) (Kt + (32 * t + fr) * KS + (16 * s + 8 * kg) * 2);
                S[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, qf[s], S[t], 0, 0, 0);
            }
        }
        // ---- scale, mask, online softmax (lane = one query; registers = 32 of the tile's 64 KV positions)
        float mx = -INFINITY;
        if (use_mask || tail) {  // (wave-uniform) a tile on the diagonal / with a bias / at the end of the cache
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int p0 = kv0 + 32 * t + 8 * g4 + 4 * kg;  // four consecutive KV positions: registers 4*g4 .. 4*g4+3
                    float mv[4] = {0.f, 0.f, 0.f, 0.f};
                    if (use_mask) {
                        mv[0] = h2f((uint16_t) (mw[t][g4].x & 0xFFFF)); mv[1] = h2f((uint16_t) (mw[t][g4].x >> 16));
                        mv[2] = h2f((uint16_t) (mw[t][g4].y & 0xFFFF)); mv[3] = h2f((uint16_t) (mw[t][g4].y >> 16));
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float sv = fmaf(S[t][4 * g4 + e], sc2, mv[e] * LOG2E);  // log2 domain: p = 2^(s - m) is ONE v_exp_f32
                        if (p0 + e >= geo.n_kv) sv = -INFINITY;
                        S[t][4 * g4 + e] = sv;
                        mx = fmaxf(mx, sv);
                    }
                }
            }
        } else {  // every cell visible to every query, no bias: one multiply per score
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float sv = S[t][r] * sc2;  // (= fmaf(s, sc2, +0.0), what the masked path computes for a visible cell)
                    S[t][r] = sv;
                    mx = fmaxf(mx, sv);
                }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m, mx);
        if (__all(m_new == -INFINITY)) continue;  // nothing visible to any query of this wave yet (barriers are at the loop top)
        const float mref = m_new == -INFINITY ? 0.0f : m_new;  // this query sees nothing yet: every exponent below is -inf -> 0
        const float alpha = __builtin_amdgcn_exp2f(m - mref);
        float rs = 0.0f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float sv = S[t][r];
                const float p = __builtin_amdgcn_exp2f(sv - mref);
                S[t][r] = p;
                rs += p;
            }
        rs += __shfl_xor(rs, 32, 64);
        l = l * alpha + rs;
        m = m_new;
        if (!__all(alpha == 1.0f)) {  // (wave-uniform; once the running maxima have settled — most tiles of a long row — nothing is rescaled)
#pragma unroll
            for (int dt = 0; dt < ND; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) O[dt][r] *= alpha;
        }
        // ---- O^T += V^T·P^T
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                half8 pb;
#pragma unroll
                for (int u = 0; u < 8; ++u) pb[u] = (_Float16) S[t][8 * s2 + u];
                const int kvg = 8 * t + 4 * s2 + kg;  // this half's first group of 4 KV positions inside a V^T row (the second: + 2)
#pragma unroll
                for (int dt = 0; dt < ND; ++dt) {
                    const char * vrow = Vt + (32 * dt + fr) * VS;
                    const half4v a0 = *(const half4v *) (vrow + 8 * (kvg ^ (2 * dt))), a1 = *(const half4v *) (vrow + 8 * ((kvg + 2) ^ (2 * dt)));
                    const half8 a = (half8){a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                    O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, pb, O[dt], 0, 0, 0);
                }
            }
        }
    }
    if (qi < geo.n_q) {
        if (geo.n_splits == 1) {
            const float inv = 1.0f / l;
            float * out = (float *) (dst.data + (int64_t) h * dst.nb[1] + (int64_t) qi * dst.nb[2] + (int64_t) bat * dst.nb[3]);
#pragma unroll
            for (int dt = 0; dt < ND; ++dt)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int d0 = 32 * dt + 8 * g4 + 4 * kg;
                    *(float4 *) (out + d0) = make_float4(O[dt][4 * g4] * inv, O[dt][4 * g4 + 1] * inv, O[dt][4 * g4 + 2] * inv, O[dt][4 * g4 + 3] * inv);
                }
        } else {  // partial record (O relative to m, m, l) in the layout k_fattn_combine reads
            float * rec = ws + ((((int64_t) bat * geo.n_q + qi) * geo.n_head + h) * geo.n_splits + split) * (D + 2);
#pragma unroll
            for (int dt = 0; dt < ND; ++dt)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int d0 = 32 * dt + 8 * g4 + 4 * kg;
#pragma unroll
                    for (int e = 0; e < 4; ++e) rec[d0 + e] = O[dt][4 * g4 + e];
                }
            if (kg == 0) {
                rec[D] = m * (1.0f / LOG2E);  // the combine pass works in the natural-log domain
                rec[D + 1] = l;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ the NON-FLASH chain of a prompt micro-batch
// llama-box runs with -fa off unless asked (engine_param.hpp:772-779): a prompt chunk then reaches the backend as
//   kq = MUL_MAT(K view, q) -> p = SOFT_MAX(kq, mask f32, scale) -> MUL_MAT(V^T view, p)
// — three dense launches over a [n_kv, tokens, heads] f32 score matrix (134 MB for 512 tokens x 2048 cells x 32 heads: 54 + 78 + 93 us per layer).
// The same arithmetic on the tiling of k_fattn_mma, in two passes over the K tiles so that the softmax is ggml_compute_forward_soft_max_f32's and
// not an online one: PASS 1 leaves (max, sum of e^(w - max)) per (query, head, KV split); PASS 2 combines those into the row's maximum and sum,
// recomputes the scores, takes p = e^(w - max) * (1 / sum) — rounded to f16 where the CPU's second MUL_MAT rounds its src1 — and multiplies with
// the TRANSPOSED V cache, whose rows need no transposition on the way into LDS.  Partial outputs of KV splits simply add (k_fattn_combine_rows
// with coefficients 1).  llama-box's zero-sum guard (ggml-cpu.patch:5-15) and the NaN row of a query that sees nothing are kept.
template <int D, int NW, int PASS>
__global__ void __launch_bounds__(NW * 64, 2) k_attn_nf_mma(const tdesc q, const tdesc k, const tdesc vt, const tdesc mask, const tdesc dst, const fam_geom geo, float * __restrict__ stats,
                                                         float * __restrict__ ws, const uint8_t * __restrict__ vis) {
    constexpr int BKV = 64;
    constexpr int KS = (D + 8) * 2, VS = (BKV + 4) * 2, NS = D / 16, ND = D / 32;
    constexpr int T = NW * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char * Kt = smem;
    char * Vt = smem + BKV * KS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, kg = lane >> 5;
    const int h = blockIdx.y, bat = blockIdx.z / geo.n_splits, split = blockIdx.z % geo.n_splits;
    const int tiles = (geo.n_kv + BKV - 1) / BKV, tps = (tiles + geo.n_splits - 1) / geo.n_splits;
    const int kv_begin = split * tps * BKV, kv_end = min(geo.n_kv, kv_begin + tps * BKV);
    const int kvh = h / (geo.n_head / geo.n_kv_head);
    const int qi = blockIdx.x * (NW * 32) + wave * 32 + fr;
    const int qrow = min(qi, geo.n_q - 1);
    const int64_t kb = bat / (q.ne[3] / k.ne[3]), vb = bat / (q.ne[3] / vt.ne[3]);
    half8 qf[NS];
    {
        const float * qp = (const float *) (q.data + (int64_t) qrow * q.nb[1] + (int64_t) h * q.nb[2] + (int64_t) bat * q.nb[3]);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float4 a = *(const float4 *) (qp + 16 * s + 8 * kg), b = *(const float4 *) (qp + 16 * s + 8 * kg + 4);
            qf[s] = (half8){(_Float16) a.x, (_Float16) a.y, (_Float16) a.z, (_Float16) a.w, (_Float16) b.x, (_Float16) b.y, (_Float16) b.z, (_Float16) b.w};
        }
    }
    const float * mrow = (const float *) (mask.data + (int64_t) qrow * mask.nb[1] + (int64_t) (bat % mask.ne[3]) * mask.nb[3]);
    const char * kbase = k.data + (int64_t) kvh * k.nb[2] + kb * k.nb[3];
    const char * vtbase = vt.data + (int64_t) kvh * vt.nb[2] + vb * vt.nb[3];  // row d of the head at + d * vt.nb[1], cells contiguous
    constexpr float LOG2E = 1.4426950408889634f;
    const int64_t srow_id = ((int64_t) bat * geo.n_q + qrow) * geo.n_head + h;  // this lane's (query, head) row of the statistics / records

    // PASS 2: the row's maximum and sum from the splits' statistics (ggml: max over the row, sum of expf(w - max), 1 / sum; zero-sum guard)
    float m_g = -INFINITY, inv = 0.0f;
    if constexpr (PASS == 2) {
        const float * st = stats + srow_id * geo.n_splits * 2;
        for (int u = 0; u < geo.n_splits; ++u) m_g = fmaxf(m_g, st[2 * u]);
        double sum = 0.0;
        for (int u = 0; u < geo.n_splits; ++u) {
            const float mu = st[2 * u];
            if (mu != -INFINITY) sum += (double) st[2 * u + 1] * (double) __builtin_amdgcn_exp2f((mu - m_g) * LOG2E);
        }
        if (m_g == -INFINITY) sum = __builtin_nan("");  // expf(-inf - -inf) in every cell
        if (isnan(sum) || sum == 0.0) sum = -INFINITY;
        inv = (float) (1.0 / sum);
    }

    float16v O[ND];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[dt][r] = 0.0f;
    const float16v zero = O[0];
    float m = -INFINITY, l = 0.0f;

    const int srow = tid / NW, spart = tid % NW;
    constexpr int CH = D / (8 * NW);
    constexpr int VCH = D * 8 / T;  // 16-byte chunks of the V^T tile (D rows x 8 chunks of 8 cells) per thread
    const int n_qt = (geo.n_q + 31) / 32;
    const int kt_begin = kv_begin / BKV, kt_end = (kv_end + BKV - 1) / BKV;
    uint8_t * const vst = (uint8_t *) (smem + BKV * KS + D * VS);
    if (vis) {
        const uint8_t * visrow = vis + (int64_t) (blockIdx.x * NW) * tiles;
        for (int i = kt_begin + tid; i < kt_end; i += T) {
            uint32_t b = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w)
                if ((int) blockIdx.x * NW + w < n_qt) b |= (uint32_t) (visrow[(int64_t) w * tiles + i] & 3) << (2 * w);
            vst[i - kt_begin] = (uint8_t) b;
        }
        __syncthreads();
    }
    auto next_tile = [&](int kt) __attribute__((always_inline)) {
        if (vis)
            while (kt < kt_end && __builtin_amdgcn_readfirstlane((int) vst[kt - kt_begin]) == 0) ++kt;
        return kt;
    };
    typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
    u32x4v kr[CH];
    u32x4v vr[PASS == 2 ? VCH : 1];
    auto load_tile = [&](const int kt) __attribute__((always_inline)) {
        const int kv0 = kt * BKV;
        const int pos = min(kv0 + srow, geo.n_kv - 1);
        const u32x4v * kp = (const u32x4v *) (kbase + (int64_t) pos * k.nb[1]) + spart * CH;
#pragma unroll
        for (int i = 0; i < CH; ++i) kr[i] = kp[i];
        if constexpr (PASS == 2) {
#pragma unroll
            for (int j = 0; j < VCH; ++j) {
                const int c = tid + j * T, drow = c >> 3, kvc = c & 7;
                // (cells past n_kv belong to the cache but not to this graph's view: they are given p = 0, and 0 x NaN must not happen)
                const bool in = kv0 + 8 * kvc + 7 < geo.n_kv;
                vr[j] = in ? *(const u32x4v *) (vtbase + (int64_t) drow * vt.nb[1] + (int64_t) (kv0 + 8 * kvc) * 2) : (u32x4v){0u, 0u, 0u, 0u};
            }
        }
    };
    auto store_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < CH; ++i) *(u32x4v *) (Kt + srow * KS + (spart * CH + i) * 16) = kr[i];
        if constexpr (PASS == 2) {
#pragma unroll
            for (int j = 0; j < VCH; ++j) {
                const int c = tid + j * T, drow = c >> 3, kvc = c & 7, sw = 2 * (drow >> 5);  // (the swizzle of k_fattn_mma's V^T image)
                *(uint2 *) (Vt + drow * VS + 8 * ((2 * kvc) ^ sw)) = make_uint2(vr[j][0], vr[j][1]);
                *(uint2 *) (Vt + drow * VS + 8 * ((2 * kvc + 1) ^ sw)) = make_uint2(vr[j][2], vr[j][3]);
            }
        }
    };

    int kt = next_tile(kt_begin);
    if (kt < kt_end) load_tile(kt);
    while (kt < kt_end) {
        const int kv0 = kt * BKV;
        const int mine = !vis ? 1 : (int) __builtin_amdgcn_readfirstlane((vst[kt - kt_begin] >> (2 * wave)) & 3);
        const bool tail = kv0 + BKV > geo.n_kv;
        __syncthreads();
        store_tile();
        // (64-query workgroups — batches below 256 tokens — hold twice the staging registers per thread: their mask words are read where they are used,
        // behind the barrier, instead of a tile ahead)
        constexpr bool MASK_EARLY = NW == 4;
        float4 mw[2][4];
        const bool use_mask = mine == 1;
        if (MASK_EARLY && use_mask) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) mw[t][g4] = *(const float4 *) (mrow + min(kv0 + 32 * t + 8 * g4 + 4 * kg, geo.n_kv - 4));
        }
        const int kt_next = next_tile(kt + 1);
        if (kt_next < kt_end) load_tile(kt_next);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        kt = kt_next;
        if (mine == 0) continue;
        float16v S[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            S[t] = zero;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const half8 a = *(const half8 *) (Kt + (32 * t + fr) * KS + (16 * s + 8 * kg) * 2);
                S[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, qf[s], S[t], 0, 0, 0);
            }
        }
        // w = kq * scale + mask (two roundings, as SOFT_MAX computes it)
        if (use_mask || tail) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int p0 = kv0 + 32 * t + 8 * g4 + 4 * kg;
                    if (!MASK_EARLY && use_mask) mw[t][g4] = *(const float4 *) (mrow + min(p0, geo.n_kv - 4));
                    const float mv[4] = {use_mask ? mw[t][g4].x : 0.0f, use_mask ? mw[t][g4].y : 0.0f, use_mask ? mw[t][g4].z : 0.0f, use_mask ? mw[t][g4].w : 0.0f};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float w = S[t][4 * g4 + e] * geo.scale + mv[e];
                        if (p0 + e >= geo.n_kv) w = -INFINITY;
                        S[t][4 * g4 + e] = w;
                    }
                }
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) S[t][r] = S[t][r] * geo.scale;
        }
        if constexpr (PASS == 1) {
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, S[t][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m, mx);
            if (__all(m_new == -INFINITY)) continue;
            const float mref = m_new == -INFINITY ? 0.0f : m_new;
            const float alpha = __builtin_amdgcn_exp2f((m - mref) * LOG2E);
            float rs = 0.0f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) rs += __builtin_amdgcn_exp2f((S[t][r] - mref) * LOG2E);
            rs += __shfl_xor(rs, 32, 64);
            l = l * alpha + rs;
            m = m_new;
        } else {
            if (__all(m_g == -INFINITY)) continue;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    half8 pb;
#pragma unroll
                    for (int u = 0; u < 8; ++u) pb[u] = (_Float16) (__builtin_amdgcn_exp2f((S[t][8 * s2 + u] - m_g) * LOG2E) * inv);
                    const int kvg = 8 * t + 4 * s2 + kg;
#pragma unroll
                    for (int dt = 0; dt < ND; ++dt) {
                        const char * vrow = Vt + (32 * dt + fr) * VS;
                        const half4v a0 = *(const half4v *) (vrow + 8 * (kvg ^ (2 * dt))), a1 = *(const half4v *) (vrow + 8 * ((kvg + 2) ^ (2 * dt)));
                        const half8 a = (half8){a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                        O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, pb, O[dt], 0, 0, 0);
                    }
                }
            }
        }
    }
    if (qi >= geo.n_q) return;
    if constexpr (PASS == 1) {
        if (kg == 0) {
            float * st = stats + (srow_id * geo.n_splits + split) * 2;
            st[0] = m;
            st[1] = l;
        }
    } else {
        const bool nan_row = m_g == -INFINITY;  // the query sees nothing: the CPU's row is NaN throughout
        if (geo.n_splits == 1) {
            float * out = (float *) (dst.data + (int64_t) h * dst.nb[1] + (int64_t) qi * dst.nb[2] + (int64_t) bat * dst.nb[3]);
#pragma unroll
            for (int dt = 0; dt < ND; ++dt)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int d0 = 32 * dt + 8 * g4 + 4 * kg;
                    const float nn = __builtin_nanf("");
                    *(float4 *) (out + d0) = nan_row ? make_float4(nn, nn, nn, nn) : make_float4(O[dt][4 * g4], O[dt][4 * g4 + 1], O[dt][4 * g4 + 2], O[dt][4 * g4 + 3]);
                }
        } else {  // a record k_fattn_combine_rows adds to the other splits': coefficient e^(0 - 0) = 1 each, sum of the "l" fields = 1
            float * rec = ws + (srow_id * geo.n_splits + split) * (D + 2);
#pragma unroll
            for (int dt = 0; dt < ND; ++dt)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int d0 = 32 * dt + 8 * g4 + 4 * kg;
#pragma unroll
                    for (int e = 0; e < 4; ++e) rec[d0 + e] = nan_row ? __builtin_nanf("") : O[dt][4 * g4 + e];
                }
            if (kg == 0) {
                rec[D] = 0.0f;
                rec[D + 1] = split == 0 ? 1.0f : 0.0f;
            }
        }
    }
}

// returns false when this variant does not apply (caller falls back to the split-KV kernel)
int fattn_mma_pick_splits(const tdesc & q, const tdesc & k) {
    const int64_t qt = q.ne[1] >= 256 ? 128 : 64;
    const int64_t wgs = ((q.ne[1] + qt - 1) / qt) * q.ne[2] * q.ne[3];
    const int64_t tiles = (k.ne[1] + 63) / 64;
    int64_t want = (512 + wgs - 1) / wgs;  // ~2 workgroups per CU
    want = std::max<int64_t>(1, std::min<int64_t>(want, std::min<int64_t>(64, tiles / 4)));
    return (int) want;
}
// a wave per (32-query tile, 64-cell tile): lane = (mask row, half of the 64 cells), 64 bytes each — coalesced, where the first version
// walked 32 rows per THREAD (86 us per micro-batch of 512 x 2048).  Result: 0 = nothing visible, 2 = every value +0.0, 1 = otherwise.
template <bool F32>  // (the mask of the non-flash path is f32: rows 16-byte aligned there)
__global__ void __launch_bounds__(256) k_fattn_vis_scan(const tdesc mask, const int n_q, const int n_kv, uint8_t * __restrict__ vis) {
    const int qt = blockIdx.x, tiles = (n_kv + 63) / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane >> 1, half = lane & 1;
    const int qi = qt * 32 + r;
    for (int kt = blockIdx.y * 4 + wave; kt < tiles; kt += gridDim.y * 4) {
        bool any = false, zero = true;
        if (qi < n_q) {
            if constexpr (F32) {
                const uint32_t * mrow = (const uint32_t *) (mask.data + (int64_t) qi * mask.nb[1]) + kt * 64 + half * 32;
#pragma unroll
                for (int c = 0; c < 32; c += 4) {
                    if (kt * 64 + half * 32 + c < n_kv) {
                        const uint4 w = *(const uint4 *) (mrow + c);
                        any = any || w.x != 0xFF800000u || w.y != 0xFF800000u || w.z != 0xFF800000u || w.w != 0xFF800000u;
                        zero = zero && (w.x | w.y | w.z | w.w) == 0u;
                    }
                }
            } else {
            const uint16_t * mrow = (const uint16_t *) (mask.data + (int64_t) qi * mask.nb[1]) + kt * 64 + half * 32;
#pragma unroll
            for (int c = 0; c < 32; c += 4) {  // n_kv % 4 == 0 and rows 8-byte aligned (launcher)
                if (kt * 64 + half * 32 + c < n_kv) {
                    const uint2 w = *(const uint2 *) (mrow + c);
                    any = any || w.x != 0xFC00FC00u || w.y != 0xFC00FC00u;
                    zero = zero && (w.x | w.y) == 0u;
                }
            }
            }
        }
        const bool w_any = __any(any), w_zero = __all(zero) && (kt + 1) * 64 <= n_kv;
        if (lane == 0) vis[(int64_t) qt * tiles + kt] = !w_any ? 0 : (w_zero ? 2 : 1);
    }
}
size_t fattn_vis_bytes(const tdesc & q, const tdesc & k) { return (size_t) ((q.ne[1] + 31) / 32) * (size_t) ((k.ne[1] + 63) / 64); }
void launch_fattn_vis_scan(hipStream_t s, const tdesc & mask, int n_q, int n_kv, uint8_t * vis) {
    const int tiles = (n_kv + 63) / 64;
    if (mask.type == GGML_TYPE_F32) hipLaunchKernelGGL(k_fattn_vis_scan<true>, dim3((unsigned) ((n_q + 31) / 32), (unsigned) std::min(64, (tiles + 3) / 4)), dim3(256), 0, s, mask, n_q, n_kv, vis);
    else hipLaunchKernelGGL(k_fattn_vis_scan<false>, dim3((unsigned) ((n_q + 31) / 32), (unsigned) std::min(64, (tiles + 3) / 4)), dim3(256), 0, s, mask, n_q, n_kv, vis);
}
int fattn_mma_min_q() {
    // up to 32 query tokens are served per token (tile-list / lane-parallel kernels): a -np 32 decode step has 32 tokens that
    // each see their own 1/32 of a unified cache, which a dense 32-query tile would multiply through in full
    static const int v = getenv("GGML_MI355X_FA_MMA_MIN_Q") ? std::max(32, atoi(getenv("GGML_MI355X_FA_MMA_MIN_Q"))) : 33;
    return v;
}
bool flash_attn_mma_applies(const tdesc & q, const tdesc & k, const tdesc * mask, const float * sinks, const tdesc & dst, const fattn_params & p) {
    const int D = (int) k.ne[0];
    if (q.ne[1] < fattn_mma_min_q() || sinks != nullptr || p.max_bias != 0.0f || p.logit_softcap != 0.0f || (D != 64 && D != 128)) return false;
    if ((k.ne[1] % 4) != 0 || (q.nb[1] % 16) || (q.nb[2] % 16) || (((uintptr_t) q.data) & 15) || (dst.nb[1] % 16) || (dst.nb[2] % 16) || (((uintptr_t) dst.data) & 15)) return false;
    if (mask && ((mask->nb[1] % 8) || (((uintptr_t) mask->data) & 7))) return false;
    return true;
}
bool launch_flash_attn_mma(hipStream_t s, const tdesc & q, const tdesc & k, const tdesc & v, const tdesc * mask, const float * sinks, const tdesc & dst,
                           const fattn_params & p, void * workspace) {
    const int D = (int) k.ne[0];
    if (!flash_attn_mma_applies(q, k, mask, sinks, dst, p)) return false;
    fam_geom geo;
    geo.n_q = (int) q.ne[1];
    geo.n_head = (int) q.ne[2];
    geo.n_kv_head = (int) k.ne[2];
    geo.n_kv = (int) k.ne[1];
    geo.has_mask = mask ? 1 : 0;
    geo.scale = p.scale;
    geo.n_splits = std::max(1, p.n_splits);
    const tdesc mk = mask ? *mask : q;
    float * ws = (float *) workspace;
    const int nw = geo.n_q >= 256 ? 4 : 2;  // 128-query workgroups when there are enough queries to keep the grid full
    dim3 grid((unsigned) ((geo.n_q + nw * 32 - 1) / (nw * 32)), (unsigned) geo.n_head, (unsigned) (q.ne[3] * geo.n_splits));
    const int tiles_all = (geo.n_kv + 63) / 64, tps = (tiles_all + geo.n_splits - 1) / geo.n_splits;
    const uint8_t * const tile_vis = tps <= 16384 ? p.tile_vis : nullptr;  // (without the states every tile is multiplied and reads its mask: correct, slower)
    const size_t vt_bytes = tile_vis ? (size_t) ((tps + 15) & ~15) : 0;  // the split's tile states (a byte per tile)
    if (D == 128) {
        const size_t lds = 64 * (128 + 8) * 2 + 128 * (64 + 4) * 2 + vt_bytes;
        if (nw == 4) hipLaunchKernelGGL((k_fattn_mma<128, 4>), grid, dim3(256), lds, s, q, k, v, mk, dst, geo, ws, tile_vis);
        else hipLaunchKernelGGL((k_fattn_mma<128, 2>), grid, dim3(128), lds, s, q, k, v, mk, dst, geo, ws, tile_vis);
    } else {
        const size_t lds = 64 * (64 + 8) * 2 + 64 * (64 + 4) * 2 + vt_bytes;
        if (nw == 4) hipLaunchKernelGGL((k_fattn_mma<64, 4>), grid, dim3(256), lds, s, q, k, v, mk, dst, geo, ws, tile_vis);
        else hipLaunchKernelGGL((k_fattn_mma<64, 2>), grid, dim3(128), lds, s, q, k, v, mk, dst, geo, ws, tile_vis);
    }
    if (geo.n_splits > 1) launch_flash_attn_combine(s, (int) k.ne[0], ws, sinks, dst, (int) q.ne[1], (int) q.ne[2], (int) q.ne[3], geo.n_splits, p.q8_out);
    return true;
}

// ---- the non-flash chain of a prompt micro-batch (k_attn_nf_mma): applicability, workspace, launch
bool attn_nf_mma_applies(const tdesc & q, const tdesc & k, const tdesc & vt, const tdesc & mask) {
    static const bool on = !getenv("GGML_MI355X_ATTN_NF_MMA") || atoi(getenv("GGML_MI355X_ATTN_NF_MMA")) != 0;
    const int64_t D = k.ne[0];
    if (!on || D != 128 || q.ne[1] < fattn_mma_min_q() || q.ne[3] != 1 || k.ne[3] != 1 || vt.ne[3] != 1 || k.type != GGML_TYPE_F16 || vt.type != GGML_TYPE_F16 || q.type != GGML_TYPE_F32 ||
        mask.type != GGML_TYPE_F32)
        return false;
    if (k.ne[2] <= 0 || q.ne[2] % k.ne[2] != 0 || vt.ne[2] != k.ne[2] || vt.ne[0] != k.ne[1] || vt.ne[1] != D || (k.ne[1] % 8) != 0) return false;
    if ((q.nb[1] % 16) || (q.nb[2] % 16) || (((uintptr_t) q.data) & 15) || k.nb[0] != 2 || (k.nb[1] % 16) || (k.nb[2] % 16) || (((uintptr_t) k.data) & 15)) return false;
    if (vt.nb[0] != 2 || (vt.nb[1] % 16) || (vt.nb[2] % 16) || (((uintptr_t) vt.data) & 15) || (mask.nb[1] % 16) || (((uintptr_t) mask.data) & 15) || mask.ne[0] < k.ne[1] || mask.ne[1] < q.ne[1]) return false;
    return true;
}
size_t attn_nf_mma_ws_bytes(const tdesc & q, int n_splits) {  // [statistics: rows x splits x 2][records: rows x splits x (D + 2)]
    const size_t rows = (size_t) q.ne[1] * (size_t) q.ne[2];
    return ((rows * (size_t) n_splits * 2 * sizeof(float) + 255) & ~(size_t) 255) + (n_splits > 1 ? rows * (size_t) n_splits * (128 + 2) * sizeof(float) : 0);
}
// dst: the result as [D, head, token] strides in nb[1] (head) / nb[2] (token) — the caller passes kqv's or the CONT copy's layout that way
void launch_attn_nf_mma(hipStream_t s, const tdesc & q, const tdesc & k, const tdesc & vt, const tdesc & mask, const tdesc & dst, float scale, int n_splits, const uint8_t * tile_vis, void * workspace,
                        void * q8_out) {
    fam_geom geo;
    geo.n_q = (int) q.ne[1];
    geo.n_head = (int) q.ne[2];
    geo.n_kv_head = (int) k.ne[2];
    geo.n_kv = (int) k.ne[1];
    geo.has_mask = 1;
    geo.scale = scale;
    geo.n_splits = std::max(1, n_splits);
    const size_t rows = (size_t) q.ne[1] * (size_t) q.ne[2];
    float * stats = (float *) workspace;
    float * recs = (float *) ((char *) workspace + ((rows * (size_t) geo.n_splits * 2 * sizeof(float) + 255) & ~(size_t) 255));
    const int tiles_all = (geo.n_kv + 63) / 64, tps = (tiles_all + geo.n_splits - 1) / geo.n_splits;
    const uint8_t * const vis = tps <= 16384 ? tile_vis : nullptr;
    const size_t lds = 64 * (128 + 8) * 2 + 128 * (64 + 4) * 2 + (vis ? (size_t) ((tps + 15) & ~15) : 0);
    // (always 128-query workgroups: the 64-query form of pass 2 holds twice the staging registers per thread and spills 60 of them; a batch
    // of 64 tokens leaves two of the four waves multiplying clamped duplicates instead)
    constexpr int nw = 4;
    dim3 grid((unsigned) ((geo.n_q + nw * 32 - 1) / (nw * 32)), (unsigned) geo.n_head, (unsigned) geo.n_splits);
    hipLaunchKernelGGL((k_attn_nf_mma<128, 4, 1>), grid, dim3(256), lds, s, q, k, vt, mask, dst, geo, stats, recs, vis);
    hipLaunchKernelGGL((k_attn_nf_mma<128, 4, 2>), grid, dim3(256), lds, s, q, k, vt, mask, dst, geo, stats, recs, vis);
    if (geo.n_splits > 1) launch_flash_attn_combine(s, 128, recs, nullptr, dst, geo.n_q, geo.n_head, 1, geo.n_splits, q8_out);  // (q8_out: only where the row-parallel form applies — the caller checked)
}

MI_TU_TOUCH(fattn_mma)

}  // namespace mi355x
