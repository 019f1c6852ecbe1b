// mmq_i8.hip — batched mat-mul (3 columns and up: continuous-batching decode steps, prompt micro-batches) for Q4_K, Q5_K and
// Q6_K weights on the gfx950 INTEGER matrix cores.
//
// Same contract as mmq.hip (ggml-cpu's ggml_vec_dot_q{4,5}_K_q8_K: integer block sums on Q8_K activations, one f32
// scale-accumulate per super-block — SURVEY.md §8a row a6), but the block sums are computed by v_mfma_i32_32x32x32_i8:
//   * the activation tile is the raw int8 of the Q8_K blocks — copied global -> LDS, no conversion at all;
//   * the weight tile must carry the 6-bit sub-block scale (it changes every 32 values of K, i.e. every MFMA), and
//     sc * q does not fit int8.  The scale is therefore split into small digits, sc = 8*s1 + s0 (Q4_K: q <= 15, digits
//     <= 7, products <= 105) or sc = 16*s2 + 4*s1 + s0 (Q5_K: q <= 31, digits <= 3, products <= 93), one int8 tile
//     ("piece") per digit, one MFMA per piece, and the int32 accumulators are recombined with shifts — still exactly
//     the CPU's integers (int32 accumulation never rounds, unlike the f32 accumulation of the f16 variant which is
//     exact only below 2^24).  i8 MFMA runs at twice the f16 rate, so two pieces cost what the f16 kernel's one costs;
//   * four nibbles of a dword are multiplied by their digit with ONE v_pk_mul_lo_u16 (each byte product < 256, so no
//     carries cross bytes): the whole weight staging is ~30 VALU per 32 weights instead of ~90;
//   * the mins term keeps the f16 MFMA step of mmq.hip (one per super-block).
// LDS tiles are [row][128 B of K] with the 16-byte chunk index XOR-swizzled by (row >> 1) & 7 — conflict-free for the
// ds_read_b128 lane groups of gfx950 without padding — and DOUBLE-buffered: trip t+1 is converted and written while
// the MFMAs of trip t run, one barrier per trip.  BN = 128 (8 waves) or 64 (4 waves; used when the grid would not
// fill 256 CUs otherwise); each wave owns 32 weight rows x 64 activation columns (BM = 128), or all 64 / 32 columns of a
// small batch with the K steps of a trip shared between two wave groups (BM = 64 / 32).  Q6_K splits the PRODUCT
// (q - 32) * sc = 64 * p1 + p0 instead of the scale.  Up to three matrices over the same activations share a launch.
#include <algorithm>

#include "dev_util.h"
#include "kernels.h"
#include "mmq_args.h"

namespace mi355x {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef int int4v __attribute__((ext_vector_type(4)));
typedef int int16v __attribute__((ext_vector_type(16)));

constexpr int MI_BM = 128;
constexpr int MI_MS = (16 + 8) * 2;  // row stride of the 16-wide f16 mins / bsums tiles (48 B)

__device__ __forceinline__ uint32_t pk_mul_u16x2(const uint32_t a, const uint32_t b) {
    typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
    const ushort2v r = __builtin_bit_cast(ushort2v, a) * __builtin_bit_cast(ushort2v, b);
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pack_h2i(const int a, const int b) {
    const _Float16 x = (_Float16) a, y = (_Float16) b;
    uint16_t ux, uy;
    __builtin_memcpy(&ux, &x, 2);
    __builtin_memcpy(&uy, &y, 2);
    return (uint32_t) ux | ((uint32_t) uy << 16);
}
__device__ __forceinline__ int sw_off(const int row, const int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

#ifndef MMQ_FOLD_PACKED
#define MMQ_FOLD_PACKED 0
#endif
// BM = activation columns per workgroup.  128: each wave owns 32 rows x 64 columns and all four 32-wide K steps of a trip.
// 64 / 32 (continuous-batching decode steps, M <= 64): every wave sees ALL columns (two / one 32-column tiles) and the two wave
// groups split the K steps of a trip between them — the staging work (all threads unpack weights) is unchanged, the MFMA,
// LDS-read and fold work shrinks with the tile instead of multiplying zeros; the two partial sums meet in LDS at the end.
template <int QT, int BN, int BM = 128>
__global__ void __launch_bounds__(BN * 4, BN == 64 ? (BM == 32 ? 3 : 2) : 1) k_mmq_i8(const mmq8_args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = BN * 4;
    constexpr int NP = QT == 5 ? 3 : 2;
    constexpr int BYTES = QT == 4 ? 144 : (QT == 5 ? 176 : 210);
    constexpr bool Q6 = QT == 6;
    constexpr int TA = BN * 128, TB = BM * 128, STAGE = NP * TA + TB;
    constexpr int NBC = (BM * 8) / NT;  // 16-byte activation chunks per thread per trip (1, 2 or 4)
    constexpr int NTT = BM >= 64 ? 2 : 1;  // 32-column tiles per wave
    constexpr int KG = BM == 128 ? 1 : 2;  // wave groups sharing the K steps of a trip
    constexpr int NKS = 4 / KG;
    static_assert(NBC == 1 || NBC == 2 || NBC == 4, "activation staging layout");
    char * Am = smem + 2 * STAGE;                   // mins  [BN][24] f16
    char * Bm = Am + BN * MI_MS;                    // bsums [BM][24] f16
    float2 * dd = (float2 *) (Bm + BM * MI_MS);     // (d, dmin) per row
    float * dyv = (float *) (dd + BN);              // dy per column

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int xcd = blockIdx.x & 7, qb = blockIdx.x >> 3;
    const int panel = (qb / a.m_tiles) * 8 + xcd, mt = qb % a.m_tiles;
    if (panel >= a.n_panels) return;
    // the matrix this panel belongs to (selects, not indexing: the argument struct stays in the kernarg segment)
    const int mi = (a.n_mat > 2 && panel >= a.mat[2].panel0) ? 2 : ((a.n_mat > 1 && panel >= a.mat[1].panel0) ? 1 : 0);
#define MAT_SEL(f) (mi == 0 ? a.mat[0].f : (mi == 1 ? a.mat[1].f : a.mat[2].f))
    const uint8_t * const mW = MAT_SEL(W);
    const int64_t m_w_nb1 = MAT_SEL(w_nb1);
    const int mN = MAT_SEL(N);
    const int m_panel0 = MAT_SEL(panel0);
    const int n0 = (panel - m_panel0) * BN, m0 = mt * BM;
    const int nblk_all = a.K / 256;
    const int sb_lo = (int) (((int64_t) blockIdx.y * nblk_all) / a.ksplit), sb_hi = (int) (((int64_t) (blockIdx.y + 1) * nblk_all) / a.ksplit);
    const int nblk = nblk_all;  // row stride of the activation blocks
    const int nslab = wave % (BN / 32), mhalf = wave / (BN / 32);
    const int cb = KG == 1 ? mhalf * 64 : 0;   // first activation column of this wave
    const int ks0 = KG == 1 ? 0 : mhalf * NKS;  // first K step of a trip this wave multiplies

    // staging roles
    const int arow = tid >> 2, aq = tid & 3;
    const uint8_t * wrow = mW + (size_t) min(n0 + arow, mN - 1) * m_w_nb1;
    // (scalars, not arrays: arrays captured by the staging lambdas end up in scratch memory)
    auto b_src = [&](const int i) { const int c = tid + i * NT; return a.act[(size_t) min(m0 + (c >> 3), a.M - 1) * nblk].qs + 16 * (c & 7); };
    auto b_off = [&](const int i) { const int c = tid + i * NT; return sw_off(c >> 3, c & 7); };
    const int8_t * bsrc0 = b_src(0), * bsrc1 = b_src(NBC > 1 ? 1 : 0), * bsrc2 = b_src(NBC > 2 ? 2 : 0), * bsrc3 = b_src(NBC > 2 ? 3 : 0);
    const int boff0 = b_off(0), boff1 = b_off(NBC > 1 ? 1 : 0), boff2 = b_off(NBC > 2 ? 2 : 0), boff3 = b_off(NBC > 2 ? 3 : 0);
    const q8k_dev * mcol = a.act + (size_t) min(m0 + (tid & (BM - 1)), a.M - 1) * nblk;  // column whose bsums / d this thread stages (tid < BM)
    const int c_lo = 4 * (aq >> 1) + (aq & 1);
    const int aoff_lo = sw_off(arow, c_lo), aoff_hi = sw_off(arow, c_lo + 2);

    float16v C[NTT];
    int16v acc[NP][NTT];
#pragma unroll
    for (int t = 0; t < NTT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            C[t][r] = 0.0f;
#pragma unroll
            for (int p = 0; p < NP; ++p) acc[p][t][r] = 0;
        }
    const float16v zerof = C[0];
    const int16v zeroi = acc[0][0];

    // raw global data of the NEXT trip to be staged
    uint4 g_hdr, g_q, g_qh, g_b0, g_b1, g_b2, g_b3, g_bs0, g_bs1;
    uint32_t g6_ql0 = 0, g6_ql1 = 0, g6_ql2 = 0, g6_ql3 = 0, g6_qh0 = 0, g6_qh1 = 0, g6_sc0 = 0, g6_sc1 = 0;  // Q6_K: 2-byte aligned blocks, dword loads
    uint16_t g6_d = 0;
    float g_dy = 0.0f;
    auto issue_loads = [&](const int sb, const int h) {
        const uint8_t * blk = wrow + (size_t) sb * BYTES;
        if constexpr (!Q6) {
            g_hdr = *(const uint4 *) blk;
            g_q = *(const uint4 *) (blk + (QT == 5 ? 48 : 16) + 64 * h + 16 * aq);
            if constexpr (QT == 5) g_qh = *(const uint4 *) (blk + 16 + 16 * (aq & 1));
        } else {
            // half h of a Q6_K block: ql[64h .. 64h+63], qh[32h .. 32h+31], scales[8h .. 8h+7]; this thread owns l = 8*aq .. 8*aq+7
            g6_ql0 = ld32_a2(blk + 64 * h + 8 * aq);
            g6_ql1 = ld32_a2(blk + 64 * h + 8 * aq + 4);
            g6_ql2 = ld32_a2(blk + 64 * h + 32 + 8 * aq);
            g6_ql3 = ld32_a2(blk + 64 * h + 32 + 8 * aq + 4);
            g6_qh0 = ld32_a2(blk + 128 + 32 * h + 8 * aq);
            g6_qh1 = ld32_a2(blk + 128 + 32 * h + 8 * aq + 4);
            g6_sc0 = ld32_a2(blk + 192 + 8 * h);
            g6_sc1 = ld32_a2(blk + 192 + 8 * h + 4);
            g6_d = ld16(blk + 208);
        }
        const size_t bo = (size_t) sb * sizeof(q8k_dev) + 128 * h;
        g_b0 = *(const uint4 *) (bsrc0 + bo);
        if constexpr (NBC > 1) g_b1 = *(const uint4 *) (bsrc1 + bo);
        if constexpr (NBC > 2) {
            g_b2 = *(const uint4 *) (bsrc2 + bo);
            g_b3 = *(const uint4 *) (bsrc3 + bo);
        }
        if (h == 1 && tid < BM) {
            if constexpr (!Q6) {
                g_bs0 = *(const uint4 *) mcol[sb].bsums;
                g_bs1 = *(const uint4 *) (mcol[sb].bsums + 8);
            }
            g_dy = mcol[sb].d;
        }
    };

    auto stage = [&](const int h) {
        char * buf = smem + h * STAGE;
        if constexpr (!Q6) {
        // ---- weight pieces: this thread owns 16 values of sub-block 2*j2 (low nibbles) and of 2*j2+1 (high nibbles)
        const int j2 = 2 * h + (aq >> 1);
        uint32_t scp;  // (sc of sub-block 2*j2) | (sc of 2*j2+1) << 8
        {
            const int sh = 16 * (j2 & 1);
            const uint32_t aa = (g_hdr.y >> sh) & 0xFFFFu, ww = (g_hdr.w >> sh) & 0xFFFFu;
            scp = j2 < 2 ? (aa & 0x3F3Fu) : ((ww & 0x0F0Fu) | ((aa & 0xC0C0u) >> 2));
        }
        const uint32_t sc0 = scp & 0xFF, sc1 = scp >> 8;
        uint32_t dlo[NP], dhi[NP];  // per-piece digits, replicated into both 16-bit lanes
        if constexpr (QT == 4) {
            dlo[0] = (sc0 >> 3) * 0x00010001u; dlo[1] = (sc0 & 7) * 0x00010001u;
            dhi[0] = (sc1 >> 3) * 0x00010001u; dhi[1] = (sc1 & 7) * 0x00010001u;
        } else {
            dlo[0] = (sc0 >> 4) * 0x00010001u; dlo[1] = ((sc0 >> 2) & 3) * 0x00010001u; dlo[2] = (sc0 & 3) * 0x00010001u;
            dhi[0] = (sc1 >> 4) * 0x00010001u; dhi[1] = ((sc1 >> 2) & 3) * 0x00010001u; dhi[2] = (sc1 & 3) * 0x00010001u;
        }
        const uint32_t qv[4] = {g_q.x, g_q.y, g_q.z, g_q.w};
        uint32_t l4[4], h4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            l4[k] = qv[k] & 0x0F0F0F0Fu;
            h4[k] = (qv[k] >> 4) & 0x0F0F0F0Fu;
            if constexpr (QT == 5) {
                const uint32_t qh = k == 0 ? g_qh.x : (k == 1 ? g_qh.y : (k == 2 ? g_qh.z : g_qh.w));
                l4[k] |= ((qh >> (2 * j2)) & 0x01010101u) << 4;
                h4[k] |= ((qh >> (2 * j2 + 1)) & 0x01010101u) << 4;
            }
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            char * Ap = buf + p * TA;
            *(uint4 *) (Ap + aoff_lo) = make_uint4(pk_mul_u16x2(l4[0], dlo[p]), pk_mul_u16x2(l4[1], dlo[p]), pk_mul_u16x2(l4[2], dlo[p]), pk_mul_u16x2(l4[3], dlo[p]));
            *(uint4 *) (Ap + aoff_hi) = make_uint4(pk_mul_u16x2(h4[0], dhi[p]), pk_mul_u16x2(h4[1], dhi[p]), pk_mul_u16x2(h4[2], dhi[p]), pk_mul_u16x2(h4[3], dhi[p]));
        }
        } else {
            // Q6_K: values q - 32 in [-32, 31] times an int8 scale reach +-4096, so the PRODUCT is split instead of the scale:
            // p = (q - 32) * sc = 64 * p1 + p0 with p0 = p & 63 in [0, 63] and p1 = p >> 6 in [-64, 62] — two int8 pieces again.
            // This thread: l = 8*aq .. +7 of the four 32-value groups w = 0..3 of the half -> elements 32*w + l, scale 2*w + (aq >> 1)
            typedef short short2v __attribute__((ext_vector_type(2)));
            const int is = aq >> 1;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const uint32_t scw = (w & 2) ? g6_sc1 : g6_sc0;                           // scales 4*(w>>1) .. +3 of the half
                const int sc = (int) (int8_t) (scw >> (8 * (2 * (w & 1) + is)));           // scale index 2*w + is
                const uint32_t scp = ((uint32_t) sc & 0xFFFFu) * 0x00010001u;
                const uint32_t cp = ((uint32_t) (-32 * sc) & 0xFFFFu) * 0x00010001u;
                uint32_t o1[2], o0[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const uint32_t ql = (w & 1) ? (i ? g6_ql3 : g6_ql2) : (i ? g6_ql1 : g6_ql0);
                    const uint32_t qh = i ? g6_qh1 : g6_qh0;
                    const uint32_t nib = (w & 2) ? ((ql >> 4) & 0x0F0F0F0Fu) : (ql & 0x0F0F0F0Fu);
                    const uint32_t u = nib | (((qh >> (2 * w)) & 0x03030303u) << 4);   // four 6-bit values
                    const short2v pa = __builtin_bit_cast(short2v, __builtin_amdgcn_perm(0u, u, 0x0c010c00u)) * __builtin_bit_cast(short2v, scp) + __builtin_bit_cast(short2v, cp);
                    const short2v pb = __builtin_bit_cast(short2v, __builtin_amdgcn_perm(0u, u, 0x0c030c02u)) * __builtin_bit_cast(short2v, scp) + __builtin_bit_cast(short2v, cp);
                    const uint32_t pa1 = __builtin_bit_cast(uint32_t, pa >> (short2v){6, 6}), pb1 = __builtin_bit_cast(uint32_t, pb >> (short2v){6, 6});
                    const uint32_t pa0 = __builtin_bit_cast(uint32_t, pa) & 0x003F003Fu, pb0 = __builtin_bit_cast(uint32_t, pb) & 0x003F003Fu;
                    // low bytes of the four 16-bit lanes -> one dword (element order 0,1,2,3)
                    o1[i] = __builtin_amdgcn_perm(pb1, pa1, 0x06040200u);
                    o0[i] = __builtin_amdgcn_perm(pb0, pa0, 0x06040200u);
                }
                const int off = sw_off(arow, 2 * w + is) + 8 * (aq & 1);
                *(uint2 *) (buf + off) = make_uint2(o1[0], o1[1]);
                *(uint2 *) (buf + TA + off) = make_uint2(o0[0], o0[1]);
            }
        }
        // ---- activation tile: raw int8
        char * Bt = buf + NP * TA;
        *(uint4 *) (Bt + boff0) = g_b0;
        if constexpr (NBC > 1) *(uint4 *) (Bt + boff1) = g_b1;
        if constexpr (NBC > 2) {
            *(uint4 *) (Bt + boff2) = g_b2;
            *(uint4 *) (Bt + boff3) = g_b3;
        }
        // ---- per-super-block metadata (staged with the second half; consumed at the end of that trip)
        if (h == 1) {
            if constexpr (Q6) {
                if (aq == 0) dd[arow] = make_float2(h2f(g6_d), 0.0f);
                if (tid < BM) dyv[tid] = g_dy;
            }
            if (!Q6 && aq == 0) {
                const float d = h2f((uint16_t) (g_hdr.x & 0xFFFF)), dmin = h2f((uint16_t) (g_hdr.x >> 16));
                dd[arow] = make_float2(d, dmin);
                const uint32_t hz = g_hdr.z, hw = g_hdr.w;
                const uint32_t mlo = hz & 0x3F3F3F3Fu, mhi = ((hw >> 4) & 0x0F0F0F0Fu) | ((hz >> 2) & 0x30303030u);  // mins 0..3 / 4..7 as bytes
                uint32_t pm[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {  // byte j into both 16-bit lanes, then 0x6400 | n == 1024 + n in f16
                    typedef _Float16 half2m __attribute__((ext_vector_type(2)));
                    const uint32_t sel = 0x0c000c00u | (uint32_t) j | ((uint32_t) j << 16);
                    half2m a2 = __builtin_bit_cast(half2m, __builtin_amdgcn_perm(0u, mlo, sel) | 0x64006400u);
                    half2m b2 = __builtin_bit_cast(half2m, __builtin_amdgcn_perm(0u, mhi, sel) | 0x64006400u);
                    a2 = a2 - (half2m){(_Float16) 1024.0f, (_Float16) 1024.0f};
                    b2 = b2 - (half2m){(_Float16) 1024.0f, (_Float16) 1024.0f};
                    pm[j] = __builtin_bit_cast(uint32_t, a2);
                    pm[j + 4] = __builtin_bit_cast(uint32_t, b2);
                }
                uint4 * dm = (uint4 *) (Am + arow * MI_MS);
                dm[0] = make_uint4(pm[0], pm[1], pm[2], pm[3]);
                dm[1] = make_uint4(pm[4], pm[5], pm[6], pm[7]);
            }
            if (!Q6 && tid < BM) {
                dyv[tid] = g_dy;
                uint4 * dbm = (uint4 *) (Bm + tid * MI_MS);  // the Q8_K bsums are stored as f16 already
                dbm[0] = g_bs0;
                dbm[1] = g_bs1;
            }
        }
    };

    const int fr = lane & 31, kg = lane >> 5;
    const int swz = (fr >> 1) & 7;
    const int arow_off = (nslab * 32 + fr) * 128;
    const int brow_off[2] = {(cb + fr) * 128, (cb + 32 + fr) * 128};

    // (the first MFMA of a super-block takes a literal zero accumulator: the integer sums never have to be cleared)
    auto mma = [&](const int h) {
        const char * buf = smem + h * STAGE;
        const char * Bt = buf + NP * TA;
#pragma unroll
        for (int kk = 0; kk < NKS; ++kk) {
            const int co = ((2 * (ks0 + kk) + kg) ^ swz) << 4;
            int4v fb[NTT];
#pragma unroll
            for (int t = 0; t < NTT; ++t) fb[t] = *(const int4v *) (Bt + brow_off[t] + co);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int4v fa = *(const int4v *) (buf + p * TA + arow_off + co);
#pragma unroll
                for (int t = 0; t < NTT; ++t) acc[p][t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb[t], (h == 0 && kk == 0) ? zeroi : acc[p][t], 0, 0, 0);
            }
        }
    };

    // super-block fold on the packed-f32 pipe (v_pk_mul_f32 / v_pk_fma_f32): two accumulator rows per instruction
    auto fold = [&]() {
        typedef float float2v __attribute__((ext_vector_type(2)));
        half8 fam;
        if constexpr (!Q6) fam = *(const half8 *) (Am + (nslab * 32 + fr) * MI_MS + kg * 16);
#pragma unroll
        for (int t = 0; t < NTT; ++t) {
            float16v am = zerof;
            if constexpr (!Q6) {
                if (KG == 1 || mhalf == 0) {  // the mins term belongs to the super-block, not to a K step: one wave group adds it
                    const half8 fbm = *(const half8 *) (Bm + (cb + t * 32 + fr) * MI_MS + kg * 16);
                    am = __builtin_amdgcn_mfma_f32_32x32x16_f16(fam, fbm, zerof, 0, 0, 0);
                }
            }
            const float dy = dyv[cb + t * 32 + fr];
            const float2v dy2 = {dy, dy};
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int i = nslab * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;  // rows i, i + 1
                const float4 sd = *(const float4 *) &dd[i];                   // (d, dmin) of both
                int is0, is1;
                if constexpr (QT == 4) { is0 = (acc[0][t][r] << 3) + acc[1][t][r]; is1 = (acc[0][t][r + 1] << 3) + acc[1][t][r + 1]; }
                else if constexpr (QT == 5) {
                    is0 = (acc[0][t][r] << 4) + (acc[1][t][r] << 2) + acc[2][t][r];
                    is1 = (acc[0][t][r + 1] << 4) + (acc[1][t][r + 1] << 2) + acc[2][t][r + 1];
                } else { is0 = (acc[0][t][r] << 6) + acc[1][t][r]; is1 = (acc[0][t][r + 1] << 6) + acc[1][t][r + 1]; }
#if MMQ_FOLD_PACKED
                float2v v = (float2v){(float) is0, (float) is1} * (float2v){sd.x, sd.z};
                if constexpr (!Q6) v = __builtin_elementwise_fma(-(float2v){sd.y, sd.w}, (float2v){am[r], am[r + 1]}, v);
                const float2v c = __builtin_elementwise_fma(dy2, v, (float2v){C[t][r], C[t][r + 1]});
#else
                float v0 = sd.x * (float) is0, v1 = sd.z * (float) is1;
                if constexpr (!Q6) { v0 = __builtin_fmaf(-sd.y, am[r], v0); v1 = __builtin_fmaf(-sd.w, am[r + 1], v1); }
                const float2v c = {__builtin_fmaf(dy, v0, C[t][r]), __builtin_fmaf(dy, v1, C[t][r + 1])};
                (void) dy2;
#endif
                C[t][r] = c.x;
                C[t][r + 1] = c.y;
            }
        }
    };

    issue_loads(sb_lo, 0);
    stage(0);
    issue_loads(sb_lo, 1);
    __syncthreads();
    for (int sb = sb_lo; sb < sb_hi; ++sb) {
        // trip (sb, 0): convert + write the second half while the MFMAs of the first half run
        stage(1);
        if (sb + 1 < sb_hi) issue_loads(sb + 1, 0);
        mma(0);
        __syncthreads();
        // trip (sb, 1)
        if (sb + 1 < sb_hi) {
            stage(0);
            issue_loads(sb + 1, 1);
        }
        mma(1);
        fold();
        __syncthreads();
    }
    if constexpr (KG == 2) {
        // the two wave groups hold partial sums over disjoint K steps: group 1 hands its tile over through LDS (the loop ended
        // with a barrier, the stage buffers are free)
        float * red = (float *) smem;
        if (mhalf == 1) {
#pragma unroll
            for (int t = 0; t < NTT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((nslab * NTT + t) * 16 + r) * 64 + lane] = C[t][r];
        }
        __syncthreads();
        if (mhalf == 1) return;
#pragma unroll
        for (int t = 0; t < NTT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) C[t][r] += red[((nslab * NTT + t) * 16 + r) * 64 + lane];
    }
    // ---- store: lane holds column (token) j and 4 runs of 4 consecutive rows
#pragma unroll
    for (int t = 0; t < NTT; ++t) {
        const int m = m0 + cb + t * 32 + fr;
        if (m >= a.M) continue;
        float * out = a.ksplit > 1 ? MAT_SEL(part) + ((size_t) blockIdx.y * a.M + m) * mN : MAT_SEL(dst) + (size_t) m * MAT_SEL(dst_stride);
        const float * m_add = MAT_SEL(add);
        const float * ad = (m_add && a.ksplit == 1) ? m_add + (size_t) m * MAT_SEL(add_stride) : nullptr;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = n0 + nslab * 32 + 8 * g + 4 * kg;
            float c4[4] = {C[t][4 * g], C[t][4 * g + 1], C[t][4 * g + 2], C[t][4 * g + 3]};
            if (ad) {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n + r < mN) c4[r] += ad[n + r];
            }
            if (n + 3 < mN && ((((uintptr_t) (out + n)) & 15) == 0)) {
                *(float4 *) (out + n) = make_float4(c4[0], c4[1], c4[2], c4[3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n + r < mN) out[n + r] = c4[r];
            }
        }
    }
}

#undef MAT_SEL

bool mmq_i8_supported(int type, int64_t K, int64_t N, int64_t M) {
    (void) N;
    return (type == GGML_TYPE_Q4_K || type == GGML_TYPE_Q5_K || type == GGML_TYPE_Q6_K) && (K % 256) == 0 && M >= 2;
}

template <int QT, int BN, int BM = 128> static void launch_mmq8_t(hipStream_t s, mmq8_args a) {
    constexpr int NP = QT == 5 ? 3 : 2;
    const size_t lds = 2 * (size_t) (NP * BN * 128 + BM * 128) + (size_t) (BN + BM) * MI_MS + BN * sizeof(float2) + BM * sizeof(float);
    static std::atomic<uint32_t> lds_raised{0};  // one bit per device (common.h: ensure_dyn_lds)
    (void) ensure_dyn_lds((const void *) k_mmq_i8<QT, BN, BM>, lds, lds_raised);  // on failure the launch below fails and graph_compute reports it
    a.n_panels = 0;
    for (int i = 0; i < a.n_mat; ++i) {
        a.mat[i].panel0 = a.n_panels;
        a.n_panels += (a.mat[i].N + BN - 1) / BN;
    }
    a.m_tiles = (a.M + BM - 1) / BM;
    const unsigned grid = (unsigned) (((a.n_panels + 7) / 8) * 8 * a.m_tiles);
    MI_LAUNCH_PROBED((k_mmq_i8<QT, BN, BM>), dim3(grid, (unsigned) a.ksplit), dim3(BN * 4), lds, s, a);
}

// few activation columns (continuous-batching decode, M <= 64) leave N/64 x 1 workgroups — far fewer than 256 CUs — so
// the K range is split over blockIdx.y and the partial products are summed in a fixed order by a second tiny kernel
int mmq_pick_ksplit(int64_t K, int64_t N, int64_t M, bool skinny, int type) {
    if (skinny && M >= 2 && M <= 32 && (N % 32) == 0) return mmq_skinny_ksplit(K, N);
    // prompt batches the wide form of mmq_skinny.hip serves run without a K split (its grid fills the chip by token-tile groups)
    if (skinny && mmq_wide_tiles(type, K, &N, 1, M, (K / 256) * (type == GGML_TYPE_Q4_K ? 144 : 176), 1) != 0) return 1;
    const int64_t wgs = ((N + 63) / 64) * ((M + 127) / 128), nblk = K / 256;
    static const int ks_target = getenv("GGML_MI355X_MMQ_KS_TARGET") ? atoi(getenv("GGML_MI355X_MMQ_KS_TARGET")) : 512;
    if (M <= 64) return (int) std::max<int64_t>(1, std::min<int64_t>(nblk, ks_target / std::max<int64_t>(1, wgs)));
    // wide batches (prefill micro-batches): only when the output is too small to occupy the chip (wk/wv: 64 workgroups at
    // M = 512), or when a long K leaves exactly one workgroup per CU (ffn_down): measured 36 -> 19 us and 134 -> 117 us
    if (wgs < 256) return (int) std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(4, nblk / 4), 256 / wgs));
    if (wgs == 256 && nblk >= 32) return 2;
    return 1;
}
// can matrices of these K-quant types (same K, same activations) share one launch of 2..32 columns?
bool mmq_skinny_mix_ok(const int * types, const int64_t * N, const int64_t * w_nb1, int n, int64_t K, int64_t M) {
    bool has4 = false, has5 = false, has6 = false;
    for (int i = 0; i < n; ++i) {
        if (!mmq_skinny_supported(types[i], K, N[i], M, w_nb1[i])) return false;
        has4 = has4 || types[i] == GGML_TYPE_Q4_K;
        has5 = has5 || types[i] == GGML_TYPE_Q5_K;
        has6 = has6 || types[i] == GGML_TYPE_Q6_K;
    }
    return has6 && (has4 != has5);
}
struct splitk_mat { const float * part; int64_t mn; int N; float * dst; int64_t dst_stride; const float * add; int64_t add_stride; };
struct splitk_args { splitk_mat mat[3]; int n_mat, ks; int64_t e1, e2; };  // matrix i owns float4 slots [e_i, e_{i+1}) (e0 = 0)
__global__ void __launch_bounds__(256) k_splitk_reduce(const splitk_args a) {
    int64_t q4 = (int64_t) blockIdx.x * 256 + threadIdx.x;
    const int mi = (a.n_mat > 2 && q4 >= a.e2) ? 2 : ((a.n_mat > 1 && q4 >= a.e1) ? 1 : 0);
    q4 -= mi == 0 ? 0 : (mi == 1 ? a.e1 : a.e2);
#define SK_SEL(f) (mi == 0 ? a.mat[0].f : (mi == 1 ? a.mat[1].f : a.mat[2].f))
    const int64_t e = q4 * 4, mn = SK_SEL(mn);
    if (e >= mn) return;
    const float * part = SK_SEL(part);
    float4 acc = *(const float4 *) (part + e);
    for (int k = 1; k < a.ks; ++k) {
        const float4 v = *(const float4 *) (part + (int64_t) k * mn + e);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const int N = SK_SEL(N);
    const int64_t m = e / N, n = e % N;  // N % 4 == 0 (checked by the launcher)
    const float * add = SK_SEL(add);
    if (add) {
        const float * ad = add + m * SK_SEL(add_stride) + n;
        acc.x += ad[0]; acc.y += ad[1]; acc.z += ad[2]; acc.w += ad[3];
    }
    *(float4 *) (SK_SEL(dst) + m * SK_SEL(dst_stride) + n) = acc;
#undef SK_SEL
}
static void launch_splitk_reduce_multi(hipStream_t s, const mmq8_args & g) {
    splitk_args a{};
    a.n_mat = g.n_mat;
    a.ks = g.ksplit;
    int64_t slots = 0;
    for (int i = 0; i < g.n_mat; ++i) {
        const mmq8_mat & m = g.mat[i];
        a.mat[i] = {m.part, (int64_t) g.M * m.N, m.N, m.dst, m.dst_stride, m.add, m.add_stride};
        if (i == 1) a.e1 = slots;
        if (i == 2) a.e2 = slots;
        slots += ((int64_t) g.M * m.N / 4 + 255) / 256 * 256;  // whole workgroups per matrix
    }
    hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned) (slots / 256)), dim3(256), 0, s, a);
}
void launch_splitk_reduce(hipStream_t s, const float * part, int ks, int M, int N, float * dst, int64_t dst_stride, const float * add, int64_t add_stride) {
    mmq8_args g{};
    g.n_mat = 1;
    g.M = M;
    g.ksplit = ks;
    g.mat[0].part = const_cast<float *>(part);
    g.mat[0].N = N;
    g.mat[0].dst = dst;
    g.mat[0].dst_stride = dst_stride;
    g.mat[0].add = add;
    g.mat[0].add_stride = add_stride;
    launch_splitk_reduce_multi(s, g);
}

// n_mat (1..3) matrices of one type against the same activations; `part` holds ksplit * M * sum(N) floats when ksplit > 1
int launch_mmq_i8_multi(hipStream_t s, int type, int n_mat, const mmq_mat_desc * mats, int K, int M, const void * act_q8k, int force_bn, int ksplit, float * part, bool reduce, bool skinny, const mmq_epi * epi) {
    mmq8_args a{};
    a.n_mat = n_mat;
    a.K = K;
    a.M = M;
    a.act = (const q8k_dev *) act_q8k;
    a.ksplit = std::max(1, ksplit);
    int64_t panels128 = 0;
    const bool skinny_opt = skinny;
    bool mixed = false;
    float * pp = part;
    for (int i = 0; i < n_mat; ++i) {
        a.mat[i].W = mats[i].W;
        a.mat[i].w_nb1 = mats[i].w_nb1;
        a.mat[i].N = mats[i].N;
        a.mat[i].dst = mats[i].dst;
        a.mat[i].dst_stride = mats[i].dst_stride;
        a.mat[i].add = mats[i].add;
        a.mat[i].add_stride = mats[i].add_stride;
        a.mat[i].part = pp;
        pp += (size_t) a.ksplit * M * mats[i].N;
        panels128 += (mats[i].N + 127) / 128;
        const int ti = mats[i].type ? mats[i].type : type;
        mixed = mixed || ti != type;
        a.mat[i].qt = ti == GGML_TYPE_Q4_K ? 4 : (ti == GGML_TYPE_Q5_K ? 5 : 6);
        skinny = skinny && mmq_skinny_supported(ti, K, mats[i].N, M, mats[i].w_nb1);
    }
    if (mixed) {  // two formats: the skinny kernel's two-pass form only (the caller asked mmq_skinny_mix_ok first)
        if (!skinny || (a.ksplit & (a.ksplit - 1)) != 0 || (epi && a.ksplit != 1)) { MI_ERR("launch_mmq_i8_multi: a two-format launch the skinny kernel does not serve"); abort(); }
        if (epi) { a.has_epi = 1; a.epi = *epi; }
        if (!launch_mmq_skinny_mixed(s, a)) { MI_ERR("launch_mmq_i8_multi: this pair of formats has no two-format kernel"); abort(); }
        if (a.ksplit > 1 && reduce) launch_splitk_reduce_multi(s, a);
        return 1;
    }
    for (int i = 0; i < n_mat; ++i) a.mat[i].qt = 0;
    if (skinny_opt && M >= 33) {  // prompt batches: the wide form of the skinny unit, where it applies
        int64_t Ns[3] = {0, 0, 0};
        bool same = true;
        for (int i = 0; i < n_mat; ++i) {
            Ns[i] = mats[i].N;
            same = same && mats[i].w_nb1 == mats[0].w_nb1;
        }
        const int tt = same ? mmq_wide_tiles(type, K, Ns, n_mat, M, mats[0].w_nb1, a.ksplit) : 0;
        if (tt) {
            launch_mmq_wide(s, type, tt, a);
            return 2;
        }
    }
    skinny = skinny && (a.ksplit & (a.ksplit - 1)) == 0;  // (its K split is a power of two: mmq_skinny_ksplit)
    if (epi && !(skinny && a.ksplit == 1)) {
        MI_ERR("launch_mmq_i8_multi: a rope / cache-store epilogue was requested for a launch the skinny kernel does not serve");
        abort();
    }
    if (skinny) {  // a decode step of a continuous batch: stream the weights once (mmq_skinny.hip)
        if (epi) {
            a.has_epi = 1;
            a.epi = *epi;
        }
        launch_mmq_skinny(s, type, a);
        if (a.ksplit > 1 && reduce) launch_splitk_reduce_multi(s, a);
        return 1;
    }
    // 128-row panels unless that leaves CUs idle (256 CUs, one 8-wave workgroup each)
    const int64_t wg128 = panels128 * ((M + MI_BM - 1) / MI_BM);
    // continuous-batching decode steps: column tiles of 32 / 64 (always on 64-row panels — the output matrix included)
    static const int force_bm = getenv("GGML_MI355X_MMQ_BM") ? atoi(getenv("GGML_MI355X_MMQ_BM")) : 0;
    const int bm = force_bm ? force_bm : (M <= 32 ? 32 : (M <= 64 ? 64 : 128));
    // 128-row panels (one 8-wave workgroup per CU) when they still fill the chip — except for Q4_K / Q6_K at K < 8192, where 64-row
    // panels (two independent 4-wave workgroups per CU, not behind one barrier) measure 2-3 % ahead on every prompt shape of the 8B
    // model; Q5_K (three pieces: the 64-row variant spills) and the 70B shapes measure 4-30 % ahead with 128
    const bool prefer128 = type == GGML_TYPE_Q5_K || K >= 8192;
    // (with the K range split — ffn_down of a prompt micro-batch: 32 panels x 4 column tiles x 2 K halves — the splits count as workgroups too:
    // Q6_K 14336 -> 4096 x 512 measures 102 -> 93-98 us on 128-row panels, round 4)
    const int bn = force_bn ? force_bn : (bm == 128 && wg128 * std::max(1, a.ksplit) >= 256 && prefer128 ? 128 : 64);
    if (bm < 128 && bn == 64) {
#define MMQ_SKINNY(QT)                                   \
    {                                                    \
        if (bm == 32) launch_mmq8_t<QT, 64, 32>(s, a);   \
        else launch_mmq8_t<QT, 64, 64>(s, a);            \
    }
        if (type == GGML_TYPE_Q4_K) MMQ_SKINNY(4) else if (type == GGML_TYPE_Q5_K) MMQ_SKINNY(5) else MMQ_SKINNY(6)
#undef MMQ_SKINNY
    } else if (type == GGML_TYPE_Q4_K) {
        if (bn == 128) launch_mmq8_t<4, 128>(s, a);
        else launch_mmq8_t<4, 64>(s, a);
    } else if (type == GGML_TYPE_Q5_K) {
        if (bn == 128) launch_mmq8_t<5, 128>(s, a);
        else launch_mmq8_t<5, 64>(s, a);
    } else {
        if (bn == 128) launch_mmq8_t<6, 128>(s, a);
        else launch_mmq8_t<6, 64>(s, a);
    }
    if (a.ksplit > 1 && reduce) launch_splitk_reduce_multi(s, a);  // (!reduce: the caller's next kernel sums the partials itself)
    return 0;
}
void launch_splitk_reduce_mats(hipStream_t s, int n_mat, const mmq_mat_desc * mats, const float * part, int ks, int M) {
    mmq8_args a{};
    a.n_mat = n_mat;
    a.M = M;
    a.ksplit = ks;
    float * pp = const_cast<float *>(part);
    for (int i = 0; i < n_mat; ++i) {
        a.mat[i].N = mats[i].N;
        a.mat[i].dst = mats[i].dst;
        a.mat[i].dst_stride = mats[i].dst_stride;
        a.mat[i].add = mats[i].add;
        a.mat[i].add_stride = mats[i].add_stride;
        a.mat[i].part = pp;
        pp += (size_t) ks * M * mats[i].N;
    }
    launch_splitk_reduce_multi(s, a);
}
int launch_mmq_i8(hipStream_t s, int type, const uint8_t * W, int64_t w_nb1, int K, int N, int M, const void * act_q8k, float * dst, int64_t dst_stride, int force_bn,
                   int ksplit, float * part, const float * add, int64_t add_stride, bool reduce, bool skinny, const mmq_epi * epi) {
    const mmq_mat_desc m{W, w_nb1, N, dst, dst_stride, add, add_stride};
    return launch_mmq_i8_multi(s, type, 1, &m, K, M, act_q8k, force_bn, ksplit, part, reduce, skinny, epi);
}

MI_TU_TOUCH(mmq_i8)

}  // namespace mi355x
