// mmq.hip — prefill / large-batch mat-mul for K-quant weights on the gfx950 matrix cores.
//
// Replaces ggml-hip's mul_mat_q / dequant+hipBLAS path and must agree with ggml-cpu's
// ggml_vec_dot_q{4,5,6}_K_q8_K, which is INTEGER arithmetic on Q8_K-quantised activations:
//     y[n,m] = sum_sb  dy[m,sb] * ( d[n,sb] * sum_j sc[n,sb,j] * <q[n,j,:], q8[m,j,:]>  -  dmin[n,sb] * sum_j mn[n,sb,j] * bsum[m,sb,j] )
// A dense f16 GEMM on dequantised weights would round every weight to 11 bits and land ~1e-5 away from that — enough
// to flip downstream Q8 activation roundings (DESIGN.md "summation-order floor").  So the LDS "dequant" here keeps the
// integers: the A tile holds f16(sc * q) (<= 63*31 = 1953 < 2048: exact), the B tile f16(q8) (|.| <= 127: exact), and
// v_mfma_f32_32x32x16_f16 accumulates their products in f32, which is exact below 2^24 — the super-block sums come
// out as the SAME integers the CPU computes.  The mins term is one more MFMA step per super-block (A' = mn repeated,
// B' = the 16 bsums, <= 2032: exact).  Q6_K scales reach 127*32 = 4064 > 2048, so its int8 scale is split
// sc = 16*sh + sl and two exact products are accumulated (A1 = sh*(q-32), A2 = sl*(q-32)).  Once per super-block the
// integer accumulators are folded into the f32 result with the block scales — the only float rounding, same as the CPU.
//
// Tiling (wave64, MFMA 32x32x16): workgroup = 8 waves = 128 weight rows x 128 activation columns, K advanced half a
// super-block (128) per trip; each wave owns 32 rows x 64 columns (2 MFMA tiles, 3 accumulator sets = 96 VGPRs).
// LDS tiles are row-major with a 16-byte pad per row (272 B stride) so the 16-lane groups of ds_read_b128 hit 16
// distinct slots.  Global loads for trip t+1 are issued before the MFMAs of trip t (issue-early / write-late).
// Workgroups that share a weight panel are placed on the same XCD (block b runs on XCD b % 8) so the panel is
// fetched from HBM once per micro-batch.
#include <algorithm>

#include "dev_util.h"
#include "kernels.h"

namespace mi355x {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

constexpr int MQ_BN = 128, MQ_BM = 128, MQ_BK = 128;
constexpr int MQ_AS = (MQ_BK + 8) * 2;  // LDS row stride in bytes (272)
constexpr int MQ_MS = (16 + 8) * 2;     // row stride of the 16-wide mins / bsums tiles (48 B)

struct mmq_args {
    const uint8_t * W;
    int64_t w_nb1;
    int K, N, M;
    const q8k_dev * act;  // [M][K/256]
    float * dst;
    int64_t dst_stride;
    int n_panels, m_tiles;
    int ksplit;    // see mmq_i8.hip: blockIdx.y owns a range of super-blocks, partial [M][N] results go to part
    float * part;
};

// ---- packed integer -> f16 conversion without per-value cvt instructions.  For 0 <= n < 1024 the f16 with bits
// 0x6400 | n is exactly 1024 + n (ulp of the [1024, 2048) binade is 1), so OR-ing the bias into both halves of a dword
// and one v_pk_add_f16 of -1024 turns two 16-bit integers into two exact halves.
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t u16x2_to_h2(const uint32_t n2, const float bias) {  // n2: two u16 < 1024; result = n - bias'
    const uint32_t biased = n2 | 0x64006400u;
    half2v h = __builtin_bit_cast(half2v, biased);
    const _Float16 b = (_Float16) bias;
    h = h - (half2v){b, b};
    return __builtin_bit_cast(uint32_t, h);
}
// bytes (b0,b1,b2,b3) of x -> (b0 | b1<<16), (b2 | b3<<16)
__device__ __forceinline__ uint32_t bytes01_to_u16x2(const uint32_t x) { return __builtin_amdgcn_perm(0u, x, 0x0c010c00u); }
__device__ __forceinline__ uint32_t bytes23_to_u16x2(const uint32_t x) { return __builtin_amdgcn_perm(0u, x, 0x0c030c02u); }
__device__ __forceinline__ uint32_t pk_mul_u16(const uint32_t a, const uint32_t b) {
    typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
    const ushort2v r = __builtin_bit_cast(ushort2v, a) * __builtin_bit_cast(ushort2v, b);
    return __builtin_bit_cast(uint32_t, r);
}

__device__ __forceinline__ uint32_t pack_h2(const int a, const int b) {
    const _Float16 x = (_Float16) a, y = (_Float16) b;
    uint16_t ux, uy;
    __builtin_memcpy(&ux, &x, 2);
    __builtin_memcpy(&uy, &y, 2);
    return (uint32_t) ux | ((uint32_t) uy << 16);
}

// QT: 4 = Q4_K, 5 = Q5_K, 6 = Q6_K
template <int QT>
__global__ void __launch_bounds__(512) k_mmq(const mmq_args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BYTES = QT == 4 ? 144 : (QT == 5 ? 176 : 210);
    constexpr bool Q6 = QT == 6;
    // LDS carve-up
    char * A1 = smem;                                   // [128][136] f16
    char * A2 = A1 + MQ_BN * MQ_AS;                     // Q6_K: second scale part; K4/K5: unused (size 0)
    char * Bt = A2 + (Q6 ? MQ_BN * MQ_AS : 0);          // [128][136] f16
    char * Am = Bt + MQ_BM * MQ_AS;                     // mins  [128][24] f16 (K4/K5)
    char * Bm = Am + (Q6 ? 0 : MQ_BN * MQ_MS);          // bsums [128][24] f16 (K4/K5)
    float2 * dd = (float2 *) (Bm + (Q6 ? 0 : MQ_BM * MQ_MS));  // (d, dmin) per row
    float * dyv = (float *) (dd + MQ_BN);               // dy per column

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware placement: blocks 8q+x (x = XCD) walk panel (q / m_tiles)*8 + x, m-tile q % m_tiles
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int panel = (q / a.m_tiles) * 8 + xcd, mt = q % a.m_tiles;
    if (panel >= a.n_panels) return;
    const int n0 = panel * MQ_BN, m0 = mt * MQ_BM;
    const int nblk = a.K / 256;
    const int sb_lo = (int) (((int64_t) blockIdx.y * nblk) / a.ksplit), sb_hi = (int) (((int64_t) (blockIdx.y + 1) * nblk) / a.ksplit);
    const int nslab = wave & 3, mhalf = wave >> 2;

    // staging roles
    const int arow = tid >> 2, aq = tid & 3;  // weight row in the panel / quarter of the 64 packed bytes
    const int grow = min(n0 + arow, a.N - 1);
    const uint8_t * wrow = a.W + (size_t) grow * a.w_nb1;
    const int bcol = tid >> 2, bq = tid & 3;  // activation column / quarter of the 128 int8
    const bool bvalid = (m0 + bcol) < a.M;
    const q8k_dev * ycol = a.act + (size_t) min(m0 + bcol, a.M - 1) * nblk;

    float16v C[2], acc1[2], acc2[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { C[t][r] = 0.0f; acc1[t][r] = 0.0f; acc2[t][r] = 0.0f; }
    const float16v zero = C[0];

    // registers holding the NEXT trip's raw global data
    uint4 g_hdr, g_q, g_qh, g_b0, g_b1;
    uint32_t g6_ql[4], g6_qh[2], g6_sc[2];
    uint16_t g6_d = 0;
    auto issue_loads = [&](const int sb, const int h) {
        const uint8_t * blk = wrow + (size_t) sb * BYTES;
        if constexpr (!Q6) {
            g_hdr = *(const uint4 *) blk;
            g_q = *(const uint4 *) (blk + (QT == 5 ? 48 : 16) + 64 * h + 16 * aq);
            if constexpr (QT == 5) g_qh = *(const uint4 *) (blk + 16 + 16 * (aq & 1));
        } else {
            // half h of a Q6_K block: ql[64h .. 64h+63], qh[32h .. 32h+31], scales[8h .. 8h+7]; this thread owns l = 8*aq .. 8*aq+7
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                g6_ql[i] = ld32_a2(blk + 64 * h + 8 * aq + 4 * i);
                g6_ql[2 + i] = ld32_a2(blk + 64 * h + 32 + 8 * aq + 4 * i);
                g6_qh[i] = ld32_a2(blk + 128 + 32 * h + 8 * aq + 4 * i);
                g6_sc[i] = ld32_a2(blk + 192 + 8 * h + 4 * i);
            }
            g6_d = ld16(blk + 208);
        }
        const uint4 * yq = (const uint4 *) (ycol[sb].qs + 128 * h + 32 * bq);
        g_b0 = yq[0];
        g_b1 = yq[1];
    };

    auto stage = [&](const int sb, const int h) {
        // ---- A tile
        if constexpr (!Q6) {
            const int j2 = 2 * h + (aq >> 1);  // pair of sub-blocks (2*j2, 2*j2+1) this 16-byte chunk feeds
            int sc0, sc1, m0_, m1_;
            {
                const uint32_t hy = g_hdr.y, hz = g_hdr.z, hw = g_hdr.w;
                const int sh = 16 * (j2 & 1);
                const uint32_t aa = (hy >> sh) & 0xFFFFu, bb = (hz >> sh) & 0xFFFFu, ww = (hw >> sh) & 0xFFFFu;
                uint32_t scp, mp;
                if (j2 < 2) { scp = aa & 0x3F3Fu; mp = bb & 0x3F3Fu; }
                else { scp = (ww & 0x0F0Fu) | ((aa & 0xC0C0u) >> 2); mp = ((ww >> 4) & 0x0F0Fu) | ((bb & 0xC0C0u) >> 2); }
                sc0 = (int) (scp & 0xFF); sc1 = (int) (scp >> 8); m0_ = (int) (mp & 0xFF); m1_ = (int) (mp >> 8);
            }
            (void) m0_; (void) m1_;
            const uint32_t qv[4] = {g_q.x, g_q.y, g_q.z, g_q.w};
            uint32_t lo[8], hi[8];
            const uint32_t sc0p = (uint32_t) sc0 * 0x00010001u, sc1p = (uint32_t) sc1 * 0x00010001u;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t l4 = qv[k] & 0x0F0F0F0Fu, h4 = (qv[k] >> 4) & 0x0F0F0F0Fu;
                if constexpr (QT == 5) {
                    const uint32_t qh = k == 0 ? g_qh.x : (k == 1 ? g_qh.y : (k == 2 ? g_qh.z : g_qh.w));
                    l4 |= ((qh >> (2 * j2)) & 0x01010101u) << 4;
                    h4 |= ((qh >> (2 * j2 + 1)) & 0x01010101u) << 4;
                }
                if constexpr (QT == 4) {  // sc*q <= 945 < 1024: packed u16 multiply + bias trick, no cvt
                    lo[2 * k] = u16x2_to_h2(pk_mul_u16(bytes01_to_u16x2(l4), sc0p), 1024.0f);
                    lo[2 * k + 1] = u16x2_to_h2(pk_mul_u16(bytes23_to_u16x2(l4), sc0p), 1024.0f);
                    hi[2 * k] = u16x2_to_h2(pk_mul_u16(bytes01_to_u16x2(h4), sc1p), 1024.0f);
                    hi[2 * k + 1] = u16x2_to_h2(pk_mul_u16(bytes23_to_u16x2(h4), sc1p), 1024.0f);
                } else {  // Q5_K reaches 63*31 = 1953: keep the converting path
                    lo[2 * k] = pack_h2(sc0 * (int) (l4 & 0xFF), sc0 * (int) ((l4 >> 8) & 0xFF));
                    lo[2 * k + 1] = pack_h2(sc0 * (int) ((l4 >> 16) & 0xFF), sc0 * (int) (l4 >> 24));
                    hi[2 * k] = pack_h2(sc1 * (int) (h4 & 0xFF), sc1 * (int) ((h4 >> 8) & 0xFF));
                    hi[2 * k + 1] = pack_h2(sc1 * (int) ((h4 >> 16) & 0xFF), sc1 * (int) (h4 >> 24));
                }
            }
            // element offsets inside the half: low nibbles -> 64*(aq>>1) + 16*(aq&1) + i, high nibbles -> +32
            char * dstA = A1 + arow * MQ_AS + (64 * (aq >> 1) + 16 * (aq & 1)) * 2;
            ((uint4 *) dstA)[0] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
            ((uint4 *) dstA)[1] = make_uint4(lo[4], lo[5], lo[6], lo[7]);
            ((uint4 *) (dstA + 64))[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
            ((uint4 *) (dstA + 64))[1] = make_uint4(hi[4], hi[5], hi[6], hi[7]);
            if (h == 0 && aq == 0) {  // per-super-block row metadata: scales d/dmin and the 8 mins (each covers two 16-value bsums)
                const float d = h2f((uint16_t) (g_hdr.x & 0xFFFF)), dmin = h2f((uint16_t) (g_hdr.x >> 16));
                dd[arow] = make_float2(d, dmin);
                int mn[8];
                const uint32_t hy = g_hdr.y, hz = g_hdr.z, hw = g_hdr.w;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    mn[j] = (int) ((hz >> (8 * j)) & 63);
                    mn[j + 4] = (int) (((hw >> (8 * j + 4)) & 0xF) | ((((hz >> (8 * j)) & 0xFF) >> 6) << 4));
                }
                (void) hy;
                uint32_t pm[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) pm[j] = pack_h2(mn[j], mn[j]);
                uint4 * dm = (uint4 *) (Am + arow * MQ_MS);
                dm[0] = make_uint4(pm[0], pm[1], pm[2], pm[3]);
                dm[1] = make_uint4(pm[4], pm[5], pm[6], pm[7]);
            }
        } else {
            // values of this thread: l = 8*aq .. +7 -> elements 128h + {0,32,64,96} + l, scale groups (8h + l/16 + 2k)
            const int is = aq >> 1;
            int sc[4];
            sc[0] = (int) (int8_t) (g6_sc[0] >> (8 * is));
            sc[1] = (int) (int8_t) (g6_sc[0] >> (8 * (is + 2)));
            sc[2] = (int) (int8_t) (g6_sc[1] >> (8 * is));
            sc[3] = (int) (int8_t) (g6_sc[1] >> (8 * (is + 2)));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int sh = sc[k] >> 4, sl = sc[k] & 15;  // sc = 16*sh + sl, sh in [-8,7], sl in [0,15]
                const uint32_t shp = ((uint32_t) sh & 0xFFFFu) * 0x00010001u, slp = (uint32_t) sl * 0x00010001u;
                uint32_t o1[4], o2[4];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const uint32_t ql = (k & 1) ? g6_ql[2 + i] : g6_ql[i];
                    const uint32_t nib = (k & 2) ? ((ql >> 4) & 0x0F0F0F0Fu) : (ql & 0x0F0F0F0Fu);
                    const uint32_t v = nib | (((g6_qh[i] >> (2 * k)) & 0x03030303u) << 4);
                    // packed 16-bit lanes: (q - 32) * s + 512 lies in [32, 992] -> bias trick with 1024 + 512, no cvt
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        typedef short short2v __attribute__((ext_vector_type(2)));
                        const uint32_t u = hh ? bytes23_to_u16x2(v) : bytes01_to_u16x2(v);
                        const short2v qm = __builtin_bit_cast(short2v, u) - (short2v){32, 32};
                        const short2v p1 = qm * __builtin_bit_cast(short2v, shp) + (short2v){512, 512};
                        const short2v p2 = qm * __builtin_bit_cast(short2v, slp) + (short2v){512, 512};
                        o1[2 * i + hh] = u16x2_to_h2(__builtin_bit_cast(uint32_t, p1), 1536.0f);
                        o2[2 * i + hh] = u16x2_to_h2(__builtin_bit_cast(uint32_t, p2), 1536.0f);
                    }
                }
                const int off = arow * MQ_AS + (32 * k + 8 * aq) * 2;
                *(uint4 *) (A1 + off) = make_uint4(o1[0], o1[1], o1[2], o1[3]);
                *(uint4 *) (A2 + off) = make_uint4(o2[0], o2[1], o2[2], o2[3]);
            }
            if (h == 0 && aq == 0) dd[arow] = make_float2(h2f(g6_d), 0.0f);
        }
        // ---- B tile: 32 int8 -> 32 f16 at k = 32*bq ..
        {
            const uint32_t bw[8] = {g_b0.x, g_b0.y, g_b0.z, g_b0.w, g_b1.x, g_b1.y, g_b1.z, g_b1.w};
            uint32_t o[16];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                // int8 n -> byte n ^ 0x80 = n + 128 in [1, 255] -> f16 via the bias trick with bias 1024 + 128
                const uint32_t w = (bvalid ? bw[k] : 0u) ^ 0x80808080u;
                o[2 * k] = u16x2_to_h2(bytes01_to_u16x2(w), 1152.0f);
                o[2 * k + 1] = u16x2_to_h2(bytes23_to_u16x2(w), 1152.0f);
            }
            uint4 * db = (uint4 *) (Bt + bcol * MQ_AS + 32 * bq * 2);
            db[0] = make_uint4(o[0], o[1], o[2], o[3]);
            db[1] = make_uint4(o[4], o[5], o[6], o[7]);
            db[2] = make_uint4(o[8], o[9], o[10], o[11]);
            db[3] = make_uint4(o[12], o[13], o[14], o[15]);
            if (h == 0 && bq == 0) {
                dyv[bcol] = bvalid ? ycol[sb].d : 0.0f;
                if constexpr (!Q6) {
                    uint32_t pb[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) pb[j] = bvalid ? ((const uint32_t *) ycol[sb].bsums)[j] : 0u;  // already f16
                    uint4 * dbm = (uint4 *) (Bm + bcol * MQ_MS);
                    dbm[0] = make_uint4(pb[0], pb[1], pb[2], pb[3]);
                    dbm[1] = make_uint4(pb[4], pb[5], pb[6], pb[7]);
                }
            }
        }
    };

    const int fr = lane & 31, kg = lane >> 5;
    const char * pa1 = A1 + (nslab * 32 + fr) * MQ_AS + kg * 16;
    const char * pa2 = A2 + (nslab * 32 + fr) * MQ_AS + kg * 16;
    const char * pb[2] = {Bt + (mhalf * 64 + fr) * MQ_AS + kg * 16, Bt + (mhalf * 64 + 32 + fr) * MQ_AS + kg * 16};

    issue_loads(sb_lo, 0);
    const int trips = sb_hi * 2;
    for (int it = sb_lo * 2; it < trips; ++it) {
        const int sb = it >> 1, h = it & 1;
        __syncthreads();  // everyone finished reading the previous tiles
        stage(sb, h);
        __syncthreads();
        if (it + 1 < trips) issue_loads((it + 1) >> 1, (it + 1) & 1);  // in flight during the MFMAs below
#pragma unroll
        for (int ks = 0; ks < MQ_BK / 16; ++ks) {
            const half8 fa1 = *(const half8 *) (pa1 + ks * 32);
            half8 fa2;
            if constexpr (Q6) fa2 = *(const half8 *) (pa2 + ks * 32);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const half8 fb = *(const half8 *) (pb[t] + ks * 32);
                const bool first = (h == 0 && ks == 0);
                acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa1, fb, first ? zero : acc1[t], 0, 0, 0);
                if constexpr (Q6) acc2[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa2, fb, first ? zero : acc2[t], 0, 0, 0);
            }
        }
        if (h == 1) {
            if constexpr (!Q6) {
                const half8 fam = *(const half8 *) (Am + (nslab * 32 + fr) * MQ_MS + kg * 16);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const half8 fbm = *(const half8 *) (Bm + (mhalf * 64 + t * 32 + fr) * MQ_MS + kg * 16);
                    acc2[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fam, fbm, zero, 0, 0, 0);
                }
            }
            // fold the exact integer sums of this super-block into the f32 result
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float dy = dyv[mhalf * 64 + t * 32 + fr];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = nslab * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                    const float2 sd = dd[i];
                    float v;
                    if constexpr (Q6) v = sd.x * (16.0f * acc1[t][r] + acc2[t][r]);
                    else v = sd.x * acc1[t][r] - sd.y * acc2[t][r];
                    C[t][r] += dy * v;
                }
            }
        }
    }
    // ---- store: lane holds column (token) j and 4 runs of 4 consecutive rows
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int m = m0 + mhalf * 64 + t * 32 + fr;
        if (m >= a.M) continue;
        float * out = a.ksplit > 1 ? a.part + ((size_t) blockIdx.y * a.M + m) * a.N : a.dst + (size_t) m * a.dst_stride;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = n0 + nslab * 32 + 8 * g + 4 * kg;
            if (n + 3 < a.N && ((((uintptr_t) (out + n)) & 15) == 0)) {
                *(float4 *) (out + n) = make_float4(C[t][4 * g], C[t][4 * g + 1], C[t][4 * g + 2], C[t][4 * g + 3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n + r < a.N) out[n + r] = C[t][4 * g + r];
            }
        }
    }
}

bool mmq_supported(int type, int64_t K, int64_t N, int64_t M) {
    (void) N;
    return (type == GGML_TYPE_Q4_K || type == GGML_TYPE_Q5_K || type == GGML_TYPE_Q6_K) && (K % 256) == 0 && M >= 9;
}
size_t mmq_workspace_bytes(int type, int64_t K, int64_t N, int64_t M, bool skinny) {
    const int ks = mmq_pick_ksplit(K, N, M, skinny, type);
    return ks > 1 ? (size_t) ks * (size_t) M * (size_t) N * sizeof(float) : 0;
}

template <int QT> static void launch_mmq_t(hipStream_t s, const mmq_args & a) {
    const bool q6 = QT == 6;
    const size_t lds = (size_t) MQ_BN * MQ_AS * (q6 ? 2 : 1) + (size_t) MQ_BM * MQ_AS + (q6 ? 0 : (size_t) (MQ_BN + MQ_BM) * MQ_MS) + MQ_BN * sizeof(float2) + MQ_BM * sizeof(float) + 64;
    static std::atomic<uint32_t> lds_raised{0};  // one bit per device (common.h: ensure_dyn_lds)
    (void) ensure_dyn_lds((const void *) k_mmq<QT>, lds, lds_raised);  // on failure the launch below fails and graph_compute reports it
    const unsigned grid = (unsigned) (((a.n_panels + 7) / 8) * 8 * a.m_tiles);
    hipLaunchKernelGGL((k_mmq<QT>), dim3(grid, (unsigned) a.ksplit), dim3(512), lds, s, a);
}

void launch_mmq(hipStream_t s, int type, const uint8_t * W, int64_t w_nb1, int K, int N, int M, const void * act_q8k, float * dst, int64_t dst_stride, int ksplit, float * part) {
    mmq_args a;
    a.W = W;
    a.w_nb1 = w_nb1;
    a.K = K;
    a.N = N;
    a.M = M;
    a.act = (const q8k_dev *) act_q8k;
    a.dst = dst;
    a.dst_stride = dst_stride;
    a.n_panels = (N + MQ_BN - 1) / MQ_BN;
    a.m_tiles = (M + MQ_BM - 1) / MQ_BM;
    a.ksplit = std::max(1, ksplit);
    a.part = part;
    if (type == GGML_TYPE_Q4_K) launch_mmq_t<4>(s, a);
    else if (type == GGML_TYPE_Q5_K) launch_mmq_t<5>(s, a);
    else launch_mmq_t<6>(s, a);
    if (a.ksplit > 1) launch_splitk_reduce(s, part, a.ksplit, M, N, dst, dst_stride, nullptr, 0);
}

MI_TU_TOUCH(mmq)

}  // namespace mi355x
