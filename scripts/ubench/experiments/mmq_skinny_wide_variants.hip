// mmq_skinny.hip — quantised weights x 2..32 activation columns (the decode step of a continuous batch, `-np` > 1; SURVEY.md
// §8a row a6 at small M) as a WEIGHT-STREAMING kernel on the integer matrix cores.
//
// The prompt-batch GEMM (mmq_i8.hip) tiles N x M and re-stages the weights per column tile; at M <= 32 there is one column
// tile, every weight byte is used exactly once, and the launch is an HBM stream like the mat-vec — only with 32 dot products
// per weight instead of one, which v_dot4 cannot keep up with (35 us for the 66 MB gate/up pair = 1.9 TB/s).
//
//   * work unit = (32 weight rows) x (one 256-value super-block) — 4.6 KB of Q4_K.  A workgroup (8 waves) owns one 32-row
//     tile and a range of super-blocks; wave w takes units w, w+8, ..  and the eight partial 32x32 results meet in LDS at the
//     end (fixed order).  The grid is one workgroup per CU walking the tiles with a stride: the first units of the next tile
//     are already being fetched while the current one is reduced and stored (with one short-lived workgroup per tile the
//     memory latency was exposed once per tile: 30 us for the 66 MB gate/up pair).  No K split across workgroups unless the matrix alone cannot fill the chip (then blockIdx.y splits K
//     and the partial results go to the same [ksplit][M][N] workspace mmq_i8.hip uses — its consumers sum them).
//   * both operands are fetched as ROW-CONTIGUOUS 16-byte pieces (runs of 9..20 lanes; scripts/ubench/ta_probe.hip: 23-34
//     clocks per wave-instruction, against 65 when every lane sits in another row, which the MFMA operand layout would ask for)
//     into a wave-private LDS area, and read back in the operand layout with conflict-free ds_read_b128 (row strides of
//     4 * odd dwords).  The next unit's global loads are issued into registers before the current unit is multiplied; the LDS
//     area is private to the wave, so the loop has no barrier at all.
//   * the WEIGHTS are the matrix-core B operand and the activations the A operand: a lane's 16 accumulators are then 16
//     tokens of ONE weight row, and that row's sub-block scale — a value the lane decoded from its own row header — folds the
//     int32 block sum with one v_mad_i32_i24 per accumulator (the reverse assignment needs a per-register scale, i.e. an LDS
//     read or a digit decomposition per MFMA).  sum_j sc_j * (q . y)_j is the integer ggml-cpu computes (ggml_vec_dot_q4_K_q8_K,
//     /root/reference/llama.cpp/ggml/src/ggml-cpu/quants.c), exact in int32.
//   * mins (Q4_K / Q5_K) and the -32 offset of Q6_K are sum_j m_j * bsum_j over the activation block's 16-value sums — one
//     f16 MFMA per unit on the f16 bsums the quantisers store (exact: |.| < 2^24), as in mmq_i8.hip.
//   * Q6_K blocks are 210 bytes at 2-byte alignment: the unit is fetched as 4-byte-aligned 16-byte pieces (dwordx4 only needs
//     dword alignment) and shifted by 0 / 2 bytes on the way into LDS (v_alignbyte + the neighbour lane's first dword).
// Forms in this file: the K-parallel kernel above for any shape (k_mmq_skinny<QA, QB, EPI>; QA != QB: matrices of two formats in one
// launch, one pass per format; EPI: rope + KV-cache stores in the epilogue), the tile-parallel LDS-DMA kernels for the gate/up pair
// (k_mmq_skinny_tp, four waves; k_mmq_skinny_tp8, two waves per tile: round 3), and the same unit turned towards prompt batches (k_mmq_wide).
#include <algorithm>
#include <type_traits>

#include "dev_util.h"
#include "kernels.h"
#include "mmq_args.h"
#include "mmvq_types.h"

namespace mi355x {

typedef _Float16 half8s __attribute__((ext_vector_type(8)));
typedef float float16s __attribute__((ext_vector_type(16)));
typedef int int4s __attribute__((ext_vector_type(4)));
typedef int int2s __attribute__((ext_vector_type(2)));
typedef int int16s __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2s __attribute__((ext_vector_type(2)));

constexpr int SK_NW = 8;            // waves per workgroup
constexpr int SK_BTOK = 304;        // LDS bytes per token of the activation unit: 256 qs + 32 bsums (f16) + 16 pad (76 dwords = 4 * 19)
constexpr int SK_B_BYTES = 32 * SK_BTOK;
template <int QT> struct sk_fmt;
template <> struct sk_fmt<4> { static constexpr int BYTES = 144, PIECES = 9, ROW = 144; };   // 36 dwords = 4 * 9
template <> struct sk_fmt<5> { static constexpr int BYTES = 176, PIECES = 11, ROW = 176; };  // 44 dwords = 4 * 11
template <> struct sk_fmt<6> { static constexpr int BYTES = 210, PIECES = 14, ROW = 240; };  // 14 pieces + 16 pad: 60 dwords = 4 * 15
template <int QT> constexpr int sk_wave_lds() { return 32 * sk_fmt<QT>::ROW + SK_B_BYTES + 128; }

__device__ __forceinline__ uint32_t sk_pack_h2(const float a, const float b) {
    typedef _Float16 half2s __attribute__((ext_vector_type(2)));
    const half2s h = {(_Float16) a, (_Float16) b};
    return __builtin_bit_cast(uint32_t, h);
}

// four bytes of a dword times a small digit with ONE v_pk_mul_lo_u16 (every byte product stays below 256: no carry crosses a byte)
__device__ __forceinline__ uint32_t sk_pk_mul(const uint32_t a, const uint32_t b) {
    typedef unsigned short ushort2s __attribute__((ext_vector_type(2)));
    const ushort2s r = __builtin_bit_cast(ushort2s, a) * __builtin_bit_cast(ushort2s, b);
    return __builtin_bit_cast(uint32_t, r);
}
// byte K of w replicated into both 16-bit lanes: 0x00bb00bb
template <int K> __device__ __forceinline__ uint32_t sk_rep_byte(const uint32_t w) { return __builtin_amdgcn_perm(0u, w, 0x0c000c00u | (uint32_t) K | ((uint32_t) K << 16)); }

// One unit (32 weight rows x one super-block x 32 tokens) of Q4_K / Q5_K: arow = this lane's row in the staged weight unit ([row][block
// bytes]), btok = this lane's token in the staged activation unit ([token][256 qs | 32 bsums f16 | ..]), dAs = the 32 activation block
// scales; g = lane >> 5 (the k-group of the MFMA operands).  Adds the unit's contribution to acc (register i: token (i & 3) + 8 (i >> 2) + 4 g).
// The 6-bit sub-block scale cannot ride in an int8 operand (sc * q reaches 945), and folding each MFMA's int32 result with the
// lane's scale costs 16 v_mul_i32_i24 per MFMA — measured at ~6.8 clocks each where an fp32 fma issues in 2
// (scripts/ubench/valu_probe.hip): 128 of them were half of a unit's time.  So the scale is split into digits, sc = 8 s1 + s0 (Q4_K:
// q <= 15, digits <= 7, products <= 105) or 16 s2 + 4 s1 + s0 (Q5_K: q <= 31, digits <= 3, products <= 93), one int8 weight operand per
// digit (four nibbles x digit = ONE v_pk_mul_lo_u16), and the digit planes accumulate over all eight sub-blocks INSIDE the matrix
// cores (C operand): sum_j sc_j (q . y)_j = 8 P1 + P0, the same integers ggml-cpu computes, with no per-sub-block fold at all.
template <int QT> __device__ __forceinline__ void sk_unit_k45(const char * __restrict__ arow, const char * __restrict__ btok, const float * __restrict__ dAs, const int g, float (&acc)[16]) {
    constexpr int NP = QT == 4 ? 2 : 3;              // digit planes
    constexpr uint32_t DM = QT == 4 ? 0x07070707u : 0x03030303u;
    constexpr int DS = QT == 4 ? 3 : 2;              // bits per digit
    const int16s zeroi = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const float16s zerof = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const uint4 hdr = *(const uint4 *) arow;
    const float d = h2f((uint16_t) (hdr.x & 0xFFFF)), dmin = h2f((uint16_t) (hdr.x >> 16));
    // the 12 packed bytes: scales 0..3 / 4..7 and mins 0..3 / 4..7 as one byte each
    const uint32_t slo = hdr.y & 0x3F3F3F3Fu, shi = (hdr.w & 0x0F0F0F0Fu) | ((hdr.y >> 2) & 0x30303030u);
    const uint32_t mlo = hdr.z & 0x3F3F3F3Fu, mhi = ((hdr.w >> 4) & 0x0F0F0F0Fu) | ((hdr.z >> 2) & 0x30303030u);
    // mins: sum_j m_j * (sum of the 32 activations of sub-block j) = sum over the sixteen 16-value bsums with m_{k/2}; this lane's k-group
    // covers bsums 8g .. 8g+7, i.e. mins 4g .. 4g+3, each twice: 0x6400 | n == 1024 + n in f16, minus 1024
    float16s ms;
    {
        typedef _Float16 half2s __attribute__((ext_vector_type(2)));
        const uint32_t msrc = g ? mhi : mlo;
        const half2s k1024 = {(_Float16) 1024.0f, (_Float16) 1024.0f};
        uint4 mfu;
        mfu.x = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2s, sk_rep_byte<0>(msrc) | 0x64006400u) - k1024);
        mfu.y = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2s, sk_rep_byte<1>(msrc) | 0x64006400u) - k1024);
        mfu.z = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2s, sk_rep_byte<2>(msrc) | 0x64006400u) - k1024);
        mfu.w = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2s, sk_rep_byte<3>(msrc) | 0x64006400u) - k1024);
        const half8s bsf = *(const half8s *) (btok + 256 + 16 * g);
        ms = __builtin_amdgcn_mfma_f32_32x32x16_f16(bsf, __builtin_bit_cast(half8s, mfu), zerof, 0, 0, 0);
    }
    // digit dwords: plane n of scales 0..3 / 4..7
    uint32_t dlo[NP], dhi[NP];
#pragma unroll
    for (int n = 0; n < NP; ++n) {
        dlo[n] = (slo >> (DS * n)) & DM;
        dhi[n] = (shi >> (DS * n)) & DM;
    }
    uint4 qh = make_uint4(0, 0, 0, 0);
    if constexpr (QT == 5) qh = *(const uint4 *) (arow + 16 + 16 * g);
    int16s pl[NP];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const uint4 raw = *(const uint4 *) (arow + (QT == 5 ? 48 : 16) + 32 * p + 16 * g);
        uint32_t wlo[4] = {raw.x & 0x0F0F0F0Fu, raw.y & 0x0F0F0F0Fu, raw.z & 0x0F0F0F0Fu, raw.w & 0x0F0F0F0Fu};
        uint32_t whi[4] = {(raw.x >> 4) & 0x0F0F0F0Fu, (raw.y >> 4) & 0x0F0F0F0Fu, (raw.z >> 4) & 0x0F0F0F0Fu, (raw.w >> 4) & 0x0F0F0F0Fu};
        if constexpr (QT == 5) {
            const uint32_t h4[4] = {qh.x, qh.y, qh.z, qh.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                wlo[k] |= ((h4[k] >> (2 * p)) & 0x01010101u) << 4;
                whi[k] |= ((h4[k] >> (2 * p + 1)) & 0x01010101u) << 4;
            }
        }
        const int4s y0 = *(const int4s *) (btok + 64 * p + 16 * g);
        const int4s y1 = *(const int4s *) (btok + 64 * p + 32 + 16 * g);
#pragma unroll
        for (int n = 0; n < NP; ++n) {
            // sub-blocks 2p (low nibbles) and 2p + 1 (high nibbles): their digit n, replicated into both 16-bit lanes
            const uint32_t src = p < 2 ? dlo[n] : dhi[n];
            const uint32_t e0 = (p & 1) ? sk_rep_byte<2>(src) : sk_rep_byte<0>(src), e1 = (p & 1) ? sk_rep_byte<3>(src) : sk_rep_byte<1>(src);
            int4s a0, a1;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                a0[k] = (int) sk_pk_mul(wlo[k], e0);
                a1[k] = (int) sk_pk_mul(whi[k], e1);
            }
            pl[n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(y0, a0, p == 0 ? zeroi : pl[n], 0, 0, 0);
            pl[n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(y1, a1, pl[n], 0, 0, 0);
        }
    }
    // ---- fold the unit
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 dy = *(const float4 *) (dAs + 8 * q + 4 * g);
        const float dyv[4] = {dy.x, dy.y, dy.z, dy.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * q + r;
            int isum;
            if constexpr (QT == 4) isum = (pl[1][i] << 3) + pl[0][i];
            else isum = (pl[2][i] << 4) + (pl[1][i] << 2) + pl[0][i];
            const float v = __builtin_fmaf(-dmin, ms[i], d * (float) isum);
            acc[i] = __builtin_fmaf(dyv[r], v, acc[i]);
        }
    }
}

// EPI: the results of the attention projections of a batch are rotated and stored by this launch (mmq_epi) — a separate instantiation,
// because the accurate cosf / sinf of the rotation bring a scratch frame that every launch of the kernel would otherwise pay for
// (measured: 4.27 -> 4.45 ms per -np 32 step with the epilogue compiled into the one kernel, used or not)
// MIXED: the matrices of the launch are stored in two formats (wq / wk as Q4_K next to a Q6_K wv: what Q4_K_M does to half the layers) —
// this pass serves the items whose matrix is stored as QT and walks past the others (k_mmq_skinny_mix runs one pass per format)
// (the pass is a lambda INSIDE the kernel: as a device function taking the argument block, by value or by reference, it made the compiler
// copy the block into scratch memory — 408 bytes per lane — and read every pointer of the launch from there)
template <int QA, int QB, bool EPI>
__global__ void __launch_bounds__(SK_NW * 64, 1) k_mmq_skinny(const mmq8_args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool MIXED = QA != QB;
    auto pass = [&](auto qt_tag) __attribute__((always_inline)) {
    constexpr int QT = decltype(qt_tag)::value;
    typedef sk_fmt<QT> F;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = lane & 31, g = lane >> 5;
    const int nblk = a.K / 256;
    // work items = (32-row tile, K slice); workgroup j serves items j, j + gridDim.x, ..  (one resident workgroup per CU: the loads
    // of an item's first units are in flight while the previous item is reduced and stored)
    const int ksl = 31 - __builtin_clz((unsigned) a.ksplit);  // ksplit is a power of two (launcher)
    const int n_items = a.n_panels << ksl;
    // the row stride is the same for every matrix of one format (checked by the launcher)
    const int w_nb1 = !MIXED || a.mat[0].qt == QT ? (int) a.mat[0].w_nb1 : (a.mat[1].qt == QT ? (int) a.mat[1].w_nb1 : (int) a.mat[2].w_nb1);

    char * const As = smem + wave * sk_wave_lds<QT>();
    char * const Bs = As + 32 * F::ROW;
    float * const dAs = (float *) (Bs + SK_B_BYTES);

    // ---- fetch roles.  Every offset is linear in the fetch index u, so a handful of registers address the whole unit:
    //   weights, last 8 pieces of a row (Q4_K / Q5_K: the 128 qs bytes): u = 0..3, row = (lane >> 3) + 8 u, piece = HEAD + (lane & 7)
    //   weights, first HEAD pieces of a row (header, Q5_K: + qh): piece index lane + 64 u over 32 * HEAD
    //   Q6_K: a 256-byte window per row (16 pieces; the block is 210 bytes, the tail belongs to the next block): u = 0..7,
    //         row = (lane >> 4) + 4 u, piece = lane & 15
    //   activations: qs u = 0..7, token = (lane >> 4) + 4 u, piece = lane & 15; bsums: token = lane >> 1, piece 16 + (lane & 1)
    // Token groups of four beyond M are not fetched at all; within the last group tokens >= M are fetched like the others (the caller's
    // activation area holds 32 tokens' worth of bytes) and their results never stored.
    constexpr int HEAD = QT == 6 ? 0 : F::PIECES - 8;
    constexpr int NLT = QT == 6 ? 8 : 4;
    constexpr int NLH = (32 * HEAD + 63) / 64;
    constexpr int NLA = NLT + NLH;
    const int at_off = QT == 6 ? (lane >> 4) * w_nb1 + (lane & 15) * 16 : (lane >> 3) * w_nb1 + (HEAD + (lane & 7)) * 16;
    const int at_lds = QT == 6 ? (lane >> 4) * F::ROW + (lane & 15) * 16 : (lane >> 3) * F::ROW + (HEAD + (lane & 7)) * 16;
    const int at_step = (QT == 6 ? 4 : 8) * w_nb1;
    constexpr int at_lstep = (QT == 6 ? 4 : 8) * F::ROW;
    int ah_off[NLH > 0 ? NLH : 1], ah_lds[NLH > 0 ? NLH : 1];
#pragma unroll
    for (int u = 0; u < NLH; ++u) {
        const int pi = min(lane + 64 * u, 32 * HEAD - 1);
        const int r = pi / (HEAD > 0 ? HEAD : 1), pc = pi - r * HEAD;
        ah_off[u] = r * w_nb1 + pc * 16;
        ah_lds[u] = r * F::ROW + pc * 16;
    }
    const bool ah_last_live = ((32 * HEAD) % 64) == 0 || lane < ((32 * HEAD) % 64);
    const char * const act_base = (const char *) a.act;
    const int tok_bytes = nblk * (int) sizeof(q8k_dev);
    const int bq_off = (lane >> 4) * tok_bytes + (lane & 15) * 16, bq_lds = (lane >> 4) * SK_BTOK + (lane & 15) * 16;
    const int bq_step = 4 * tok_bytes;
    const int bs_off = (lane >> 1) * tok_bytes + 256 + (lane & 1) * 16, bs_lds = (lane >> 1) * SK_BTOK + 256 + (lane & 1) * 16;
    const int d_off = row * tok_bytes + 304;  // q8k_dev::d

    const int16s zeroi = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const float16s zerof = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

#define MAT_OF(t) ((a.n_mat > 2 && (t) >= a.mat[2].panel0) ? 2 : ((a.n_mat > 1 && (t) >= a.mat[1].panel0) ? 1 : 0))
#define MAT_SEL(mi, f) ((mi) == 0 ? a.mat[0].f : ((mi) == 1 ? a.mat[1].f : a.mat[2].f))
    // first row of the tile of an item (wave-uniform: scalar registers)
    auto item_rows = [&](const int item) -> const uint8_t * {
        const int tile = item >> ksl, mi = MAT_OF(tile);
        return MAT_SEL(mi, W) + (size_t) (tile - MAT_SEL(mi, panel0)) * 32 * (size_t) w_nb1;
    };
    auto item_lo = [&](const int item) { return ((item & (a.ksplit - 1)) * nblk) >> ksl; };
    auto item_hi = [&](const int item) { return (((item & (a.ksplit - 1)) + 1) * nblk) >> ksl; };
    auto mine = [&](const int item) {  // (MIXED) is this item's matrix stored as QT?
        if constexpr (!MIXED) return true;
        const int mi = MAT_OF(item >> ksl);
        return MAT_SEL(mi, qt) == QT;
    };

    u32x4s ga[NLA], gb[9] = {};
    float gd = 0.0f;
    // (a straight-line macro, not a lambda: register arrays captured by reference end up in scratch memory)
#define SK_ISSUE(rows, sb)                                                                                              \
    {                                                                                                                   \
        const uint8_t * wb = (rows) + (QT == 6 ? (size_t) (sb) * 210 - 2 * ((sb) & 1) : (size_t) (sb) * F::BYTES);      \
        _Pragma("unroll") for (int u = 0; u < NLT; ++u) ga[u] = *(const u32x4s *) (wb + at_off + u * at_step);          \
        _Pragma("unroll") for (int u = 0; u < NLH; ++u) ga[NLT + u] = *(const u32x4s *) (wb + ah_off[u]);               \
        const char * ab = act_base + (size_t) (sb) * sizeof(q8k_dev);                                                   \
        _Pragma("unroll") for (int u = 0; u < 8; ++u)                                                                   \
            if (4 * u < a.M) gb[u] = *(const u32x4s *) (ab + bq_off + u * bq_step); /* tokens 4u .. 4u+3: skipped beyond M */ \
        gb[8] = *(const u32x4s *) (ab + bs_off);                                                                        \
        gd = *(const float *) (ab + d_off);                                                                             \
    }
    // the unit whose loads are in flight: (nx_item, nx_sb); wave w takes super-blocks lo + w, lo + w + 8, .. of every item
    int nx_item = blockIdx.x, nx_sb = 0;
#define SK_NORMALISE()                                                                 \
    while (nx_item < n_items && (nx_sb >= item_hi(nx_item) || !mine(nx_item))) {       \
        nx_item += gridDim.x;                                                          \
        if (nx_item < n_items) nx_sb = item_lo(nx_item) + wave;                        \
    }
    if (nx_item < n_items) nx_sb = item_lo(nx_item) + wave;
    SK_NORMALISE()
    if (nx_item < n_items) SK_ISSUE(item_rows(nx_item), nx_sb)

    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        if (!mine(item)) continue;
        float acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
        while (nx_item == item) {
            const int sb = nx_sb;
            // ---- registers -> the wave's LDS area
            if constexpr (QT == 6) {
                const int sh = 2 * (sb & 1);  // the block starts `sh` bytes into the fetched window
#pragma unroll
                for (int u = 0; u < NLT; ++u) {
                    const uint32_t nx = (uint32_t) __shfl_down((int) ga[u].x, 1);  // first dword of the row's next piece
                    u32x4s o;
                    o.x = __builtin_amdgcn_alignbyte(ga[u].y, ga[u].x, sh);
                    o.y = __builtin_amdgcn_alignbyte(ga[u].z, ga[u].y, sh);
                    o.z = __builtin_amdgcn_alignbyte(ga[u].w, ga[u].z, sh);
                    o.w = __builtin_amdgcn_alignbyte(nx, ga[u].w, sh);
                    if ((lane & 15) < 14) *(u32x4s *) (As + at_lds + u * at_lstep) = o;  // 14 pieces cover the 210 bytes
                }
            } else {
#pragma unroll
                for (int u = 0; u < NLT; ++u) *(u32x4s *) (As + at_lds + u * at_lstep) = ga[u];
#pragma unroll
                for (int u = 0; u < NLH; ++u)
                    if (u + 1 < NLH || ah_last_live) *(u32x4s *) (As + ah_lds[u]) = ga[NLT + u];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (4 * u < a.M) *(u32x4s *) (Bs + bq_lds + u * 4 * SK_BTOK) = gb[u];
            *(u32x4s *) (Bs + bs_lds) = gb[8];
            if (g == 0) dAs[row] = gd;
            // ---- the next unit's loads (of this item or of the workgroup's next one) fly while this one is multiplied
            nx_sb += SK_NW;
            SK_NORMALISE()
            if (nx_item < n_items) SK_ISSUE(item_rows(nx_item), nx_sb)

            const char * const arow = As + row * F::ROW;   // this lane's weight row (B operand: n = row)
            const char * const btok = Bs + row * SK_BTOK;  // this lane's token (A operand: m = token)
            if constexpr (QT == 4 || QT == 5) {
                sk_unit_k45<QT>(arow, btok, dAs, g, acc);
            } else {
                // (the fold stays on v_mad_i32_i24 here although it issues at ~6.8 clocks (scripts/ubench/valu_probe.hip): the fp32 form of it —
                // cvt + fma, 2 clocks each — was built and made the compiler park the prefetched unit in scratch (156 bytes per lane))
                int16s isum = zeroi;
                float16s ms;
                float d;
                const uint4 scb = *(const uint4 *) (arow + 192);
                d = h2f(*(const uint16_t *) (arow + 208));
                const uint32_t scw[4] = {scb.x, scb.y, scb.z, scb.w};
                int sc[16];
#pragma unroll
                for (int s = 0; s < 16; ++s) sc[s] = (int) (int8_t) (scw[s >> 2] >> (8 * (s & 3)));
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        // l = 16 t + 8 g .. + 7 of half h: ql[64h + l], ql[64h + 32 + l], qh[32h + l] carry the four sub-blocks 8h + 2k + t
                        const uint2 qa = *(const uint2 *) (arow + 64 * h + 16 * t + 8 * g);
                        const uint2 qb = *(const uint2 *) (arow + 64 * h + 32 + 16 * t + 8 * g);
                        const uint2 qc = *(const uint2 *) (arow + 128 + 32 * h + 16 * t + 8 * g);
                        int2s v[4];
                        v[0][0] = (int) ((qa.x & 0x0F0F0F0Fu) | ((qc.x << 4) & 0x30303030u));
                        v[0][1] = (int) ((qa.y & 0x0F0F0F0Fu) | ((qc.y << 4) & 0x30303030u));
                        v[1][0] = (int) ((qb.x & 0x0F0F0F0Fu) | ((qc.x << 2) & 0x30303030u));
                        v[1][1] = (int) ((qb.y & 0x0F0F0F0Fu) | ((qc.y << 2) & 0x30303030u));
                        v[2][0] = (int) (((qa.x >> 4) & 0x0F0F0F0Fu) | (qc.x & 0x30303030u));
                        v[2][1] = (int) (((qa.y >> 4) & 0x0F0F0F0Fu) | (qc.y & 0x30303030u));
                        v[3][0] = (int) (((qb.x >> 4) & 0x0F0F0F0Fu) | ((qc.x >> 2) & 0x30303030u));
                        v[3][1] = (int) (((qb.y >> 4) & 0x0F0F0F0Fu) | ((qc.y >> 2) & 0x30303030u));
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int s = 8 * h + 2 * k + t;
                            const int2s y = *(const int2s *) (btok + 16 * s + 8 * g);
                            const int16s tk = __builtin_amdgcn_mfma_i32_32x32x16_i8(__builtin_bit_cast(long, y), __builtin_bit_cast(long, v[k]), zeroi, 0, 0, 0);
#pragma unroll
                            for (int i = 0; i < 16; ++i) isum[i] += __mul24(tk[i], sc[s]);
                        }
                    }
                // the codes are q + 32: subtract 32 * sum_s sc_s * bsum_s (one f16 MFMA, exact integers)
                const half8s bsf = *(const half8s *) (btok + 256 + 16 * g);
                uint4 sfu;
                sfu.x = g ? sk_pack_h2((float) sc[8], (float) sc[9]) : sk_pack_h2((float) sc[0], (float) sc[1]);
                sfu.y = g ? sk_pack_h2((float) sc[10], (float) sc[11]) : sk_pack_h2((float) sc[2], (float) sc[3]);
                sfu.z = g ? sk_pack_h2((float) sc[12], (float) sc[13]) : sk_pack_h2((float) sc[4], (float) sc[5]);
                sfu.w = g ? sk_pack_h2((float) sc[14], (float) sc[15]) : sk_pack_h2((float) sc[6], (float) sc[7]);
                ms = __builtin_amdgcn_mfma_f32_32x32x16_f16(bsf, __builtin_bit_cast(half8s, sfu), zerof, 0, 0, 0);
                // fold the unit: tokens of register i are (i & 3) + 8 (i >> 2) + 4 g
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 dy = *(const float4 *) (dAs + 8 * q + 4 * g);
                    const float dyv[4] = {dy.x, dy.y, dy.z, dy.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 4 * q + r;
                        const float v = d * (float) (isum[i] - 32 * (int) ms[i]);
                        acc[i] = __builtin_fmaf(dyv[r], v, acc[i]);
                    }
                }
            }
        }

        // ---- the eight K slices of the tile meet in LDS: each wave leaves its 32 x 32 partial [token][row] at the start of its
        // own area (no other wave touches that), then every thread sums two outputs over the waves in a fixed order
        float * const own = (float *) As;
#pragma unroll
        for (int i = 0; i < 16; ++i) own[((i & 3) + 8 * (i >> 2) + 4 * g) * 32 + row] = acc[i];
        __syncthreads();
        const int tile = item >> ksl, ky = item & (a.ksplit - 1), mi = MAT_OF(tile);
        const int mN = MAT_SEL(mi, N);
        const int n0 = (tile - MAT_SEL(mi, panel0)) * 32;
        float * const m_part = MAT_SEL(mi, part);
        float * const m_dst = MAT_SEL(mi, dst);
        const int64_t m_dst_stride = MAT_SEL(mi, dst_stride);
        const float * const m_add = MAT_SEL(mi, add);
        const int64_t m_add_stride = MAT_SEL(mi, add_stride);
#pragma unroll
        for (int q = 0; q < 1024 / (SK_NW * 64); ++q) {
            const int o = tid + q * SK_NW * 64;
            const int tok = o >> 5, n = n0 + (o & 31);
            float v = *(const float *) (smem + o * 4);
#pragma unroll
            for (int w = 1; w < SK_NW; ++w) v += *(const float *) (smem + w * sk_wave_lds<QT>() + o * 4);
            if (a.ksplit > 1) {
                if (tok < a.M) m_part[((size_t) ky * a.M + tok) * mN + n] = v;
                continue;
            }
            const int tk = min(tok, a.M - 1);
            if (m_add) v += m_add[(size_t) tk * m_add_stride + n];
            const int kind = EPI ? (mi == 0 ? a.epi.kind[0] : (mi == 1 ? a.epi.kind[1] : a.epi.kind[2])) : 0;
            if (kind == 0) {
                if (tok < a.M) m_dst[(size_t) tok * m_dst_stride + n] = v;
                continue;
            }
            if constexpr (EPI) {
            // the attention projections of a batch: rotate (q, k) and store (rope(q) as f32, k and v as f16 rows of the cache) — the
            // arithmetic of k_rope_qk_store (ops.hip), element for element
            const float partner = __shfl_xor(v, 1);  // the other element of the rotation pair: rows n ^ 1, adjacent lanes
            char * const e_out = mi == 0 ? a.epi.out[0] : (mi == 1 ? a.epi.out[1] : a.epi.out[2]);
            const int64_t e_nb1 = mi == 0 ? a.epi.nb1[0] : (mi == 1 ? a.epi.nb1[1] : a.epi.nb1[2]);
            const int64_t e_nb2 = mi == 0 ? a.epi.nb2[0] : (mi == 1 ? a.epi.nb2[1] : a.epi.nb2[2]);
            const int head = n / a.epi.head_dim, dd = n - head * a.epi.head_dim;
            float r = v;
            if (kind < 3 && dd < a.epi.n_dims) {
                // (cos, sin) of (token, pair): the table launch_rope_table wrote for this graph run — the chain of multiplies and the accurate
                // cosf / sinf of rope_cos_sin once per step, not once per layer and element (and no scratch frame in this kernel)
                const float2 cssn = *(const float2 *) (a.epi.tab + ((size_t) tk * (a.epi.n_dims >> 1) + (dd >> 1)) * 2);
                const float cs = cssn.x, sn = cssn.y;
                const float x0 = (dd & 1) ? partner : v, x1 = (dd & 1) ? v : partner;
                r = (dd & 1) ? x0 * sn + x1 * cs : x0 * cs - x1 * sn;
            }
            if (tok >= a.M) continue;
            if (kind == 1) *(float *) (e_out + (int64_t) head * e_nb1 + (int64_t) tok * e_nb2 + (int64_t) dd * 4) = r;
            else if (kind == 4) ((uint16_t *) e_out)[a.epi.v_idx[(int64_t) tok * mN + n]] = f2h(r);
            else ((uint16_t *) (e_out + a.epi.idx[tok] * e_nb1))[n] = f2h(r);
            }
        }
        __syncthreads();  // the areas are free for the next item's units
    }
#undef SK_ISSUE
#undef SK_NORMALISE
#undef MAT_SEL
#undef MAT_OF
    };  // pass
    // two formats in one launch: a workgroup's items of the first format, then its items of the second (the wave areas are laid out per
    // format in the same LDS; every served item ends with a barrier, so the second pass's staging cannot overtake the first's last reduction)
    pass(std::integral_constant<int, QA>{});
    if constexpr (MIXED) pass(std::integral_constant<int, QB>{});
}

// ------------------------------------------------------------------------------------------------ tile-parallel form (large N)
// The loop above keeps ONE unit per wave in flight: the next unit's loads sit in registers while the current one is multiplied, and a
// memory round trip under load (~3 us) is longer than a unit's arithmetic — the 66 MB gate/up pair streamed at 2.6 TB/s.  More
// units in flight need LDS, and the 9.7 KB activation unit is what fills it; so when a matrix has enough 128-row groups to occupy
// the chip, four waves take FOUR TILES of the SAME super-block and share its activations:
//   * a workgroup = 4 waves = 4 tiles (128 rows) x the whole K range; step j = super-block j; no partial sums anywhere — a wave
//     owns its tile's results from the first block to the last, and stores them itself;
//   * everything arrives by LDS-DMA (global_load_lds_dwordx4: lane-linear 1 KB pieces, so the LDS image is the [row][piece] /
//     [token][piece] order of the fetch) into rings of TP_NS stages: the activations of step j + TP_NS - 1 (each wave a quarter) and
//     every wave's own weight unit of that step are requested while step j is multiplied — TP_NS - 1 units per wave in flight
//     (three stages, two in flight, were still latency-bound: 25.9 us for the gate/up pair), no staging registers, no ds_write;
//   * one barrier per step: after a wave's own vmcnt says its pieces of step j have landed, the barrier says everybody's have (and
//     that everybody is done reading step j - 1, whose stage the next requests then overwrite).
// hipcc does not count inline-asm memory operations, and in-order vmcnt is the reason for the ring: whatever a wave consumes must be
// older than everything it keeps in flight, so the activations are requested with their step's weights, two steps ahead.
constexpr int TP_NW = 4, TP_NS = 3;  // (five stages measured no faster than three: 19.0 us either way with the arithmetic switched off)
template <int QT> constexpr int tp_a_stage() { return 32 * sk_fmt<QT>::ROW; }
template <int QT> constexpr int tp_lds_bytes() { return TP_NS * (SK_B_BYTES + 128) + TP_NW * TP_NS * tp_a_stage<QT>(); }

// lane i's 16 (4) bytes at g land at LDS[lds + 16 (4) * i]; `lds` is wave-uniform (MI355X guide: M0 is written in the statement that uses it)
__device__ __forceinline__ void tp_dma16(const void * g, const uint32_t lds) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}
__device__ __forceinline__ void tp_dma4(const void * g, const uint32_t lds) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}
// s_waitcnt vmcnt(N): the instruction takes an immediate (a run-time switch over the counts cost 270 clocks per step)
template <int N> __device__ __forceinline__ void tp_wait_c() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int QT>
__global__ void __launch_bounds__(TP_NW * 64, 2) k_mmq_skinny_tp(const mmq8_args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef sk_fmt<QT> F;
    constexpr int NPA = 32 * F::PIECES, NLA = (NPA + 63) / 64;  // weight pieces per unit and the wave-instructions fetching them
    constexpr int NPB = 32 * 19 / TP_NW;                         // activation pieces per wave and step: 152 = 2 x 64 + 24
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row = lane & 31, g = lane >> 5;
    const int nblk = a.K / 256;
    const int w_nb1 = (int) a.mat[0].w_nb1;
    const int tok_bytes = nblk * (int) sizeof(q8k_dev);
    const uint32_t lds0 = (uint32_t) (uintptr_t) smem;  // LDS byte address of the dynamic array
    // LDS: [activation ring: NS x 9728][scale ring: NS x 128][weight rings: wave x NS x stage]
    const uint32_t b_ring = lds0, d_ring = lds0 + TP_NS * SK_B_BYTES, a_ring = d_ring + TP_NS * 128 + wave * TP_NS * tp_a_stage<QT>();
    char * const a_ring_p = smem + TP_NS * (SK_B_BYTES + 128) + wave * TP_NS * tp_a_stage<QT>();

    // fetch roles (offsets from the unit's first byte)
    int a_off[NLA];
#pragma unroll
    for (int u = 0; u < NLA; ++u) {
        const int pi = min(lane + 64 * u, NPA - 1);
        a_off[u] = (pi / F::PIECES) * w_nb1 + (pi % F::PIECES) * 16;
    }
    int b_off[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int pi = wave * NPB + min(lane + 64 * u, NPB - 1);
        b_off[u] = (pi / 19) * tok_bytes + (pi % 19) * 16;
    }
    const int d_off = row * tok_bytes + 304;  // q8k_dev::d (lanes 0..31)
    constexpr int n_ops = NLA + 4;  // vector-memory operations of one step, the same for every wave (all four request the 128 scale bytes)

#define MAT_OF(t) ((a.n_mat > 2 && (t) >= a.mat[2].panel0) ? 2 : ((a.n_mat > 1 && (t) >= a.mat[1].panel0) ? 1 : 0))
#define MAT_SEL(mi, f) ((mi) == 0 ? a.mat[0].f : ((mi) == 1 ? a.mat[1].f : a.mat[2].f))
    // items = groups of four 32-row tiles (panel0 / n_panels count groups here); this workgroup serves group blockIdx.x, + gridDim.x, ..
    const int n_items = a.n_panels;
    const int my_items = n_items > (int) blockIdx.x ? (n_items - 1 - (int) blockIdx.x) / (int) gridDim.x + 1 : 0;
    const int total = my_items * nblk;  // steps
    auto issue = [&](const int item, const int sb, const int slot) {
        const int mi = MAT_OF(item);
        const uint8_t * wb = MAT_SEL(mi, W) + ((size_t) (item - MAT_SEL(mi, panel0)) * 128 + wave * 32) * (size_t) w_nb1 + (size_t) sb * F::BYTES;
        const uint32_t al = a_ring + slot * tp_a_stage<QT>();
#pragma unroll
        for (int u = 0; u < NLA; ++u)
            if (u + 1 < NLA || (NPA % 64) == 0 || lane < (NPA % 64)) tp_dma16(wb + a_off[u], al + u * 1024);
        const char * ab = (const char *) a.act + (size_t) sb * sizeof(q8k_dev);
        const uint32_t bl = b_ring + slot * SK_B_BYTES + wave * NPB * 16;
        tp_dma16(ab + b_off[0], bl);
        tp_dma16(ab + b_off[1], bl + 1024);
        if (lane < NPB - 128) tp_dma16(ab + b_off[2], bl + 2048);
        if (lane < 32) tp_dma4(ab + d_off, d_ring + slot * 128);  // (every wave: identical bytes to the same place, and one wait count for all)
    };

    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    // request steps 0 and 1
    int is_item = blockIdx.x, is_sb = 0, is_s = 0;  // the next step to be requested
#define TP_ISSUE_NEXT()                                          \
    {                                                            \
        issue(is_item, is_sb, is_s % TP_NS);                     \
        ++is_s;                                                  \
        if (++is_sb == nblk) { is_sb = 0; is_item += gridDim.x; } \
    }
#pragma unroll
    for (int k = 0; k < TP_NS - 1; ++k)
        if (k < total) TP_ISSUE_NEXT()
    int cu_item = blockIdx.x, cu_sb = 0;
    for (int s = 0; s < total; ++s) {
        // this wave's pieces of step s have landed (the later steps' may still fly)
        if (total - 1 - s >= TP_NS - 2) tp_wait_c<(TP_NS - 2) * n_ops>();
        else if (total - 1 - s == 1) tp_wait_c<n_ops>();
        else tp_wait_c<0>();
        __syncthreads();                                    // ... everybody's have, and nobody reads step s - 1 any more
        if (s + TP_NS - 1 < total) TP_ISSUE_NEXT()
        const int slot = s % TP_NS;
        sk_unit_k45<QT>(a_ring_p + slot * tp_a_stage<QT>() + row * F::ROW, smem + slot * SK_B_BYTES + row * SK_BTOK, (const float *) (smem + TP_NS * SK_B_BYTES + slot * 128), g, acc);
        if (++cu_sb == nblk) {
            // ---- the tile is complete: lane = weight row, register i = token (i & 3) + 8 (i >> 2) + 4 g
            const int mi = MAT_OF(cu_item);
            const int n = (cu_item - MAT_SEL(mi, panel0)) * 128 + wave * 32 + row;
            float * const m_dst = MAT_SEL(mi, dst);
            const int64_t m_dst_stride = MAT_SEL(mi, dst_stride);
            const float * const m_add = MAT_SEL(mi, add);
            const int64_t m_add_stride = MAT_SEL(mi, add_stride);
            if (m_add) {  // (addends first, then nothing but stores: see k_mmq_wide)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] += m_add[(size_t) min((i & 3) + 8 * (i >> 2) + 4 * g, a.M - 1) * m_add_stride + n];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int tok = (i & 3) + 8 * (i >> 2) + 4 * g;
                if (tok < a.M) m_dst[(size_t) tok * m_dst_stride + n] = acc[i];
                acc[i] = 0.0f;
            }
            cu_sb = 0;
            cu_item += gridDim.x;
        }
    }
#undef TP_ISSUE_NEXT
#undef MAT_SEL
#undef MAT_OF
}

// ------------------------------------------------------------------------------------------------ tile-parallel form, eight waves
// The four-wave form above keeps ONE wave on a SIMD, and a wave's step is a serial instruction stream: 13 requests (each an M0 write
// and a wait for room in the CU's address pipeline), ~270 VALU, 17 MFMAs, the barrier — profiles/r03_np32_pmc_pass1.csv: its waves
// execute an instruction in 50 % of their cycles and wait for memory in 19 %, while the same bytes stream at 4.8 TB/s through four waves
// per CU that do nothing else (scripts/ubench/stride_probe.hip, any walking order).  The launch is bound by that stream of
// instructions, not by HBM.  Here every tile gets TWO waves — wave t and wave t + 4 take the even and the odd super-blocks of tile t —
// so that each SIMD has a second instruction stream to issue from while the first waits for the matrix pipe or the address pipeline;
// a step = two consecutive super-blocks, whose activation blocks are neighbours in memory and land in one stage.  LDS (Q4_K):
// weights 8 waves x 3 stages x 4.5 KB + activations 2 stages x 2 blocks x 9.5 KB = 146.5 KB — the activations (L2 hits) run one step
// ahead, the weights two.  The halves of a tile meet once, at the end: wave t + 4 leaves its sums in the weight stage it consumed last.
// (Q5_K's 5.5 KB units do not fit that way: it stays on the four-wave form.)
constexpr int TP8_NW = 8, TP8_NSA = 3, TP8_NSB = 2;
constexpr int TP8_B_STAGE = 2 * SK_B_BYTES, TP8_D_STAGE = 256;
template <int QT> constexpr int tp8_lds_bytes() { return TP8_NSB * (TP8_B_STAGE + TP8_D_STAGE) + TP8_NW * TP8_NSA * tp_a_stage<QT>(); }

template <int QT>
__global__ void __launch_bounds__(TP8_NW * 64, 1) k_mmq_skinny_tp8(const mmq8_args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef sk_fmt<QT> F;
    constexpr int NPA = 32 * F::PIECES, NLA = (NPA + 63) / 64;  // weight pieces per unit and the wave-instructions fetching them
    constexpr int NPB = 2 * 32 * 19 / TP8_NW;                    // activation pieces per wave and step: 152 = 2 x 64 + 24
    constexpr int A_STAGE = tp_a_stage<QT>();
    static_assert(tp8_lds_bytes<QT>() <= 160 * 1024, "LDS");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = wave & 3, kh = wave >> 2;  // this wave's tile of the group and its parity of super-blocks
    const int row = lane & 31, g = lane >> 5;
    const int nst = a.K / 512;  // steps per item
    const int w_nb1 = (int) a.mat[0].w_nb1;
    const int tok_bytes = (a.K / 256) * (int) sizeof(q8k_dev);
    const uint32_t lds0 = (uint32_t) (uintptr_t) smem;
    // LDS: [activation ring: 2 x (2 blocks x 9728)][scale ring: 2 x 256][weight rings: wave x 3 x stage]
    constexpr int A_RINGS = TP8_NSB * (TP8_B_STAGE + TP8_D_STAGE);
    const uint32_t b_ring = lds0, d_ring = lds0 + TP8_NSB * TP8_B_STAGE, a_ring = lds0 + A_RINGS + wave * TP8_NSA * A_STAGE;
    char * const a_ring_p = smem + A_RINGS + wave * TP8_NSA * A_STAGE;

    int a_off[NLA];
#pragma unroll
    for (int u = 0; u < NLA; ++u) {
        const int pi = min(lane + 64 * u, NPA - 1);
        a_off[u] = (pi / F::PIECES) * w_nb1 + (pi % F::PIECES) * 16;
    }
    // the stage's 1216 activation pieces in LDS order [block][token][19 pieces]; this wave requests 152 consecutive ones
    int b_off[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int pi = wave * NPB + min(lane + 64 * u, NPB - 1);
        const int blk = pi / 608, rem = pi - blk * 608;
        b_off[u] = (rem / 19) * tok_bytes + blk * (int) sizeof(q8k_dev) + (rem % 19) * 16;
    }
    const int d_off = row * tok_bytes + g * (int) sizeof(q8k_dev) + 304;  // q8k_dev::d of (block g, token row): 64 lanes
    constexpr int n_ops_a = NLA;

#define MAT_OF(t) ((a.n_mat > 2 && (t) >= a.mat[2].panel0) ? 2 : ((a.n_mat > 1 && (t) >= a.mat[1].panel0) ? 1 : 0))
#define MAT_SEL(mi, f) ((mi) == 0 ? a.mat[0].f : ((mi) == 1 ? a.mat[1].f : a.mat[2].f))
    const int n_items = a.n_panels;
    const int my_items = n_items > (int) blockIdx.x ? (n_items - 1 - (int) blockIdx.x) / (int) gridDim.x + 1 : 0;
    const int total = my_items * nst;  // steps
    auto issue_a = [&](const int item, const int j, const int slot) {
        const int mi = MAT_OF(item);
        const uint8_t * wb = MAT_SEL(mi, W) + ((size_t) (item - MAT_SEL(mi, panel0)) * 128 + tile * 32) * (size_t) w_nb1 + (size_t) (2 * j + kh) * F::BYTES;
        const uint32_t al = a_ring + slot * A_STAGE;
#pragma unroll
        for (int u = 0; u < NLA; ++u)
            if (u + 1 < NLA || (NPA % 64) == 0 || lane < (NPA % 64)) tp_dma16(wb + a_off[u], al + u * 1024);
    };
    auto issue_b = [&](const int j, const int slot) {
        const char * ab = (const char *) a.act + (size_t) (2 * j) * sizeof(q8k_dev);
        const uint32_t bl = b_ring + slot * TP8_B_STAGE + wave * NPB * 16;
        tp_dma16(ab + b_off[0], bl);
        tp_dma16(ab + b_off[1], bl + 1024);
        if (lane < NPB - 128) tp_dma16(ab + b_off[2], bl + 2048);
        tp_dma4(ab + d_off, d_ring + slot * TP8_D_STAGE);  // (every wave: identical bytes to the same place, and one wait count for all)
    };

    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    // request cursors: weights run two steps ahead, activations one; queue order per step [activations s + 1, weights s + 2] — what
    // step s + 1 consumes is then older than the one thing that may still fly when it starts (in-order vmcnt)
    int ia_item = blockIdx.x, ia_j = 0, na = 0, sa = 0;  // next weight step to request (item, step of the item, running number, slot)
    int ib_j = 0, nb = 0, sb_ = 0;
#define TP8_ISSUE_A() { issue_a(ia_item, ia_j, sa); ++na; sa = sa == TP8_NSA - 1 ? 0 : sa + 1; if (++ia_j == nst) { ia_j = 0; ia_item += gridDim.x; } }
#define TP8_ISSUE_B() { issue_b(ib_j, sb_); ++nb; sb_ ^= 1; if (++ib_j == nst) ib_j = 0; }
    if (na < total) TP8_ISSUE_A()
    if (nb < total) TP8_ISSUE_B()
    if (na < total) TP8_ISSUE_A()
    int cu_item = blockIdx.x, cu_j = 0, ca = 0, cb = 0;  // the step being multiplied and its slots
    for (int s = 0; s < total; ++s) {
        if (na > s + 1) tp_wait_c<n_ops_a>();
        else tp_wait_c<0>();
        __syncthreads();  // everything of step s is in LDS, and nobody reads step s - 1 any more
        if (nb < total) TP8_ISSUE_B()
        if (na < total) TP8_ISSUE_A()
        sk_unit_k45<QT>(a_ring_p + ca * A_STAGE + row * F::ROW, smem + cb * TP8_B_STAGE + kh * SK_B_BYTES + row * SK_BTOK,
                        (const float *) (smem + TP8_NSB * TP8_B_STAGE + cb * TP8_D_STAGE + kh * 128), g, acc);
        if (++cu_j == nst) {
            // ---- the tile is complete: the odd half's sums cross over through the weight stage that wave has just consumed (its next
            // request goes out after the next step's barrier); lane = weight row, register i = token (i & 3) + 8 (i >> 2) + 4 g
            if (kh) {
                float * const own = (float *) (a_ring_p + ca * A_STAGE);
#pragma unroll
                for (int i = 0; i < 16; ++i) own[i * 64 + lane] = acc[i];
            }
            __syncthreads();
            if (!kh) {
                const float * const other = (const float *) (a_ring_p + 4 * TP8_NSA * A_STAGE + ca * A_STAGE);
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] += other[i * 64 + lane];
                const int mi = MAT_OF(cu_item);
                const int n = (cu_item - MAT_SEL(mi, panel0)) * 128 + tile * 32 + row;
                float * const m_dst = MAT_SEL(mi, dst);
                const int64_t m_dst_stride = MAT_SEL(mi, dst_stride);
                const float * const m_add = MAT_SEL(mi, add);
                const int64_t m_add_stride = MAT_SEL(mi, add_stride);
                if (m_add) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[i] += m_add[(size_t) min((i & 3) + 8 * (i >> 2) + 4 * g, a.M - 1) * m_add_stride + n];
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int tok = (i & 3) + 8 * (i >> 2) + 4 * g;
                    if (tok < a.M) m_dst[(size_t) tok * m_dst_stride + n] = acc[i];
                }
                // stores count in vmcnt and are not ordered with the requests in flight: drain once per item, so that the counted waits
                // of the next item's steps see requests only
                if (s + 1 < total) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
            cu_j = 0;
            cu_item += gridDim.x;
        }
        ca = ca == TP8_NSA - 1 ? 0 : ca + 1;
        cb ^= 1;
    }
#undef TP8_ISSUE_A
#undef TP8_ISSUE_B
#undef MAT_SEL
#undef MAT_OF
}

// the tile-parallel form serves a launch when every matrix is whole 128-row groups, there is no K split, and the groups alone
// occupy most of the chip
static bool skinny_tp_applies(int type, const mmq8_args & a, int n_cu) {
    if (!(type == GGML_TYPE_Q4_K || type == GGML_TYPE_Q5_K) || a.ksplit != 1) return false;
    static const bool on = !getenv("GGML_MI355X_SKINNY_TP") || atoi(getenv("GGML_MI355X_SKINNY_TP")) != 0;
    if (!on) return false;
    int64_t groups = 0;
    for (int i = 0; i < a.n_mat; ++i) {
        if (a.mat[i].N % 128) return false;
        groups += a.mat[i].N / 128;
    }
    return groups * 4 >= (int64_t) n_cu * 3;
}

static int skinny_n_cu();

// ------------------------------------------------------------------------------------------------ wide form (prompt batches)
// The same unit for 33 .. thousands of columns: a workgroup = NW computing waves = NW row tiles x TT token tiles (32 TT columns)
// x the whole K range (+ loader waves, below).  Per step (super-block) a wave converts ITS weight unit once, in registers — nibbles x scale digits = the
// int8 B operands of all eight sub-blocks, 64 registers for Q4_K — and multiplies it with TT token tiles whose activations sit in
// LDS exactly as the quantisers wrote them ([token][256 qs | bsums]: LDS-DMA, no conversion, no ds_write anywhere in the kernel).
// mmq_i8.hip, which this replaces where it applies, stages both operands through registers into LDS every 128 values of K and
// runs at 24 % of the matrix pipe with its components adding up serially (fragment reads 4.0, conversion + LDS writes
// 2.6, global loads 2.0 ms of a 13 ms prefill GEMM budget); here an MFMA operand costs one LDS read per TT/1 use on the weight
// side and one per use on the activation side, and a step is 17 TT MFMAs against ~180 + 80 TT VALU.
//   rings: weights 3 stages per wave (requested two steps ahead), activations 2 stages per workgroup (one step ahead: they come from
//   L2); request order per step [activations of s+1, weights of s+2] so that what step s+1 consumes is older than what stays in flight
//   (in-order vmcnt).  Items (128-row group, token tile group) of one weight group run on the same XCD at the same time: its L2
//   serves the weight bytes to all of them.
constexpr int WD_NB = 2;  // stages of the activation ring (3 = requested two steps ahead, like the weights: measured equal, A/B on one box)
template <int QT, int TT, int NW, int NA = 3, int KH = 1> constexpr int wd_lds_bytes() { return WD_NB * KH * TT * (SK_B_BYTES + 128) + NW * KH * NA * tp_a_stage<QT>(); }
constexpr int WD_NL = 2;  // loader waves per workgroup
// timing-only knock-outs (GGML_MI355X_MMQ_WIDE_KO; results are wrong): 1 = the loaders request nothing after the prologue, 2 = the computing
// waves only meet the barriers.  Which side of the workgroup bounds a step?
__device__ int g_wd_ko = 0;

// NW computing waves (one 32-row tile each) + WD_NL loader waves.  In-kernel timestamps of the first version, where every wave
// requested its own operands: per step and wave 1 050 clocks issuing 9 LDS-DMA instructions (each waits for room in the CU's one
// address pipeline), 1 650 at the barrier (for the slowest issuer), 430 converting the weight unit, 2 700 in the 34 MFMAs of its two
// token tiles — the matrix pipe busy 37 %.  The loaders take the first two off the computing waves: they request, wait (vmcnt) and
// meet the others at the step's barrier; the computing waves only convert and multiply.
// NL = 0: no loader waves — every computing wave requests its own weight tile and its share of the activations (NW waves = one per
// SIMD: 512 registers each, accumulators in AGPRs, room for 4 token tiles per converted weight unit).  Measured (A/B, 2048-token
// prefill): 23.2 k tok/s against 31.6 k for 4 + 2 waves x 2 tiles — the requests' issue stalls sit in the computing waves' own
// instruction streams again.  Kept behind GGML_MI355X_MMQ_WIDE_SELF=1 (correct: the wide tests pass with it).
// NA = stages of a weight ring.  NA = 2 (weights requested ONE step ahead, like the activations) brings the workgroup to 76 KB of LDS, so
// that TWO workgroups share a CU (12 waves, 168 registers each): the second one runs while the first waits at its barrier, for an LDS
// read or for the matrix pipe — the latency hiding a single computing wave per SIMD does not have (round 4).
// KH = 2 (round 4): TWO computing waves per row tile and SIMD — wave (tile, kh) takes super-blocks 2 s + kh of step s, so a step is two
// super-blocks and a SIMD always has one wave converting / folding (VALU) while the other multiplies (matrix pipe); PMC of the KH = 1 form:
// its single computing wave per SIMD is busy 85 % of the time with VALU and MFMA one after the other.  Both waves of a pair hold f32 tiles; at the
// end of an item the kh = 1 wave hands its tile over through the activation stage that was just consumed (two more barriers per item) and the
// kh = 0 wave adds and stores: sum over even super-blocks + sum over odd ones.  (Two workgroups per CU do not do this: a workgroup of 5 or 6 waves
// at 168 registers is never co-resident with a second one — scripts/ubench/occupancy_probe.hip — and without loader waves, 4 + 4 co-resident
// waves run at 25 k tok/s against 30.5 k.)
template <int QT, int TT, int NW, int NL, int NA, int KH>
__global__ void __launch_bounds__((NW * KH + NL) * 64, (NA == 2 || KH == 2) ? 3 : 1) k_mmq_wide(const mmq8_args a) {
    constexpr bool SELF = NL == 0;
    constexpr int NCW = NW * KH;           // computing waves
    constexpr int NLD = SELF ? NCW : NL;   // waves that request
    static_assert(KH == 1 || (KH == 2 && NL > 0), "the K-paired form has loader waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef sk_fmt<QT> F;
    constexpr int NP = QT == 4 ? 2 : 3;
    constexpr uint32_t DM = QT == 4 ? 0x07070707u : 0x03030303u;
    constexpr int DS = QT == 4 ? 3 : 2;
    constexpr int NPA = 32 * F::PIECES, NLA = (NPA + 63) / 64;
    constexpr int NPB = KH * TT * 32 * 19 / NLD, NLB = (NPB + 63) / 64;  // activation pieces per loader and step, and the instructions fetching them
    constexpr int TPL = NCW / NLD;                                        // weight units per loader
    constexpr int DCNT = (KH * TT + NLD - 1) / NLD;                       // scale requests per loader and step
    constexpr int B_STAGE = KH * TT * SK_B_BYTES, A_STAGE = tp_a_stage<QT>();  // an activation stage: [kh][token tile][token]
    constexpr int D_STAGE = KH * TT * 128;
    static_assert(NCW % NLD == 0 && (KH * TT * 32 * 19) % NLD == 0, "loader roles");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ko = __builtin_amdgcn_readfirstlane(g_wd_ko);
    const bool loader = !SELF && wave >= NCW;
    const int ld = SELF ? wave : wave - NCW;  // loader index
    const int tile = wave % NW, kh = KH == 1 ? 0 : (wave / NW) & 1;  // (computing waves)
    const int row = lane & 31, g = lane >> 5;
    const int nblk = a.K / 256;
    const int nsteps = nblk / KH;  // steps of an item
    const int w_nb1 = (int) a.mat[0].w_nb1;
    const int tok_bytes = nblk * (int) sizeof(q8k_dev);
    const uint32_t lds0 = (uint32_t) (uintptr_t) smem;
    // LDS: [activation ring 2 x KH x TT x 9728][scale ring 2 x KH x TT x 128][weight rings: computing wave x NA x stage]
    const uint32_t b_ring = lds0, d_ring = lds0 + WD_NB * B_STAGE, a_rings = d_ring + WD_NB * D_STAGE;
    char * const b_ring_p = smem;
    const char * const d_ring_p = smem + WD_NB * B_STAGE;
    const char * const a_ring_p = smem + WD_NB * B_STAGE + WD_NB * D_STAGE + (loader ? 0 : wave) * NA * A_STAGE;

    // loader fetch roles (offsets from the unit's first byte)
    int a_off[NLA];
#pragma unroll
    for (int u = 0; u < NLA; ++u) {
        const int pi = min(lane + 64 * u, NPA - 1);
        a_off[u] = (pi / F::PIECES) * w_nb1 + (pi % F::PIECES) * 16;
    }
    int b_off[NLB];
#pragma unroll
    for (int u = 0; u < NLB; ++u) {
        const int pi = max(ld, 0) * NPB + min(lane + 64 * u, NPB - 1);
        const int q = pi / 19;  // [kh][token of the TT tiles]
        b_off[u] = (q % (TT * 32)) * tok_bytes + (q / (TT * 32)) * (int) sizeof(q8k_dev) + (pi % 19) * 16;
    }
    const int d_off = row * tok_bytes + 304;  // q8k_dev::d of token `row` of a token tile
    constexpr int n_ops_a = TPL * NLA;        // a loader's weight requests of one step

#define MAT_OF(t) ((a.n_mat > 2 && (t) >= a.mat[2].panel0) ? 2 : ((a.n_mat > 1 && (t) >= a.mat[1].panel0) ? 1 : 0))
#define MAT_SEL(mi, f) ((mi) == 0 ? a.mat[0].f : ((mi) == 1 ? a.mat[1].f : a.mat[2].f))
    // virtual items v = blockIdx.x + gridDim.x r: XCD v % 8 (workgroups go round the XCDs) serves groups xcd, xcd + 8, ..; the token-tile
    // groups of one weight group are consecutive there
    const int n_groups = a.n_panels, m_tiles = a.m_tiles;
    const int n_virtual = ((n_groups + 7) / 8) * 8 * m_tiles;
    auto decode = [&](const int v, int & group, int & mt) {
        const int xcd = v & 7, q = v >> 3;
        group = (q / m_tiles) * 8 + xcd;
        mt = q - (q / m_tiles) * m_tiles;
    };
    auto next_valid = [&](int v) {
        for (; v < n_virtual; v += gridDim.x) {
            int gr, mt;
            decode(v, gr, mt);
            if (gr < n_groups) return v;
        }
        return n_virtual;
    };
    int total = 0;  // steps of this workgroup (the same number in every wave: one barrier each)
    for (int v = next_valid(blockIdx.x); v < n_virtual; v = next_valid(v + gridDim.x)) total += nsteps;

    auto issue_a = [&](const int v, const int sb, const int slot) {
        int gr, mt;
        decode(v, gr, mt);
        const int mi = MAT_OF(gr);
        const uint8_t * wg = MAT_SEL(mi, W) + (size_t) (gr - MAT_SEL(mi, panel0)) * (32 * NW) * (size_t) w_nb1 + (size_t) sb * KH * F::BYTES;
#pragma unroll
        for (int k = 0; k < TPL; ++k) {
            const int t = ld * TPL + k;  // computing wave (tile t % NW, K half t / NW)
            const uint8_t * wb = wg + (size_t) (t % NW) * 32 * (size_t) w_nb1 + (t / NW) * F::BYTES;
            const uint32_t al = a_rings + (t * NA + slot) * A_STAGE;
#pragma unroll
            for (int u = 0; u < NLA; ++u)
                if (u + 1 < NLA || (NPA % 64) == 0 || lane < (NPA % 64)) tp_dma16(wb + a_off[u], al + u * 1024);
        }
    };
    auto issue_b = [&](const int v, const int sb, const int slot) {
        int gr, mt;
        decode(v, gr, mt);
        const char * ab = (const char *) a.act + (size_t) mt * (32 * TT) * (size_t) tok_bytes + (size_t) sb * KH * sizeof(q8k_dev);
        const uint32_t bl = b_ring + slot * B_STAGE + ld * NPB * 16;
#pragma unroll
        for (int u = 0; u < NLB; ++u)
            if (u + 1 < NLB || (NPB % 64) == 0 || lane < (NPB % 64)) tp_dma16(ab + b_off[u], bl + u * 1024);
        // the block scales of token tile tt: loader tt % WD_NL — every loader issues DCNT of these instructions (a duplicate when it has
        // no tile of its own), so that the wait counts are the same compile-time constants for all
#pragma unroll
        for (int t = 0; t < DCNT; ++t) {
            const int tt = min(ld + t * NLD, KH * TT - 1);  // [kh][token tile]
            if (lane < 32) tp_dma4(ab + (size_t) (tt % TT) * 32 * (size_t) tok_bytes + (tt / TT) * sizeof(q8k_dev) + d_off, d_ring + slot * D_STAGE + tt * 128);
        }
    };

    float acc[TT][16];
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;

    // request cursors (loaders): weights run two steps ahead, activations one
    int va = next_valid(blockIdx.x), sa = 0, na = 0;  // next weight step to request (virtual item, super-block, running step number)
    int vb = va, sbb = 0, nb = 0;
    int vc = va, sc_ = 0;                             // the step being multiplied (computing waves)
#define WD_ADV(v, sb) { if (++(sb) == nsteps) { (sb) = 0; (v) = next_valid((v) + gridDim.x); } }
    if (loader || SELF) {
        if (va < n_virtual) { issue_a(va, sa, na % NA); ++na; WD_ADV(va, sa) }
        if (vb < n_virtual) { issue_b(vb, sbb, nb % WD_NB); ++nb; WD_ADV(vb, sbb) }
        if (NA == 3 && va < n_virtual) { issue_a(va, sa, na % NA); ++na; WD_ADV(va, sa) }
        if (WD_NB == 3 && vb < n_virtual) { issue_b(vb, sbb, nb % WD_NB); ++nb; WD_ADV(vb, sbb) }
    }
    const int16s zeroi = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const float16s zerof = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (!SELF) {
        // the loaders' loop, apart from the computing waves' (one barrier per step in both): its address registers are not live in the
        // multiply loop and the operands are not live here
        if (loader) {
            for (int s = 0; s < total; ++s) {
                // this loader's pieces of step s (weights requested two steps ago, activations one) have landed; only its weight requests of
                // step s + 1 may still fly
                // (WD_NB == 3: the activations run two steps ahead as well; what may still fly is step s + 1 of both kinds, requested together)
                if (NA == 3 && na > s + 1) tp_wait_c<WD_NB == 3 ? n_ops_a + NLB + DCNT : n_ops_a>();
                else tp_wait_c<0>();
                __syncthreads();  // everything of step s is in LDS, and nobody reads step s - 1 any more
                if (!(ko & 1)) {
                    if (vb < n_virtual) { issue_b(vb, sbb, nb % WD_NB); ++nb; WD_ADV(vb, sbb) }
                    if (va < n_virtual) { issue_a(va, sa, na % NA); ++na; WD_ADV(va, sa) }
                }
                if (KH == 2 && ++sc_ == nsteps) {  // the pairs' hand-over at the end of an item (below): two barriers
                    __syncthreads();
                    __syncthreads();
                    sc_ = 0;
                }
            }
            return;
        }
    }
    for (int s = 0; s < total; ++s) {
        if (SELF) {
            if (NA == 3 && na > s + 1) tp_wait_c<WD_NB == 3 ? n_ops_a + NLB + DCNT : n_ops_a>();
            else tp_wait_c<0>();
        }
        __syncthreads();  // everything of step s is in LDS, and nobody reads step s - 1 any more
        if (SELF) {
            if (vb < n_virtual) { issue_b(vb, sbb, nb % WD_NB); ++nb; WD_ADV(vb, sbb) }
            if (va < n_virtual) { issue_a(va, sa, na % NA); ++na; WD_ADV(va, sa) }
        }

        if (ko & 2) continue;
        // ---- this wave's weight unit -> digit-plane operands, once per step
        const char * const arow = a_ring_p + (s % NA) * A_STAGE + row * F::ROW;
        const uint4 hdr = *(const uint4 *) arow;
        const float d = h2f((uint16_t) (hdr.x & 0xFFFF)), dmin = h2f((uint16_t) (hdr.x >> 16));
        const uint32_t slo = hdr.y & 0x3F3F3F3Fu, shi = (hdr.w & 0x0F0F0F0Fu) | ((hdr.y >> 2) & 0x30303030u);
        const uint32_t mlo = hdr.z & 0x3F3F3F3Fu, mhi = ((hdr.w >> 4) & 0x0F0F0F0Fu) | ((hdr.z >> 2) & 0x30303030u);
        uint4 mfu;
        {
            typedef _Float16 half2s __attribute__((ext_vector_type(2)));
            const uint32_t msrc = g ? mhi : mlo;
            const half2s k1024 = {(_Float16) 1024.0f, (_Float16) 1024.0f};
            mfu.x = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2s, sk_rep_byte<0>(msrc) | 0x64006400u) - k1024);
            mfu.y = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2s, sk_rep_byte<1>(msrc) | 0x64006400u) - k1024);
            mfu.z = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2s, sk_rep_byte<2>(msrc) | 0x64006400u) - k1024);
            mfu.w = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2s, sk_rep_byte<3>(msrc) | 0x64006400u) - k1024);
        }
        uint32_t dlo[NP], dhi[NP];
#pragma unroll
        for (int n = 0; n < NP; ++n) {
            dlo[n] = (slo >> (DS * n)) & DM;
            dhi[n] = (shi >> (DS * n)) & DM;
        }
        uint4 qh = make_uint4(0, 0, 0, 0);
        if constexpr (QT == 5) qh = *(const uint4 *) (arow + 16 + 16 * g);
        int4s W0[4][NP], W1[4][NP];  // pair p, digit plane n: sub-block 2p (low nibbles) / 2p + 1 (high nibbles)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const uint4 raw = *(const uint4 *) (arow + (QT == 5 ? 48 : 16) + 32 * p + 16 * g);
            uint32_t wlo[4] = {raw.x & 0x0F0F0F0Fu, raw.y & 0x0F0F0F0Fu, raw.z & 0x0F0F0F0Fu, raw.w & 0x0F0F0F0Fu};
            uint32_t whi[4] = {(raw.x >> 4) & 0x0F0F0F0Fu, (raw.y >> 4) & 0x0F0F0F0Fu, (raw.z >> 4) & 0x0F0F0F0Fu, (raw.w >> 4) & 0x0F0F0F0Fu};
            if constexpr (QT == 5) {
                const uint32_t h4[4] = {qh.x, qh.y, qh.z, qh.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    wlo[k] |= ((h4[k] >> (2 * p)) & 0x01010101u) << 4;
                    whi[k] |= ((h4[k] >> (2 * p + 1)) & 0x01010101u) << 4;
                }
            }
#pragma unroll
            for (int n = 0; n < NP; ++n) {
                const uint32_t src = p < 2 ? dlo[n] : dhi[n];
                const uint32_t e0 = (p & 1) ? sk_rep_byte<2>(src) : sk_rep_byte<0>(src), e1 = (p & 1) ? sk_rep_byte<3>(src) : sk_rep_byte<1>(src);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    W0[p][n][k] = (int) sk_pk_mul(wlo[k], e0);
                    W1[p][n][k] = (int) sk_pk_mul(whi[k], e1);
                }
            }
        }
        // ---- TT token tiles against it
        const char * const bst = b_ring_p + (s % WD_NB) * B_STAGE + kh * TT * SK_B_BYTES;
        const float * const dst_ = (const float *) (d_ring_p + (s % WD_NB) * D_STAGE + kh * TT * 128);
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const char * const btok = bst + (t * 32 + row) * SK_BTOK;
            int16s pl[NP];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int4s y0 = *(const int4s *) (btok + 64 * p + 16 * g);
                const int4s y1 = *(const int4s *) (btok + 64 * p + 32 + 16 * g);
#pragma unroll
                for (int n = 0; n < NP; ++n) pl[n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(y0, W0[p][n], p == 0 ? zeroi : pl[n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < NP; ++n) pl[n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(y1, W1[p][n], pl[n], 0, 0, 0);
            }
            const half8s bsf = *(const half8s *) (btok + 256 + 16 * g);
            const float16s ms = __builtin_amdgcn_mfma_f32_32x32x16_f16(bsf, __builtin_bit_cast(half8s, mfu), zerof, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 dy = *(const float4 *) (dst_ + t * 32 + 8 * q + 4 * g);
                const float dyv[4] = {dy.x, dy.y, dy.z, dy.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 4 * q + r;
                    int isum;
                    if constexpr (QT == 4) isum = (pl[1][i] << 3) + pl[0][i];
                    else isum = (pl[2][i] << 4) + (pl[1][i] << 2) + pl[0][i];
                    const float v = __builtin_fmaf(-dmin, ms[i], d * (float) isum);
                    acc[t][i] = __builtin_fmaf(dyv[r], v, acc[t][i]);
                }
            }
        }
        if (++sc_ == nsteps) {
            // ---- the tiles are complete: lane = weight row, register i = token (i & 3) + 8 (i >> 2) + 4 g of token tile t
            if constexpr (KH == 2) {
                // the odd super-blocks' tile goes to the wave of the even ones through the activation stage of this step (nobody reads it
                // after the first barrier; the loaders write it again only after the NEXT step's barrier)
                float * const xch = (float *) (b_ring_p + (s % WD_NB) * B_STAGE) + tile * (TT * 16 * 64) + lane;
                static_assert(NW * TT * 16 * 64 * 4 <= B_STAGE, "hand-over area");
                __syncthreads();
                if (kh == 1) {
#pragma unroll
                    for (int t = 0; t < TT; ++t)
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            xch[(t * 16 + i) * 64] = acc[t][i];
                            acc[t][i] = 0.0f;
                        }
                }
                __syncthreads();
                if (kh == 0) {
#pragma unroll
                    for (int t = 0; t < TT; ++t)
#pragma unroll
                        for (int i = 0; i < 16; ++i) acc[t][i] += xch[(t * 16 + i) * 64];
                }
            }
            int gr, mt;
            decode(vc, gr, mt);
            const int mi = MAT_OF(gr);
            const int n = (gr - MAT_SEL(mi, panel0)) * (32 * NW) + tile * 32 + row;
            float * const m_dst = MAT_SEL(mi, dst);
            const int64_t m_dst_stride = MAT_SEL(mi, dst_stride);
            const float * const m_add = MAT_SEL(mi, add);
            const int64_t m_add_stride = MAT_SEL(mi, add_stride);
            // (addends first, all of them in flight, then nothing but stores: with the optional load inside the store loop the compiler
            // waited vmcnt(0) before every store — stores count in vmcnt on this ISA, so each of the 32 waited for its predecessor's
            // write acknowledgement, ~16k clocks per item)
            if (KH == 1 || kh == 0) {
                if (m_add) {
#pragma unroll
                    for (int t = 0; t < TT; ++t)
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int tok = min(mt * 32 * TT + t * 32 + (i & 3) + 8 * (i >> 2) + 4 * g, a.M - 1);
                            acc[t][i] += m_add[(size_t) tok * m_add_stride + n];
                        }
                }
#pragma unroll
                for (int t = 0; t < TT; ++t)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int tok = mt * 32 * TT + t * 32 + (i & 3) + 8 * (i >> 2) + 4 * g;
                        if (tok < a.M) m_dst[(size_t) tok * m_dst_stride + n] = acc[t][i];
                        acc[t][i] = 0.0f;
                    }
            }
            sc_ = 0;
            vc = next_valid(vc + gridDim.x);
        }
    }
#undef WD_ADV
#undef MAT_SEL
#undef MAT_OF
}

// Prompt batches (33 columns and up) take the wide form when every matrix is whole 128-row groups of Q4_K / Q5_K, there is no K
// split and (row groups x 64-column groups) can occupy the chip.  One workgroup shape is in use — 4 computing waves (128 rows) x 2
// token tiles (64 columns) + 2 loaders; measured on the 512-token prefill of Llama-3-8B Q4_K_M against the tiled GEMM at 28.9 k tok/s:
// this shape 30.5 k; 8 computing waves x 2 tiles, everyone requesting its own operands, 28.7 k; the same with 2 / 4 loaders 26.1 /
// 28.9 k (one address pipeline per CU: 52 LDS-DMA instructions per step); 4 waves x 4 tiles 20.2 k (spills at the 256-register cap
// of six waves).  Qwen2-7B Q5_K_M (three digit planes: the tiled GEMM spills there) 20.5 -> 25.3 k.  Returns the shape code
// NW * 16 + TT, or 0.  The caller's activation area must hold whole 128-column groups (columns beyond M are fetched, never stored).
int mmq_wide_tiles(int type, int64_t K, const int64_t * N, int n_mat, int64_t M, int64_t w_nb1, int ksplit) {
    static const bool on = !getenv("GGML_MI355X_MMQ_WIDE") || atoi(getenv("GGML_MI355X_MMQ_WIDE")) != 0;
    if (!on || !(type == GGML_TYPE_Q4_K || type == GGML_TYPE_Q5_K) || ksplit != 1 || M < 33 || (K % 256) != 0) return 0;
    if (w_nb1 != (K / 256) * (type == GGML_TYPE_Q4_K ? 144 : 176)) return 0;
    int64_t g128 = 0;
    for (int i = 0; i < n_mat; ++i) {
        if (N[i] % 128) return 0;
        g128 += N[i] / 128;
    }
    return g128 * ((M + 63) / 64) * 2 >= skinny_n_cu() ? 4 * 16 + 2 : 0;
}

template <int QT, int TT, int NW, int NL, int NA = 3, int KH = 1> static void launch_wide_t(hipStream_t s, mmq8_args a, const int per_cu = 1) {
    static std::atomic<uint32_t> lds_raised{0};
    const void * fn = (const void *) k_mmq_wide<QT, TT, NW, NL, NA, KH>;
    const size_t lds = (size_t) wd_lds_bytes<QT, TT, NW, NA, KH>();
    (void) ensure_dyn_lds(fn, lds, lds_raised);
    static const bool occ_log = getenv("GGML_MI355X_OCC_LOG") != nullptr;
    if (occ_log) {
        static std::atomic<int> once{0};
        if (!once.exchange(1)) {
            int nb = -1;
            hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, (NW * KH + NL) * 64, lds);
            hipFuncAttributes fa{};
            (void) hipFuncGetAttributes(&fa, fn);
            fprintf(stderr, "[mi355x] k_mmq_wide<%d,%d,%d,%d,%d,%d>: %d blocks/CU by the occupancy API (err %d), lds %zu, regs %d, static lds %zu, max dyn lds %d\n", QT, TT, NW, NL, NA, KH, nb, (int) e, lds,
                    fa.numRegs, fa.sharedSizeBytes, fa.maxDynamicSharedSizeBytes);
        }
    }
    static const int ko_env = getenv("GGML_MI355X_MMQ_WIDE_KO") ? atoi(getenv("GGML_MI355X_MMQ_WIDE_KO")) : 0;
    if (ko_env) {
        static std::atomic<int> ko_set{0};
        if (!ko_set.exchange(1)) (void) hipMemcpyToSymbol(HIP_SYMBOL(g_wd_ko), &ko_env, sizeof(int));
    }
    a.n_panels = 0;
    for (int i = 0; i < a.n_mat; ++i) {
        a.mat[i].panel0 = a.n_panels;
        a.n_panels += a.mat[i].N / (32 * NW);
    }
    a.m_tiles = (a.M + 32 * TT - 1) / (32 * TT);
    const int n_virtual = ((a.n_panels + 7) / 8) * 8 * a.m_tiles;
    MI_LAUNCH_PROBED((k_mmq_wide<QT, TT, NW, NL, NA, KH>), dim3((unsigned) std::min(n_virtual, skinny_n_cu() * per_cu)), dim3((NW * KH + NL) * 64), lds, s, a);
}
void launch_mmq_wide(hipStream_t s, int type, int shape, const mmq8_args & a) {
    static const int self4 = getenv("GGML_MI355X_MMQ_WIDE_SELF") ? atoi(getenv("GGML_MI355X_MMQ_WIDE_SELF")) : 0;  // experiment: 4 waves x 4 token tiles, no loaders
    (void) shape;
    if (self4 && type == GGML_TYPE_Q4_K) { launch_wide_t<4, 4, 4, 0>(s, a); return; }
    static const int wg2 = getenv("GGML_MI355X_MMQ_WIDE_WG2") ? atoi(getenv("GGML_MI355X_MMQ_WIDE_WG2")) : 0;  // two workgroups per CU (weight ring of two stages)
    static const int kh2 = getenv("GGML_MI355X_MMQ_WIDE_KH") ? atoi(getenv("GGML_MI355X_MMQ_WIDE_KH")) : 0;  // two computing waves per row tile (even / odd super-blocks)
    if (kh2 && type == GGML_TYPE_Q4_K && (a.K / 256) % 2 == 0) {
        if (kh2 == 4) launch_wide_t<4, 2, 4, 4, 2, 2>(s, a);
        else launch_wide_t<4, 2, 4, 2, 2, 2>(s, a);
        return;
    }
    if (wg2) {
        if (type == GGML_TYPE_Q4_K && wg2 >= 5) launch_wide_t<4, 2, 4, 0, 2>(s, a, wg2 == 6 ? 1 : 2);  // no loaders: 4 waves, one per SIMD and workgroup
        else if (type == GGML_TYPE_Q4_K && wg2 >= 3) launch_wide_t<4, 2, 4, 1, 2>(s, a, wg2 == 4 ? 1 : 2);  // one loader: 5 waves
        else if (type == GGML_TYPE_Q4_K) launch_wide_t<4, 2, 4, WD_NL, 2>(s, a, wg2 == 2 ? 1 : 2);
        else launch_wide_t<5, 2, 4, WD_NL, 2>(s, a, wg2 == 2 ? 1 : 2);
        return;
    }
    if (type == GGML_TYPE_Q4_K) launch_wide_t<4, 2, 4, WD_NL>(s, a);
    else launch_wide_t<5, 2, 4, WD_NL>(s, a);
}

bool mmq_skinny_supported(int type, int64_t K, int64_t N, int64_t M, int64_t w_nb1) {
    if (M < 2 || M > 32 || (K % 256) != 0 || (N % 32) != 0) return false;  // whole 32-row tiles (every model dimension is one)
    // rows packed back to back (then every matrix of a launch has the same row stride)
    if (type == GGML_TYPE_Q4_K) return w_nb1 == (K / 256) * 144;
    if (type == GGML_TYPE_Q5_K) return w_nb1 == (K / 256) * 176;
    // (a Q6_K matrix of 64 k rows and more — the output matrix — stays on the tiled GEMM: 144 us against 170 us here for 128256 x 4096 at
    // 32 columns, profiles/r03_np32_ab_q6_tiled.txt; this kernel's Q6_K unit is bound by its integer VALU, the tiled one converts once per 64 columns)
    if (type == GGML_TYPE_Q6_K) return N < 65536 && w_nb1 == (K / 256) * 210 && (w_nb1 % 4) == 0;  // dword-aligned rows: the 0 / 2-byte shift of a super-block is the same in every row
    return false;
}

// workgroups = 32-row tiles x ksplit: split K only while the tiles alone leave CUs without a workgroup
int mmq_skinny_ksplit(int64_t K, int64_t n_total) {
    const int64_t tiles = (n_total + 31) / 32, nblk = K / 256;
    int ks = 1;
    while (tiles * ks < 200 && ks * 2 <= 8 && nblk / (ks * 2) >= 4) ks *= 2;
    return ks;
}

static int skinny_n_cu() {
    static std::atomic<int> n_cu{0};
    if (n_cu.load(std::memory_order_relaxed) == 0) {
        int dev = 0, v = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void) hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
        n_cu.store(v > 0 ? v : 256, std::memory_order_relaxed);
    }
    return n_cu.load(std::memory_order_relaxed);
}

template <int QT> static void launch_skinny_t(hipStream_t s, mmq8_args a) {
    const int n_cu = skinny_n_cu();
    a.m_tiles = 1;
    if constexpr (QT == 4 || QT == 5) {
        if (!a.has_epi && skinny_tp_applies(QT == 4 ? GGML_TYPE_Q4_K : GGML_TYPE_Q5_K, a, n_cu)) {
            a.n_panels = 0;
            for (int i = 0; i < a.n_mat; ++i) {
                a.mat[i].panel0 = a.n_panels;
                a.n_panels += a.mat[i].N / 128;
            }
            if constexpr (QT == 4) {
                // two waves per tile (GGML_MI355X_SKINNY_TP=1: the four-wave form)
                static const bool tp8 = !getenv("GGML_MI355X_SKINNY_TP") || atoi(getenv("GGML_MI355X_SKINNY_TP")) != 1;
                if (tp8 && (a.K % 512) == 0) {
                    static std::atomic<uint32_t> lds_raised_tp8{0};
                    (void) ensure_dyn_lds((const void *) k_mmq_skinny_tp8<QT>, (size_t) tp8_lds_bytes<QT>(), lds_raised_tp8);
                    MI_LAUNCH_PROBED((k_mmq_skinny_tp8<QT>), dim3((unsigned) std::min(a.n_panels, n_cu)), dim3(TP8_NW * 64), (size_t) tp8_lds_bytes<QT>(), s, a);
                    return;
                }
            }
            static std::atomic<uint32_t> lds_raised_tp{0};
            (void) ensure_dyn_lds((const void *) k_mmq_skinny_tp<QT>, (size_t) tp_lds_bytes<QT>(), lds_raised_tp);
            MI_LAUNCH_PROBED((k_mmq_skinny_tp<QT>), dim3((unsigned) std::min(a.n_panels, n_cu)), dim3(TP_NW * 64), (size_t) tp_lds_bytes<QT>(), s, a);
            return;
        }
    }
    const size_t lds = (size_t) SK_NW * sk_wave_lds<QT>();
    static std::atomic<uint32_t> lds_raised{0}, lds_raised_epi{0};
    const void * fn_plain = (const void *) k_mmq_skinny<QT, QT, false>;
    const void * fn_epi = (const void *) k_mmq_skinny<QT, QT, true>;
    (void) ensure_dyn_lds(a.has_epi ? fn_epi : fn_plain, lds, a.has_epi ? lds_raised_epi : lds_raised);
    a.n_panels = 0;
    for (int i = 0; i < a.n_mat; ++i) {
        a.mat[i].panel0 = a.n_panels;
        a.n_panels += a.mat[i].N / 32;
    }
    // one workgroup per CU (its LDS areas fill the CU), walking the (tile, K slice) items with a stride of the grid
    const int items = a.n_panels * a.ksplit;
    if (a.has_epi) MI_LAUNCH_PROBED((k_mmq_skinny<QT, QT, true>), dim3((unsigned) std::min(items, n_cu)), dim3(SK_NW * 64), lds, s, a);
    else MI_LAUNCH_PROBED((k_mmq_skinny<QT, QT, false>), dim3((unsigned) std::min(items, n_cu)), dim3(SK_NW * 64), lds, s, a);
}

template <int QA, int QB> static void launch_skinny_mix_t(hipStream_t s, mmq8_args a) {
    constexpr size_t lds = (size_t) SK_NW * (size_t) std::max(sk_wave_lds<QA>(), sk_wave_lds<QB>());
    static std::atomic<uint32_t> lds_raised{0}, lds_raised_epi{0};
    const void * fn = a.has_epi ? (const void *) k_mmq_skinny<QA, QB, true> : (const void *) k_mmq_skinny<QA, QB, false>;
    (void) ensure_dyn_lds(fn, lds, a.has_epi ? lds_raised_epi : lds_raised);
    a.m_tiles = 1;
    a.n_panels = 0;
    for (int i = 0; i < a.n_mat; ++i) {
        a.mat[i].panel0 = a.n_panels;
        a.n_panels += a.mat[i].N / 32;
    }
    const int items = a.n_panels * a.ksplit;
    if (a.has_epi) MI_LAUNCH_PROBED((k_mmq_skinny<QA, QB, true>), dim3((unsigned) std::min(items, skinny_n_cu())), dim3(SK_NW * 64), lds, s, a);
    else MI_LAUNCH_PROBED((k_mmq_skinny<QA, QB, false>), dim3((unsigned) std::min(items, skinny_n_cu())), dim3(SK_NW * 64), lds, s, a);
}
bool launch_mmq_skinny_mixed(hipStream_t s, const mmq8_args & a) {
    bool has[7] = {};
    for (int i = 0; i < a.n_mat; ++i) {
        if (a.mat[i].qt < 4 || a.mat[i].qt > 6) return false;
        has[a.mat[i].qt] = true;
    }
    if (has[4] && has[6] && !has[5]) { launch_skinny_mix_t<4, 6>(s, a); return true; }
    if (has[5] && has[6] && !has[4]) { launch_skinny_mix_t<5, 6>(s, a); return true; }
    return false;
}

void launch_mmq_skinny(hipStream_t s, int type, const mmq8_args & a) {
    if (type == GGML_TYPE_Q4_K) launch_skinny_t<4>(s, a);
    else if (type == GGML_TYPE_Q5_K) launch_skinny_t<5>(s, a);
    else launch_skinny_t<6>(s, a);
}

MI_TU_TOUCH(mmq_skinny)

}  // namespace mi355x
