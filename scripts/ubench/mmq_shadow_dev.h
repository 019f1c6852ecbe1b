// mmq_shadow_dev.h — device code of the prompt-batch GEMM over SHADOW PLANES (mmq_shadow.hip hosts it; scripts/ubench/shadow_probe.hip
// times it stand-alone).
//
// What it replaces: mul_mat_q for prompt micro-batches (reference call site llama-box/httpserver.hpp:3539-3623
// with -b/-ub, engine_param.hpp:1181,:1190).  Contract unchanged — ggml-cpu's ggml_vec_dot_q{4,5,6}_K_q8_K: integer block sums on
// Q8_K activations, one f32 scale-accumulate per super-block.
//
// Why a second weight layout.  Round 2's kernels (mmq_i8.hip, mmq_skinny.hip wide form) turn GGUF super-blocks into int8 matrix-core
// operands INSIDE the GEMM: nibble unpacking, scale digits, v_pk_mul, LDS writes — 280 VALU + 31 LDS instructions beside 34 MFMAs per
// step of a wave, the matrix pipe 28 % busy (profiles/r02_pmc_passes.txt).  An MI355X has 288 GB of HBM and an 8B model is 5 GB: the
// operands are built ONCE (k_shadow_build, at the first prompt batch that reads a weight) and kept beside the GGUF bytes, which stay
// what get_tensor returns and what every decode kernel streams.
//
// The planes.  For a super-block the reference computes  sumi = sum_k s(k) q(k) a(k)  with s the 6-bit (Q4_K/Q5_K) or int8 (Q6_K)
// sub-block scale and q the 4/5/6-bit value (Q6_K: q - 32).  p(k) = s(k) q(k) is up to 11 bits (Q6_K: 13, signed) — not an int8 — but
// ANY split p = 128 hi + lo with lo in [-64, 63] has hi in [-32, 31]: two int8 planes, and
//      sumi = sum lo(k) a(k) + 128 sum hi(k) a(k)           (exact in int32: |.| <= 256 x 64 x 127 and 128 x 256 x 32 x 127)
// for every format alike — one kernel, 16 v_mfma_i32_32x32x32_i8 per 32 x 32 tile and super-block, and the fold
//      C += dy (d float(sumi) - dmin sum_j m_j bsum_j)      (mins by ONE f16 MFMA on the Q8_K bsums; Q6_K has none)
// exactly as mmq_i8.hip does it (same integers, same f32 operations in the same order per element).
//
// Layout in HBM, per matrix (rows = output features N, a multiple of 128; nblk = K / 256):
//   planes  [N/32 panels][nblk][4 quarters (64 values of K)][2 planes: lo, hi][32 rows][64 bytes]
//           the four 16-byte slots of a row are stored XOR-swizzled by (row >> 2) & 3 — the image a conflict-free ds_read_b128 wants —
//           so a stage of the GEMM is copied HBM -> LDS by LDS-DMA in contiguous 1 KB pieces with nothing to compute on the way
//   meta    [N/32 panels][nblk][ mins: 32 rows x 16 f16 (m_j twice: one per 16-value bsum) | dd: 32 rows x (d, dmin) f32 ]  = 1280 B
// 512 + 40 bytes per 256 weights: x3.83 of Q4_K, x3.14 of Q5_K, x2.63 of Q6_K.
#pragma once
#include "dev_util.h"
#include "common.h"

namespace mi355x {

typedef _Float16 sh_half8 __attribute__((ext_vector_type(8)));
typedef float sh_f16v __attribute__((ext_vector_type(16)));
typedef int sh_i4v __attribute__((ext_vector_type(4)));
typedef int sh_i16v __attribute__((ext_vector_type(16)));

constexpr int SH_META = 1280;                    // bytes of metadata per (32-row panel, super-block)
constexpr int SH_PANEL_SB = 4 * 2 * 32 * 64;     // bytes of planes per (32-row panel, super-block) = 16 KB

struct shadow_mat {
    const char * planes;
    const char * meta;
    int N, group0;          // rows; first 128-row group of this matrix in the launch's list
    float * dst;
    int64_t dst_stride;
    const float * add;      // optional epilogue addend: bias row (stride 0) or residual (stride = row length)
    int64_t add_stride;
    float * part;           // ksplit > 1: [ksplit][M][N] partial results of this matrix (summed in a fixed order by the consumer / k_splitk_reduce)
};
struct shadow_args {
    shadow_mat mat[3];
    int n_mat;
    int K, M;
    const q8k_dev * act;    // [M][K / 256]
    int n_groups, m_tiles;  // 128-row groups of all matrices; 256-token tiles
    int ksplit;             // > 1: blockIdx.y owns a contiguous range of super-blocks (small-N matrices: the tiles alone leave CUs idle)
    unsigned long long * stamps;  // probe builds (SH_STAMP): [8 waves][8 phases] cycle sums of workgroup 0
};

// ------------------------------------------------------------------------------------------------ building the planes
// one workgroup per (32-row panel, super-block): thread t = row (t >> 3) x sub-range (t & 7) of 32 values
template <int QT>
__global__ void __launch_bounds__(256) k_shadow_build(const uint8_t * __restrict__ W, const int64_t w_nb1, const int nblk, char * __restrict__ planes, char * __restrict__ meta) {
    constexpr int BYTES = QT == 4 ? 144 : (QT == 5 ? 176 : 210);
    const int P = blockIdx.x / nblk, sb = blockIdx.x % nblk;
    const int r = threadIdx.x >> 3, j = threadIdx.x & 7;
    const uint8_t * blk = W + (size_t) (P * 32 + r) * w_nb1 + (size_t) sb * BYTES;
    int p[32];  // s(k) q(k) of the 32 values k = 32 j .. 32 j + 31
    float d = 0.0f, dmin = 0.0f;
    int mn = 0;
    if constexpr (QT == 4 || QT == 5) {
        d = h2f(ld16(blk));
        dmin = h2f(ld16(blk + 2));
        const uint8_t * sc12 = blk + 4;
        int sc;
        if (j < 4) { sc = sc12[j] & 63; mn = sc12[j + 4] & 63; }
        else { sc = (sc12[j + 4] & 0xF) | ((sc12[j - 4] >> 6) << 4); mn = (sc12[j + 4] >> 4) | ((sc12[j] >> 6) << 4); }
        const uint8_t * qs = blk + (QT == 5 ? 48 : 16) + 32 * (j >> 1);
        const uint8_t * qh = blk + 16;
#pragma unroll
        for (int l = 0; l < 32; ++l) {
            int q = (j & 1) ? (qs[l] >> 4) : (qs[l] & 0xF);
            if constexpr (QT == 5) q |= ((qh[l] >> j) & 1) << 4;  // bit 2 (j >> 1) + (j & 1) = j
            p[l] = sc * q;
        }
    } else {
        d = h2f(ld16(blk + 208));
        const int hh = j >> 2, w = j & 3;
        const uint8_t * ql = blk + 64 * hh + (w & 1) * 32;
        const uint8_t * qh = blk + 128 + 32 * hh;
        const int8_t * sc = (const int8_t *) (blk + 192) + 8 * hh + 2 * w;
#pragma unroll
        for (int l = 0; l < 32; ++l) {
            const int nib = (w & 2) ? (ql[l] >> 4) : (ql[l] & 0xF);
            const int q = (nib | (((qh[l] >> (2 * w)) & 3) << 4)) - 32;
            p[l] = (int) sc[l >> 4] * q;
        }
    }
    // k = 32 j + l: quarter j >> 1, byte (j & 1) * 32 + l of the row's 64 -> 16-byte slots 2 (j & 1) and 2 (j & 1) + 1
    char * base = planes + ((size_t) (P * nblk + sb) * 4 + (j >> 1)) * (2 * 32 * 64) + r * 64;
    const int swz = (r >> 2) & 3;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        uint32_t lo4[4], hi4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            lo4[u] = 0;
            hi4[u] = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int v = p[16 * c + 4 * u + b];
                const int lo = ((v + 64) & 127) - 64;
                const int hi = (v - lo) >> 7;
                lo4[u] |= (uint32_t) (lo & 0xFF) << (8 * b);
                hi4[u] |= (uint32_t) (hi & 0xFF) << (8 * b);
            }
        }
        const int slot = ((2 * (j & 1) + c) ^ swz) << 4;
        *(uint4 *) (base + slot) = make_uint4(lo4[0], lo4[1], lo4[2], lo4[3]);
        *(uint4 *) (base + 32 * 64 + slot) = make_uint4(hi4[0], hi4[1], hi4[2], hi4[3]);
    }
    char * mb = meta + (size_t) (P * nblk + sb) * SH_META;
    if constexpr (QT == 4 || QT == 5) {
        const uint16_t hm = f2h((float) mn);  // 0 .. 63: exact
        *(uint32_t *) (mb + r * 32 + j * 4) = (uint32_t) hm | ((uint32_t) hm << 16);
    } else {
        *(uint32_t *) (mb + r * 32 + j * 4) = 0u;
    }
    if (j == 0) *(float2 *) (mb + 1024 + r * 8) = make_float2(d, dmin);
}

// ------------------------------------------------------------------------------------------------ the GEMM
// Workgroup = 8 waves (two per SIMD) = 256 tokens x 128 rows x the whole K range; wave (wm, wn) owns tokens [64 wm, +64) x rows [64 wn, +64):
// 2 x 2 MFMA tiles, lo / hi accumulators for each (128 registers) + the f32 results (64).  The TOKENS are the A operand and the weight
// rows the B operand: a lane holds ONE weight row (lane & 31) and 16 tokens of it per tile, so d / dmin are per-lane values and a
// store instruction writes two runs of 32 consecutive floats.
// A step = 64 values of K = 2 MFMA K steps = 16 MFMAs per wave.  Stage of the ring: [256 tokens x 64 B | 4 panels x 2 planes x 32 rows x
// 64 B] = 32 KB, four stages, requested three steps ahead; everything — operands and per-super-block metadata — arrives by LDS-DMA
// (global_load_lds), so no wave ever waits on a register-destination load in the loop.  One barrier per step: after a wave's own
// vmcnt says its pieces of the step have landed, the barrier says everybody's have, and that everybody is done with the previous step,
// whose stage the next requests overwrite.  Metadata of super-block sb is requested at its first step and read by the fold after its
// last (single-buffered: the next request follows the barrier behind that fold).
constexpr int SH_BM = 256, SH_BN = 128, SH_NW = 8;
constexpr int SH_STAGE = SH_BM * 64 + SH_BN * 64 * 2;  // 32768
constexpr int SH_NS = 4;
constexpr int SH_LDS_BST = SH_NS * SH_STAGE;            // token bsums  [256][32 B]
constexpr int SH_LDS_DY = SH_LDS_BST + SH_BM * 32;      // token scales [256] f32
constexpr int SH_LDS_WM = SH_LDS_DY + SH_BM * 4;        // weight metadata [4 panels][1280 B]
constexpr int SH_LDS_BYTES = SH_LDS_WM + 4 * SH_META;   // 145 408

__device__ __forceinline__ void sh_dma16(const void * g, const uint32_t lds) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}
__device__ __forceinline__ void sh_dma4(const void * g, const uint32_t lds) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}
// the same with a wave-uniform 64-bit base in SGPRs and a 32-bit per-lane offset: one VGPR per request stream instead of two (the
// ping-pong form runs at the 256-register limit).  s_nop 4: the base may come fresh out of a v_readfirstlane (guide §5.7 item 2).
__device__ __forceinline__ void sh_dma16s(const uint32_t voff, const void * sbase, const uint32_t lds) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(lds), "s"(sbase) : "memory");
}
__device__ __forceinline__ void sh_dma4s(const uint32_t voff, const void * sbase, const uint32_t lds) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dword %1, %3\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(lds), "s"(sbase) : "memory");
}
template <int N> __device__ __forceinline__ void sh_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// SH_STAMP (probe builds only): workgroup 0's waves accumulate s_memtime deltas per phase into a.stamps[wave][phase]
#ifndef SH_STAMP
#define SH_STAMP 0
#endif
#if SH_STAMP
#define SH_T(i) { const unsigned long long t_ = __builtin_readcyclecounter(); st_acc[i] += t_ - st_t; st_t = t_; }
#else
#define SH_T(i)
#endif
#ifndef SH_INTERLEAVE
#define SH_INTERLEAVE 1  // a step's requests between its MFMAs (0: all four right behind the barrier)
#endif
#ifndef SH_KO
#define SH_KO 0  // timing-only knock-outs (wrong results): 1 no MFMAs, 2 no fold, 3 no requests after the prologue, 4 no fragment reads
#endif

template <bool MINS>
__global__ void __launch_bounds__(SH_NW * 64, 2) k_mmq_shadow(const shadow_args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NM = MINS ? 4 : 2;  // metadata requests per wave and super-block
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, kg = lane >> 5;
    const int wm = wave & 3, wn = wave >> 2;
    // items of one weight group run on the same XCD back to back: its L2 serves the planes to every token tile
    const int xcd = blockIdx.x & 7, qb = blockIdx.x >> 3;
    const int group = (qb / a.m_tiles) * 8 + xcd, mt = qb % a.m_tiles;
    if (group >= a.n_groups) return;
    const int mi = (a.n_mat > 2 && group >= a.mat[2].group0) ? 2 : ((a.n_mat > 1 && group >= a.mat[1].group0) ? 1 : 0);
#define MAT_SEL(f) (mi == 0 ? a.mat[0].f : (mi == 1 ? a.mat[1].f : a.mat[2].f))
    const int gl = group - MAT_SEL(group0);  // group within its matrix
    const int nblk = a.K / 256;
    const int m0 = mt * SH_BM;
    const int tok_bytes = nblk * (int) sizeof(q8k_dev);
    const uint32_t lds0 = (uint32_t) (uintptr_t) smem;

    // ---- request roles.  Tokens: 16 instructions per stage (16 rows x 64 B each), wave w issues 2 w and 2 w + 1; lane p -> row p >> 2,
    // LDS slot p & 3 holds source chunk (p & 3) ^ ((row >> 2) & 3)
    const char * tsrc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int row = 16 * (2 * wave + u) + (lane >> 2);
        const int chunk = (lane & 3) ^ ((row >> 2) & 3);
        tsrc[u] = (const char *) a.act + (size_t) min(m0 + row, a.M - 1) * tok_bytes + chunk * 16;
    }
    // weights: panel wave >> 1, pieces 2 (wave & 1) and + 1 of its 4 KB per step; contiguous in HBM over the whole K range
    const char * wsrc = MAT_SEL(planes) + (size_t) (gl * 4 + (wave >> 1)) * nblk * SH_PANEL_SB + (wave & 1) * 2048 + lane * 16;
    // metadata: token bsums 512 pieces (token p >> 1, half p & 1): one instruction per wave; token scales: four dword instructions, waves
    // w and w + 4 both issue number w & 3 (same bytes, same place: every wave then counts the same operations); weights: panel w & 3
    const char * msrc_bs = (const char *) a.act + (size_t) min(m0 + 32 * wave + (lane >> 1), a.M - 1) * tok_bytes + 256 + (lane & 1) * 16;
    const char * msrc_dy = (const char *) a.act + (size_t) min(m0 + 64 * (wave & 3) + lane, a.M - 1) * tok_bytes + 304;
    const char * msrc_w = MAT_SEL(meta) + (size_t) (gl * 4 + (wave & 3)) * nblk * SH_META + lane * 16;

    const int sb_lo_ = (int) (((int64_t) blockIdx.y * nblk) / a.ksplit);
    // one of a step's four requests of this wave (0, 1: its token rows; 2, 3: its weight pieces)
    auto issue_piece = [&](const int sb, const int q, const int buf, const int i) {
#if SH_KO == 5
        if (i < 2 && !(sb == sb_lo_ && q < 3)) return;  // timing only: no token requests after the prologue
#elif SH_KO == 6
        if (i >= 2 && !(sb == sb_lo_ && q < 3)) return;  // timing only: no weight requests after the prologue
#endif
        const uint32_t tb = lds0 + buf * SH_STAGE;
        if (i < 2) {
            sh_dma16(tsrc[i] + (size_t) sb * sizeof(q8k_dev) + q * 64, tb + (2 * wave + i) * 1024);
        } else {
            const char * ws = wsrc + ((size_t) sb * 4 + q) * 4096 + (i - 2) * 1024;
            sh_dma16(ws, tb + SH_BM * 64 + (wave >> 1) * 4096 + (wave & 1) * 2048 + (i - 2) * 1024);
        }
    };
    auto issue_stage = [&](const int sb, const int q, const int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_piece(sb, q, buf, i);
    };
    auto issue_meta = [&](const int sb) {
        const size_t to = (size_t) sb * sizeof(q8k_dev);
        if constexpr (MINS) sh_dma16(msrc_bs + to, lds0 + SH_LDS_BST + wave * 1024);
        sh_dma4(msrc_dy + to, lds0 + SH_LDS_DY + (wave & 3) * 256);
        const char * ms = msrc_w + (size_t) sb * SH_META;
        const uint32_t mb = lds0 + SH_LDS_WM + (wave & 3) * SH_META;
        if constexpr (MINS) sh_dma16(ms, mb);
        if (lane < 16) sh_dma16(ms + 1024, mb + 1024);
    };

    sh_f16v C[2][2];       // [row tile][token tile]
    sh_i16v acc[2][2][2];  // [plane][row tile][token tile]
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                C[rt][ta][r] = 0.0f;
                acc[0][rt][ta][r] = 0;
                acc[1][rt][ta][r] = 0;
            }
    const sh_f16v zerof = C[0][0];
    const sh_i16v zeroi = acc[0][0][0];

    // fragment addresses: rows 64 wm + 32 ta + fr of the token area, rows fr of panel 2 wn + rt; 16-byte slot (2 kk + kg) ^ ((fr >> 2) & 3)
    const int swz = (fr >> 2) & 3;
    const int co0 = ((0 + kg) ^ swz) << 4, co1 = ((2 + kg) ^ swz) << 4;
    const char * const tfrag = smem + (64 * wm + fr) * 64;
    const char * const wfrag = smem + SH_BM * 64 + (2 * wn) * 4096 + fr * 64;

#if SH_STAMP
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t = __builtin_readcyclecounter();
#endif
    const int sb_lo = (int) (((int64_t) blockIdx.y * nblk) / a.ksplit), sb_hi = (int) (((int64_t) (blockIdx.y + 1) * nblk) / a.ksplit);
    const int g_lo = sb_lo * 4, g_last = sb_hi * 4 - 1;
    // prologue: steps 0, 1, 2
#pragma unroll
    for (int g = 0; g < 3; ++g) issue_stage(min(g_lo + g, g_last) >> 2, min(g_lo + g, g_last) & 3, g);

    for (int sb = sb_lo; sb < sb_hi; ++sb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // this wave's pieces of step (sb, q) have landed: what it requested since are the two later steps (+ this super-block's metadata)
            SH_T(0)
            if (q == 0 || q == 3) sh_wait<8>(); else sh_wait<8 + NM>();
            SH_T(1)
            __syncthreads();
            SH_T(2)
            if (q == 0) issue_meta(sb);
            const int g3 = min(sb * 4 + q + 3, g_last);  // (past the end: the last stage again, into a stage nobody reads any more)
#if SH_KO == 3
            const bool do_issue = sb == sb_lo && q == 0;
#else
            const bool do_issue = true;
#endif
#if !SH_INTERLEAVE
            if (do_issue) issue_stage(g3 >> 2, g3 & 3, (q + 3) & 3);
#endif
            SH_T(3)
            const char * tb = tfrag + q * SH_STAGE;
            const char * wb = wfrag + q * SH_STAGE;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int co = kk == 0 ? co0 : co1;
                sh_i4v fa[2], fb[2][2];
#if SH_KO == 4
                fa[0] = fa[1] = fb[0][0] = fb[0][1] = fb[1][0] = fb[1][1] = (sh_i4v){lane, kk, q, sb};
#else
#pragma unroll
                for (int ta = 0; ta < 2; ++ta) fa[ta] = *(const sh_i4v *) (tb + ta * 2048 + co);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int p = 0; p < 2; ++p) fb[rt][p] = *(const sh_i4v *) (wb + rt * 4096 + p * 2048 + co);
#endif
#if SH_KO != 1
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
                    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                        for (int p = 0; p < 2; ++p)
                            acc[p][rt][ta] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[ta], fb[rt][p], (q == 0 && kk == 0) ? zeroi : acc[p][rt][ta], 0, 0, 0);
#if SH_INTERLEAVE
                    // a request between the MFMAs: while it waits for room in the CU's address pipeline the matrix pipe works on what
                    // was issued before it, and the other wave of this SIMD can issue (requests behind the barrier, all eight waves at
                    // once: 630 - 1090 cycles of a 2 600-cycle step were spent issuing them — profiles/r03_shadow_probe.txt)
                    __builtin_amdgcn_sched_barrier(0);
                    if (do_issue) issue_piece(g3 >> 2, g3 & 3, (q + 3) & 3, 2 * kk + rt);
                    __builtin_amdgcn_sched_barrier(0);
#endif
                }
#else
                asm volatile("" ::"v"(fa[0]), "v"(fa[1]), "v"(fb[0][0]), "v"(fb[0][1]), "v"(fb[1][0]), "v"(fb[1][1]));
#if SH_INTERLEAVE
                if (do_issue) { issue_piece(g3 >> 2, g3 & 3, (q + 3) & 3, 2 * kk); issue_piece(g3 >> 2, g3 & 3, (q + 3) & 3, 2 * kk + 1); }
#endif
#endif
            }
        }
        SH_T(4)
#if SH_KO != 2
        // ---- fold of super-block sb (its metadata landed before the operands of its last step: in-order vmcnt + that step's barrier)
        {
            const char * wmeta = smem + SH_LDS_WM + (2 * wn) * SH_META;
            float2 dd[2];
            sh_half8 fbm[2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                dd[rt] = *(const float2 *) (wmeta + rt * SH_META + 1024 + fr * 8);
                if constexpr (MINS) fbm[rt] = *(const sh_half8 *) (wmeta + rt * SH_META + fr * 32 + kg * 16);
            }
#pragma unroll
            for (int ta = 0; ta < 2; ++ta) {
                sh_half8 fam;
                if constexpr (MINS) fam = *(const sh_half8 *) (smem + SH_LDS_BST + (64 * wm + 32 * ta + fr) * 32 + kg * 16);
                float dy[16];
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const float4 t = *(const float4 *) (smem + SH_LDS_DY + (64 * wm + 32 * ta + 8 * g4 + 4 * kg) * 4);
                    dy[4 * g4] = t.x; dy[4 * g4 + 1] = t.y; dy[4 * g4 + 2] = t.z; dy[4 * g4 + 3] = t.w;
                }
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    sh_f16v am = zerof;
                    if constexpr (MINS) am = __builtin_amdgcn_mfma_f32_32x32x16_f16(fam, fbm[rt], zerof, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int is = (acc[1][rt][ta][r] << 7) + acc[0][rt][ta][r];
                        float v = dd[rt].x * (float) is;
                        if constexpr (MINS) v = __builtin_fmaf(-dd[rt].y, am[r], v);
                        C[rt][ta][r] = __builtin_fmaf(dy[r], v, C[rt][ta][r]);
                    }
                }
            }
        }
#endif
    }
    SH_T(5)
    sh_wait<0>();
#if SH_STAMP
    if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && a.stamps)
        for (int i = 0; i < 8; ++i) a.stamps[wave * 8 + i] = st_acc[i];
#endif
    // ---- store: lane = weight row, register r = token (r & 3) + 8 (r >> 2) + 4 kg of its tile
    const int mN = MAT_SEL(N);
    float * const m_dst = a.ksplit > 1 ? MAT_SEL(part) + (size_t) blockIdx.y * a.M * mN : MAT_SEL(dst);
    const int64_t m_dst_stride = a.ksplit > 1 ? (int64_t) mN : MAT_SEL(dst_stride);
    const float * const m_add = a.ksplit > 1 ? nullptr : MAT_SEL(add);
    const int64_t m_add_stride = MAT_SEL(add_stride);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int n = gl * SH_BN + 64 * wn + 32 * rt + fr;
#pragma unroll
        for (int ta = 0; ta < 2; ++ta) {
            const int mb = m0 + 64 * wm + 32 * ta + 4 * kg;
            if (m_add) {  // (addends first, then nothing but stores: a load between stores makes the compiler wait vmcnt(0) before each store)
#pragma unroll
                for (int r = 0; r < 16; ++r) C[rt][ta][r] += m_add[(size_t) min(mb + (r & 3) + 8 * (r >> 2), a.M - 1) * m_add_stride + n];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb + (r & 3) + 8 * (r >> 2);
                if (m < a.M) m_dst[(size_t) m * m_dst_stride + n] = C[rt][ta][r];
            }
        }
    }
#undef MAT_SEL
}


// ------------------------------------------------------------------------------------------------ the GEMM, second form: two wave groups in turn
// Stand-alone measurements of the form above (scripts/ubench/shadow_probe.hip, profiles/r03_shadow_probe.txt): all eight waves do the same
// thing at the same time — request, read fragments, multiply — and a step's components ADD UP in every wave's in-order instruction stream
// (knock-outs: MFMAs 55, fold 61, requests 42, fragment reads 59 of 197 us), the matrix pipe a quarter busy.  Here the two waves of a SIMD
// take turns: waves 0-3 (group 0) and 4-7 (group 1, one barrier behind) alternate between a LOAD segment — fragment reads of the next 32
// values of K, two of the step's four requests, the fold at a super-block boundary — and a COMPUTE segment of 8 MFMAs on registers only, a
// barrier between segments: while one wave of a SIMD multiplies, the other one waits on the address pipeline / LDS / VALU (the structure of
// the guide's 8-phase GEMM template, cdna_hip_programming.md §5).
// Global barrier 4 g opens step g for group 0 (4 g + 1 for group 1); every wave's pieces of step g have landed before barrier 4 g (group 0
// waits in front of its own first barrier of the step, group 1 in front of its last barrier of step g - 1), and the stage of step g - 1 is
// overwritten only by requests issued behind that barrier.  The metadata is double-buffered by super-block parity: group 1 folds
// super-block sb - 1 while group 0 has already requested sb's.
constexpr int SH2_META_BYTES = SH_BM * 32 + SH_BM * 4 + 4 * SH_META;  // token bsums | token scales | weight metadata of 4 panels = 14 336
constexpr int SH2_LDS_META = SH_NS * SH_STAGE;
constexpr int SH2_LDS_BYTES = SH2_LDS_META + 2 * SH2_META_BYTES;      // 159 744

#define SH_BAR()                                 \
    {                                            \
        __builtin_amdgcn_sched_barrier(0);       \
        __syncthreads();                         \
        __builtin_amdgcn_sched_barrier(0);       \
    }

template <bool MINS>
__global__ void __launch_bounds__(SH_NW * 64, 2) k_mmq_shadow_pp(const shadow_args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NM = 2;  // metadata requests per wave and super-block in the main loop (token scales, weight d)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, kg = lane >> 5;
    const int wm = wave & 3, wn = wave >> 2;
    const int grp = wn;  // group 1 runs one barrier behind group 0
    const int xcd = blockIdx.x & 7, qb = blockIdx.x >> 3;
    const int group = (qb / a.m_tiles) * 8 + xcd, mt = qb % a.m_tiles;
    if (group >= a.n_groups) return;
    const int mi = (a.n_mat > 2 && group >= a.mat[2].group0) ? 2 : ((a.n_mat > 1 && group >= a.mat[1].group0) ? 1 : 0);
#define MAT_SEL(f) (mi == 0 ? a.mat[0].f : (mi == 1 ? a.mat[1].f : a.mat[2].f))
    const int gl = group - MAT_SEL(group0);
    const int nblk = a.K / 256;
    const int m0 = mt * SH_BM;
    const int tok_bytes = nblk * (int) sizeof(q8k_dev);
    const uint32_t lds0 = (uint32_t) (uintptr_t) smem;

    // request roles as in the form above, as 32-bit per-lane offsets from wave-uniform bases.  The offsets are RECOMPUTED from the lane id at
    // every request (a handful of VALU beside a ~100-cycle issue) instead of living in registers: the kernel runs at the 256-register limit
    // and eight spilled accumulators cost a vmcnt(0) — the whole request queue drained — per super-block.
    const char * const act_b = (const char *) a.act;
    const char * const wsrc_b = MAT_SEL(planes) + (size_t) (gl * 4 + (wave >> 1)) * nblk * SH_PANEL_SB + (wave & 1) * 2048;
    const char * const msrc_wb = MAT_SEL(meta) + (size_t) (gl * 4 + (wave & 3)) * nblk * SH_META;
    auto opaque_lane = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };

    auto issue_piece = [&](const int sb, const int q, const int buf, const int i) {
        const uint32_t tb = lds0 + buf * SH_STAGE;
#if SH_KO == 8 || SH_KO == 10
        if (i < 2) return;   // timing only: requests alone, weights only
#elif SH_KO == 9
        if (i >= 2) return;  // timing only: requests alone, tokens only
#endif
        const int l = opaque_lane();
        if (i < 2) {
            const int row = 16 * (2 * wave + i) + (l >> 2);
            const uint32_t toff = (uint32_t) min(m0 + row, a.M - 1) * (uint32_t) tok_bytes + (((l & 3) ^ ((row >> 2) & 3)) << 4);
            sh_dma16s(toff, act_b + (size_t) sb * sizeof(q8k_dev) + q * 64, tb + (2 * wave + i) * 1024);
        } else {
            sh_dma16s((uint32_t) l * 16u, wsrc_b + ((size_t) sb * 4 + q) * 4096 + (i - 2) * 1024, tb + SH_BM * 64 + (wave >> 1) * 4096 + (wave & 1) * 2048 + (i - 2) * 1024);
        }
    };
    auto issue_meta = [&](const int sb) {  // main loop: token scales + weight (d, dmin)
        const uint32_t mbase = lds0 + SH2_LDS_META + (sb & 1) * SH2_META_BYTES;
        const int l = opaque_lane();
        sh_dma4s((uint32_t) min(m0 + 64 * (wave & 3) + l, a.M - 1) * (uint32_t) tok_bytes + 304, act_b + (size_t) sb * sizeof(q8k_dev), mbase + SH_BM * 32 + (wave & 3) * 256);
        if (l < 16) sh_dma16s((uint32_t) l * 16u, msrc_wb + (size_t) sb * SH_META + 1024, mbase + SH_BM * 32 + SH_BM * 4 + (wave & 3) * SH_META + 1024);
    };
    auto issue_meta_mins = [&](const int sb) {  // mins pass: + token bsums + weight mins (4 requests per wave)
        const uint32_t mbase = lds0 + SH2_LDS_META + (sb & 1) * SH2_META_BYTES;
        const int l = opaque_lane();
        sh_dma16s((uint32_t) min(m0 + 32 * wave + (l >> 1), a.M - 1) * (uint32_t) tok_bytes + 256 + (l & 1) * 16, act_b + (size_t) sb * sizeof(q8k_dev), mbase + wave * 1024);
        sh_dma16s((uint32_t) l * 16u, msrc_wb + (size_t) sb * SH_META, mbase + SH_BM * 32 + SH_BM * 4 + (wave & 3) * SH_META);
        issue_meta(sb);
    };

    sh_f16v C[2][2];
    sh_i16v acc[2][2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                C[rt][ta][r] = 0.0f;
                acc[0][rt][ta][r] = 0;
                acc[1][rt][ta][r] = 0;
            }
    const sh_f16v zerof = C[0][0];
    const sh_i16v zeroi = acc[0][0][0];

    // fold of a finished super-block: C += dy (d float(lo + 128 hi)).  The mins term of Q4_K / Q5_K is NOT here: it is summed for all
    // super-blocks by the pass in front of the main loop (below) — ggml-cpu keeps it in a running sum of its own too — because its product
    // tile (16 registers) beside the 192 accumulator registers spilled accumulators, and every reload drained the request queue (vmcnt(0)).
    auto fold = [&](const int sb) {
        const char * mbase = smem + SH2_LDS_META + (sb & 1) * SH2_META_BYTES;
        const char * wmeta = mbase + SH_BM * 32 + SH_BM * 4 + (2 * wn) * SH_META;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const float d = *(const float *) (wmeta + rt * SH_META + 1024 + fr * 8);
#pragma unroll
            for (int ta = 0; ta < 2; ++ta) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const float4 t = *(const float4 *) (mbase + SH_BM * 32 + (64 * wm + 32 * ta + 8 * g4 + 4 * kg) * 4);
                    const float dy[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * g4 + e;
                        const int is = (acc[1][rt][ta][r] << 7) + acc[0][rt][ta][r];
                        C[rt][ta][r] = __builtin_fmaf(dy[e], d * (float) is, C[rt][ta][r]);
                    }
                }
            }
        }
    };
    // mins pass of one super-block: C -= dy (dmin sum_j m_j bsum_j), one f16 MFMA per tile
    auto mins_pass = [&](const int sb) {
        const char * mbase = smem + SH2_LDS_META + (sb & 1) * SH2_META_BYTES;
        const char * wmeta = mbase + SH_BM * 32 + SH_BM * 4 + (2 * wn) * SH_META;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const float dmin = *(const float *) (wmeta + rt * SH_META + 1024 + fr * 8 + 4);
            const sh_half8 fbm = *(const sh_half8 *) (wmeta + rt * SH_META + fr * 32 + kg * 16);
#pragma unroll
            for (int ta = 0; ta < 2; ++ta) {
                const sh_half8 fam = *(const sh_half8 *) (mbase + (64 * wm + 32 * ta + fr) * 32 + kg * 16);
                const sh_f16v am = __builtin_amdgcn_mfma_f32_32x32x16_f16(fam, fbm, zerof, 0, 0, 0);
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const float4 t = *(const float4 *) (mbase + SH_BM * 32 + (64 * wm + 32 * ta + 8 * g4 + 4 * kg) * 4);
                    const float dy[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * g4 + e;
                        C[rt][ta][r] = __builtin_fmaf(-dy[e], dmin * am[r], C[rt][ta][r]);
                    }
                }
            }
        }
    };

    const int swz = (fr >> 2) & 3;
    const int co0 = ((0 + kg) ^ swz) << 4, co1 = ((2 + kg) ^ swz) << 4;
    const char * const tfrag = smem + (64 * wm + fr) * 64;
    const char * const wfrag = smem + SH_BM * 64 + (2 * wn) * 4096 + fr * 64;

    const int sb_lo = (int) (((int64_t) blockIdx.y * nblk) / a.ksplit), sb_hi = (int) (((int64_t) (blockIdx.y + 1) * nblk) / a.ksplit);
    const int g_lo = sb_lo * 4, g_last = sb_hi * 4 - 1;
    if constexpr (MINS) {
        // ---- the mins term of every super-block of this K range, before the main loop (all eight waves in step; the metadata of
        // super-block sb + 1 is requested while sb is summed: two buffers by parity)
        issue_meta_mins(sb_lo);
        for (int sb = sb_lo; sb < sb_hi; ++sb) {
            issue_meta_mins(min(sb + 1, sb_hi - 1));
            sh_wait<4>();
            SH_BAR()
            mins_pass(sb);
            SH_BAR()  // (everybody is done reading before the buffer of this parity is requested again)
        }
        sh_wait<0>();
        SH_BAR()
    }
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_piece(min(g_lo + g, g_last) >> 2, min(g_lo + g, g_last) & 3, g, i);
    if (grp == 1) {  // one barrier behind: its pieces of the first step have landed before the barrier that opens that step for group 0
        sh_wait<8>();
        SH_BAR()
    }

    sh_i4v fa[2], fb[2][2];
    for (int sb = sb_lo; sb < sb_hi; ++sb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int g3 = min(sb * 4 + q + 3, g_last);
            const char * tb = tfrag + q * SH_STAGE;
            const char * wb = wfrag + q * SH_STAGE;
            if (grp == 0) { if (q == 0 || q == 3) sh_wait<8>(); else sh_wait<8 + NM>(); }
            SH_BAR()
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                // ---- LOAD segment: requests first (they take the longest to come back), the fold of the finished super-block while no
                // fragment is live, then the fragment reads
                const int co = kk == 0 ? co0 : co1;
#if SH_KO == 3 || SH_KO == 11 || SH_KO == 12 || SH_KO == 13
                if (sb == sb_lo && q == 0)
#endif
                {
                if (q == 0 && kk == 0) issue_meta(sb);
                issue_piece(g3 >> 2, g3 & 3, (q + 3) & 3, 2 * kk);  // ONE request per segment and wave: a wave's second request in a row waits ~270 cycles for the first
                }
#if SH_KO != 2 && SH_KO < 7
                if (q == 0 && kk == 0 && sb > sb_lo) {
                    fold(sb - 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
#endif
#if SH_KO == 4 || (SH_KO >= 7 && SH_KO != 11)
                fa[0] = fa[1] = fb[0][0] = fb[0][1] = fb[1][0] = fb[1][1] = (sh_i4v){lane, kk, q, sb};
#else
#pragma unroll
                for (int ta = 0; ta < 2; ++ta) fa[ta] = *(const sh_i4v *) (tb + ta * 2048 + co);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int p = 0; p < 2; ++p) fb[rt][p] = *(const sh_i4v *) (wb + rt * 4096 + p * 2048 + co);
#endif
                if (kk == 1 && grp == 1) {  // group 1: the next step opens (for group 0) at the barrier that ends this segment
                    // (behind the next step's four pieces: step g - 1's four, this step's first three, + the metadata of a super-block begun in either)
                    if (q == 3 || q == 2) sh_wait<7>(); else sh_wait<7 + NM>();
                }
                SH_BAR()
                // ---- COMPUTE segment: registers only
#if SH_KO == 1 || (SH_KO >= 7 && SH_KO != 12)
                asm volatile("" ::"v"(fa[0]), "v"(fa[1]), "v"(fb[0][0]), "v"(fb[0][1]), "v"(fb[1][0]), "v"(fb[1][1]));
#else
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                        for (int p = 0; p < 2; ++p)
                            acc[p][rt][ta] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[ta], fb[rt][p], (q == 0 && kk == 0) ? zeroi : acc[p][rt][ta], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
#endif
                // the step's other two requests ride behind the MFMAs of the COMPUTE segments (issued, the matrix pipe works them off)
                __builtin_amdgcn_sched_barrier(0);
#if SH_KO == 3 || SH_KO == 11 || SH_KO == 12 || SH_KO == 13
                if (sb == sb_lo && q == 0)
#endif
                issue_piece(g3 >> 2, g3 & 3, (q + 3) & 3, 2 * kk + 1);
                if (kk == 0) SH_BAR()
            }
        }
    }
    fold(sb_hi - 1);
    if (grp == 0) SH_BAR()  // (group 1 took one barrier more at the start)
    sh_wait<0>();
    const int mN = MAT_SEL(N);
    float * const m_dst = a.ksplit > 1 ? MAT_SEL(part) + (size_t) blockIdx.y * a.M * mN : MAT_SEL(dst);
    const int64_t m_dst_stride = a.ksplit > 1 ? (int64_t) mN : MAT_SEL(dst_stride);
    const float * const m_add = a.ksplit > 1 ? nullptr : MAT_SEL(add);
    const int64_t m_add_stride = MAT_SEL(add_stride);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int n = gl * SH_BN + 64 * wn + 32 * rt + fr;
#pragma unroll
        for (int ta = 0; ta < 2; ++ta) {
            const int mb = m0 + 64 * wm + 32 * ta + 4 * kg;
            if (m_add) {
#pragma unroll
                for (int r = 0; r < 16; ++r) C[rt][ta][r] += m_add[(size_t) min(mb + (r & 3) + 8 * (r >> 2), a.M - 1) * m_add_stride + n];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb + (r & 3) + 8 * (r >> 2);
                if (m < a.M) m_dst[(size_t) m * m_dst_stride + n] = C[rt][ta][r];
            }
        }
    }
#undef MAT_SEL
}


// ------------------------------------------------------------------------------------------------ the GEMM, third form: two independent workgroups per CU
// 4 waves = 128 tokens x 128 rows (wave (wm, wn): tokens [64 wm, +64) x rows [64 wn, +64), the same 2 x 2 tiles per wave), 75 KB of LDS, so
// that TWO workgroups share a CU: the two waves of a SIMD belong to different workgroups, meet at no common barrier and drift apart by
// themselves — one's requests / fragment reads / fold beside the other's MFMAs without any hand-made schedule.  Price: 24 KB per step for
// half the MACs of the 256 x 128 tile (1.5 x the bytes per MAC).  One barrier per step as in the first form; mins pre-pass and lean fold as
// in the second.
constexpr int SH3_BM = 128, SH3_NW = 4, SH3_NS = 3;
constexpr int SH3_STAGE = SH3_BM * 64 + SH_BN * 64 * 2;          // 24 576
constexpr int SH3_LDS_DY = SH3_NS * SH3_STAGE;                    // token scales [128] f32
constexpr int SH3_LDS_DD = SH3_LDS_DY + SH3_BM * 4;               // weight (d, dmin) [4 panels][32 rows] float2
constexpr int SH3_LDS_BYTES = SH3_LDS_DD + 4 * 256;               // 75 264

template <bool MINS>
__global__ void __launch_bounds__(SH3_NW * 64, 2) k_mmq_shadow3(const shadow_args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int OPS = 6;  // requests per wave and step: 2 token pieces + its panel's 4 weight pieces
    constexpr int NM = 2;   // + per super-block: token scales, weight (d, dmin)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, kg = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;
    const int xcd = blockIdx.x & 7, qb = blockIdx.x >> 3;
    const int group = (qb / a.m_tiles) * 8 + xcd, mt = qb % a.m_tiles;  // (m_tiles counts 128-token tiles for this kernel)
    if (group >= a.n_groups) return;
    const int mi = (a.n_mat > 2 && group >= a.mat[2].group0) ? 2 : ((a.n_mat > 1 && group >= a.mat[1].group0) ? 1 : 0);
#define MAT_SEL(f) (mi == 0 ? a.mat[0].f : (mi == 1 ? a.mat[1].f : a.mat[2].f))
    const int gl = group - MAT_SEL(group0);
    const int nblk = a.K / 256;
    const int m0 = mt * SH3_BM;
    const int tok_bytes = nblk * (int) sizeof(q8k_dev);
    const uint32_t lds0 = (uint32_t) (uintptr_t) smem;
    const char * const act_b = (const char *) a.act;
    const char * const wsrc_b = MAT_SEL(planes) + (size_t) (gl * 4 + wave) * nblk * SH_PANEL_SB;
    const char * const msrc_wb = MAT_SEL(meta) + (size_t) (gl * 4 + wave) * nblk * SH_META;
    auto opaque_lane = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };

    auto issue_stage = [&](const int sb, const int q, const int buf) {
        const uint32_t tb = lds0 + buf * SH3_STAGE;
        const int l = opaque_lane();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = 16 * (2 * wave + i) + (l >> 2);
            const uint32_t toff = (uint32_t) min(m0 + row, a.M - 1) * (uint32_t) tok_bytes + (((l & 3) ^ ((row >> 2) & 3)) << 4);
            sh_dma16s(toff, act_b + (size_t) sb * sizeof(q8k_dev) + q * 64, tb + (2 * wave + i) * 1024);
        }
        const char * ws = wsrc_b + ((size_t) sb * 4 + q) * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) sh_dma16s((uint32_t) l * 16u, ws + i * 1024, tb + SH3_BM * 64 + wave * 4096 + i * 1024);
    };
    auto issue_meta = [&](const int sb) {  // token scales (waves w and w + 2 both request piece w & 1) + this wave's panel (d, dmin)
        const int l = opaque_lane();
        sh_dma4s((uint32_t) min(m0 + 64 * (wave & 1) + l, a.M - 1) * (uint32_t) tok_bytes + 304, act_b + (size_t) sb * sizeof(q8k_dev), lds0 + SH3_LDS_DY + (wave & 1) * 256);
        if (l < 16) sh_dma16s((uint32_t) l * 16u, msrc_wb + (size_t) sb * SH_META + 1024, lds0 + SH3_LDS_DD + wave * 256);
    };

    sh_f16v C[2][2];
    sh_i16v acc[2][2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                C[rt][ta][r] = 0.0f;
                acc[0][rt][ta][r] = 0;
                acc[1][rt][ta][r] = 0;
            }
    const sh_f16v zerof = C[0][0];
    const sh_i16v zeroi = acc[0][0][0];

    const int sb_lo = (int) (((int64_t) blockIdx.y * nblk) / a.ksplit), sb_hi = (int) (((int64_t) (blockIdx.y + 1) * nblk) / a.ksplit);
    const int g_lo = sb_lo * 4, g_last = sb_hi * 4 - 1;
    if constexpr (MINS) {
        // ---- mins pre-pass: C = - sum_sb dy dmin (sum_j m_j bsum_j).  The stage ring is still empty: two buffers of [token bsums 128 x 32 B |
        // token scales 512 B | weight metadata 4 x 1280 B] = 9.5 KB live at its start
        constexpr int MB = SH3_BM * 32 + SH3_BM * 4 + 4 * SH_META;
        auto issue_mins = [&](const int sb) {
            const uint32_t mb = lds0 + (sb & 1) * MB;
            const int l = opaque_lane();
            const char * ab = act_b + (size_t) sb * sizeof(q8k_dev);
            sh_dma16s((uint32_t) min(m0 + 32 * wave + (l >> 1), a.M - 1) * (uint32_t) tok_bytes + 256 + (l & 1) * 16, ab, mb + wave * 1024);
            sh_dma4s((uint32_t) min(m0 + 64 * (wave & 1) + l, a.M - 1) * (uint32_t) tok_bytes + 304, ab, mb + SH3_BM * 32 + (wave & 1) * 256);
            const char * ms = msrc_wb + (size_t) sb * SH_META;
            sh_dma16s((uint32_t) l * 16u, ms, mb + SH3_BM * 32 + SH3_BM * 4 + wave * SH_META);
            if (l < 16) sh_dma16s((uint32_t) l * 16u, ms + 1024, mb + SH3_BM * 32 + SH3_BM * 4 + wave * SH_META + 1024);
        };
        issue_mins(sb_lo);
        for (int sb = sb_lo; sb < sb_hi; ++sb) {
            issue_mins(min(sb + 1, sb_hi - 1));
            sh_wait<4>();
            __syncthreads();
            const char * mbase = smem + (sb & 1) * MB;
            const char * wmeta = mbase + SH3_BM * 32 + SH3_BM * 4 + (2 * wn) * SH_META;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const float dmin = *(const float *) (wmeta + rt * SH_META + 1024 + fr * 8 + 4);
                const sh_half8 fbm = *(const sh_half8 *) (wmeta + rt * SH_META + fr * 32 + kg * 16);
#pragma unroll
                for (int ta = 0; ta < 2; ++ta) {
                    const sh_half8 fam = *(const sh_half8 *) (mbase + (64 * wm + 32 * ta + fr) * 32 + kg * 16);
                    const sh_f16v am = __builtin_amdgcn_mfma_f32_32x32x16_f16(fam, fbm, zerof, 0, 0, 0);
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const float4 t = *(const float4 *) (mbase + SH3_BM * 32 + (64 * wm + 32 * ta + 8 * g4 + 4 * kg) * 4);
                        const float dy[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) C[rt][ta][4 * g4 + e] = __builtin_fmaf(-dy[e], dmin * am[4 * g4 + e], C[rt][ta][4 * g4 + e]);
                    }
                }
            }
            __syncthreads();
        }
        sh_wait<0>();
        __syncthreads();
    }

    const int swz = (fr >> 2) & 3;
    const int co0 = ((0 + kg) ^ swz) << 4, co1 = ((2 + kg) ^ swz) << 4;
    const char * const tfrag = smem + (64 * wm + fr) * 64;
    const char * const wfrag = smem + SH3_BM * 64 + (2 * wn) * 4096 + fr * 64;
    // prologue: steps 0 and 1 (three stages, requested two steps ahead)
#pragma unroll
    for (int g = 0; g < 2; ++g) issue_stage(min(g_lo + g, g_last) >> 2, min(g_lo + g, g_last) & 3, g);
    int buf = 0;  // stage of the current step (steps are not a multiple of three: a running index)
    for (int sb = sb_lo; sb < sb_hi; ++sb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // this wave's pieces of step (sb, q) have landed: requested since are the next step's (+ this super-block's metadata at q = 1:
            // it was requested at q = 0 behind the barrier, in front of that step's stage)
            if (q == 1) sh_wait<OPS + NM>(); else sh_wait<OPS>();
            __syncthreads();
            if (q == 0) issue_meta(sb);
            {
                const int g2 = min(sb * 4 + q + 2, g_last);
                int nb = buf + 2;
                nb = nb >= 3 ? nb - 3 : nb;
                issue_stage(g2 >> 2, g2 & 3, nb);
            }
            const char * tb = tfrag + buf * SH3_STAGE;
            const char * wb = wfrag + buf * SH3_STAGE;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int co = kk == 0 ? co0 : co1;
                sh_i4v fa[2], fb[2][2];
#pragma unroll
                for (int ta = 0; ta < 2; ++ta) fa[ta] = *(const sh_i4v *) (tb + ta * 2048 + co);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int p = 0; p < 2; ++p) fb[rt][p] = *(const sh_i4v *) (wb + rt * 4096 + p * 2048 + co);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                        for (int p = 0; p < 2; ++p)
                            acc[p][rt][ta] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[ta], fb[rt][p], (q == 0 && kk == 0) ? zeroi : acc[p][rt][ta], 0, 0, 0);
            }
            buf = buf == 2 ? 0 : buf + 1;
        }
        // ---- fold (metadata of sb: requested at q = 0, older than the stage waited for at q = 2, visible since that step's barrier)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const float d = *(const float *) (smem + SH3_LDS_DD + (2 * wn + rt) * 256 + fr * 8);
#pragma unroll
            for (int ta = 0; ta < 2; ++ta) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const float4 t = *(const float4 *) (smem + SH3_LDS_DY + (64 * wm + 32 * ta + 8 * g4 + 4 * kg) * 4);
                    const float dy[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * g4 + e;
                        const int is = (acc[1][rt][ta][r] << 7) + acc[0][rt][ta][r];
                        C[rt][ta][r] = __builtin_fmaf(dy[e], d * (float) is, C[rt][ta][r]);
                    }
                }
            }
        }
    }
    sh_wait<0>();
    const int mN = MAT_SEL(N);
    float * const m_dst = a.ksplit > 1 ? MAT_SEL(part) + (size_t) blockIdx.y * a.M * mN : MAT_SEL(dst);
    const int64_t m_dst_stride = a.ksplit > 1 ? (int64_t) mN : MAT_SEL(dst_stride);
    const float * const m_add = a.ksplit > 1 ? nullptr : MAT_SEL(add);
    const int64_t m_add_stride = MAT_SEL(add_stride);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int n = gl * SH_BN + 64 * wn + 32 * rt + fr;
#pragma unroll
        for (int ta = 0; ta < 2; ++ta) {
            const int mb = m0 + 64 * wm + 32 * ta + 4 * kg;
            if (m_add) {
#pragma unroll
                for (int r = 0; r < 16; ++r) C[rt][ta][r] += m_add[(size_t) min(mb + (r & 3) + 8 * (r >> 2), a.M - 1) * m_add_stride + n];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb + (r & 3) + 8 * (r >> 2);
                if (m < a.M) m_dst[(size_t) m * m_dst_stride + n] = C[rt][ta][r];
            }
        }
    }
#undef MAT_SEL
}

}  // namespace mi355x
