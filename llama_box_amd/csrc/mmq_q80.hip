// mmq_q80.hip — batched mat-mul for Q8_0 weights on the gfx950 INTEGER matrix cores.
//
// ggml-cpu's ggml_vec_dot_q8_0_q8_0 quantises the activations to Q8_0 as well and sums,
// per 32-value block, sumi * (d_w * d_x) in f32.  A block is exactly the K extent of v_mfma_i32_32x32x32_i8, so one MFMA
// with a zero accumulator yields the 32x32 integer block sums, and the f32 scale-accumulate follows at once — there is no
// super-block to amortise it over, which makes this kernel VALU-bound (4 VALU per output and block against one MFMA per
// 1024 outputs): the matrix core only replaces the 32 multiply-adds per output and block.  Still an order of magnitude
// faster than running the batch as 8-column mat-vec passes, which re-stream the weights once per pass.
// Tile 128 weight rows x 128 columns x 128 K (4 blocks) per trip, 8 waves (32 x 64 each), double-buffered LDS: raw int8 of
// both operands in the XOR-swizzled [row][128 B] image of mmq_i8.hip plus the 4 block scales of every row / column.
#include <algorithm>
#include <cstdlib>

#include "dev_util.h"
#include "kernels.h"

namespace mi355x {

typedef float float16q __attribute__((ext_vector_type(16)));
typedef int int4q __attribute__((ext_vector_type(4)));
typedef int int16q __attribute__((ext_vector_type(16)));

struct mmq80_args {
    const uint8_t * W;
    int64_t w_nb1;
    int K, N, M;
    const q80_dev * act;  // [M][K/32]
    float * dst;
    int64_t dst_stride;
    int n_panels, m_tiles;
    const float * add;
    int64_t add_stride;
};

constexpr int Q80_T = 128 * 128;                // bytes of one operand tile
constexpr int Q80_STAGE = 2 * Q80_T + 2 * 128 * 16;  // A | B | dw[128][4] | da[128][4]

__device__ __forceinline__ int sw_offq(const int row, const int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// PANEL (round 6): the weights come from their panel copy (repack.hip: k_repack_q80_panels) and the activations in panel order (quantize.hip: k_quantize_q8_0<true>) —
// a thread's block is two aligned 16-byte loads + its scale instead of eight 2-byte-aligned dwords + the scale, consecutive threads read consecutive bytes
template <bool PANEL> __global__ void __launch_bounds__(512, 1) k_mmq_q80(const mmq80_args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int xcd = blockIdx.x & 7, qb = blockIdx.x >> 3;
    const int panel = (qb / a.m_tiles) * 8 + xcd, mt = qb % a.m_tiles;
    if (panel >= a.n_panels) return;
    const int n0 = panel * 128, m0 = mt * 128;
    const int nb32 = a.K / 32, trips = a.K / 128;
    const int nslab = wave & 3, mhalf = wave >> 2;

    // staging roles: thread -> (row | column, block of the trip)
    const int srow = tid >> 2, sq = tid & 3;
    const uint8_t * wblk = a.W + (size_t) min(n0 + srow, a.N - 1) * a.w_nb1 + (size_t) sq * 34;
    const q80_dev * yblk = a.act + (size_t) min(m0 + srow, a.M - 1) * nb32 + sq;
    const int off0 = sw_offq(srow, 2 * sq), off1 = sw_offq(srow, 2 * sq + 1);

    uint32_t ga0, ga1, ga2, ga3, ga4, ga5, ga6, ga7, gb0, gb1, gb2, gb3, gb4, gb5, gb6, gb7;
    uint16_t gad = 0;
    float gbd = 0.0f;
    // PANEL: the (32 rows, 4 blocks) tile of this thread's row for trip t, and the (block, 32 columns) tile of its column
    const int wrow_c = min(n0 + srow, a.N - 1), ycol_c = min(m0 + srow, a.M - 1);
    const uint8_t * wtile = a.W + (size_t) (wrow_c >> 5) * (size_t) trips * 4352 + (size_t) sq * 1024 + (size_t) (wrow_c & 31) * 16;
    const uint8_t * wtile_d = a.W + (size_t) (wrow_c >> 5) * (size_t) trips * 4352 + 4096 + (size_t) sq * 64 + (size_t) (wrow_c & 31) * 2;
    const char * ytile = (const char *) a.act + ((size_t) (ycol_c >> 5) * (size_t) nb32 + (size_t) sq) * 1152 + (size_t) (ycol_c & 31) * 16;
    auto issue_loads = [&](const int t) {
        if constexpr (PANEL) {
            const uint4 a0 = *(const uint4 *) (wtile + (size_t) t * 4352), a1 = *(const uint4 *) (wtile + (size_t) t * 4352 + 512);
            ga0 = a0.x; ga1 = a0.y; ga2 = a0.z; ga3 = a0.w; ga4 = a1.x; ga5 = a1.y; ga6 = a1.z; ga7 = a1.w;
            gad = *(const uint16_t *) (wtile_d + (size_t) t * 4352);
            const char * yt = ytile + (size_t) t * 4 * 1152;
            const uint4 b0 = *(const uint4 *) yt, b1 = *(const uint4 *) (yt + 512);
            gb0 = b0.x; gb1 = b0.y; gb2 = b0.z; gb3 = b0.w; gb4 = b1.x; gb5 = b1.y; gb6 = b1.z; gb7 = b1.w;
            gbd = *(const float *) (yt + 1024 - (size_t) (ycol_c & 31) * 16 + (size_t) (ycol_c & 31) * 4);
            return;
        }
        const uint8_t * p = wblk + (size_t) t * 4 * 34;  // Q8_0 blocks are 2-byte aligned: dword loads typed accordingly
        gad = ld16(p);
        ga0 = ld32_a2(p + 2); ga1 = ld32_a2(p + 6); ga2 = ld32_a2(p + 10); ga3 = ld32_a2(p + 14);
        ga4 = ld32_a2(p + 18); ga5 = ld32_a2(p + 22); ga6 = ld32_a2(p + 26); ga7 = ld32_a2(p + 30);
        const q80_dev * y = yblk + (size_t) t * 4;
        const uint32_t * yq = (const uint32_t *) y->qs;
        gb0 = yq[0]; gb1 = yq[1]; gb2 = yq[2]; gb3 = yq[3]; gb4 = yq[4]; gb5 = yq[5]; gb6 = yq[6]; gb7 = yq[7];
        gbd = y->d;
    };
    auto stage = [&](const int t) {
        char * buf = smem + (t & 1) * Q80_STAGE;
        *(uint4 *) (buf + off0) = make_uint4(ga0, ga1, ga2, ga3);
        *(uint4 *) (buf + off1) = make_uint4(ga4, ga5, ga6, ga7);
        *(uint4 *) (buf + Q80_T + off0) = make_uint4(gb0, gb1, gb2, gb3);
        *(uint4 *) (buf + Q80_T + off1) = make_uint4(gb4, gb5, gb6, gb7);
        ((float *) (buf + 2 * Q80_T))[sq * 128 + srow] = h2f(gad);              // scales are stored [block][row] so that the four
        ((float *) (buf + 2 * Q80_T + 128 * 16))[sq * 128 + srow] = gbd;        // consecutive rows of an accumulator quad are one 16-byte read
    };

    const int fr = lane & 31, kg = lane >> 5;
    const int swz = (fr >> 1) & 7;
    const int arow_off = (nslab * 32 + fr) * 128;
    const int brow_off0 = (mhalf * 64 + fr) * 128, brow_off1 = (mhalf * 64 + 32 + fr) * 128;
    float16q C0, C1;
    int16q zi;
#pragma unroll
    for (int r = 0; r < 16; ++r) { C0[r] = 0.0f; C1[r] = 0.0f; zi[r] = 0; }

    issue_loads(0);
    stage(0);
    if (trips > 1) issue_loads(1);
    __syncthreads();
    for (int t = 0; t < trips; ++t) {
        if (t + 1 < trips) {
            stage(t + 1);
            if (t + 2 < trips) issue_loads(t + 2);
        }
        const char * buf = smem + (t & 1) * Q80_STAGE;
        const float * dwt = (const float *) (buf + 2 * Q80_T);
        const float * dat = (const float *) (buf + 2 * Q80_T + 128 * 16);
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            const int co = ((2 * blk + kg) ^ swz) << 4;
            const int4q fa = *(const int4q *) (buf + arow_off + co);
            const int4q fb0 = *(const int4q *) (buf + Q80_T + brow_off0 + co), fb1 = *(const int4q *) (buf + Q80_T + brow_off1 + co);
            const float y0 = dat[blk * 128 + mhalf * 64 + fr], y1 = dat[blk * 128 + mhalf * 64 + 32 + fr];
            float x[16];  // block scale of the 16 rows this lane's accumulator registers belong to
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 v = *(const float4 *) (dwt + blk * 128 + nslab * 32 + 8 * g4 + 4 * kg);
                x[4 * g4] = v.x; x[4 * g4 + 1] = v.y; x[4 * g4 + 2] = v.z; x[4 * g4 + 3] = v.w;
            }
            const int16q s0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb0, zi, 0, 0, 0);
            const int16q s1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb1, zi, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                C0[r] += (float) s0[r] * (x[r] * y0);  // sumf += sumi * (d_w * d_x), as the CPU does
                C1[r] += (float) s1[r] * (x[r] * y1);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int m = m0 + mhalf * 64 + tt * 32 + fr;
        if (m >= a.M) continue;
        float * out = a.dst + (size_t) m * a.dst_stride;
        const float * ad = a.add ? a.add + (size_t) m * a.add_stride : nullptr;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = n0 + nslab * 32 + 8 * g + 4 * kg;
            float c4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) c4[r] = tt ? C1[4 * g + r] : C0[4 * g + r];
            if (ad) {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n + r < a.N) c4[r] += ad[n + r];
            }
            if (n + 3 < a.N && ((((uintptr_t) (out + n)) & 15) == 0)) {
                *(float4 *) (out + n) = make_float4(c4[0], c4[1], c4[2], c4[3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n + r < a.N) out[n + r] = c4[r];
            }
        }
    }
}

bool mmq_q80_supported(int type, int64_t K, int64_t N, int64_t M) {
    (void) N;
    return type == GGML_TYPE_Q8_0 && (K % 128) == 0 && M >= 9;
}

void launch_mmq_q80(hipStream_t s, const uint8_t * W, const uint8_t * W_panels, int64_t w_nb1, int K, int N, int M, const void * act_q80, float * dst, int64_t dst_stride, const float * add,
                    int64_t add_stride) {  // W_panels != nullptr: the panel copy, and act_q80 holds the activations in panel order
    mmq80_args a;
    a.W = W_panels ? W_panels : W;
    a.w_nb1 = w_nb1;
    a.K = K;
    a.N = N;
    a.M = M;
    a.act = (const q80_dev *) act_q80;
    a.dst = dst;
    a.dst_stride = dst_stride;
    a.n_panels = (N + 127) / 128;
    a.m_tiles = (M + 127) / 128;
    a.add = add;
    a.add_stride = add_stride;
    const size_t lds = 2 * (size_t) Q80_STAGE;
    static std::atomic<uint32_t> lds_raised{0};  // one bit per device (common.h: ensure_dyn_lds)
    static std::atomic<uint32_t> lds_raised_p{0};
    const unsigned grid = (unsigned) (((a.n_panels + 7) / 8) * 8 * a.m_tiles);
    if (W_panels) {
        (void) ensure_dyn_lds((const void *) k_mmq_q80<true>, lds, lds_raised_p);  // on failure the launch below fails and graph_compute reports it
        hipLaunchKernelGGL(k_mmq_q80<true>, dim3(grid), dim3(512), lds, s, a);
    } else {
        (void) ensure_dyn_lds((const void *) k_mmq_q80<false>, lds, lds_raised);
        hipLaunchKernelGGL(k_mmq_q80<false>, dim3(grid), dim3(512), lds, s, a);
    }
}

// ------------------------------------------------------------------------------------------------ 9 .. 32 columns (round 6)
// A -np decode step of a Q8_0 model: until this round every such mat-mul ran as 8-column mat-vec passes (the weights streamed once per pass, the dot products
// on the vector ALU: 61 us per matrix of Llama-3-8B Q8_0 at 32 columns, a 16 ms step) and the 128 x 128-tile GEMM above has too few workgroups for 32 columns
// (97 us).  Here the WEIGHTS stream once: a workgroup owns 32 weight rows, its eight waves take the K range in interleaved chunks of four blocks (128 values:
// one cache line of every row), one v_mfma_i32_32x32x32_i8 per block gives the 32 x 32 integer block sums, and the f32 scale-accumulate follows as in the GEMM
// (sumf += sumi * (d_w * d_x), ggml_vec_dot_q8_0_q8_0's expression).  Weight bytes go from global memory straight into the MFMA operand registers (a lane = a
// row and a 16-value half of the block, 2-byte aligned 16-byte loads); the activations' blocks (32 columns x K / 32 x 36 B: L2-resident) likewise; the 32 row
// scales of a block reach the 16 accumulator rows of a lane through a wave-private LDS line.  The eight partial tiles meet in LDS and are added in wave order.
struct mmq80s_mat {
    const uint8_t * W;   // the tensor (block layout), or its panel copy (repack.hip: k_repack_q80_panels) — template WP
    int64_t w_nb1;
    int N;
    int panel_end;       // workgroups [previous matrix's panel_end, panel_end) are this matrix's 32-row panels
    float * dst;
    int64_t dst_stride;
    const float * add;
    int64_t add_stride;
};
struct mmq80s_args {    // up to three matrices over the same activations in one launch (wq / wk / wv, gate / up: graph.cpp try_merge_q80_skinny)
    mmq80s_mat m[3];
    int K, M;
    const char * act;    // the activations in panel order (quantize.hip: k_quantize_q8_0<true>): 1152 B per block
};
typedef uint32_t __attribute__((ext_vector_type(4), aligned(2))) q80s_u32x4_a2;
typedef uint32_t __attribute__((ext_vector_type(4), aligned(4))) q80s_u32x4_a4;

constexpr int Q80S_WV = 8;
template <bool WP> __global__ void __launch_bounds__(Q80S_WV * 64, 2) k_mmq_q80_skinny(const mmq80s_args a) {
    constexpr bool PF = true;
    __shared__ float dwl[Q80S_WV][4][32];       // a chunk's row scales, per wave
    __shared__ float red[Q80S_WV][16][64];      // the waves' partial tiles
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, kg = lane >> 5;
    const int mi = (int) blockIdx.x < a.m[0].panel_end ? 0 : ((int) blockIdx.x < a.m[1].panel_end ? 1 : 2);  // (uniform)
    const mmq80s_mat & mt = a.m[mi];
    const int panel = (int) blockIdx.x - (mi ? a.m[mi - 1].panel_end : 0);
    const int n0 = panel * 32;
    const int nblk = a.K / 32, nchunk = nblk / 4;
    const uint8_t * wrow = WP ? mt.W + (size_t) panel * nchunk * 4352 : mt.W + (size_t) min(n0 + fr, mt.N - 1) * mt.w_nb1;
    const int colc = min(fr, a.M - 1);
    float C[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) C[r] = 0.0f;
    int16q zi;
#pragma unroll
    for (int r = 0; r < 16; ++r) zi[r] = 0;

    int4q fa[4], fb[4], nfa[4], nfb[4];
    uint16_t dw[4], ndw[4];
    float dy[4], ndy[4];
    auto load_chunk = [&](const int c, int4q * A, int4q * B, uint16_t * dW, float * dY) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if constexpr (WP) {  // the chunk's tile of the panel copy: a wave-instruction reads 1 KB of consecutive bytes
                const uint8_t * tile = wrow + (size_t) c * 4352;
                A[b] = *(const int4q *) (tile + b * 1024 + kg * 512 + fr * 16);
                dW[b] = *(const uint16_t *) (tile + 4096 + b * 64 + fr * 2);
            } else {
                const uint8_t * blk = wrow + (size_t) (c * 4 + b) * 34;
                dW[b] = ld16(blk);
                A[b] = __builtin_bit_cast(int4q, *(const q80s_u32x4_a2 *) (blk + 2 + 16 * kg));
            }
            const char * yt = a.act + (size_t) (c * 4 + b) * 1152;
            B[b] = *(const int4q *) (yt + kg * 512 + colc * 16);
            dY[b] = *(const float *) (yt + 1024 + colc * 4);
        }
    };
    int c = wave;
    if (c < nchunk) load_chunk(c, fa, fb, dw, dy);
    for (; c < nchunk; c += Q80S_WV) {
        const int cn = c + Q80S_WV;
        if constexpr (PF) { if (cn < nchunk) load_chunk(cn, nfa, nfb, ndw, ndy); }
        // the block scales of the 32 rows: written by the row's first lane, read back as the 16 rows of this lane's accumulator registers
        if (kg == 0) {
#pragma unroll
            for (int b = 0; b < 4; ++b) dwl[wave][b][fr] = h2f(dw[b]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            float x[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 v = *(const float4 *) &dwl[wave][b][8 * g4 + 4 * kg];
                x[4 * g4] = v.x; x[4 * g4 + 1] = v.y; x[4 * g4 + 2] = v.z; x[4 * g4 + 3] = v.w;
            }
            const int16q s = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[b], fb[b], zi, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) C[r] += (float) s[r] * (x[r] * dy[b]);  // sumf += sumi * (d_w * d_x), as the CPU does
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // (the next chunk's scales overwrite the line)
        if (cn < nchunk) {
            if constexpr (PF) {
#pragma unroll
                for (int b = 0; b < 4; ++b) { fa[b] = nfa[b]; fb[b] = nfb[b]; dw[b] = ndw[b]; dy[b] = ndy[b]; }
            } else {
                load_chunk(cn, fa, fb, dw, dy);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = C[r];
    __syncthreads();
    for (int e = tid; e < 16 * 64; e += Q80S_WV * 64) {
        const int r = e >> 6, l = e & 63;
        float v = red[0][r][l];
#pragma unroll
        for (int w = 1; w < Q80S_WV; ++w) v += red[w][r][l];
        const int m = l & 31, n = n0 + 8 * (r >> 2) + 4 * (l >> 5) + (r & 3);
        if (m < a.M && n < mt.N) {
            if (mt.add) v += mt.add[(size_t) m * mt.add_stride + n];
            mt.dst[(size_t) m * mt.dst_stride + n] = v;
        }
    }
}

bool mmq_q80_skinny_supported(int type, int64_t K, int64_t N, int64_t M) {
    static const bool on = !getenv("GGML_MI355X_Q80_SKINNY") || atoi(getenv("GGML_MI355X_Q80_SKINNY")) != 0;
    static const int max_cols = getenv("GGML_MI355X_Q80_SKINNY_MAX") ? atoi(getenv("GGML_MI355X_Q80_SKINNY_MAX")) : 128;  // (up to 128 columns: four passes of 32 still beat the 128 x 128-tile GEMM's few workgroups — Llama-3-8B Q8_0, 128-token micro-batches: 8.5 k against 5.1 k tok/s)
    return on && type == GGML_TYPE_Q8_0 && (K % 128) == 0 && N >= 1 && M >= 9 && M <= std::min(max_cols, 128);
}
// n matrices (1 .. 3) over the same activations; panels[q] != nullptr for ALL of them or for none (the template form is per launch)
void launch_mmq_q80_skinny_multi(hipStream_t s, int n, const mmq80s_desc * mats, int K, int M, const void * act_q80) {
    mmq80s_args a;
    int pe = 0;
    const bool wp = mats[0].W_panels != nullptr;
    for (int q = 0; q < 3; ++q) {
        const mmq80s_desc & d = mats[q < n ? q : n - 1];
        if (q < n) pe += (d.N + 31) / 32;
        a.m[q] = {wp ? d.W_panels : d.W, d.w_nb1, d.N, pe, d.dst, d.dst_stride, d.add, d.add_stride};
    }
    a.K = K;
    for (int m0 = 0; m0 < M; m0 += 32) {  // groups of 32 columns: the activations' panel tiles of group g follow those of group g - 1 (quantize.hip)
        a.M = std::min(32, M - m0);
        a.act = (const char *) act_q80 + (size_t) (m0 / 32) * (size_t) (K / 32) * 1152;
        for (int q = 0; q < 3; ++q) {
            const mmq80s_desc & d = mats[q < n ? q : n - 1];
            a.m[q].dst = d.dst + (size_t) m0 * d.dst_stride;
            a.m[q].add = d.add ? d.add + (size_t) m0 * d.add_stride : nullptr;
        }
        if (wp) hipLaunchKernelGGL(k_mmq_q80_skinny<true>, dim3((unsigned) pe), dim3(Q80S_WV * 64), 0, s, a);
        else hipLaunchKernelGGL(k_mmq_q80_skinny<false>, dim3((unsigned) pe), dim3(Q80S_WV * 64), 0, s, a);
    }
}
void launch_mmq_q80_skinny(hipStream_t s, const uint8_t * W, const uint8_t * W_panels, int64_t w_nb1, int K, int N, int M, const void * act_q80, float * dst, int64_t dst_stride, const float * add,
                           int64_t add_stride) {
    const mmq80s_desc d{W, W_panels, w_nb1, N, dst, dst_stride, add, add_stride};
    launch_mmq_q80_skinny_multi(s, 1, &d, K, M, act_q80);
}

MI_TU_TOUCH(mmq_q80)

}  // namespace mi355x
