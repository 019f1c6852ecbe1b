// mmvq_dma.hip — EXPERIMENT (round 2, not part of the product): batch-1 mat-vec whose weights travel HBM -> LDS by LDS-DMA
// (global_load_lds_dwordx4).  Built and timed by scripts/ubench/decode_lab.hip against the product's register-streaming kernel.
// RESULT (MI355X, dependent chains inside a hipGraph, us per launch, register-streaming vs LDS-DMA; bit-identical results):
//   wo Q4_K 9 MB 6.0 / 5.9 · gate+up Q4_K 63 MB 16.4 / 16.8 · ffn_down Q4_K 31.5 MB 11.7 / 11.4 · ffn_down Q6_K 45.9 MB 15.8 / 23.3 ·
//   lm_head Q6_K 411 MB 94.8 / 141.7 · 70B gate+up 252 MB 49.3 / 48.4 · Qwen2 gate+up Q5_K 89 MB 22.2 / 30.6
// i.e. perfectly coalesced, format-agnostic transport with deeper prefetch buys NOTHING for the 16-byte-aligned formats — the
// address pattern of the loads is not what bounds these launches — and unpacking 2-byte-aligned Q6_K / Q5_K fields from LDS is
// slower than fetching them from global memory.  What bounds them is the fixed cost of a dependent launch (stamp_lab.hip,
// prologue_probe.hip).  Kept for the record and for re-measurement; the hypothesis it tested follows.
//
// What round 1's k_mmvq_stream (mmvq.hip) could not fix inside its structure (profiles/r01_*: 4.2 TB/s on the 66 MB gate/up
// launch, waves parked 2.4x as long as they issue):
//   * every lane fetched "its" (super-block, chunk) pair straight into VGPRs, so the address pattern of a load instruction was
//     dictated by the quant format: Q4_K 3 x 16 B at 32-byte strides + a header fetched by 4 lanes at once (each 128-byte line
//     touched by three instructions), Q6_K / Q8_0 8-byte pieces at 2-byte alignment (five / nine instructions per 34 bytes);
//   * bytes in flight were VGPRs (48 .. 96 B per lane), and more of them cost occupancy.
// Here the transport is format-agnostic: a wave moves a row segment as consecutive 1 KiB pieces (64 lanes x 16 B, fully
// coalesced, non-temporal) with LDS-DMA — no VGPRs, no L1 pollution by a second touch — into a wave-private ring of tiles, and
// the format-specific unpacking reads the tile back from LDS (ds_read_b128 / b64; 2-byte aligned fields are fine there).
// Waves never synchronise with each other in the main loop: a wave waits on ITS OWN vmcnt for ITS tile (counted waits: the
// DMA instructions are inline asm, so hipcc does not drain them with vmcnt(0) — cdna_hip_programming.md §5.7), consumes it,
// and re-issues into the slot it just freed.  Row results wait in LDS until the end of the launch so that no store or
// residual load sits between the DMAs (either would make a counted wait conservative and stall the ring).
//
// Arithmetic is mmvq.hip's, bit for bit: same T::dot on the same bytes, same activation prologue (PRO 1: CPU-identical
// Q8_K / Q8_0 quantisation of the f32 row; PRO 2: RMS_NORM * w first, sum of squares in double), same wave reduction and
// epilogue order (bias / residual adds, SwiGLU).  Replaces ggml_vec_dot_q*_K_q8_K / q8_0_q8_0 + mul_mat_vec_q.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>

#include "../../../llama_box_amd/csrc/mmvq_types.h"

namespace mi355x {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// one 1 KiB piece: lane i's 16 bytes at g land at LDS[lds + 16 * i]; `lds` must be wave-uniform
__device__ __forceinline__ void dma16(const void * g, const uint32_t lds) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g), "s"(lds)
                 : "memory");
}
// an f32x4 load hipcc does not count (so that it cannot answer it with vmcnt(0) while DMAs issued later are in flight)
__device__ __forceinline__ f32x4 ld_uncounted(const void * g) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(g) : "memory");
    return v;
}
// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate only)
__device__ __forceinline__ void wait_vm(const int n) {
#define MI_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) {
        MI_W(0) MI_W(1) MI_W(2) MI_W(3) MI_W(4) MI_W(5) MI_W(6) MI_W(7) MI_W(8) MI_W(9) MI_W(10) MI_W(11) MI_W(12) MI_W(13) MI_W(14) MI_W(15)
        MI_W(16) MI_W(17) MI_W(18) MI_W(19) MI_W(20) MI_W(21) MI_W(22) MI_W(23) MI_W(24) MI_W(25) MI_W(26) MI_W(27) MI_W(28) MI_W(29) MI_W(30) MI_W(31)
        MI_W(32) MI_W(33) MI_W(34) MI_W(35) MI_W(36) MI_W(37) MI_W(38) MI_W(39) MI_W(40) MI_W(41) MI_W(42) MI_W(43) MI_W(44) MI_W(45) MI_W(46) MI_W(47)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;  // (more than 47 outstanding never happens; 0 is always safe)
    }
#undef MI_W
}

struct mvd_geom {
    int sb;          // super-blocks (Q8_0: blocks) of a row per tile
    int nseg;        // tiles per row
    int seg_bytes;   // sb * T::BYTES, a multiple of 16
    int chunks;      // 1 KiB pieces per matrix and tile
    int depth;       // tiles in the ring of a wave
    int max_rows;    // row results a wave keeps in LDS
    int act_bytes;   // quantised activations
};

template <typename T, bool GLU, int PRO>
__global__ void __launch_bounds__(1024) k_mmvq_dma(const mmvq_args a, const mvd_geom ge) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename T::act act;
    constexpr int WAVES = 16;
    constexpr int XU = PRO == 2 ? 4 : 8;    // 256-value chunks of the activation row a wave quantises (K <= 16384 / 32768)
    constexpr int BPC = 256 / T::BLK;       // activation blocks per chunk
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nblk = a.K / T::BLK, nchk = a.K / 256;
    // ---- LDS: activations | 16 doubles | row results [16][max_rows] | rings [16][depth][tile]
    act * yl = (act *) smem;
    double * red = (double *) (smem + ge.act_bytes);
    float * res = (float *) (smem + ge.act_bytes + 128) + wave * ge.max_rows;
    const int tile_stride = ge.seg_bytes * (GLU ? 2 : 1);
    char * ring = smem + ge.act_bytes + 128 + WAVES * ge.max_rows * 4 + (size_t) wave * ge.depth * tile_stride;
    const uint32_t ring_lds = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) char *) ring;

    // ---- rows of this wave: one per pass (row = pass * GW + block * 16 + wave), the last partial pass spread evenly (mmvq.hip)
    const int GW = gridDim.x * WAVES;
    const int full = (a.N / GW) * GW, rem = a.N - full;
    const int rem_per = (rem + (int) gridDim.x - 1) / (int) gridDim.x;
    const int tail_row = (wave < rem_per && full + (int) blockIdx.x * rem_per + wave < a.N) ? full + (int) blockIdx.x * rem_per + wave : a.N;
    const int row_first = (int) blockIdx.x * WAVES + wave;
    const int n_full = full / GW;  // rows of the full passes (row_first + k * GW), then tail_row if < N
    const int n_rows = n_full + (tail_row < a.N ? 1 : 0);
    const int n_tiles = n_rows * ge.nseg;
    auto row_of = [&](const int i) { return i < n_full ? row_first + i * GW : tail_row; };

    auto issue = [&](const int t) {  // DMA tile t (row t / nseg, segment t % nseg) into slot t % depth
        const int r = row_of(t / ge.nseg), sg = t % ge.nseg;
        const size_t goff = (size_t) r * a.w_nb1 + (size_t) sg * ge.seg_bytes + (size_t) lane * 16;
        const uint32_t dst = ring_lds + (uint32_t) ((t % ge.depth) * tile_stride);
        for (int c = 0; c < ge.chunks; ++c) {
            if (c * 1024 + lane * 16 < ge.seg_bytes) dma16(a.W + goff + c * 1024, __builtin_amdgcn_readfirstlane(dst + c * 1024));
        }
        if (GLU) {
            for (int c = 0; c < ge.chunks; ++c) {
                if (c * 1024 + lane * 16 < ge.seg_bytes) dma16(a.W2 + goff + c * 1024, __builtin_amdgcn_readfirstlane(dst + ge.seg_bytes + c * 1024));
            }
        }
    };
    const int per_tile = ge.chunks * (GLU ? 2 : 1);  // DMA instructions per tile

    // ---- the activation row first (it is the older request: it returns first), then the ring is primed
    f32x4 xv[XU], gv[XU];
#pragma unroll
    for (int u = 0; u < XU; ++u) {
        const int b = wave + u * WAVES;
        if (b < nchk) {
            xv[u] = ld_uncounted(a.x + (size_t) (b * 64 + lane) * 4);
            if (PRO == 2) gv[u] = ld_uncounted(a.norm_w + (size_t) (b * 64 + lane) * 4);
        } else {
            xv[u] = (f32x4) (0.0f);
            gv[u] = (f32x4) (0.0f);
        }
    }
    int issued = 0;
    for (; issued < ge.depth && issued < n_tiles; ++issued) issue(issued);
    wait_vm(issued * per_tile);  // everything older than the DMAs has landed: x (and the norm weights)
#pragma unroll
    for (int u = 0; u < XU; ++u) {
        asm volatile("" : "+v"(xv[u]));
        if (PRO == 2) asm volatile("" : "+v"(gv[u]));
    }
    float scale = 1.0f;
    if constexpr (PRO == 2) {
        double ss = 0.0;
#pragma unroll
        for (int u = 0; u < XU; ++u) ss += (double) (xv[u].x * xv[u].x) + (double) (xv[u].y * xv[u].y) + (double) (xv[u].z * xv[u].z) + (double) (xv[u].w * xv[u].w);
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
    for (int u = 0; u < XU; ++u) {
        const int b = wave + u * WAVES;
        if (b < nchk) {
            float t[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
            if constexpr (PRO == 2) {
                t[0] = (t[0] * scale) * gv[u].x;
                t[1] = (t[1] * scale) * gv[u].y;
                t[2] = (t[2] * scale) * gv[u].z;
                t[3] = (t[3] * scale) * gv[u].w;
                // (block 0 leaves the graph's RMS_NORM * w tensor behind — graph.cpp, deferred_norm::out; this counted store only
                // makes that one workgroup's waits conservative)
                if (a.norm_out && blockIdx.x == 0) ((float4 *) a.norm_out)[b * 64 + lane] = make_float4(t[0], t[1], t[2], t[3]);
            }
            if constexpr (BPC == 1) wave_quantize_q8_K(t, lane, yl + b);
            else wave_quantize_q8_0(t, lane, yl + (size_t) b * BPC);
        }
    }
    __syncthreads();
    const act * y = (const act *) smem;

    // ---- main loop: tile t is ready when at most (issued - t - 1) tiles' worth of DMAs remain outstanding
    const int npl = ge.sb * T::PPB;  // lane pairs of a tile
    float acc = 0.0f, acc2 = 0.0f;
    int ri = 0;
    for (int t = 0; t < n_tiles; ++t) {
        wait_vm((issued - t - 1) * per_tile);
        const uint8_t * tl = (const uint8_t *) (ring + (t % ge.depth) * tile_stride);
        const int sg = t % ge.nseg;
        for (int p = lane; p < npl; p += 64) {
            const typename T::raw w = T::load(tl, p);
            T::template dot<1>(w, sg * npl + p, y, nblk, &acc);
            if (GLU) {
                const typename T::raw w2 = T::load(tl + ge.seg_bytes, p);
                T::template dot<1>(w2, sg * npl + p, y, nblk, &acc2);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the tile has been read: its slot may be overwritten
        if (issued < n_tiles) {
            issue(issued);
            ++issued;
        }
        if (sg == ge.nseg - 1) {
            float v = wave_sum(acc);
            if (GLU) {
                const float g = wave_sum(acc2);
                v = silu_f(v) * g;
            }
            if (lane == 0) res[ri] = v;
            ++ri;
            acc = 0.0f;
            acc2 = 0.0f;
        }
    }
    // ---- epilogue: nothing is in flight any more; lanes take the wave's rows
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    for (int i = lane; i < n_rows; i += 64) {
        const int row = row_of(i);
        float v = res[i];
        if (a.add) v += a.add[row];
        if (a.add2) v += a.add2[row];
        a.dst[row] = v;
    }
}

// ------------------------------------------------------------------------------------------------ host side
template <typename T> static bool plan_geom(const mmvq_args & a, bool glu, int pro, mvd_geom & ge, size_t & lds) {
    const int nblk = a.K / T::BLK;
    if (a.K % 256 != 0 || (a.w_nb1 % 16) != 0 || (((uintptr_t) a.W) & 15) || (glu && (((uintptr_t) a.W2) & 15))) return false;
    if (a.w_nb1 != (int64_t) nblk * T::BYTES) return false;  // rows must be dense (a tile never spans rows, but offsets assume it)
    if (a.K > (pro == 2 ? 16384 : 32768)) return false;
    if ((((uintptr_t) a.x) & 15) || (pro == 2 && (((uintptr_t) a.norm_w) & 15))) return false;
    const int max_seg = glu ? 2304 : 4608;
    int best = 0;
    bool best_full = false;
    for (int d = 1; d <= nblk; ++d) {
        if (nblk % d) continue;
        const int bytes = d * T::BYTES;
        if (bytes > max_seg || (bytes % 16) != 0) continue;
        const bool full = ((d * T::PPB) % 64) == 0;
        if ((full && !best_full) || (full == best_full && d > best)) {
            best = d;
            best_full = full;
        }
    }
    if (!best) return false;
    const unsigned grid = (unsigned) std::min<int64_t>(256, ((int64_t) a.N + 15) / 16);
    ge.sb = best;
    ge.nseg = nblk / best;
    ge.seg_bytes = best * T::BYTES;
    ge.chunks = (ge.seg_bytes + 1023) / 1024;
    ge.max_rows = (int) ((a.N + (int64_t) grid * 16 - 1) / ((int64_t) grid * 16)) + 1;
    ge.max_rows = (ge.max_rows + 3) & ~3;
    ge.act_bytes = (int) (((size_t) nblk * sizeof(typename T::act) + 15) & ~(size_t) 15);
    const size_t fixed = (size_t) ge.act_bytes + 128 + (size_t) 16 * ge.max_rows * 4;
    const size_t tile = (size_t) ge.seg_bytes * (glu ? 2 : 1);
    const size_t budget = 160 * 1024;
    if (fixed + 16 * 2 * tile > budget) return false;
    static const int max_depth = getenv("GGML_MI355X_DMA_DEPTH") ? atoi(getenv("GGML_MI355X_DMA_DEPTH")) : 4;
    ge.depth = (int) std::min<size_t>((size_t) std::max(2, max_depth), (budget - fixed) / (16 * tile));
    if (ge.depth * ge.chunks * (glu ? 2 : 1) > 47) ge.depth = std::max(2, 47 / (ge.chunks * (glu ? 2 : 1)));
    lds = fixed + (size_t) 16 * ge.depth * tile;
    return true;
}

template <typename T, bool GLU, int PRO> static void launch_dma_t(hipStream_t s, const mmvq_args & a, const mvd_geom & ge, size_t lds) {
    static std::atomic<uint32_t> lds_raised{0};
    (void) ensure_dyn_lds((const void *) k_mmvq_dma<T, GLU, PRO>, 160 * 1024, lds_raised);
    const unsigned grid = (unsigned) std::min<int64_t>(256, ((int64_t) a.N + 15) / 16);
    if (g_launch_probe.armed && !g_launch_probe.used) {
        hipExtLaunchKernelGGL((k_mmvq_dma<T, GLU, PRO>), dim3(grid), dim3(1024), lds, s, g_launch_probe.e0, g_launch_probe.e1, 0, a, ge);
        g_launch_probe.used = true;
    } else {
        hipLaunchKernelGGL((k_mmvq_dma<T, GLU, PRO>), dim3(grid), dim3(1024), lds, s, a, ge);
    }
}

template <typename T> static bool try_type(hipStream_t s, const mmvq_args & a) {
    const bool glu = a.W2 != nullptr;
    const int pro = a.norm_w ? 2 : 1;
    mvd_geom ge{};
    size_t lds = 0;
    if (!plan_geom<T>(a, glu, pro, ge, lds)) return false;
    if (pro == 2) { if (glu) launch_dma_t<T, true, 2>(s, a, ge, lds); else launch_dma_t<T, false, 2>(s, a, ge, lds); }
    else          { if (glu) launch_dma_t<T, true, 1>(s, a, ge, lds); else launch_dma_t<T, false, 1>(s, a, ge, lds); }
    return true;
}

// single-column mat-vec with the f32 / norm prologue (a.x != null).  false: shape not served here (caller falls back to mmvq.hip)
bool launch_mmvq_dma(hipStream_t s, const mmvq_args & a) {
    if (!a.x || a.ncols != 1) return false;
    switch (a.type) {
        case GGML_TYPE_Q4_K: return try_type<T_Q4K>(s, a);
        case GGML_TYPE_Q5_K: return try_type<T_Q5K>(s, a);
        case GGML_TYPE_Q6_K: return try_type<T_Q6K>(s, a);
        case GGML_TYPE_Q8_0: return try_type<T_Q80>(s, a);
        default: return false;
    }
}

}  // namespace mi355x
