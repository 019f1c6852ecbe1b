// repack.hip — writes THE DECODE COPY of a K-quant weight matrix: the same bytes, row by row, in the line-aligned plane layout the batch-1 mat-vec kernels stream
// with non-temporal loads (mmvq_types.h: T_Q4KP / T_Q5KP / T_Q6KP).  The original tensor stays what the host uploaded (get_tensor, the batch kernels, GET_ROWS and
// the tensor-split paths read it); the copy lives at the same offset of a shadow allocation of the weights buffer (backend.cpp).  One thread moves one 16-byte
// piece (the Q6_K tail: one super-block's 16 scale bytes + its d); sources are read at the formats' true alignment (2 bytes for Q6_K).
#include "common.h"
#include "dev_util.h"
#include "kernels.h"

namespace mi355x {

struct piece16 { uint16_t h[8]; };
__device__ __forceinline__ piece16 ld_piece_a2(const uint8_t * p) {
    piece16 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v.h[i] = *(const uint16_t *) (p + 2 * i);
    return v;
}
__device__ __forceinline__ void st_piece_a2(uint8_t * p, const piece16 & v) {
#pragma unroll
    for (int i = 0; i < 8; ++i) *(uint16_t *) (p + 2 * i) = v.h[i];
}

// grid.x covers the SOURCE pieces of one row (per super-block: Q4_K 9, Q5_K 11, Q6_K 12 + one tail item), grid.y the rows.  Group g holds super-blocks 8 g .. 8 g + nb - 1
// (nb = 8, or nblk % 8 for a short last group); lane l = 4 (b - 8 g) + j of the group owns the 16-byte pieces j of super-block b.
template <int QT> __global__ void __launch_bounds__(256) k_repack_planes(const uint8_t * __restrict__ src, uint8_t * __restrict__ dst, const int nblk, const int64_t nb1) {
    const uint8_t * srow = src + (size_t) blockIdx.y * nb1;
    uint8_t * drow = dst + (size_t) blockIdx.y * nb1;
    const int e = (int) (blockIdx.x * 256 + threadIdx.x);
    constexpr int PPB = QT == 4 ? 9 : QT == 5 ? 11 : 13;
    const int b = e / PPB, k = e - b * PPB;
    if (b >= nblk) return;
    const int g = b >> 3, bl = b & 7, nb = min(8, nblk - (g << 3));
    if constexpr (QT == 4) {  // pieces of a block: header, then qs in 16-byte pieces (j = piece / 2, low / high half)
        const uint8_t * blk = srow + (size_t) b * 144;
        uint8_t * grp = drow + (size_t) g * 1152;
        const int j = (k - 1) >> 1, hi = (k - 1) & 1;
        uint8_t * o = k == 0 ? grp + 16 * bl : grp + 16 * nb + hi * 64 * nb + 16 * (4 * bl + j);
        *(uint4 *) o = *(const uint4 *) (blk + 16 * k);
    } else if constexpr (QT == 5) {  // header, qh low half, qh high half, then qs as for Q4_K
        const uint8_t * blk = srow + (size_t) b * 176;
        uint8_t * grp = drow + (size_t) g * 1408;
        const int j = (k - 3) >> 1, hi = (k - 3) & 1;
        uint8_t * o = k < 3 ? grp + 16 * nb * k + 16 * bl : grp + 48 * nb + hi * 64 * nb + 16 * (4 * bl + j);
        *(uint4 *) o = *(const uint4 *) (blk + 16 * k);
    } else {  // Q6_K: 12 pieces (plane A / B / C x lane 2 h + t), then the tail item (16 scale bytes + d)
        const uint8_t * blk = srow + (size_t) b * 210;
        if (k < 12) {
            const int plane = k >> 2, l4 = k & 3, h = l4 >> 1, t = l4 & 1;
            const int so = plane == 0 ? 64 * h + 16 * t : plane == 1 ? 64 * h + 32 + 16 * t : 128 + 32 * h + 16 * t;
            st_piece_a2(drow + (size_t) g * 1536 + 64 * nb * plane + 16 * (4 * bl + l4), ld_piece_a2(blk + so));
        } else {
            uint8_t * tail = drow + (size_t) nblk * 192;
            st_piece_a2(tail + (size_t) b * 16, ld_piece_a2(blk + 192));
            *(uint16_t *) (tail + (size_t) nblk * 16 + (size_t) b * 2) = *(const uint16_t *) (blk + 208);
        }
    }
}

bool repack_supported(int type, int64_t K, int64_t nb1) {
    if (K <= 0 || (K % 256) != 0) return false;
    const int64_t nblk = K / 256;
    if (type == GGML_TYPE_Q4_K) return nb1 == nblk * 144;
    if (type == GGML_TYPE_Q5_K) return nb1 == nblk * 176;
    if (type == GGML_TYPE_Q6_K) return nb1 == nblk * 210;
    return false;
}
// rows [r0, r0 + n_rows) of a [K, N] matrix whose rows are nb1 bytes apart in both `src` (block layout) and `dst` (planes)
void launch_repack_planes(hipStream_t s, int type, const void * src, void * dst, int64_t K, int64_t nb1, int64_t r0, int64_t n_rows) {
    if (n_rows <= 0) return;
    const int nblk = (int) (K / 256);
    const uint8_t * sp = (const uint8_t *) src + (size_t) r0 * nb1;
    uint8_t * dp = (uint8_t *) dst + (size_t) r0 * nb1;
    const int ppb = type == GGML_TYPE_Q4_K ? 9 : type == GGML_TYPE_Q5_K ? 11 : 13;
    const unsigned gx = (unsigned) ((nblk * ppb + 255) / 256);
    for (int64_t y0 = 0; y0 < n_rows; y0 += 32768) {  // (grid.y limit)
        const unsigned ny = (unsigned) std::min<int64_t>(32768, n_rows - y0);
        const uint8_t * s1 = sp + (size_t) y0 * nb1;
        uint8_t * d1 = dp + (size_t) y0 * nb1;
        if (type == GGML_TYPE_Q4_K) hipLaunchKernelGGL((k_repack_planes<4>), dim3(gx, ny), dim3(256), 0, s, s1, d1, nblk, nb1);
        else if (type == GGML_TYPE_Q5_K) hipLaunchKernelGGL((k_repack_planes<5>), dim3(gx, ny), dim3(256), 0, s, s1, d1, nblk, nb1);
        else hipLaunchKernelGGL((k_repack_planes<6>), dim3(gx, ny), dim3(256), 0, s, s1, d1, nblk, nb1);
    }
}

// ---- Q8_0: the PANEL copy read by the 9 .. 32-column matrix-core kernel (mmq_q80.hip: k_mmq_q80_skinny).  Same bytes as the tensor (N x K / 32 x 34), regrouped per
// (panel of 32 rows, chunk of 4 blocks) into a tile of 4352 bytes: [block 0 .. 3][K half 0 / 1][row 0 .. 31][16 quants] (4 KB: a wave-instruction's operand load is
// 1 KB of consecutive bytes) + [block][row] f16 scales (256 B).  Rows a multiple of 32, K a multiple of 128, rows densely packed.
__global__ void __launch_bounds__(256) k_repack_q80_panels(const uint8_t * __restrict__ src, uint8_t * __restrict__ dst, const int nchunk, const int64_t nb1) {
    // grid.x = chunk, grid.y = panel; thread t: (block b = t / 64, half kg = (t / 32) & 1, row fr = t % 32)
    const int t = threadIdx.x, b = t >> 6, kg = (t >> 5) & 1, fr = t & 31;
    const uint8_t * blk = src + (size_t) (blockIdx.y * 32 + fr) * nb1 + (size_t) (blockIdx.x * 4 + b) * 34;
    uint8_t * tile = dst + ((size_t) blockIdx.y * nchunk + blockIdx.x) * 4352;
    piece16 v = ld_piece_a2(blk + 2 + 16 * kg);
    st_piece_a2(tile + b * 1024 + kg * 512 + fr * 16, v);
    if (kg == 0) *(uint16_t *) (tile + 4096 + b * 64 + fr * 2) = *(const uint16_t *) blk;
}
bool repack_q80_supported(int type, int64_t K, int64_t N, int64_t nb1) {
    return type == GGML_TYPE_Q8_0 && K > 0 && (K % 128) == 0 && N > 0 && (N % 32) == 0 && nb1 == (K / 32) * 34;
}
void launch_repack_q80_panels(hipStream_t s, const void * src, void * dst, int64_t K, int64_t N, int64_t nb1) {
    const int nchunk = (int) (K / 128);
    const int64_t panels = N / 32;
    for (int64_t p0 = 0; p0 < panels; p0 += 32768) {  // (grid.y limit)
        const unsigned np = (unsigned) std::min<int64_t>(32768, panels - p0);
        hipLaunchKernelGGL(k_repack_q80_panels, dim3((unsigned) nchunk, np), dim3(256), 0, s, (const uint8_t *) src + (size_t) p0 * 32 * nb1, (uint8_t *) dst + (size_t) p0 * nchunk * 4352, nchunk, nb1);
    }
}

MI_TU_TOUCH(repack)

}  // namespace mi355x
