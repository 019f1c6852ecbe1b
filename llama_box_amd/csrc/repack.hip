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

// grid.x covers the pieces of one row, grid.y the rows
template <int QT> __global__ void __launch_bounds__(256) k_repack_planes(const uint8_t * __restrict__ src, uint8_t * __restrict__ dst, const int nblk, const int64_t nb1) {
    const uint8_t * srow = src + (size_t) blockIdx.y * nb1;
    uint8_t * drow = dst + (size_t) blockIdx.y * nb1;
    const int e = (int) (blockIdx.x * 256 + threadIdx.x);
    if constexpr (QT == 4) {  // per group of 8 blocks: 8 header pieces, 32 q0 pieces, 32 q1 pieces
        const int grp = e / 72, d = e - grp * 72;
        if (grp >= (nblk >> 3)) return;
        const uint8_t * g0 = srow + (size_t) grp * 1152;
        uint8_t * o = drow + (size_t) grp * 1152;
        int so;
        if (d < 8) so = d * 144;
        else {
            const int l = (d - 8) & 31, hi = (d - 8) >> 5;
            so = (l >> 2) * 144 + 16 + 32 * (l & 3) + 16 * hi;
        }
        *(uint4 *) (o + 16 * d) = *(const uint4 *) (g0 + so);
    } else if constexpr (QT == 5) {  // 8 headers, 8 qh low halves, 8 qh high halves, 32 q0, 32 q1
        const int grp = e / 88, d = e - grp * 88;
        if (grp >= (nblk >> 3)) return;
        const uint8_t * g0 = srow + (size_t) grp * 1408;
        uint8_t * o = drow + (size_t) grp * 1408;
        int so;
        if (d < 24) so = (d & 7) * 176 + 16 * (d >> 3);
        else {
            const int l = (d - 24) & 31, hi = (d - 24) >> 5;
            so = (l >> 2) * 176 + 48 + 32 * (l & 3) + 16 * hi;
        }
        *(uint4 *) (o + 16 * d) = *(const uint4 *) (g0 + so);
    } else {  // Q6_K: per group 32 A, 32 B, 32 C pieces; then per super-block of the row one tail item (scales + d)
        const int n_grp = nblk >> 3;
        if (e < n_grp * 96) {
            const int grp = e / 96, d = e - grp * 96;
            const int plane = d >> 5, l = d & 31, h = (l >> 1) & 1, t = l & 1;
            const uint8_t * blk = srow + (size_t) (grp * 8 + (l >> 2)) * 210;
            const int so = plane == 0 ? 64 * h + 16 * t : plane == 1 ? 64 * h + 32 + 16 * t : 128 + 32 * h + 16 * t;
            st_piece_a2(drow + (size_t) grp * 1536 + 16 * d, ld_piece_a2(blk + so));
        } else {
            const int b = e - n_grp * 96;
            if (b >= nblk) return;
            const uint8_t * blk = srow + (size_t) b * 210;
            uint8_t * tail = drow + (size_t) n_grp * 1536;
            st_piece_a2(tail + (size_t) b * 16, ld_piece_a2(blk + 192));
            *(uint16_t *) (tail + (size_t) nblk * 16 + (size_t) b * 2) = *(const uint16_t *) (blk + 208);
        }
    }
}

bool repack_supported(int type, int64_t K, int64_t nb1) {
    if (K <= 0 || (K % 2048) != 0) return false;
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
    for (int64_t y0 = 0; y0 < n_rows; y0 += 32768) {  // (grid.y limit)
        const unsigned ny = (unsigned) std::min<int64_t>(32768, n_rows - y0);
        const uint8_t * s1 = sp + (size_t) y0 * nb1;
        uint8_t * d1 = dp + (size_t) y0 * nb1;
        if (type == GGML_TYPE_Q4_K) hipLaunchKernelGGL((k_repack_planes<4>), dim3((unsigned) (((nblk >> 3) * 72 + 255) / 256), ny), dim3(256), 0, s, s1, d1, nblk, nb1);
        else if (type == GGML_TYPE_Q5_K) hipLaunchKernelGGL((k_repack_planes<5>), dim3((unsigned) (((nblk >> 3) * 88 + 255) / 256), ny), dim3(256), 0, s, s1, d1, nblk, nb1);
        else hipLaunchKernelGGL((k_repack_planes<6>), dim3((unsigned) (((nblk >> 3) * 96 + nblk + 255) / 256), ny), dim3(256), 0, s, s1, d1, nblk, nb1);
    }
}

MI_TU_TOUCH(repack)

}  // namespace mi355x
