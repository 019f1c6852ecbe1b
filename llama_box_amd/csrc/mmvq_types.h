// mmvq_types.h — per-format load / integer-dot building blocks shared by the mat-vec kernels (mmvq.hip, qkv.hip).
// Each type T provides:  raw (the registers one lane holds for one (super-block, chunk) "pair"),  load(row, p),
// dot<NC>(raw, p, activations, nblk, acc)  and the same pair expressed over a flat dword buffer (loadu/dotu, DW dwords).
#pragma once
#include "dev_util.h"
#include "kernels.h"

namespace mi355x {

// sc/m pair extraction for K-quants' 12 packed bytes, for sub-blocks (2j, 2j+1); hy/hz/hw = bytes 0-3 / 4-7 / 8-11
__device__ __forceinline__ void k4_scale_pair(uint32_t hy, uint32_t hz, uint32_t hw, int j, int & sc0, int & sc1, int & m0, int & m1) {
    const int sh = 16 * (j & 1);
    const uint32_t a = (hy >> sh) & 0xFFFFu, b = (hz >> sh) & 0xFFFFu, w = (hw >> sh) & 0xFFFFu;
    uint32_t scp, mp;
    if (j < 2) {
        scp = a & 0x3F3Fu;
        mp = b & 0x3F3Fu;
    } else {
        scp = (w & 0x0F0Fu) | ((a & 0xC0C0u) >> 2);
        mp = ((w >> 4) & 0x0F0Fu) | ((b & 0xC0C0u) >> 2);
    }
    sc0 = (int) (scp & 0xFF);
    sc1 = (int) (scp >> 8);
    m0 = (int) (mp & 0xFF);
    m1 = (int) (mp >> 8);
}

// ------------------------------------------------------------------------------------------------ Q4_K
struct T_Q4K {
    typedef q8k_dev act;
    static constexpr int BLK = 256, BYTES = 144, PPB = 8;  // pairs (lanes) per block
    struct raw { uint4 hdr, q; };
    static constexpr int DW = 8;
    static __device__ __forceinline__ void pack(const raw & r, uint32_t * d) {
        d[0] = r.hdr.x; d[1] = r.hdr.y; d[2] = r.hdr.z; d[3] = r.hdr.w; d[4] = r.q.x; d[5] = r.q.y; d[6] = r.q.z; d[7] = r.q.w;
    }
    static __device__ __forceinline__ raw unpack(const uint32_t * d) {
        raw r;
        r.hdr = make_uint4(d[0], d[1], d[2], d[3]);
        r.q = make_uint4(d[4], d[5], d[6], d[7]);
        return r;
    }
    static __device__ __forceinline__ raw load(const uint8_t * __restrict__ row, int p) {
        const uint8_t * blk = row + (size_t) (p >> 3) * BYTES;
        raw r;
        r.hdr = *(const uint4 *) blk;
        r.q = *(const uint4 *) (blk + 16 + 16 * (p & 7));
        return r;
    }
    template <int NC> static __device__ __forceinline__ void dot(const raw & r, int p, const act * __restrict__ y, int nblk, float * acc) {
        const int b = p >> 3, c = p & 7, j = c >> 1;
        const float d = h2f((uint16_t) (r.hdr.x & 0xFFFF)), dmin = h2f((uint16_t) (r.hdr.x >> 16));
        int sc0, sc1, m0, m1;
        k4_scale_pair(r.hdr.y, r.hdr.z, r.hdr.w, j, sc0, sc1, m0, m1);
        const int e0 = 64 * j + 16 * (c & 1);
        const uint32_t qv[4] = {r.q.x, r.q.y, r.q.z, r.q.w};
#pragma unroll
        for (int col = 0; col < NC; ++col) {
            const act * yb = y + (size_t) col * nblk + b;
            const uint4 ylo = *(const uint4 *) (yb->qs + e0);
            const uint4 yhi = *(const uint4 *) (yb->qs + e0 + 32);
            const uint32_t yl[4] = {ylo.x, ylo.y, ylo.z, ylo.w}, yh[4] = {yhi.x, yhi.y, yhi.z, yhi.w};
            int s_lo = 0, s_hi = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s_lo = dot4((int) (qv[k] & 0x0F0F0F0Fu), (int) yl[k], s_lo);
                s_hi = dot4((int) ((qv[k] >> 4) & 0x0F0F0F0Fu), (int) yh[k], s_hi);
            }
            const int bs_lo = yb->bsums[4 * j + (c & 1)], bs_hi = yb->bsums[4 * j + 2 + (c & 1)];
            const int isum = sc0 * s_lo + sc1 * s_hi;
            const int msum = m0 * bs_lo + m1 * bs_hi;
            acc[col] += yb->d * (d * (float) isum - dmin * (float) msum);
        }
    }
};

// ------------------------------------------------------------------------------------------------ Q5_K
struct T_Q5K {
    typedef q8k_dev act;
    static constexpr int BLK = 256, BYTES = 176, PPB = 8;
    struct raw { uint4 hdr, qh, q; };
    static constexpr int DW = 12;
    static __device__ __forceinline__ void pack(const raw & r, uint32_t * d) {
        d[0] = r.hdr.x; d[1] = r.hdr.y; d[2] = r.hdr.z; d[3] = r.hdr.w; d[4] = r.qh.x; d[5] = r.qh.y; d[6] = r.qh.z; d[7] = r.qh.w;
        d[8] = r.q.x; d[9] = r.q.y; d[10] = r.q.z; d[11] = r.q.w;
    }
    static __device__ __forceinline__ raw unpack(const uint32_t * d) {
        raw r;
        r.hdr = make_uint4(d[0], d[1], d[2], d[3]);
        r.qh = make_uint4(d[4], d[5], d[6], d[7]);
        r.q = make_uint4(d[8], d[9], d[10], d[11]);
        return r;
    }
    static __device__ __forceinline__ raw load(const uint8_t * __restrict__ row, int p) {
        const uint8_t * blk = row + (size_t) (p >> 3) * BYTES;
        raw r;
        r.hdr = *(const uint4 *) blk;
        r.qh = *(const uint4 *) (blk + 16 + 16 * (p & 1));
        r.q = *(const uint4 *) (blk + 48 + 16 * (p & 7));
        return r;
    }
    template <int NC> static __device__ __forceinline__ void dot(const raw & r, int p, const act * __restrict__ y, int nblk, float * acc) {
        const int b = p >> 3, c = p & 7, j = c >> 1;
        const float d = h2f((uint16_t) (r.hdr.x & 0xFFFF)), dmin = h2f((uint16_t) (r.hdr.x >> 16));
        int sc0, sc1, m0, m1;
        k4_scale_pair(r.hdr.y, r.hdr.z, r.hdr.w, j, sc0, sc1, m0, m1);
        const int e0 = 64 * j + 16 * (c & 1);
        const uint32_t qv[4] = {r.q.x, r.q.y, r.q.z, r.q.w}, qh[4] = {r.qh.x, r.qh.y, r.qh.z, r.qh.w};
        uint32_t lo[4], hi[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            lo[k] = (qv[k] & 0x0F0F0F0Fu) | (((qh[k] >> (2 * j)) & 0x01010101u) << 4);
            hi[k] = ((qv[k] >> 4) & 0x0F0F0F0Fu) | (((qh[k] >> (2 * j + 1)) & 0x01010101u) << 4);
        }
#pragma unroll
        for (int col = 0; col < NC; ++col) {
            const act * yb = y + (size_t) col * nblk + b;
            const uint4 ylo = *(const uint4 *) (yb->qs + e0);
            const uint4 yhi = *(const uint4 *) (yb->qs + e0 + 32);
            const uint32_t yl[4] = {ylo.x, ylo.y, ylo.z, ylo.w}, yh[4] = {yhi.x, yhi.y, yhi.z, yhi.w};
            int s_lo = 0, s_hi = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s_lo = dot4((int) lo[k], (int) yl[k], s_lo);
                s_hi = dot4((int) hi[k], (int) yh[k], s_hi);
            }
            const int bs_lo = yb->bsums[4 * j + (c & 1)], bs_hi = yb->bsums[4 * j + 2 + (c & 1)];
            const int isum = sc0 * s_lo + sc1 * s_hi;
            const int msum = m0 * bs_lo + m1 * bs_hi;
            acc[col] += yb->d * (d * (float) isum - dmin * (float) msum);
        }
    }
};

// ------------------------------------------------------------------------------------------------ Q6_K
// 16 lanes per super-block: lane (h, t) owns l = 4t..4t+3 of half h, i.e. 16 of the 256 values
struct T_Q6K {
    typedef q8k_dev act;
    static constexpr int BLK = 256, BYTES = 210, PPB = 16;
    struct raw { uint32_t ql0, ql1, qh, s0, s1; uint16_t d; };
    static constexpr int DW = 6;
    static __device__ __forceinline__ void pack(const raw & r, uint32_t * d) {
        d[0] = r.ql0; d[1] = r.ql1; d[2] = r.qh; d[3] = r.s0; d[4] = r.s1; d[5] = r.d;
    }
    static __device__ __forceinline__ raw unpack(const uint32_t * d) {
        raw r;
        r.ql0 = d[0]; r.ql1 = d[1]; r.qh = d[2]; r.s0 = d[3]; r.s1 = d[4]; r.d = (uint16_t) d[5];
        return r;
    }
    static __device__ __forceinline__ raw load(const uint8_t * __restrict__ row, int p) {
        const uint8_t * blk = row + (size_t) (p >> 4) * BYTES;
        const int h = (p >> 3) & 1, t = p & 7;
        raw r;
        r.ql0 = ld32_a2(blk + 64 * h + 4 * t);
        r.ql1 = ld32_a2(blk + 64 * h + 32 + 4 * t);
        r.qh = ld32_a2(blk + 128 + 32 * h + 4 * t);
        r.s0 = ld32_a2(blk + 192 + 8 * h);
        r.s1 = ld32_a2(blk + 196 + 8 * h);
        r.d = ld16(blk + 208);
        return r;
    }
    template <int NC> static __device__ __forceinline__ void dot(const raw & r, int p, const act * __restrict__ y, int nblk, float * acc) {
        const int b = p >> 4, h = (p >> 3) & 1, t = p & 7, is = t >> 2;
        const float d = h2f(r.d);
        const uint32_t v[4] = {
            (r.ql0 & 0x0F0F0F0Fu) | ((r.qh & 0x03030303u) << 4),
            (r.ql1 & 0x0F0F0F0Fu) | (((r.qh >> 2) & 0x03030303u) << 4),
            ((r.ql0 >> 4) & 0x0F0F0F0Fu) | (((r.qh >> 4) & 0x03030303u) << 4),
            ((r.ql1 >> 4) & 0x0F0F0F0Fu) | (((r.qh >> 6) & 0x03030303u) << 4),
        };
        const int sc[4] = {
            (int) (int8_t) (r.s0 >> (8 * is)), (int) (int8_t) (r.s0 >> (8 * (is + 2))),
            (int) (int8_t) (r.s1 >> (8 * is)), (int) (int8_t) (r.s1 >> (8 * (is + 2))),
        };
#pragma unroll
        for (int col = 0; col < NC; ++col) {
            const act * yb = y + (size_t) col * nblk + b;
            int isum = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int yk = *(const int *) (yb->qs + 128 * h + 32 * k + 4 * t);
                // sum (q - 32) * y = dot(q, y) - 32 * sum(y)
                const int s = dot4((int) v[k], yk, 0) - 32 * dot4(0x01010101, yk, 0);
                isum += sc[k] * s;
            }
            acc[col] += yb->d * d * (float) isum;
        }
    }
};

// ------------------------------------------------------------------------------------------------ Q8_0
struct T_Q80 {
    typedef q80_dev act;
    static constexpr int BLK = 32, BYTES = 34, PPB = 1;
    struct raw { uint32_t q[8]; uint16_t d; };
    static constexpr int DW = 9;
    static __device__ __forceinline__ raw load(const uint8_t * __restrict__ row, int p) {
        const uint8_t * blk = row + (size_t) p * BYTES;
        raw r;
        r.d = ld16(blk);
#pragma unroll
        for (int k = 0; k < 8; ++k) r.q[k] = ld32_a2(blk + 2 + 4 * k);
        return r;
    }
    template <int NC> static __device__ __forceinline__ void dot(const raw & r, int p, const act * __restrict__ y, int nblk, float * acc) {
        const float d = h2f(r.d);
#pragma unroll
        for (int col = 0; col < NC; ++col) {
            const act * yb = y + (size_t) col * nblk + p;
            int s = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) s = dot4((int) r.q[k], ((const int *) yb->qs)[k], s);
            acc[col] += (float) s * (d * yb->d);
        }
    }
};


}  // namespace mi355x
