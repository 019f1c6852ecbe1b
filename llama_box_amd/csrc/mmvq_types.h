// mmvq_types.h — per-format load / integer-dot building blocks shared by the mat-vec kernels (mmvq.hip, qkv.hip).
// Each type T provides:  raw (the registers one lane holds for one (super-block, chunk) "pair"),  load(row, p),
// dot<NC>(raw, p, activations, nblk, acc)  and the same pair expressed over a flat dword buffer (loadu/dotu, DW dwords).
#pragma once
#include "dev_util.h"
#include "kernels.h"

namespace mi355x {

// Non-temporal weight loads (the `nt` bit; MI355X_MICROARCH.md "nt-weights" reports -5..-10 % per decode layer for a bf16 stream of
// whole 128-byte lines).  Measured here, round 2: 480 -> 415 tok/s (gate/up 15.2 -> 18.9 us, output 83 -> 141 us).  A 144 / 176 /
// 210-byte super-block is not line-aligned and a lane's header, low and high quants are three instructions that touch the same
// lines: without L2 residency every one of them goes back to memory.  Kept as a build switch, off.
#ifndef MI_NT_WEIGHTS
#define MI_NT_WEIGHTS 0
#endif
// THE DECODE COPY (round 6): weights the batch-1 mat-vec kernels stream are kept a second time in a line-aligned PLANE layout (repack.hip writes it at load time;
// 288 GB of HBM hold both copies of anything llama-box serves on one GPU), and only that copy is read with the nt bit:
//   Q4_K, per group of 8 super-blocks (1152 B = 9 lines): [8 x 16 B {d, dmin, scales}] [32 lanes x 16 B: qs bytes 32 j .. 32 j + 15 of block b, lane 4 b + j]
//                                                          [32 lanes x 16 B: qs bytes 32 j + 16 .. 32 j + 31]
//   Q5_K (1408 B = 11 lines): [8 x 16 B header] [8 x 16 B qh low half] [8 x 16 B qh high half] [32 x 16 B q0] [32 x 16 B q1]
//   Q6_K (per group 1536 B = 12 lines): [32 lanes x 16 B ql 64 h + 16 t ..] [32 x 16 B ql 64 h + 32 + 16 t ..] [32 x 16 B qh 32 h + 16 t ..], lane 4 b + 2 h + t;
//         behind the row's groups: 16 B of scales per super-block, then the f16 d's
// A lane receives exactly the registers T::load of the block layout gives it, so the dot products are the same code and the same bits; every 128-byte line is
// touched by ONE wave-instruction, which is what makes the non-temporal policy pay (on the block layout it cost 15 %: header, low and high quants of a lane are
// three instructions on the same lines).  Measured before building (same bytes from these addresses, wrong values): 541 -> 580 tok/s; the output matrix
// 83.5 -> 67.5 us, Q6_K ffn_down 13.8 -> 11.65, gate/up 14.55 -> 14.0 (profiles/r06_lab_planes_nt.txt).
typedef uint32_t mi_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t mi_u32x2 __attribute__((ext_vector_type(2)));
typedef mi_u32x4 mi_u32x4_a2 __attribute__((aligned(2)));
typedef mi_u32x2 mi_u32x2_a2 __attribute__((aligned(2)));
template <typename V, typename N> __device__ __forceinline__ V ld_stream_as(const V * p) {
#if MI_NT_WEIGHTS
    const N v = __builtin_nontemporal_load((const N *) p);
    return __builtin_bit_cast(V, v);
#else
    return *p;
#endif
}
__device__ __forceinline__ uint4 ld_stream(const uint4 * p) { return ld_stream_as<uint4, mi_u32x4>(p); }
__device__ __forceinline__ uint4 ld_nt(const void * p) { return __builtin_bit_cast(uint4, __builtin_nontemporal_load((const mi_u32x4 *) p)); }
__device__ __forceinline__ uint2 ld_nt8(const void * p) { return __builtin_bit_cast(uint2, __builtin_nontemporal_load((const mi_u32x2 *) p)); }

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
// 4 lanes per super-block: lane j owns the 32 qs bytes of 64-value group j (low nibbles = sub-block 2j, high = 2j+1).
// Round-1 ISA count: with 16 bytes per lane the loop spent ~94 VALU instructions per 16 B of weights (scale unpacking
// and predication repeated per lane), enough to make the 66 MB gate/up launch ALU-limited; 32 B per lane amortises the
// header work over twice the bytes and K = 4096 becomes exactly one 64-lane trip per row.
struct T_Q4K {
    typedef q8k_dev act;
    static constexpr int BLK = 256, BYTES = 144, PPB = 4;
    struct raw { uint4 hdr, q0, q1; };
    static constexpr int DW = 12;
    static __device__ __forceinline__ raw load(const uint8_t * __restrict__ row, int p, int nblk = 0) {
        (void) nblk;
        raw r;
        const uint8_t * blk = row + (size_t) (p >> 2) * BYTES;
        r.hdr = ld_stream((const uint4 *) blk);
        r.q0 = ld_stream((const uint4 *) (blk + 16 + 32 * (p & 3)));
        r.q1 = ld_stream((const uint4 *) (blk + 32 + 32 * (p & 3)));
        return r;
    }
    template <int NC> static __device__ __forceinline__ void dot(const raw & r, int p, const act * __restrict__ y, int nblk, float * acc) {
        const int b = p >> 2, j = p & 3;
        const float d = h2f((uint16_t) (r.hdr.x & 0xFFFF)), dmin = h2f((uint16_t) (r.hdr.x >> 16));
        int sc0, sc1, m0, m1;
        k4_scale_pair(r.hdr.y, r.hdr.z, r.hdr.w, j, sc0, sc1, m0, m1);
        const uint32_t qv[8] = {r.q0.x, r.q0.y, r.q0.z, r.q0.w, r.q1.x, r.q1.y, r.q1.z, r.q1.w};
#pragma unroll
        for (int col = 0; col < NC; ++col) {
            const act * yb = y + (size_t) col * nblk + b;
            const uint4 * yq = (const uint4 *) (yb->qs + 64 * j);
            const uint4 y0 = yq[0], y1 = yq[1], y2 = yq[2], y3 = yq[3];
            const uint32_t yl[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
            const uint32_t yh[8] = {y2.x, y2.y, y2.z, y2.w, y3.x, y3.y, y3.z, y3.w};
            int s_lo = 0, s_hi = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                s_lo = dot4((int) (qv[k] & 0x0F0F0F0Fu), (int) yl[k], s_lo);
                s_hi = dot4((int) ((qv[k] >> 4) & 0x0F0F0F0Fu), (int) yh[k], s_hi);
            }
            const uint32_t bs = *(const uint32_t *) (yb->bs32 + 2 * j);  // sums of q8 over sub-blocks 2j, 2j+1
            const int bs_lo = (int) (int16_t) (bs & 0xFFFF), bs_hi = (int) (int16_t) (bs >> 16);
            const int isum = __mul24(sc0, s_lo) + __mul24(sc1, s_hi);
            const int msum = __mul24(m0, bs_lo) + __mul24(m1, bs_hi);
            acc[col] += yb->d * (d * (float) isum - dmin * (float) msum);
        }
    }
};

// ------------------------------------------------------------------------------------------------ Q5_K
struct T_Q5K {
    typedef q8k_dev act;
    static constexpr int BLK = 256, BYTES = 176, PPB = 4;
    struct raw { uint4 hdr, h0, h1, q0, q1; };
    static constexpr int DW = 20;
    static __device__ __forceinline__ raw load(const uint8_t * __restrict__ row, int p, int = 0) {
        const uint8_t * blk = row + (size_t) (p >> 2) * BYTES;
        raw r;
        r.hdr = ld_stream((const uint4 *) blk);
        r.h0 = ld_stream((const uint4 *) (blk + 16));
        r.h1 = ld_stream((const uint4 *) (blk + 32));
        r.q0 = ld_stream((const uint4 *) (blk + 48 + 32 * (p & 3)));
        r.q1 = ld_stream((const uint4 *) (blk + 64 + 32 * (p & 3)));
        return r;
    }
    template <int NC> static __device__ __forceinline__ void dot(const raw & r, int p, const act * __restrict__ y, int nblk, float * acc) {
        const int b = p >> 2, j = p & 3;
        const float d = h2f((uint16_t) (r.hdr.x & 0xFFFF)), dmin = h2f((uint16_t) (r.hdr.x >> 16));
        int sc0, sc1, m0, m1;
        k4_scale_pair(r.hdr.y, r.hdr.z, r.hdr.w, j, sc0, sc1, m0, m1);
        const uint32_t qv[8] = {r.q0.x, r.q0.y, r.q0.z, r.q0.w, r.q1.x, r.q1.y, r.q1.z, r.q1.w};
        const uint32_t qh[8] = {r.h0.x, r.h0.y, r.h0.z, r.h0.w, r.h1.x, r.h1.y, r.h1.z, r.h1.w};
        uint32_t lo[8], hi[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            lo[k] = (qv[k] & 0x0F0F0F0Fu) | (((qh[k] >> (2 * j)) & 0x01010101u) << 4);
            hi[k] = ((qv[k] >> 4) & 0x0F0F0F0Fu) | (((qh[k] >> (2 * j + 1)) & 0x01010101u) << 4);
        }
#pragma unroll
        for (int col = 0; col < NC; ++col) {
            const act * yb = y + (size_t) col * nblk + b;
            const uint4 * yq = (const uint4 *) (yb->qs + 64 * j);
            const uint4 y0 = yq[0], y1 = yq[1], y2 = yq[2], y3 = yq[3];
            const uint32_t yl[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
            const uint32_t yh[8] = {y2.x, y2.y, y2.z, y2.w, y3.x, y3.y, y3.z, y3.w};
            int s_lo = 0, s_hi = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                s_lo = dot4((int) lo[k], (int) yl[k], s_lo);
                s_hi = dot4((int) hi[k], (int) yh[k], s_hi);
            }
            const uint32_t bs = *(const uint32_t *) (yb->bs32 + 2 * j);
            const int bs_lo = (int) (int16_t) (bs & 0xFFFF), bs_hi = (int) (int16_t) (bs >> 16);
            const int isum = __mul24(sc0, s_lo) + __mul24(sc1, s_hi);
            const int msum = __mul24(m0, bs_lo) + __mul24(m1, bs_hi);
            acc[col] += yb->d * (d * (float) isum - dmin * (float) msum);
        }
    }
};

// ------------------------------------------------------------------------------------------------ Q6_K
// 4 lanes per super-block, 64 values each (round 2; round 1 used 8 lanes x 34 bytes fetched as five 8- / 2-byte loads, and the
// Q6_K launches streamed at 3.1 TB/s where Q4_K reached 4.2): lane (h, t) owns l = 16t .. 16t+15 of half h, i.e. 16 bytes of
// ql twice (low / high nibbles = the value groups 128h + {0, 64} + l and 128h + {32, 96} + l) and the 16 bytes of qh that carry
// their upper two bits — three 16-byte loads at the format's 2-byte alignment (gfx950 serves unaligned global_load_dwordx4),
// one 8-byte load for the half's scales and the f16 d.  The -32 offset of the 6-bit codes comes from the activation block's
// 16-value sums (bsums) instead of a second dot product.
struct __attribute__((packed, aligned(2))) u128_a2 { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(2))) u64_a2 { uint32_t x, y; };
__device__ __forceinline__ u128_a2 ld_stream(const u128_a2 * p) { return ld_stream_as<u128_a2, mi_u32x4_a2>(p); }
__device__ __forceinline__ u64_a2 ld_stream(const u64_a2 * p) { return ld_stream_as<u64_a2, mi_u32x2_a2>(p); }
struct T_Q6K {
    typedef q8k_dev act;
    static constexpr int BLK = 256, BYTES = 210, PPB = 4;
    struct raw { u128_a2 a, b, c; u64_a2 s; uint16_t d; };
    static constexpr int DW = 15;
    static __device__ __forceinline__ raw load(const uint8_t * __restrict__ row, int p, int nblk = 0) {
        const uint8_t * blk = row + (size_t) (p >> 2) * BYTES;
        const int h = (p >> 1) & 1, t = p & 1;
        raw r;
        (void) nblk;
        r.a = ld_stream((const u128_a2 *) (blk + 64 * h + 16 * t));
        r.b = ld_stream((const u128_a2 *) (blk + 64 * h + 32 + 16 * t));
        r.c = ld_stream((const u128_a2 *) (blk + 128 + 32 * h + 16 * t));
        r.s = ld_stream((const u64_a2 *) (blk + 192 + 8 * h));
        r.d = ld16(blk + 208);
        return r;
    }
    template <int NC> static __device__ __forceinline__ void dot(const raw & r, int p, const act * __restrict__ y, int nblk, float * acc) {
        const int b = p >> 2, h = (p >> 1) & 1, t = p & 1;
        const float d = h2f(r.d);
        const uint32_t A[4] = {r.a.x, r.a.y, r.a.z, r.a.w}, B[4] = {r.b.x, r.b.y, r.b.z, r.b.w}, C[4] = {r.c.x, r.c.y, r.c.z, r.c.w};
        uint32_t v[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[0][i] = (A[i] & 0x0F0F0F0Fu) | ((C[i] & 0x03030303u) << 4);
            v[1][i] = (B[i] & 0x0F0F0F0Fu) | (((C[i] >> 2) & 0x03030303u) << 4);
            v[2][i] = ((A[i] >> 4) & 0x0F0F0F0Fu) | (((C[i] >> 4) & 0x03030303u) << 4);
            v[3][i] = ((B[i] >> 4) & 0x0F0F0F0Fu) | (((C[i] >> 6) & 0x03030303u) << 4);
        }
        // scales of the half: bytes 8h .. 8h+7 of the block's 16; group k of this lane uses byte t + 2k
        const int sc[4] = {
            (int) (int8_t) (r.s.x >> (8 * t)), (int) (int8_t) (r.s.x >> (8 * (t + 2))),
            (int) (int8_t) (r.s.y >> (8 * t)), (int) (int8_t) (r.s.y >> (8 * (t + 2))),
        };
#pragma unroll
        for (int col = 0; col < NC; ++col) {
            const act * yb = y + (size_t) col * nblk + b;
            int isum = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint4 yk = *(const uint4 *) (yb->qs + 128 * h + 32 * k + 16 * t);
                // sum (q - 32) * y = dot(q, y) - 32 * sum(y); sum(y) over these 16 values = bsums[8h + 2k + t] (an exact f16 integer)
                int s = dot4((int) v[k][0], (int) yk.x, 0);
                s = dot4((int) v[k][1], (int) yk.y, s);
                s = dot4((int) v[k][2], (int) yk.z, s);
                s = dot4((int) v[k][3], (int) yk.w, s);
                const int ys = (int) h2f(yb->bsums[8 * h + 2 * k + t]);
                isum += __mul24(sc[k], s - 32 * ys);
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
    static __device__ __forceinline__ raw load(const uint8_t * __restrict__ row, int p, int = 0) {
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



// ------------------------------------------------------------------------------------------------ the plane layouts of the decode copy (see the top of this file)
// (a row whose super-block count is not a multiple of 8 — Qwen2-7B: 14 and 74 — ends in a SHORT group of nb = nblk % 8 super-blocks with the same plane order,
// 4 nb lanes wide; only its lines are shared between planes)
__device__ __forceinline__ int plane_group_blocks(const int p, const int nblk) { return min(8, nblk - ((p >> 5) << 3)); }
// (WHOLE: every group has 8 super-blocks — the offsets are constants; the short-group forms are separate kernels, T_Q*KS, so Llama's launches pay nothing for them)
template <bool WHOLE> struct T_Q4KP_t : T_Q4K {
    static __device__ __forceinline__ raw load(const uint8_t * __restrict__ row, int p, int nblk) {
        const uint8_t * grp = row + (size_t) (p >> 5) * 1152;
        const int nb = WHOLE ? 8 : plane_group_blocks(p, nblk);
        raw r;
        r.hdr = ld_nt(grp + 16 * ((p >> 2) & 7));
        r.q0 = ld_nt(grp + 16 * nb + 16 * (p & 31));
        r.q1 = ld_nt(grp + 80 * nb + 16 * (p & 31));
        return r;
    }
};
template <bool WHOLE> struct T_Q5KP_t : T_Q5K {
    static __device__ __forceinline__ raw load(const uint8_t * __restrict__ row, int p, int nblk) {
        const uint8_t * grp = row + (size_t) (p >> 5) * 1408;
        const int nb = WHOLE ? 8 : plane_group_blocks(p, nblk), b8 = (p >> 2) & 7;
        raw r;
        r.hdr = ld_nt(grp + 16 * b8);
        r.h0 = ld_nt(grp + 16 * nb + 16 * b8);
        r.h1 = ld_nt(grp + 32 * nb + 16 * b8);
        r.q0 = ld_nt(grp + 48 * nb + 16 * (p & 31));
        r.q1 = ld_nt(grp + 112 * nb + 16 * (p & 31));
        return r;
    }
};
// (Q6_K rows are 210 nblk bytes: 16-byte aligned only when nblk is a multiple of 8 — the plane loads carry the format's 2-byte alignment, which gfx950 serves)
__device__ __forceinline__ u128_a2 ld_nt_a2(const void * p) { return __builtin_bit_cast(u128_a2, __builtin_nontemporal_load((const mi_u32x4_a2 *) p)); }
template <bool WHOLE> struct T_Q6KP_t : T_Q6K {
    static __device__ __forceinline__ raw load(const uint8_t * __restrict__ row, int p, int nblk) {
        const uint8_t * grp = row + (size_t) (p >> 5) * 1536;
        const uint8_t * tail = row + (size_t) nblk * 192;
        const int nb = WHOLE ? 8 : plane_group_blocks(p, nblk), h = (p >> 1) & 1;
        raw r;
        r.a = ld_nt_a2(grp + 16 * (p & 31));
        r.b = ld_nt_a2(grp + 64 * nb + 16 * (p & 31));
        r.c = ld_nt_a2(grp + 128 * nb + 16 * (p & 31));
        r.s = __builtin_bit_cast(u64_a2, __builtin_nontemporal_load((const mi_u32x2_a2 *) (tail + (size_t) (p >> 2) * 16 + 8 * h)));
        r.d = ld16(tail + (size_t) nblk * 16 + (size_t) (p >> 2) * 2);
        return r;
    }
};
struct T_Q4KP : T_Q4KP_t<true> {};
struct T_Q5KP : T_Q5KP_t<true> {};
struct T_Q6KP : T_Q6KP_t<true> {};
struct T_Q4KS : T_Q4KP_t<false> {};  // rows with a short last group (K % 2048 != 0)
struct T_Q5KS : T_Q5KP_t<false> {};
struct T_Q6KS : T_Q6KP_t<false> {};
template <typename T> struct plane_of { typedef T type; typedef T short_type; };
template <> struct plane_of<T_Q4K> { typedef T_Q4KP type; typedef T_Q4KS short_type; };
template <> struct plane_of<T_Q5K> { typedef T_Q5KP type; typedef T_Q5KS short_type; };
template <> struct plane_of<T_Q6K> { typedef T_Q6KP type; typedef T_Q6KS short_type; };

}  // namespace mi355x
