"""Generates tests/golden/quant_golden.npz — golden vectors for the quantised formats on the hot path.

The reference snapshot holds NO golden vectors for this path (parity unpinned), so these are
produced by an INDEPENDENT NumPy restatement of the published ggml formats, written
vector-wise from the format description rather than from the C oracle's loops.  The C oracle (oracle/) and the
HIP kernels are both checked against them: a three-way agreement (NumPy / C / HIP) is the strongest pin available
offline.  Run:  python tests/golden/make_golden.py   (deterministic; commit the .npz)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))


def f16(b):  # two uint8 columns -> float32
    return np.ascontiguousarray(b).view(np.float16).astype(np.float32).reshape(-1)


def scale_min_k4(s12):
    """12 packed bytes -> (8 six-bit scales, 8 six-bit mins)."""
    s12 = s12.astype(np.int32)
    sc = np.zeros((s12.shape[0], 8), np.int32)
    mn = np.zeros((s12.shape[0], 8), np.int32)
    sc[:, :4] = s12[:, 0:4] & 63
    mn[:, :4] = s12[:, 4:8] & 63
    sc[:, 4:] = (s12[:, 8:12] & 0xF) | ((s12[:, 0:4] >> 6) << 4)
    mn[:, 4:] = (s12[:, 8:12] >> 4) | ((s12[:, 4:8] >> 6) << 4)
    return sc, mn


def unpack_q(qtype, blocks):
    """-> integer quants [n, 256 or 32] (Q4_K/Q5_K unsigned, Q6_K already minus 32, Q8_0 signed)."""
    import llama_box_amd as L
    b = blocks
    n = b.shape[0]
    if qtype == L.Q8_0:
        return b[:, 2:34].view(np.int8).astype(np.int32)
    if qtype in (L.Q4_K, L.Q5_K):
        qs = b[:, 48:176] if qtype == L.Q5_K else b[:, 16:144]
        q = np.zeros((n, 256), np.int32)
        for j in range(4):
            q[:, 64 * j:64 * j + 32] = qs[:, 32 * j:32 * j + 32] & 0xF
            q[:, 64 * j + 32:64 * j + 64] = qs[:, 32 * j:32 * j + 32] >> 4
        if qtype == L.Q5_K:
            qh = b[:, 16:48].astype(np.int32)
            for s in range(8):
                q[:, 32 * s:32 * s + 32] += ((qh >> s) & 1) * 16
        return q
    if qtype == L.Q6_K:
        ql = b[:, 0:128].astype(np.int32)
        qh = b[:, 128:192].astype(np.int32)
        q = np.zeros((n, 256), np.int32)
        for h in range(2):
            l0, l1, hh = ql[:, 64 * h:64 * h + 32], ql[:, 64 * h + 32:64 * h + 64], qh[:, 32 * h:32 * h + 32]
            q[:, 128 * h + 0:128 * h + 32] = (l0 & 0xF) | ((hh & 3) << 4)
            q[:, 128 * h + 32:128 * h + 64] = (l1 & 0xF) | (((hh >> 2) & 3) << 4)
            q[:, 128 * h + 64:128 * h + 96] = (l0 >> 4) | (((hh >> 4) & 3) << 4)
            q[:, 128 * h + 96:128 * h + 128] = (l1 >> 4) | (((hh >> 6) & 3) << 4)
        return q - 32
    raise ValueError(qtype)


def dequant(qtype, blocks):
    """float32 dequantisation with the same two-rounding arithmetic as dequantize_row_* (d*sc first, then *q, then -m)."""
    import llama_box_amd as L
    b = blocks
    q = unpack_q(qtype, b).astype(np.float32)
    if qtype == L.Q8_0:
        return q * f16(b[:, 0:2])[:, None]
    if qtype in (L.Q4_K, L.Q5_K):
        d, dmin = f16(b[:, 0:2]), f16(b[:, 2:4])
        sc, mn = scale_min_k4(b[:, 4:16])
        d1 = (d[:, None] * sc.astype(np.float32)).astype(np.float32)
        m1 = (dmin[:, None] * mn.astype(np.float32)).astype(np.float32)
        y = (np.repeat(d1, 32, axis=1) * q).astype(np.float32) - np.repeat(m1, 32, axis=1)
        return y.astype(np.float32)
    if qtype == L.Q6_K:
        d = f16(b[:, 208:210])
        sc = b[:, 192:208].view(np.int8).astype(np.float32)
        ds = (d[:, None] * sc).astype(np.float32)
        return (np.repeat(ds, 16, axis=1) * q).astype(np.float32)
    raise ValueError(qtype)


def quantize_q8_K(x):
    """x [n, 256] float32 -> (d float32 [n], qs int8 [n,256], bsums int16 [n,16]); ties: first index, round-half-even."""
    x = x.astype(np.float32)
    idx = np.argmax(np.abs(x), axis=1)  # first occurrence of the max |x|
    mx = x[np.arange(x.shape[0]), idx]
    d = np.zeros(x.shape[0], np.float32)
    qs = np.zeros(x.shape, np.int8)
    nz = mx != 0
    iscale = np.zeros_like(mx)
    iscale[nz] = (np.float32(-127.0) / mx[nz]).astype(np.float32)
    v = np.rint((iscale[:, None] * x).astype(np.float32)).astype(np.int32)
    qs[nz] = np.minimum(127, v[nz]).astype(np.int8)
    d[nz] = (np.float32(1.0) / iscale[nz]).astype(np.float32)
    bsums = qs.astype(np.int32).reshape(-1, 16, 16).sum(axis=2).astype(np.int16)
    return d, qs, bsums


def quantize_q8_0(x):
    """x [n, 32] -> (d as float16 [n], qs int8 [n,32]); d = amax/127, q = round-half-away(x * (1/d))."""
    x = x.astype(np.float32)
    amax = np.max(np.abs(x), axis=1)
    d = (amax / np.float32(127.0)).astype(np.float32)
    inv = np.where(d != 0, np.float32(1.0) / np.where(d != 0, d, 1), 0).astype(np.float32)
    t = (x * inv[:, None]).astype(np.float32)
    q = np.where(t >= 0, np.floor(t + np.float32(0.5)), np.ceil(t - np.float32(0.5))).astype(np.int32)
    return d.astype(np.float16), q.astype(np.int8)


def vec_dot_exact(qtype, wblocks, x):
    """float64 value of the CPU mat-vec semantics for ONE row: integer sub-block sums exact, scales in float64."""
    import llama_box_amd as L
    nb = wblocks.shape[0]
    if qtype == L.Q8_0:
        dq, q8 = quantize_q8_0(x.reshape(nb, 32))
        w = unpack_q(qtype, wblocks)
        isum = (w * q8.astype(np.int32)).sum(axis=1).astype(np.float64)
        return float(np.sum(isum * f16(wblocks[:, 0:2]).astype(np.float64) * dq.astype(np.float32).astype(np.float64)))
    d8, q8, bs = quantize_q8_K(x.reshape(nb, 256))
    w = unpack_q(qtype, wblocks)
    prod = w * q8.astype(np.int32)
    if qtype == L.Q6_K:
        sc = wblocks[:, 192:208].view(np.int8).astype(np.int64)
        isum = (prod.reshape(nb, 16, 16).sum(axis=2) * sc).sum(axis=1).astype(np.float64)
        return float(np.sum(isum * f16(wblocks[:, 208:210]).astype(np.float64) * d8.astype(np.float64)))
    sc, mn = scale_min_k4(wblocks[:, 4:16])
    isum = (prod.reshape(nb, 8, 32).sum(axis=2) * sc).sum(axis=1).astype(np.float64)
    msum = (bs.astype(np.int64).reshape(nb, 8, 2).sum(axis=2) * mn).sum(axis=1).astype(np.float64)
    d, dmin = f16(wblocks[:, 0:2]).astype(np.float64), f16(wblocks[:, 2:4]).astype(np.float64)
    return float(np.sum(d8.astype(np.float64) * (d * isum - dmin * msum)))


def main():
    import llama_box_amd as L
    import harness as T
    rng = np.random.default_rng(0x5EED)
    out = {}
    for name, qt in (("q8_0", L.Q8_0), ("q4_K", L.Q4_K), ("q5_K", L.Q5_K), ("q6_K", L.Q6_K)):
        K = 512
        nbk = K // L.TYPE_BLCK[qt]
        blocks = T.rand_blocks(qt, 6 * nbk, K, rng)
        # hand-made edge blocks: all-zero quants, max scales, alternating nibbles
        blocks[0, :] = 0
        blocks[1, :] = 0xFF
        if qt != L.Q8_0:
            blocks[1, 0:2] = np.array([0.001], np.float16).view(np.uint8)  # keep d finite
            if qt != L.Q6_K:
                blocks[1, 2:4] = np.array([0.0005], np.float16).view(np.uint8)
            else:
                blocks[1, 208:210] = np.array([0.001], np.float16).view(np.uint8)
        else:
            blocks[1, 0:2] = np.array([0.001], np.float16).view(np.uint8)
        blocks[2, L.TYPE_SIZE[qt] // 2:] = 0xA5
        x = (rng.standard_normal((6, K)) * rng.uniform(0.1, 4.0, (6, 1))).astype(np.float32)
        x[3, :256] = 0.0  # an all-zero activation block
        x[4, 7] = -x[4, 300:].max() if False else x[4, 7]
        out[name + "_blocks"] = blocks
        out[name + "_dequant"] = dequant(qt, blocks)
        out[name + "_x"] = x
        out[name + "_dot"] = np.array([vec_dot_exact(qt, blocks[r * nbk:(r + 1) * nbk], x[r]) for r in range(6)], np.float64)
    xk = (rng.standard_normal((8, 256)) * 3).astype(np.float32)
    xk[0, :] = 0
    xk[1] = np.clip(xk[1], -4.0, 4.0)
    xk[1, 10] = 5.0
    xk[1, 200] = -5.0  # tie in |x|: the first (positive) one must win
    xk[2, :] = np.float32(0.5)  # exact .5 multiples exercise round-half-even
    d, qs, bs = quantize_q8_K(xk)
    out["q8K_x"], out["q8K_d"], out["q8K_qs"], out["q8K_bsums"] = xk, d, qs, bs
    x0 = (rng.standard_normal((8, 32)) * 2).astype(np.float32)
    x0[0, :] = 0
    d0, q0 = quantize_q8_0(x0)
    out["q80_x"], out["q80_d"], out["q80_qs"] = x0, d0, q0
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "quant_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
