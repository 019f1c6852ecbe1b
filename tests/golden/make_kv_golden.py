"""Generates tests/golden/kv_golden.npz — golden vectors for the formats a KV cache may be kept in besides f16 / q8_0
(llama-box/engine_param.hpp:51-54: -ctk / -ctv f32, bf16, q4_0, q4_1, iq4_nl, q5_0, q5_1) and for the Q8_1 activation block.

As for quant_golden.npz the reference snapshot holds no vectors for these (parity unpinned): they come from an INDEPENDENT NumPy restatement
of the published formats — whole arrays of blocks at a time, float32 element operations (one IEEE rounding each, as the C evaluates them), written
from the format description (block layout, "d = max / -8", "q = min(15, trunc(x / d + 8.5))", ...) rather than from the C oracle's loops.  The C
oracle and the HIP kernels are both compared with them byte for byte.  Run:  python tests/golden/make_kv_golden.py   (deterministic; commit the .npz)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))

F = np.float32
KVALUES = np.array([-127, -104, -83, -65, -49, -35, -22, -10, 1, 13, 25, 38, 53, 69, 89, 113], np.int32)


def h2b(x):  # float32 [n] -> the two bytes of its f16 rounding [n, 2]
    with np.errstate(over="ignore"):  # (scales beyond 65504 round to infinity, as GGML_FP32_TO_FP16 does)
        return x.astype(np.float16).view(np.uint8).reshape(-1, 2)


def b2f(b):  # [n, 2] bytes of an f16 -> float32 [n]
    return np.ascontiguousarray(b).view(np.float16).astype(F).reshape(-1)


def signed_max(x):
    """the element of largest magnitude with its sign — the FIRST one on a tie (strict > while scanning)"""
    i = np.argmax(np.abs(x), axis=1)  # argmax returns the first maximum
    m = x[np.arange(x.shape[0]), i]
    return np.where(m == 0, F(0.0), m).astype(F)  # (a block of zeros of either sign: the scan never moves off its initial +0.0)


def inv(d):
    return np.where(d != 0, F(1.0) / np.where(d != 0, d, F(1.0)), F(0.0)).astype(F)


def pack_nibbles(lo, hi):
    return (lo.astype(np.uint8) | (hi.astype(np.uint8) << 4)).astype(np.uint8)


def fifth_bits(q):  # q [n, 32] with values < 32 -> the 4 qh bytes: bit j = element j's bit 4, bit j + 16 = element 16 + j's
    b = ((q.astype(np.uint32) >> 4) & 1)
    w = (b << np.arange(32, dtype=np.uint32)[None, :]).sum(axis=1).astype(np.uint32)
    return w.view(np.uint8).reshape(-1, 4)


def quantize(name, x):
    """x float32 [n, 32] -> blocks uint8 [n, block bytes]"""
    x = x.astype(F)
    n = x.shape[0]
    if name == "q4_0":
        d = (signed_max(x) / F(-8.0)).astype(F)
        t = (x * inv(d)[:, None]).astype(F)
        q = np.minimum(15, np.trunc((t + F(8.5)).astype(F)).astype(np.int32))
        return np.concatenate([h2b(d), pack_nibbles(q[:, :16], q[:, 16:])], axis=1)
    if name == "q5_0":
        d = (signed_max(x) / F(-16.0)).astype(F)
        t = (x * inv(d)[:, None]).astype(F)
        q = np.minimum(31, np.trunc((t + F(16.5)).astype(F)).astype(np.int32))
        return np.concatenate([h2b(d), fifth_bits(q), pack_nibbles(q[:, :16] & 15, q[:, 16:] & 15)], axis=1)
    if name in ("q4_1", "q5_1"):
        levels = 15 if name == "q4_1" else 31
        mn, mx = x[np.arange(n), np.argmin(x, axis=1)], x[np.arange(n), np.argmax(x, axis=1)]  # (the FIRST element that attains the extreme: its sign if that is a zero)
        d = ((mx - mn).astype(F) / F(levels)).astype(F)
        t = ((x - mn[:, None]).astype(F) * inv(d)[:, None]).astype(F)
        q = np.trunc((t + F(0.5)).astype(F)).astype(np.int32)
        if name == "q4_1":
            q = np.minimum(15, q)
            return np.concatenate([h2b(d), h2b(mn), pack_nibbles(q[:, :16], q[:, 16:])], axis=1)
        return np.concatenate([h2b(d), h2b(mn), fifth_bits(q), pack_nibbles(q[:, :16] & 15, q[:, 16:] & 15)], axis=1)
    if name == "q8_1":
        amax = np.abs(x).max(axis=1)
        d = (amax / F(127.0)).astype(F)
        t = (x * inv(d)[:, None]).astype(F)
        q = np.where(t >= 0, np.floor(t + F(0.5)), np.ceil(t - F(0.5))).astype(np.int32)  # roundf: halves away from zero
        s = (q.sum(axis=1).astype(F) * d).astype(F)
        return np.concatenate([h2b(d), h2b(s), q.astype(np.int8).view(np.uint8)], axis=1)
    if name == "iq4_nl":
        mx = signed_max(x)
        live = np.abs(mx) >= F(1e-15)
        d0 = (mx / F(-127.0)).astype(F)
        idv = np.where(live, F(1.0) / np.where(live, d0, F(1.0)), F(0.0)).astype(F)
        al = (idv[:, None] * x).astype(F)
        lv = nearest_level(al)
        qv = KVALUES[lv].astype(F)
        w = (x * x).astype(F)
        sumqx = np.zeros(n, F)
        sumq2 = np.zeros(n, F)
        for j in range(32):  # sequential f32 accumulation, element by element
            sumqx = (sumqx + ((w[:, j] * qv[:, j]).astype(F) * x[:, j]).astype(F)).astype(F)
            sumq2 = (sumq2 + ((w[:, j] * qv[:, j]).astype(F) * qv[:, j]).astype(F)).astype(F)
        with np.errstate(invalid="ignore", divide="ignore"):
            d = np.where(live, sumqx / np.where(live, sumq2, F(1.0)), F(0.0)).astype(F)
        lv = np.where(live[:, None], lv, 0)
        return np.concatenate([h2b(d), pack_nibbles(lv[:, :16], lv[:, 16:])], axis=1)
    raise ValueError(name)


def nearest_level(a):
    """index of the nearest of the 16 levels; exactly between two levels: the upper one; beyond the ends: the end"""
    vals = KVALUES.astype(F)
    hi = np.clip(np.searchsorted(vals, a, side="right"), 1, 15)  # first level above a
    lo = hi - 1
    pick_lo = (a - vals[lo]).astype(F) < (vals[hi] - a).astype(F)
    out = np.where(pick_lo, lo, hi)
    out = np.where(a <= vals[0], 0, out)
    out = np.where(a >= vals[15], 15, out)
    return out.astype(np.int32)


def levels(name, b):
    """the integer level of every element [n, 32]: q4_0 q - 8, q5_0 q - 16, q4_1 / q5_1 q, iq4_nl the table value"""
    o = {"q4_0": 2, "iq4_nl": 2, "q4_1": 4, "q5_0": 6, "q5_1": 8}[name]
    qs = b[:, o:o + 16]
    q = np.concatenate([qs & 15, qs >> 4], axis=1).astype(np.int32)
    if name in ("q5_0", "q5_1"):
        qh = np.ascontiguousarray(b[:, o - 4:o]).view(np.uint32).reshape(-1)
        q = q | ((((qh[:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1).astype(np.int32)) << 4)
    if name == "q4_0":
        return q - 8
    if name == "q5_0":
        return q - 16
    if name == "iq4_nl":
        return KVALUES[q]
    return q


def dequantize(name, b):
    """blocks uint8 [n, bytes] -> float32 [n, 32]: level * d (+ m for the offset formats), one rounding per operation"""
    d = b2f(b[:, 0:2])
    q = levels(name, b).astype(F)
    if name in ("q4_1", "q5_1"):
        return ((q * d[:, None]).astype(F) + b2f(b[:, 2:4])[:, None]).astype(F)
    return (q * d[:, None]).astype(F) if name != "iq4_nl" else (d[:, None] * q).astype(F)


def to_bf16(x):
    u = x.astype(F).view(np.uint32)
    nan = (u & 0x7FFFFFFF) > 0x7F800000
    r = ((u.astype(np.uint64) + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    return np.where(nan, ((u >> 16) | 64).astype(np.uint16), r)


def dot_exact(name, kb, q):
    """float64 value of the K.q dot of ONE cache row as ggml-cpu forms it: K blocks against the query quantised to Q8_0 (q4_0, q5_0, iq4_nl) or
    Q8_1 (q4_1, q5_1: + m * s per block); integer block sums exact, the scale products in float64"""
    nb = kb.shape[0]
    act = quantize("q8_1", q.reshape(nb, 32))
    dq, sq, qi = b2f(act[:, 0:2]).astype(np.float64), b2f(act[:, 2:4]).astype(np.float64), act[:, 4:].view(np.int8).astype(np.int64)
    d = b2f(kb[:, 0:2]).astype(np.float64)
    isum = (levels(name, kb).astype(np.int64) * qi).sum(axis=1)
    if name in ("q4_1", "q5_1"):
        return float(np.sum(isum * d * dq + b2f(kb[:, 2:4]).astype(np.float64) * sq))
    return float(np.sum(isum * d * dq))


def main():
    rng = np.random.default_rng(0xCAC4E)
    out = {}
    x = (rng.standard_normal((64, 32)) * rng.uniform(0.05, 6.0, (64, 1))).astype(F)
    x[0, :] = 0  # an all-zero block: d = 0, id = 0
    x[1, :] = F(0.75)  # a constant block: the offset formats get d = 0
    x[2, 5], x[2, 20] = F(3.0), F(-3.0)  # a tie in |x|: the first one (positive) sets the sign of the scale
    x[2] = np.clip(x[2], -3.0, 3.0)
    x[3, 9], x[3, 1] = F(-4.0), F(4.0)  # ... and with the negative one first
    x[3] = np.clip(x[3], -4.0, 4.0)
    x[4] = (np.arange(32, dtype=F) - F(15.5)) * F(0.25)  # evenly spaced: values that land exactly on .5 before truncation
    x[5] = np.where(np.arange(32) % 2 == 0, F(1e-20), F(-1e-20))  # below iq4_nl's group epsilon, above zero
    out["x"] = x
    for name in ("q4_0", "q4_1", "q5_0", "q5_1", "iq4_nl", "q8_1"):
        out[name + "_blocks"] = quantize(name, x)
    bf_in = np.concatenate([x.reshape(-1), np.array([np.inf, -np.inf, np.nan, 1.0 + 2.0 ** -8, 1.0 + 3 * 2.0 ** -9, 65504.0, 1e-40, -0.0], F)])
    out["bf16_x"], out["bf16_bits"] = bf_in, to_bf16(bf_in)
    # arbitrary block bytes (finite scales) through the dequantisers, and the K.q dot of whole 128-value cache rows
    for name, size in (("q4_0", 18), ("q4_1", 20), ("q5_0", 22), ("q5_1", 24), ("iq4_nl", 18)):
        b = rng.integers(0, 256, (48, size), dtype=np.uint8)
        b[:, 0:2] = rng.uniform(0.001, 0.05, 48).astype(np.float16).view(np.uint8).reshape(48, 2)
        if name in ("q4_1", "q5_1"):
            b[:, 2:4] = rng.uniform(-1.0, 1.0, 48).astype(np.float16).view(np.uint8).reshape(48, 2)
        out[name + "_rand_blocks"] = b
        out[name + "_rand_dequant"] = dequantize(name, b)
        qv = (rng.standard_normal((12, 128)) * 2).astype(F)
        out[name + "_q"] = qv
        out[name + "_dot"] = np.array([dot_exact(name, b[4 * r:4 * r + 4], qv[r]) for r in range(12)], np.float64)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kv_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
