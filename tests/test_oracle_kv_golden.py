"""CPU: the oracle's restatement of the formats a KV cache may be kept in besides f16 / q8_0 (-ctk / -ctv q4_0, q4_1, q5_0, q5_1, iq4_nl, bf16:
llama-box/engine_param.hpp:51-54) against tests/golden/kv_golden.npz (the independent NumPy restatement of tests/golden/make_kv_golden.py) and
against hand-computed blocks.  As for the K-quants: the reference holds no vectors, so parity stays unpinned; this agreement is the pin."""
import ctypes as C
import os

import numpy as np
import pytest

import harness as T
import llama_box_amd as L

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "kv_golden.npz"))
BLOCK_TYPES = [("q4_0", L.Q4_0), ("q4_1", L.Q4_1), ("q5_0", L.Q5_0), ("q5_1", L.Q5_1), ("iq4_nl", L.IQ4_NL)]


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _quantize(qt, x):
    x = np.ascontiguousarray(x, np.float32)
    out = np.zeros((x.size // 32, L.TYPE_SIZE[qt]), np.uint8)
    assert T.oracle().oracle_quantize_row(qt, _ptr(x), _ptr(out), x.size) == 1
    return out


@pytest.mark.parametrize("name,qt", BLOCK_TYPES + [("q8_1", L.Q8_1)])
def test_quantize_matches_golden_byte_for_byte(built, name, qt):
    got = _quantize(qt, GOLD["x"])
    ref = GOLD[name + "_blocks"]
    bad = np.nonzero((got != ref).any(axis=1))[0]
    assert bad.size == 0, f"{name}: blocks {bad[:8]} differ from the NumPy golden, e.g. {got[bad[0]]} vs {ref[bad[0]]}"


@pytest.mark.parametrize("name,qt", BLOCK_TYPES)
def test_dequantize_matches_golden(built, name, qt):
    b = np.ascontiguousarray(GOLD[name + "_rand_blocks"])
    y = np.empty(b.shape[0] * 32, np.float32)
    T.oracle().oracle_dequantize_row(qt, _ptr(b), _ptr(y), y.size)
    assert np.array_equal(y.view(np.uint32), GOLD[name + "_rand_dequant"].reshape(-1).view(np.uint32))
    # ... and a quantise -> dequantise round trip stays within one step (half a step, except at the end the asymmetric level range clips: +max on -8 .. 7)
    x = GOLD["x"]
    q = _quantize(qt, x)
    back = np.empty(x.size, np.float32)
    T.oracle().oracle_dequantize_row(qt, _ptr(q), _ptr(back), x.size)
    err = np.abs(back.reshape(x.shape) - x).max(axis=1)
    span = {"q4_0": np.abs(x).max(axis=1) / 8, "q5_0": np.abs(x).max(axis=1) / 16, "q4_1": (x.max(axis=1) - x.min(axis=1)) / 15, "q5_1": (x.max(axis=1) - x.min(axis=1)) / 31,
            "iq4_nl": np.abs(x).max(axis=1) * 24 / 127}[name]  # (iq4_nl: its widest gap between levels is 24 of 127)
    assert np.all(err <= 1.01 * span + 1e-3 * np.abs(x).max(axis=1) + 1e-7), f"{name}: round trip error {err.max()}"


@pytest.mark.parametrize("name,qt", BLOCK_TYPES)
def test_kq_dot_matches_golden(built, name, qt):
    """one attention logit K.q of a 128-value cache row: integer block sums are exact, so the C value may differ from the float64 golden by the f32
    rounding of four block terms only"""
    b = np.ascontiguousarray(GOLD[name + "_rand_blocks"])
    qv = np.ascontiguousarray(GOLD[name + "_q"])
    for r in range(qv.shape[0]):
        row = np.ascontiguousarray(b[4 * r:4 * r + 4])
        got = T.oracle().oracle_kq_dot(qt, 128, _ptr(row), _ptr(qv[r]))
        ref = GOLD[name + "_dot"][r]
        assert abs(got - ref) <= 4e-6 * max(1.0, abs(ref)), (name, r, got, ref)


def test_bf16_rounding_matches_golden(built):
    o = T.oracle()
    x, bits = GOLD["bf16_x"], GOLD["bf16_bits"]
    got = np.array([o.oracle_fp32_to_bf16(float(v)) for v in x], np.uint16)
    fin = ~np.isnan(x)
    assert np.array_equal(got[fin], bits[fin])
    assert all((int(g) & 0x7F80) == 0x7F80 and (int(g) & 0x7F) != 0 for g in got[~fin])  # NaN stays NaN (quiet bit forced)
    back = np.array([o.oracle_bf16_to_fp32(int(b)) for b in bits[fin]], np.float32)
    assert np.array_equal(back.view(np.uint32), (bits[fin].astype(np.uint32) << 16))


def test_known_answer_blocks(built):
    """hand-computed, no generator involved"""
    o = T.oracle()
    # Q4_0: x = -8 .. 7 twice over -> max magnitude -8 (first), d = -8 / -8 = 1, q = x + 8 = 0 .. 15; element j in the low nibble of byte j, j + 16 in the high one
    x = np.concatenate([np.arange(-8, 8), np.arange(-8, 8)]).astype(np.float32)
    b = _quantize(L.Q4_0, x)[0]
    assert np.array_equal(b[0:2], np.array([1.0], np.float16).view(np.uint8))
    assert np.array_equal(b[2:], (np.arange(16) | (np.arange(16) << 4)).astype(np.uint8))
    y = np.empty(32, np.float32)
    o.oracle_dequantize_row(L.Q4_0, _ptr(b), _ptr(y), 32)
    assert np.array_equal(y, x)
    # Q4_0 with the POSITIVE extreme: max = +7.5 -> d = -0.9375; 7.5 / d = -8 -> q 0 (the extreme sits on level -8), 0 -> q 8
    x = np.zeros(32, np.float32)
    x[3] = 7.5
    b = _quantize(L.Q4_0, x)[0]
    assert np.array_equal(b[0:2], np.array([-0.9375], np.float16).view(np.uint8))
    assert (b[2 + 3] & 15) == 0 and (b[2] & 15) == 8 and (b[2] >> 4) == 8
    # Q4_1: x = 0 .. 15 and 0 .. 15 halves -> min 0, d = 1, q = x
    x = np.concatenate([np.arange(16), np.arange(16)]).astype(np.float32) * 0.5 + 2.0
    b = _quantize(L.Q4_1, x)[0]
    assert np.array_equal(b[0:2], np.array([0.5], np.float16).view(np.uint8)) and np.array_equal(b[2:4], np.array([2.0], np.float16).view(np.uint8))
    assert np.array_equal(b[4:], (np.arange(16) | (np.arange(16) << 4)).astype(np.uint8))
    # Q5_0: x = -16 .. 15 -> d = 1, q = 0 .. 31: low nibbles j & 15 | (j & 15) << 4, fifth bits: elements 16 .. 31 all set -> qh = 0xFFFF0000
    x = np.arange(-16, 16).astype(np.float32)
    b = _quantize(L.Q5_0, x)[0]
    assert np.array_equal(b[0:2], np.array([1.0], np.float16).view(np.uint8))
    assert np.array_equal(b[2:6], np.array([0xFFFF0000], np.uint32).view(np.uint8))
    assert np.array_equal(b[6:], (np.arange(16) | (np.arange(16) << 4)).astype(np.uint8))
    o.oracle_dequantize_row(L.Q5_0, _ptr(b), _ptr(y), 32)
    assert np.array_equal(y, x)
    # Q5_1: x = 0 .. 31 -> min 0, d = 1, q = x
    x = np.arange(32).astype(np.float32)
    b = _quantize(L.Q5_1, x)[0]
    assert np.array_equal(b[0:4], np.array([1.0, 0.0], np.float16).view(np.uint8)) and np.array_equal(b[4:8], np.array([0xFFFF0000], np.uint32).view(np.uint8))
    o.oracle_dequantize_row(L.Q5_1, _ptr(b), _ptr(y), 32)
    assert np.array_equal(y, x)
    # IQ4_NL: the sixteen levels themselves (twice) -> max magnitude -127 first: d0 = -127 / -127 = 1, every value is its own level, least-squares d = 1
    lv = np.array([-127, -104, -83, -65, -49, -35, -22, -10, 1, 13, 25, 38, 53, 69, 89, 113], np.float32)
    x = np.concatenate([lv, lv])
    b = _quantize(L.IQ4_NL, x)[0]
    assert np.array_equal(b[0:2], np.array([1.0], np.float16).view(np.uint8))
    assert np.array_equal(b[2:], (np.arange(16) | (np.arange(16) << 4)).astype(np.uint8))
    o.oracle_dequantize_row(L.IQ4_NL, _ptr(b), _ptr(y), 32)
    assert np.array_equal(y, x)
    # Q8_1: s = d * sum(q): x = 127 at one place, -127 at another, 1.0 elsewhere -> d = 1, sum = 30
    x = np.ones(32, np.float32)
    x[0], x[17] = 127.0, -127.0
    b = _quantize(L.Q8_1, x)[0]
    assert np.array_equal(b[0:4], np.array([1.0, 30.0], np.float16).view(np.uint8)) and b[4:].view(np.int8)[0] == 127 and b[4:].view(np.int8)[17] == -127


@pytest.mark.parametrize("name,qt", BLOCK_TYPES + [("q8_1", L.Q8_1)])
def test_quantize_matches_the_numpy_twin_on_random_blocks(built, name, qt):
    """beyond the committed vectors: 4 000 seeded blocks over twelve decades of scale, with planted ties, zeros and exact half-way values, through the C oracle and
    through the NumPy restatement (tests/golden/make_kv_golden.py) — byte for byte"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_kv_golden", os.path.join(os.path.dirname(__file__), "golden", "make_kv_golden.py"))
    twin = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(twin)
    rng = np.random.default_rng(qt * 1000 + 7)
    x = (rng.standard_normal((4000, 32)) * 10.0 ** rng.uniform(-6, 6, (4000, 1))).astype(np.float32)
    x[::17, :] = np.round(x[::17, :] * 4) / 4          # coarse grids: many equal magnitudes and values that land on a .5 before the truncation
    x[::29, rng.integers(0, 32)] = 0.0
    k = rng.integers(0, 32, 4000)
    x[np.arange(0, 4000, 13), k[::13][: len(range(0, 4000, 13))]] *= -1.0
    x[5::41] = np.abs(x[5::41])                         # all-positive blocks (the offset formats' minimum is then > 0)
    x[7::53, 1] = -x[7::53, 0]                          # an exact tie in magnitude between the first two elements
    got = _quantize(qt, x)
    ref = twin.quantize(name, x)
    bad = np.nonzero((got != ref).any(axis=1))[0]
    assert bad.size == 0, f"{name}: {bad.size} of 4000 blocks differ, first {bad[0]}: oracle {got[bad[0]].tolist()} twin {ref[bad[0]].tolist()} x {x[bad[0]].tolist()}"
