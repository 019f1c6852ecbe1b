"""CPU: the C oracle against the committed golden vectors (tests/golden/quant_golden.npz, made by the independent
NumPy restatement in tests/golden/make_golden.py) and against hand-computed known answers.  PARITY UNPINNED by the
reference itself (it has no tests); this three-way agreement is the pin (oracle/oracle.h)."""
import ctypes as C
import os

import numpy as np
import pytest

import harness as T
import llama_box_amd as L

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "quant_golden.npz"))
TYPES = [("q8_0", L.Q8_0), ("q4_K", L.Q4_K), ("q5_K", L.Q5_K), ("q6_K", L.Q6_K)]


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def test_fp16_conversion_exhaustive(built):
    o = T.oracle()
    allh = np.arange(65536, dtype=np.uint16)
    ref = allh.view(np.float16).astype(np.float32)
    got = np.array([o.oracle_fp16_to_fp32(int(h)) for h in allh], dtype=np.float32)
    assert np.array_equal(got.view(np.uint32)[~np.isnan(ref)], ref.view(np.uint32)[~np.isnan(ref)])
    rng = np.random.default_rng(1)
    f = np.concatenate([rng.standard_normal(20000).astype(np.float32) * 10.0 ** rng.integers(-9, 6, 20000),
                        np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e-8, 5.96e-8, 2.98e-8, 2.9802322e-8, 6.1e-5, np.inf, -np.inf], np.float32)]).astype(np.float32)
    with np.errstate(over="ignore"):
        ref16 = f.astype(np.float16).view(np.uint16)
    got16 = np.array([o.oracle_fp32_to_fp16(float(v)) for v in f], dtype=np.uint16)
    assert np.array_equal(got16, ref16)


@pytest.mark.parametrize("name,qt", TYPES)
def test_dequantize_matches_golden(built, name, qt):
    o = T.oracle()
    blocks = np.ascontiguousarray(GOLD[name + "_blocks"])
    n = blocks.shape[0] * L.TYPE_BLCK[qt]
    y = np.empty(n, np.float32)
    o.oracle_dequantize_row(qt, _ptr(blocks), _ptr(y), n)
    ref = GOLD[name + "_dequant"].reshape(-1)
    assert np.array_equal(y.view(np.uint32), ref.view(np.uint32)), f"{name}: dequant not bit-identical to the NumPy golden"


def test_dequantize_known_answers(built):
    """Hand-computed single blocks (no generator involved)."""
    o = T.oracle()
    # Q8_0: d = 0.5, qs = 0..31 -> y = 0.5*i
    b = np.zeros(34, np.uint8)
    b[0:2] = np.array([0.5], np.float16).view(np.uint8)
    b[2:] = np.arange(32, dtype=np.uint8)
    y = np.empty(32, np.float32)
    o.oracle_dequantize_row(L.Q8_0, _ptr(b), _ptr(y), 32)
    assert np.array_equal(y, 0.5 * np.arange(32, dtype=np.float32))
    # Q4_K: d = 1, dmin = 2, scales[0..3] -> sc 1,2,3,4 / mins 5,6,7,8 for sub-blocks 0-3, sub-blocks 4-7 sc=0x11,m=0x21
    b = np.zeros(144, np.uint8)
    b[0:2] = np.array([1.0], np.float16).view(np.uint8)
    b[2:4] = np.array([2.0], np.float16).view(np.uint8)
    b[4:8] = [1 | (1 << 6), 2 | (1 << 6), 3 | (1 << 6), 4 | (1 << 6)]      # low 6 bits sc0..3; top 2 bits -> high bits of sc4..7 (=1 -> +16)
    b[8:12] = [5 | (2 << 6), 6 | (2 << 6), 7 | (2 << 6), 8 | (2 << 6)]     # low 6 bits m0..3; top 2 bits -> high bits of m4..7 (=2 -> +32)
    b[12:16] = [0x11, 0x11, 0x11, 0x11]                                    # low nibble sc4..7 low = 1, high nibble m4..7 low = 1
    b[16:144] = 0x21                                                       # every byte: low nibble 1, high nibble 2
    y = np.empty(256, np.float32)
    o.oracle_dequantize_row(L.Q4_K, _ptr(b), _ptr(y), 256)
    exp = np.empty(256, np.float32)
    sc = [1, 2, 3, 4, 17, 17, 17, 17]
    mn = [5, 6, 7, 8, 33, 33, 33, 33]
    for s in range(8):
        q = 1 if s % 2 == 0 else 2
        exp[32 * s:32 * s + 32] = 1.0 * sc[s] * q - 2.0 * mn[s]
    assert np.array_equal(y, exp)
    # Q6_K: d = 1, scales = 1..16, ql = 0x10 (low nibble 0 / high nibble 1), qh = 0b11_10_01_00
    b = np.zeros(210, np.uint8)
    b[0:128] = 0x10
    b[128:192] = 0b11100100
    b[192:208] = np.arange(1, 17, dtype=np.uint8)
    b[208:210] = np.array([1.0], np.float16).view(np.uint8)
    o.oracle_dequantize_row(L.Q6_K, _ptr(b), _ptr(y), 256)
    exp = np.empty(256, np.float32)
    for h in range(2):
        for k in range(4):
            nib = 0 if k < 2 else 1
            q = (nib | (k << 4)) - 32
            for l in range(32):
                exp[128 * h + 32 * k + l] = float(8 * h + l // 16 + 2 * k + 1) * q
    assert np.array_equal(y, exp)


def test_quantize_q8_K_matches_golden(built):
    o = T.oracle()
    x = np.ascontiguousarray(GOLD["q8K_x"])
    out = np.zeros((x.shape[0], 292), np.uint8)
    o.oracle_quantize_row_q8_K(_ptr(x), _ptr(out), x.size)
    d = out[:, 0:4].copy().view(np.float32).reshape(-1)
    qs = out[:, 4:260].view(np.int8)
    bs = out[:, 260:292].copy().view(np.int16).reshape(-1, 16)
    assert np.array_equal(d.view(np.uint32), GOLD["q8K_d"].view(np.uint32))
    assert np.array_equal(qs, GOLD["q8K_qs"])
    assert np.array_equal(bs, GOLD["q8K_bsums"])
    assert qs[1, 10] == -127 and qs[1, 200] == 127  # tie: the first (positive) maximum defines the sign


def test_quantize_q8_0_matches_golden(built):
    o = T.oracle()
    x = np.ascontiguousarray(GOLD["q80_x"])
    out = np.zeros((x.shape[0], 34), np.uint8)
    o.oracle_quantize_row_q8_0(_ptr(x), _ptr(out), x.size)
    assert np.array_equal(out[:, 0:2].copy().view(np.float16).reshape(-1).view(np.uint16), GOLD["q80_d"].view(np.uint16))
    assert np.array_equal(out[:, 2:].view(np.int8), GOLD["q80_qs"])


@pytest.mark.parametrize("name,qt", TYPES)
def test_vec_dot_matches_golden(built, name, qt):
    o = T.oracle()
    blocks = np.ascontiguousarray(GOLD[name + "_blocks"])
    x = np.ascontiguousarray(GOLD[name + "_x"])
    K = x.shape[1]
    nbk = K // L.TYPE_BLCK[qt]
    fn = getattr(o, "oracle_vec_dot_%s_%s" % (name, "q8_0" if qt == L.Q8_0 else "q8_K"))
    for r in range(x.shape[0]):
        if qt == L.Q8_0:
            act = np.zeros((K // 32, 34), np.uint8)
            o.oracle_quantize_row_q8_0(_ptr(x[r]), _ptr(act), K)
        else:
            act = np.zeros((K // 256, 292), np.uint8)
            o.oracle_quantize_row_q8_K(_ptr(x[r]), _ptr(act), K)
        w = np.ascontiguousarray(blocks[r * nbk:(r + 1) * nbk])
        got = fn(K, _ptr(w), _ptr(act))
        ref = float(GOLD[name + "_dot"][r])
        scale = max(1e-6, abs(ref), float(np.abs(GOLD[name + "_dequant"][r * nbk:(r + 1) * nbk]).max()) * float(np.abs(x[r]).max()))
        assert abs(got - ref) <= 2e-5 * scale, (name, r, got, ref)
        # property: the integer path approximates the real dot product within the Q8 rounding bound
        wd = GOLD[name + "_dequant"][r * nbk:(r + 1) * nbk].reshape(-1).astype(np.float64)
        exact = float(wd @ x[r].astype(np.float64))
        bound = float(np.sum(np.abs(wd).reshape(-1, L.TYPE_BLCK[qt]) * (np.abs(x[r]).reshape(-1, L.TYPE_BLCK[qt]).max(axis=1, keepdims=True) / 127.0)))
        assert abs(got - exact) <= 0.51 * bound + 1e-5 * scale


def test_mul_mat_graph_matches_vec_dot(built, H):
    """MUL_MAT through the oracle's graph interpreter == per-row vec_dot (wiring of quantise + dot + broadcasting)."""
    rng = np.random.default_rng(5)
    K, N, M = 512, 24, 3
    for qt in (L.Q4_K, L.Q6_K, L.Q8_0, L.F16):
        w = T.rand_weight(qt, K, N, rng)
        x = rng.standard_normal((M, K)).astype(np.float32)

        def build(g):
            return H.ggml_mul_mat(g.ctx, g.new(qt, [K, N], w), g.new(L.F32, [K, M], x))

        y = T.run_case(build, "oracle")[0].reshape(M, N)
        if qt == L.F16:
            ref = (x.astype(np.float16).astype(np.float64) @ w.astype(np.float64).T)
        else:
            import golden.make_golden as GG  # noqa
            ref = np.array([[GG.vec_dot_exact(qt, w[n].reshape(-1, L.TYPE_SIZE[qt]), x[m]) for n in range(N)] for m in range(M)])
        assert T.nmse(y, ref) < 1e-10


# ------------------------------------------------------------------------------------------------ property tests (hypothesis)
# SURVEY.md §8c asks for them because the reference pins nothing: whatever bytes a block holds, the C oracle and the independent
# NumPy restatement must agree bit for bit, scaling d by a power of two must scale the result exactly, and the integer dot
# product must stay within the Q8 rounding bound of the real one.
from hypothesis import given, settings, strategies as st  # noqa: E402

import golden.make_golden as GG  # noqa: E402


def _arbitrary_blocks(qt, data, n_blocks):
    """Blocks of ARBITRARY bytes with finite, modest f16 scales (every bit pattern of quants / packed scales is legal)."""
    bs = L.TYPE_SIZE[qt]
    raw = np.frombuffer(data.draw(st.binary(min_size=bs * n_blocks, max_size=bs * n_blocks)), np.uint8).reshape(n_blocks, bs).copy()
    scales = np.array(data.draw(st.lists(st.floats(-4.0, 4.0, width=16), min_size=2 * n_blocks, max_size=2 * n_blocks)), np.float16)
    if qt == L.Q8_0:
        raw[:, 0:2] = scales[:n_blocks].view(np.uint8).reshape(n_blocks, 2)
    elif qt in (L.Q4_K, L.Q5_K):
        raw[:, 0:2] = scales[:n_blocks].view(np.uint8).reshape(n_blocks, 2)
        raw[:, 2:4] = scales[n_blocks:].view(np.uint8).reshape(n_blocks, 2)
    else:
        raw[:, 208:210] = scales[:n_blocks].view(np.uint8).reshape(n_blocks, 2)
    return raw


@pytest.mark.parametrize("name,qt", TYPES)
@settings(max_examples=25, deadline=None)
@given(data=st.data())
def test_dequantize_arbitrary_bytes_matches_numpy(built, name, qt, data):
    o = T.oracle()
    blocks = _arbitrary_blocks(qt, data, 3)
    n = blocks.shape[0] * L.TYPE_BLCK[qt]
    y = np.empty(n, np.float32)
    o.oracle_dequantize_row(qt, _ptr(blocks), _ptr(y), n)
    ref = GG.dequant(qt, blocks).reshape(-1).astype(np.float32)
    assert np.array_equal(y.view(np.uint32), ref.view(np.uint32))
    # doubling every scale doubles every value exactly (no rounding is involved in a power-of-two factor)
    b2 = blocks.copy()
    cols = [(0, 2)] if qt == L.Q8_0 else ([(0, 2), (2, 4)] if qt in (L.Q4_K, L.Q5_K) else [(208, 210)])
    for lo, hi in cols:
        d = b2[:, lo:hi].copy().view(np.float16).astype(np.float32) * 2.0
        b2[:, lo:hi] = d.astype(np.float16).view(np.uint8).reshape(-1, 2)
    y2 = np.empty(n, np.float32)
    o.oracle_dequantize_row(qt, _ptr(b2), _ptr(y2), n)
    assert np.array_equal(y2, 2.0 * y)


@settings(max_examples=25, deadline=None)
@given(x=st.lists(st.floats(-1e4, 1e4, width=32), min_size=256, max_size=256), scale=st.sampled_from([1e-6, 1e-3, 1.0, 37.5]))
def test_quantize_q8_K_properties(built, x, scale):
    """quantize_row_q8_K: C oracle == NumPy restatement bit for bit; |x - d*q| <= |d|/2 (+ the clamp at 127); bsums are the sums."""
    o = T.oracle()
    xv = (np.array(x, np.float32) * np.float32(scale)).astype(np.float32)
    out = np.zeros((1, 292), np.uint8)
    o.oracle_quantize_row_q8_K(_ptr(xv), _ptr(out), 256)
    d = out[0, 0:4].copy().view(np.float32)[0]
    qs = out[0, 4:260].view(np.int8).astype(np.int32)
    bs = out[0, 260:292].copy().view(np.int16).astype(np.int32)
    gd, gq, gb = GG.quantize_q8_K(xv.reshape(1, 256))
    assert np.float32(gd.reshape(-1)[0]).view(np.uint32) == np.float32(d).view(np.uint32)
    assert np.array_equal(gq.reshape(-1).astype(np.int32), qs) and np.array_equal(gb.reshape(-1).astype(np.int32), bs)
    assert np.array_equal(qs.reshape(16, 16).sum(axis=1), bs)
    if d != 0.0:
        err = np.abs(xv.astype(np.float64) - float(d) * qs)
        assert np.all(err <= abs(float(d)) * (0.5 + 1e-6) + abs(float(d)) * (qs == 127))  # (a value rounding to 128 is clamped to 127)


def test_fast_block_dots_compute_the_same_integers():
    """bench.py's cpu_baseline leg times an AVX2 restatement of ggml-cpu's x86 block dots (oracle_set_fast).  It is never the
    checker, but it must be the same computation: with unit scales every float involved is an exact integer, so the result must
    EQUAL the generic routine's; with real scales it may differ by f32 accumulation order only."""
    import ctypes as C

    lib = T.oracle()
    lib.oracle_fast_vec_dot.restype = C.c_float
    lib.oracle_fast_vec_dot.argtypes = [C.c_int, C.c_int64, C.c_void_p, C.c_void_p]
    lib.oracle_set_fast.restype = C.c_int
    if not lib.oracle_set_fast(1):
        pytest.skip("oracle built without AVX2 + FMA")
    lib.oracle_set_fast(0)
    rng = np.random.default_rng(11)
    one16 = np.array([1.0], np.float16).view(np.uint8)
    for qt, name in ((L.Q4_K, "q4_K_q8_K"), (L.Q5_K, "q5_K_q8_K"), (L.Q6_K, "q6_K_q8_K"), (L.Q8_0, "q8_0_q8_0")):
        gen = getattr(lib, "oracle_vec_dot_" + name)
        for K in (256, 4096, 14336):
            for unit in (True, False):
                w = T.rand_weight(qt, K, 1, rng)[0].copy()
                x = (rng.standard_normal(K) * 2).astype(np.float32)
                if qt == L.Q8_0:
                    y = np.zeros(K // 32 * 34, np.uint8)
                    lib.oracle_quantize_row_q8_0(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), K)
                else:
                    y = np.zeros(K // 256 * 292, np.uint8)
                    lib.oracle_quantize_row_q8_K(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), K)
                if unit:
                    wb = w.reshape(-1, L.TYPE_SIZE[qt])
                    if qt in (L.Q4_K, L.Q5_K):
                        wb[:, 0:2] = one16
                        wb[:, 2:4] = one16
                    elif qt == L.Q6_K:
                        wb[:, 208:210] = one16
                    else:
                        wb[:, 0:2] = one16
                        y.reshape(-1, 34)[:, 0:2] = one16
                    if qt != L.Q8_0:
                        y.reshape(-1, 292)[:, 0:4] = np.array([1.0], np.float32).view(np.uint8)
                a = gen(K, w.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p))
                b = lib.oracle_fast_vec_dot(qt, K, w.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p))
                if unit and abs(a) < 2 ** 24:
                    assert a == b, (name, K, a, b)
                else:
                    assert abs(a - b) <= 2e-6 * max(1.0, abs(a)) * np.sqrt(K / 256), (name, K, a, b)
