"""GPU parity: a KV cache kept in one of the OTHER types llama-box lets the user pick (-ctk / -ctv, llama-box/engine_param.hpp:51-54: f32, bf16, q4_0, q4_1,
iq4_nl, q5_0, q5_1; f16 and q8_0 are the fast paths tested elsewhere).  csrc/kv_types.hip keeps such a cache on the device:

  * SET_ROWS f32 -> type ............ the type's from_float per block of 32: BYTE-exact against the oracle (integer / byte work)
  * CPY type <-> f32 (K-shift) ...... cast bit-exact, the copy back byte-exact
  * FLASH_ATTN_EXT .................. where K and V share one of q4_0, q4_1, q5_0, q5_1, iq4_nl (head_dim 128, up to 32 query tokens) the decode kernel reads the cache IN
                                     PLACE with ggml-cpu's arithmetic (round 6): the query row quantised to Q8_0 / Q8_1, integer block dots with K's levels, one f32 term
                                     per block in the reference's operation order, V de-quantised to f32 — gate NMSE <= 1e-6 against the oracle, as for q8_0.
                                     Everything else (prompt batches, mixed pairs, bf16 / f32, head_dim 64) is expanded into an f16 image in scratch and runs the f16
                                     kernels with the query in f16: gate 1e-3 against the oracle (the two differ by the CPU's query quantisation), and against float64
                                     attention over the dequantised cache the kernel must be no further away than the oracle is (<= 1e-6 for the float types).
  * whole models .................... prompt + greedy steps with every cache type against the oracle; context shift on block-format caches
"""
import ctypes as C

import numpy as np
import pytest

import harness as T
import llama_box_amd as L
from model_util import Context, Model, preset

pytestmark = pytest.mark.gpu

STORE_TYPES = [L.Q4_0, L.Q4_1, L.Q5_0, L.Q5_1, L.IQ4_NL, L.BF16]
NAME = L.TYPE_NAME


def both(build, backend):
    return T.run_case(build, "oracle"), T.run_case(build, backend)


def row_bytes(t, n):
    return n // L.TYPE_BLCK[t] * L.TYPE_SIZE[t]


def deq(t, raw, n):
    """raw cache bytes -> float64 values, through the oracle's to_float (pinned by tests/test_oracle_kv_golden.py)"""
    raw = np.ascontiguousarray(np.asarray(raw).reshape(-1))
    if t == L.F32:
        return raw.view(np.float32).astype(np.float64)
    if t == L.F16:
        return raw.view(np.float16).astype(np.float64)
    y = np.empty(n, np.float32)
    T.oracle().oracle_dequantize_row(t, raw.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), n)
    return y.astype(np.float64)


def edge_rows(rng, n_rows, width):
    x = (rng.standard_normal((n_rows, width)) * rng.uniform(0.05, 5.0, (n_rows, 1))).astype(np.float32)
    x[0, :] = 0.0                      # all-zero blocks: d = 0
    x[1, :] = np.float32(0.75)         # constant blocks: the offset formats get d = 0, the signed ones their extreme on level -8 / -16
    x[2, :64] = np.clip(x[2, :64], -3.0, 3.0)
    x[2, 5], x[2, 20] = 3.0, -3.0      # a tie in |x|: the first element wins the sign of the scale
    x[2, 40], x[2, 33] = -3.0, 3.0
    x[3, :32] = (np.arange(32, dtype=np.float32) - 15.5) * 0.25  # values that land on .5 before the truncation
    x[4, :32] = np.where(np.arange(32) % 2 == 0, 1e-20, -1e-20).astype(np.float32)  # under iq4_nl's group epsilon
    x[5, :] = np.where(rng.integers(0, 2, width) == 0, 0.0, -0.0).astype(np.float32)  # zeros of both signs: which zero the scale / minimum inherits is part of the bytes
    return x


@pytest.mark.parametrize("t", STORE_TYPES)
def test_set_rows_into_kv_types_is_byte_exact(backend, H, plog, t):
    rng = np.random.default_rng(100 + t)
    W, NCTX, n = 1024, 48, 21
    x = edge_rows(rng, n, W)
    rows = rng.permutation(NCTX)[:n].astype(np.int64)

    def build(g):
        cache = g.new(t, [W, NCTX])
        return [H.ggml_set_rows(g.ctx, cache, g.new(L.F32, [W, n], x), g.new(L.I64, [n], rows))]

    ref, got = both(build, backend)
    ref, got = np.asarray(ref[0]).reshape(NCTX, -1), np.asarray(got[0]).reshape(NCTX, -1)
    bad = np.nonzero((ref != got).any(axis=1))[0]
    plog(f"  set_rows f32 -> {NAME[t]} cache rows of {W}: {NCTX - bad.size}/{NCTX} rows byte-equal")
    if bad.size:
        bs = L.TYPE_SIZE[t] if L.TYPE_BLCK[t] == 32 else 64
        blk = np.nonzero((ref[bad[0]].reshape(-1, bs) != got[bad[0]].reshape(-1, bs)).any(axis=1))[0][0]
        raise AssertionError(f"{NAME[t]}: cache rows {bad[:6]} differ (source rows {[int(np.nonzero(rows == b)[0][0]) if b in rows else -1 for b in bad[:6]]}); row {bad[0]} block {blk}: "
                             f"oracle {ref[bad[0]].reshape(-1, bs)[blk].tolist()} gpu {got[bad[0]].reshape(-1, bs)[blk].tolist()}")
    # the 3-D form llama.cpp uses (K of one micro-batch: [head_dim, n_head_kv, n_tokens] viewed as rows) and a broadcast index tensor
    x3 = (rng.standard_normal((2, 5, 256)) * 2).astype(np.float32)
    idx3 = np.stack([rng.permutation(16)[:5] for _ in range(2)]).astype(np.int64)

    def build3(g):
        cache = g.new(t, [256, 16, 2])
        return [H.ggml_set_rows(g.ctx, cache, g.new(L.F32, [256, 5, 2], x3), g.new(L.I64, [5, 2], idx3))]

    ref, got = both(build3, backend)
    assert np.array_equal(np.asarray(got[0]), np.asarray(ref[0])), f"{NAME[t]}: 3-D set_rows differs"


@pytest.mark.parametrize("t", STORE_TYPES)
def test_cpy_between_kv_types_and_f32_for_the_k_shift(backend, H, plog, t):
    """llama.cpp build_rope_shift on a quantised cache: cast(view -> f32), rope in place, cpy back.  The cast is bit-exact and the copy back byte-exact;
    with the rope in between a sin / cos that differs from libm in the last bit may flip a re-quantised level by one."""
    rng = np.random.default_rng(300 + t)
    HD, NKV, NCTX = 128, 2, 40
    x = edge_rows(rng, NCTX, NKV * HD)
    shift = rng.integers(-9, 3, NCTX).astype(np.int32)
    shift[::3] = 0
    rb = row_bytes(t, NKV * HD)

    def fill(g):
        cache = g.new(t, [NKV * HD, NCTX])
        return H.ggml_set_rows(g.ctx, cache, g.new(L.F32, [NKV * HD, NCTX], x), g.new(L.I64, [NCTX], np.arange(NCTX, dtype=np.int64)))

    def build_cast(g):
        k = H.ggml_view_3d(g.ctx, fill(g), HD, NKV, NCTX, row_bytes(t, HD), rb, 0)
        return [H.ggml_cast(g.ctx, k, L.F32)]

    def build_back(g):  # f32 -> type without anything in between: pure byte work
        k = H.ggml_view_3d(g.ctx, g.new(t, [NKV * HD, NCTX]), HD, NKV, NCTX, row_bytes(t, HD), rb, 0)
        return [H.ggml_cpy(g.ctx, g.new(L.F32, [HD, NKV, NCTX], x), k)]

    def build_shift(g):
        k = H.ggml_view_3d(g.ctx, fill(g), HD, NKV, NCTX, row_bytes(t, HD), rb, 0)
        f = H.ggml_cast(g.ctx, k, L.F32)
        r = H.ggml_rope_ext_inplace(g.ctx, f, g.new(L.I32, [NCTX], shift), None, HD, 0, 8192, 500000.0, 1.0, 0.0, 1.0, 32.0, 1.0)
        return [H.ggml_cpy(g.ctx, r, k)]

    ref, got = both(build_cast, backend)
    assert np.array_equal(np.asarray(got[0]).view(np.uint32), np.asarray(ref[0]).view(np.uint32)), f"cast {NAME[t]} -> f32 is not bit-exact"
    ref, got = both(build_back, backend)
    assert np.array_equal(np.asarray(got[0]), np.asarray(ref[0])), f"cpy f32 -> {NAME[t]} is not byte-exact"
    if t == L.BF16:
        return  # (llama.cpp ropes a non-quantised cache in place and ROPE has no bf16 form: no context shift on a bf16 cache)
    ref, got = both(build_shift, backend)
    n = NCTX * NKV * HD
    same = np.count_nonzero(np.asarray(got[0]) == np.asarray(ref[0]))
    plog(f"  {NAME[t]} K-shift round trip: {same}/{np.asarray(ref[0]).size} bytes equal")
    T.compare(f"cpy {NAME[t]} -> f32 -> rope -> {NAME[t]}", deq(t, got[0], n), deq(t, ref[0], n), max_nmse=1e-4, log=plog)
    assert same >= 0.99 * np.asarray(ref[0]).size
    rows0, rows0_ref = (np.asarray(a).reshape(NCTX, -1)[shift == 0] for a in (got[0], ref[0]))
    assert np.array_equal(rows0, rows0_ref)  # a zero shift is the identity rotation: pure re-quantisation


FA_KV_CASES = [  # (type_k, type_v, head_dim, NH, NKV, n_q, n_kv, splits, sinks)
    (L.Q4_0, L.Q4_0, 128, 32, 8, 1, 1024, 0, False),
    (L.Q4_1, L.Q4_1, 128, 32, 8, 1, 300, 0, False),
    (L.Q5_0, L.Q5_0, 128, 28, 4, 1, 2048, 0, False),
    (L.Q5_1, L.Q5_1, 128, 16, 2, 3, 512, 2, False),
    (L.IQ4_NL, L.IQ4_NL, 128, 32, 8, 1, 777, 0, True),
    (L.BF16, L.BF16, 128, 32, 8, 1, 1024, 0, False),
    (L.BF16, L.BF16, 128, 28, 4, 5, 2048, 3, False),   # (round 6: bf16 read in place too — ggml_vec_dot_bf16's arithmetic: the query rounded to bf16, f32 products)
    (L.BF16, L.BF16, 128, 64, 8, 1, 4096, 0, True),
    (L.F32, L.F32, 128, 8, 2, 2, 256, 0, False),
    (L.Q8_0, L.Q4_0, 128, 32, 8, 1, 1024, 0, False),   # llama-box users' favourite pair: -ctk q8_0 -ctv q4_0
    (L.F16, L.Q4_0, 128, 32, 8, 1, 640, 0, False),
    (L.Q5_1, L.F16, 128, 16, 4, 4, 384, 3, False),
    (L.Q8_0, L.Q8_0, 64, 8, 2, 1, 512, 0, False),      # head_dim 64 on a q8_0 cache: not a shape of the lane-parallel q8_0 kernel -> the image
    (L.Q4_0, L.Q4_0, 64, 32, 4, 1, 1024, 0, False),
    (L.Q4_0, L.Q4_0, 128, 24, 8, 1, 900, 0, False),    # 3 / 6 query heads per KV head (round 6): the in-place forms in the next power of two's template
    (L.Q5_1, L.Q5_1, 128, 12, 2, 2, 300, 2, False),
    (L.Q4_0, L.Q4_0, 128, 8, 8, 1, 400, 0, False),     # multi-head attention
    (L.BF16, L.BF16, 128, 24, 8, 1, 512, 0, False),
    (L.Q4_0, L.Q4_0, 64, 24, 8, 1, 700, 0, False),
    (L.Q4_0, L.Q4_0, 128, 32, 8, 64, 1024, 0, False),  # a prompt micro-batch: the matrix-core kernel on the image
    (L.Q5_0, L.Q4_1, 128, 28, 4, 160, 2048, 0, False),
    (L.IQ4_NL, L.Q5_1, 128, 8, 8, 33, 512, 0, False),
]


@pytest.mark.parametrize("tk,tv,HD,NH,NKV,nq,nkv,splits,sinks", FA_KV_CASES)
def test_flash_attn_over_kv_types(backend, H, plog, tk, tv, HD, NH, NKV, nq, nkv, splits, sinks):
    rng = np.random.default_rng(tk * 131 + tv * 7 + NH + nkv + nq)
    NCTX = nkv + 64
    q = rng.standard_normal((NH, nq, HD)).astype(np.float32)
    kf = (rng.standard_normal((nkv, NKV * HD)) * rng.uniform(0.3, 2.0, (nkv, 1))).astype(np.float32)
    vf = (rng.standard_normal((nkv, NKV * HD)) * rng.uniform(0.3, 2.0, (nkv, 1))).astype(np.float32)
    kf[3, :64] = 0.0
    MR = (nq + 63) // 64 * 64
    mask = np.full((MR, nkv), -np.inf, np.float16)
    for t in range(nq):
        mask[t, : max(8, nkv - nq + t + 1 - 9)] = 0
        mask[t, 5] = -np.inf
    sk = rng.standard_normal(NH).astype(np.float32)
    backend.set_option("fa_splits", splits)
    img0, nat0 = backend.stat("kv_image_nodes"), backend.stat("kv_native_nodes")

    def cache_of(g, t, data):
        if t in (L.F16, L.F32):  # float caches: filled directly (SET_ROWS to f16 / f32 is tested in test_gpu_ops.py)
            full = np.zeros((NCTX, NKV * HD), np.float16 if t == L.F16 else np.float32)
            full[:nkv] = data
            return g.new(t, [NKV * HD, NCTX], full)
        return H.ggml_set_rows(g.ctx, g.new(t, [NKV * HD, NCTX]), g.new(L.F32, [NKV * HD, nkv], data), g.new(L.I64, [nkv], np.arange(nkv, dtype=np.int64)))

    def build(g):
        tq = g.new(L.F32, [HD, nq, NH], q)
        ks, vs = cache_of(g, tk, kf), cache_of(g, tv, vf)
        k = H.ggml_view_3d(g.ctx, ks, HD, nkv, NKV, row_bytes(tk, NKV * HD), row_bytes(tk, HD), 0)
        v = H.ggml_view_3d(g.ctx, vs, HD, nkv, NKV, row_bytes(tv, NKV * HD), row_bytes(tv, HD), 0)
        r = H.ggml_flash_attn_ext(g.ctx, tq, k, v, g.new(L.F16, [nkv, MR], mask), 1.0 / np.sqrt(HD), 0.0, 0.0)
        H.ggml_flash_attn_ext_set_prec(r, 10)
        if sinks:
            H.ggml_flash_attn_ext_add_sinks(r, g.new(L.F32, [NH], sk))
        return [r, ks, vs]

    try:
        ref, got = both(build, backend)
    finally:
        backend.set_option("fa_splits", 0)
    for name, i in (("K", 1), ("V", 2)):
        assert np.array_equal(np.asarray(got[i]), np.asarray(ref[i])), f"{name} cache bytes differ"
    # served either through the f16 image or, where the lane-parallel kernel reads such a cache IN PLACE (fattn.hip: fattn_native_kv_ok), by that form:
    # exactly one of the two, and the in-place form only at its shapes (head_dim 128, at most 32 query tokens)
    d_nat, d_img = backend.stat("kv_native_nodes") - nat0, backend.stat("kv_image_nodes") - img0
    in_place = HD == 128 and nq <= 32 and tk == tv and tk in (L.Q4_0, L.Q4_1, L.Q5_0, L.Q5_1, L.IQ4_NL, L.BF16)  # (fattn.hip: fattn_native_kv_ok)
    assert (d_nat, d_img) == ((1, 0) if in_place else (0, 1)), (d_nat, d_img)
    tag = f"flash_attn K={NAME[tk]} V={NAME[tv]} hd={HD} H={NH}/{NKV} nq={nq} nkv={nkv} splits={splits} sinks={sinks}"
    # in place (round 6): ggml-cpu's own arithmetic — the query row quantised to Q8_0 / Q8_1, integer block dots, V de-quantised to f32 — so the gate is the
    # one every other integer-dot path of the repo has; through the image (prompt batches, mixed pairs, float types) the query stays f16: north_star's band
    T.compare(tag, got[0], ref[0], max_nmse=1e-6 if in_place else 1e-3, log=plog)
    if sinks:
        return
    n = NCTX * NKV * HD
    kd, vd = deq(tk, ref[1], n).reshape(NCTX, NKV, HD)[:nkv], deq(tv, ref[2], n).reshape(NCTX, NKV, HD)[:nkv]
    exact = np.zeros((nq, NH, HD))
    for h in range(NH):
        kh, vh = kd[:, h // (NH // NKV)], vd[:, h // (NH // NKV)]
        sc = (q[h].astype(np.float64) @ kh.T) / np.sqrt(HD) + mask[:nq].astype(np.float64)
        sc -= sc.max(axis=1, keepdims=True)
        pr = np.exp(sc)
        exact[:, h] = (pr @ vh) / pr.sum(axis=1, keepdims=True)
    e_gpu, e_cpu = T.nmse(got[0].reshape(nq, NH, HD), exact), T.nmse(ref[0].reshape(nq, NH, HD), exact)
    plog(f"    vs float64 attention over the dequantised cache: kernel nmse={e_gpu:.3e}  cpu-oracle nmse={e_cpu:.3e}")
    if in_place:  # the same approximation as the oracle's (its 8-bit query), to the summation order
        assert e_gpu <= e_cpu * 1.05 + 1e-9
        return
    assert e_gpu <= 1e-6
    if L.TYPE_BLCK[tk] == 32:  # the oracle's 8-bit query is the larger approximation
        assert e_gpu <= e_cpu * 1.01 + 1e-12


def kv_name(t):  # (context parameter: 0 = the default f16, -1 = f32, else the ggml type)
    return "f16" if t == 0 else "f32" if t == -1 else NAME[t]


class oracle_variant:
    """an equally valid evaluation of the reference (oracle_set_variant): 1 = the mat-muls' block dots summed last block first.  A cache row is
    RE-QUANTISED from f32 values that differ in the last bit between two correct implementations, and on a 4-bit cache a flipped level is a
    sixteenth of the block's range: the distance of this variant from the reference is the yardstick of that effect"""
    def __init__(self, v):
        self.v = v

    def __enter__(self):
        T.oracle().oracle_set_variant(self.v)

    def __exit__(self, *a):
        T.oracle().oracle_set_variant(0)


class exact_query_oracle:
    """the oracle with the logit K.q taken from the dequantised K row and the UNQUANTISED query (oracle_set_variant(4)): ggml-cpu quantises every query row
    to 8 bits for a block-format K, csrc/kv_types.hip keeps it in f16 — so the GPU may differ from the reference by as much as that quantisation moves
    the reference itself, and must sit close to this variant"""
    def __enter__(self):
        T.oracle().oracle_set_variant(4)

    def __exit__(self, *a):
        T.oracle().oracle_set_variant(0)


PROMPT = [1, 17, 300, 42, 9, 250, 33, 7, 120, 64, 5, 99, 14, 201, 77, 3, 180, 29]


def greedy(ctx, prompt, n):
    rc, lg = ctx.decode(prompt, range(len(prompt)), want=[0] * (len(prompt) - 1) + [1])
    assert rc == 0
    ids, rows, row = [], [], lg[-1]
    for i in range(n):
        rows.append(row)
        ids.append(int(np.argmax(row)))
        rc, l1 = ctx.decode([ids[-1]], [len(prompt) + i])
        assert rc == 0
        row = l1[0]
    return ids, rows


@pytest.mark.parametrize("tk,tv", [(L.Q4_0, L.Q4_0), (L.Q4_1, L.Q4_1), (L.Q5_0, L.Q5_0), (L.Q5_1, L.Q5_1), (L.IQ4_NL, L.IQ4_NL), (L.BF16, L.BF16), (-1, -1), (L.Q8_0, L.Q4_0), (0, L.Q5_1)])
def test_model_logits_with_kv_cache_types(backend, H, plog, tk, tv):
    """-ctk / -ctv on a whole model (head_dim 128, grouped heads): the prompt's logits and greedy steps against the oracle; the decode steps replay as
    hipGraphs with the image kernels in them; nothing of the graph is refused (supports_op) — ggml_backend_sched would not send a node to the CPU."""
    hp = preset("test-llama", n_head=4, n_head_kv=2, n_embd_head=128)
    mc = Model(hp, 1234, H.ggml_backend_cpu_buffer_type())
    mg = Model(hp, 1234, backend.buft)
    cc = Context(mc, compute=T.oracle_compute_fn(), flash_attn=1, type_k=tk, type_v=tv)
    cg = Context(mg, backend=backend, flash_attn=1, type_k=tk, type_v=tv)
    name = f"K={kv_name(tk)} V={kv_name(tv)}"
    try:
        rc, ref = cc.decode(PROMPT, range(len(PROMPT)))
        rc2, got = cg.decode(PROMPT, range(len(PROMPT)))
        assert rc == 0 and rc2 == 0
        cc.clear()
        with exact_query_oracle():
            rc3, refx = cc.decode(PROMPT, range(len(PROMPT)))
        assert rc3 == 0
        e, ex, eq = T.nmse(got, ref), T.nmse(got, refx), T.nmse(refx, ref)
        plog(f"{name} cache, prompt logits: nmse(gpu, cpu)={e:.3e}; against the oracle with an unquantised query {ex:.3e}; what the 8-bit query moves the oracle itself {eq:.3e}")
        assert e <= 1e-3 and ex <= 1e-3
        cc.clear(); cg.clear()
        i0, g0 = backend.stat("kv_image_nodes") + backend.stat("kv_native_nodes"), backend.stat("graph_launches")
        ids_ref, rows_ref = greedy(cc, PROMPT, 12)
        ids_got, rows_got = greedy(cg, PROMPT, 12)
        served = backend.stat("kv_image_nodes") + backend.stat("kv_native_nodes") - i0
        plog(f"{name} greedy ids ref={ids_ref} got={ids_got}; attention nodes read in place or through the image {served}, graph replays {backend.stat('graph_launches') - g0}")
        # (the prompt and the first decode steps run node by node — counted; the replayed steps carry those kernels inside the captured graph)
        assert served >= hp.n_layer * 2 and backend.stat("graph_launches") - g0 >= 8
        n_same = next((i for i, (a, b) in enumerate(zip(ids_ref, ids_got)) if a != b), len(ids_ref))
        for i in range(min(n_same + 1, len(rows_ref))):
            assert T.nmse(rows_got[i], rows_ref[i]) <= 1e-3
        if n_same < len(ids_ref):
            r = np.sort(rows_ref[n_same])[::-1]
            dev = float(np.max(np.abs(rows_got[n_same] - rows_ref[n_same])))
            assert r[0] - r[1] <= 2 * dev, f"greedy ids diverge at step {n_same} with margin {r[0] - r[1]:.3e} > deviation {dev:.3e}"
    finally:
        for o in (cc, cg, mc, mg):
            o.free()


@pytest.mark.parametrize("tk,tv", [(L.Q4_0, L.Q4_0), (L.Q4_1, L.Q4_1), (L.Q5_0, L.Q5_0), (L.Q5_1, L.Q5_1), (L.IQ4_NL, L.IQ4_NL), (L.BF16, L.BF16), (L.Q8_0, L.Q4_0), (0, L.Q5_1)])
def test_decode_cache_rows_stored_by_the_fused_qkv_launch_are_byte_exact(backend, H, plog, tk, tv):
    """Round 6: a decode step's K / V rows of a block-format (or bf16) cache leave the fused Q/K/V launch already in the cache's format (qkv.hip epilogue +
    kv_quant.h: a lane per value) instead of as f32 rows that one more launch stored (7 us per layer).  Same bytes in every layer's K and V cache, and the
    same logits bit for bit, as with the fused launch switched off (option qkv 0: the projections, ROPE and SET_ROWS run as separate kernels — the SET_ROWS
    whose bytes test_set_rows_into_kv_types_is_byte_exact pins on the oracle)."""
    hp = preset("test-llama", n_head=4, n_head_kv=2, n_embd_head=128)
    mg = Model(hp, 77, backend.buft)
    outs = {}
    try:
        for mode in (1, 0):
            backend.set_option("qkv", mode)
            cg = Context(mg, backend=backend, flash_attn=1, type_k=tk, type_v=tv)
            k0 = backend.stat("kernel_launches")
            rc, _ = cg.decode(PROMPT, range(len(PROMPT)), want=[0] * (len(PROMPT) - 1) + [1])
            assert rc == 0
            rows = []
            for i in range(6):
                rc, lg = cg.decode([3 + 5 * i], [len(PROMPT) + i])
                assert rc == 0
                rows.append(lg[0])
            launches = backend.stat("kernel_launches") - k0
            caches = []
            for il in range(hp.n_layer):
                for which in (0, 1):
                    t = H.llm_context_cache_tensor(cg.c, il, which)
                    n = H.ggml_nbytes(t)
                    raw = np.empty(n, np.uint8)
                    H.ggml_backend_tensor_get(t, raw.ctypes.data_as(C.c_void_p), 0, n)
                    caches.append(raw)
            outs[mode] = (np.stack(rows), caches, launches)
            cg.free()
    finally:
        backend.set_option("qkv", 1)
        mg.free()
    plog(f"K={kv_name(tk)} V={kv_name(tv)}: kernel launches for the prompt + 6 decode steps: fused {outs[1][2]}, unfused {outs[0][2]}")
    for a, b in zip(outs[1][1], outs[0][1]):
        assert np.array_equal(a, b), "cache bytes differ between the fused launch's epilogue stores and SET_ROWS"
    assert np.array_equal(outs[1][0].view(np.uint32), outs[0][0].view(np.uint32))


@pytest.mark.parametrize("t", [L.Q4_0, L.Q5_1, L.IQ4_NL, -1])
def test_context_shift_on_kv_cache_types(backend, H, plog, t):
    """llama-box context shift (httpserver.hpp:3453-3537) on such a cache: seq_rm + seq_add, K re-rotated on the device through cast -> rope -> cpy (block
    formats) or in place (f32), decoding continues at the shifted positions."""
    hp = preset("test-llama", n_head=2, n_head_kv=1, n_embd_head=128)
    mc = Model(hp, 1234, H.ggml_backend_cpu_buffer_type())
    mg = Model(hp, 1234, backend.buft)
    cc = Context(mc, compute=T.oracle_compute_fn(), flash_attn=1, type_k=t, type_v=t)
    cg = Context(mg, backend=backend, flash_attn=1, type_k=t, type_v=t)
    try:
        n_keep, n_discard = 4, 6
        for c in (cc, cg):
            assert c.decode(PROMPT, range(len(PROMPT)))[0] == 0
            assert c.seq_rm(0, n_keep, n_keep + n_discard) == 1
            assert c.seq_add(0, n_keep + n_discard, len(PROMPT), -n_discard) == 0
        pos = len(PROMPT) - n_discard
        # two more runs of the same prompt, shift and steps: the oracle with an unquantised query, and the oracle summing its block dots in the other order
        cx = Context(mc, compute=T.oracle_compute_fn(), flash_attn=1, type_k=t, type_v=t)
        cv = Context(mc, compute=T.oracle_compute_fn(), flash_attn=1, type_k=t, type_v=t)
        try:
            for c, how in ((cx, exact_query_oracle()), (cv, oracle_variant(1))):
                with how:
                    assert c.decode(PROMPT, range(len(PROMPT)))[0] == 0
                    assert c.seq_rm(0, n_keep, n_keep + n_discard) == 1 and c.seq_add(0, n_keep + n_discard, len(PROMPT), -n_discard) == 0
            for i, tok in enumerate([11, 200, 45]):
                rc, ref = cc.decode([tok], [pos + i])
                rc2, got = cg.decode([tok], [pos + i])
                with exact_query_oracle():
                    rc3, refx = cx.decode([tok], [pos + i])
                with oracle_variant(1):
                    rc4, refv = cv.decode([tok], [pos + i])
                assert rc == 0 and rc2 == 0 and rc3 == 0 and rc4 == 0
                e, ex, eq, ev = T.nmse(got, ref), T.nmse(got, refx), T.nmse(refx, ref), T.nmse(refv, ref)
                plog(f"cache={kv_name(t)} after context shift, step {i}: nmse(gpu, cpu)={e:.3e}; against the oracle with an unquantised query {ex:.3e}; the oracle from itself: "
                     f"unquantised query {eq:.3e}, other summation order {ev:.3e}")
                # the implementation's own share — its distance from the reference arithmetic WITHOUT the 8-bit query — stays inside north_star's band; the
                # distance from the reference itself may add what that query quantisation moves the reference (independent errors add: e ~ ex + eq; measured
                # on a q4_0 cache at step 2: 1.00e-3 ~ 6.2e-4 + 5.9e-4), bounded by twice that
                assert ex <= 1e-3 and e <= max(1e-3, 2.0 * max(eq, ev))
        finally:
            cx.free()
            cv.free()
    finally:
        for o in (cc, cg, mc, mg):
            o.free()


# ------------------------------------------------------------------------------------------------ -fa off: K.q as a MUL_MAT over cache blocks
@pytest.mark.parametrize("t", [L.Q8_0, L.Q4_0, L.Q4_1, L.Q5_0, L.Q5_1, L.IQ4_NL, L.BF16])
@pytest.mark.parametrize("HD,NH,NKV,nq,nkv", [(128, 32, 8, 1, 1024), (128, 28, 4, 5, 300), (64, 8, 2, 40, 512)])
def test_mul_mat_over_a_k_cache_in_kv_types(backend, H, plog, t, HD, NH, NKV, nq, nkv):
    """llama-box's default (-fa off) with -ctk <type>: llama.cpp asks flash attention only of a quantised V, so K.q is MUL_MAT(cache view [HD, n_kv, n_head_kv],
    q [HD, n_q, n_head]).  ggml-cpu quantises q to Q8_0 / Q8_1 and takes block dots; the device expands the view to f16 and runs the f16 product (q rounded
    to f16, f32 sums).  Against the float64 product over the dequantised cache the kernel must be no further away than the oracle."""
    rng = np.random.default_rng(t * 17 + NH + nkv + nq)
    NCTX = nkv + 32
    q = rng.standard_normal((NH, nq, HD)).astype(np.float32)
    kf = (rng.standard_normal((nkv, NKV * HD)) * rng.uniform(0.3, 2.0, (nkv, 1))).astype(np.float32)
    kf[3, :64] = 0.0
    img0 = backend.stat("kv_image_nodes")

    def build(g):
        ks = H.ggml_set_rows(g.ctx, g.new(t, [NKV * HD, NCTX]), g.new(L.F32, [NKV * HD, nkv], kf), g.new(L.I64, [nkv], np.arange(nkv, dtype=np.int64)))
        k = H.ggml_view_3d(g.ctx, ks, HD, nkv, NKV, row_bytes(t, NKV * HD), row_bytes(t, HD), 0)
        return [H.ggml_mul_mat(g.ctx, k, g.new(L.F32, [HD, nq, NH], q)), ks]

    ref, got = both(build, backend)
    assert np.array_equal(np.asarray(got[1]), np.asarray(ref[1]))
    assert backend.stat("kv_image_nodes") == img0 + 1
    T.compare(f"mul_mat K={NAME[t]} cache view hd={HD} H={NH}/{NKV} nq={nq} nkv={nkv}", got[0], ref[0], max_nmse=1e-3, log=plog)
    kd = deq(t, ref[1], NCTX * NKV * HD).reshape(NCTX, NKV, HD)[:nkv]
    exact = np.stack([q[h].astype(np.float64) @ kd[:, h // (NH // NKV)].T for h in range(NH)])  # [NH, nq, nkv]
    e_gpu, e_cpu = T.nmse(np.asarray(got[0]).reshape(NH, nq, nkv), exact), T.nmse(np.asarray(ref[0]).reshape(NH, nq, nkv), exact)
    plog(f"    vs the float64 product over the dequantised cache: kernel nmse={e_gpu:.3e}  cpu-oracle nmse={e_cpu:.3e}")
    assert e_gpu <= 1e-6
    if L.TYPE_BLCK[t] == 32:
        assert e_gpu <= e_cpu * 1.01 + 1e-12


@pytest.mark.parametrize("tk", [L.Q8_0, L.Q4_0, L.Q5_1, L.IQ4_NL, L.BF16])
def test_model_logits_without_flash_attention_and_a_k_cache_in_kv_types(backend, H, plog, tk):
    """-ctk <type> with llama-box's default attention path (-fa off; the V cache stays f16 and transposed): prompt + greedy steps against the oracle."""
    hp = preset("test-llama", n_head=4, n_head_kv=2, n_embd_head=128)
    mc = Model(hp, 1234, H.ggml_backend_cpu_buffer_type())
    mg = Model(hp, 1234, backend.buft)
    cc = Context(mc, compute=T.oracle_compute_fn(), flash_attn=0, type_k=tk, type_v=0)
    cg = Context(mg, backend=backend, flash_attn=0, type_k=tk, type_v=0)
    try:
        rc, ref = cc.decode(PROMPT, range(len(PROMPT)))
        rc2, got = cg.decode(PROMPT, range(len(PROMPT)))
        assert rc == 0 and rc2 == 0
        e = T.nmse(got, ref)
        plog(f"-fa 0, K={kv_name(tk)} cache, prompt logits: nmse(gpu, cpu)={e:.3e}")
        assert e <= 1e-3
        cc.clear(); cg.clear()
        ids_ref, rows_ref = greedy(cc, PROMPT, 10)
        ids_got, rows_got = greedy(cg, PROMPT, 10)
        plog(f"-fa 0, K={kv_name(tk)} greedy ids ref={ids_ref} got={ids_got}")
        n_same = next((i for i, (a, b) in enumerate(zip(ids_ref, ids_got)) if a != b), len(ids_ref))
        for i in range(min(n_same + 1, len(rows_ref))):
            assert T.nmse(rows_got[i], rows_ref[i]) <= 1e-3
        if n_same < len(ids_ref):
            r = np.sort(rows_ref[n_same])[::-1]
            dev = float(np.max(np.abs(rows_got[n_same] - rows_ref[n_same])))
            assert r[0] - r[1] <= 2 * dev, f"greedy ids diverge at step {n_same} with margin {r[0] - r[1]:.3e} > deviation {dev:.3e}"
    finally:
        for o in (cc, cg, mc, mg):
            o.free()
