"""CPU: the C-ABI library loads and exports every symbol include/ggml_mi355x.h declares (no compute without a GPU);
the host-side mirror (graph construction, allocator, GGUF, llama_decode-shaped driver) behaves; ABI struct layout."""
import ctypes as C
import os
import re
import tempfile

import numpy as np
import pytest

import harness as T
import llama_box_amd as L
from model_util import Context, Model, greedy, preset


def test_backend_library_exports_declared_symbols(built):
    hdr = open(os.path.join(L.REPO, "include", "ggml_mi355x.h")).read()
    declared = re.findall(r"^(?:ggml_backend_reg_t|int)\s+(ggml_backend_\w+)\s*\(void\);", hdr, flags=re.M)
    assert set(declared) == {"ggml_backend_init", "ggml_backend_score", "ggml_backend_mi355x_reg"}
    lib = C.CDLL(L.BACKEND_SO)
    for sym in declared:
        assert getattr(lib, sym) is not None
    # on a box without a gfx950 device: score 0 and init NULL — the product path fails loudly, it never falls back
    lib.ggml_backend_score.restype = C.c_int
    lib.ggml_backend_init.restype = C.c_void_p
    import subprocess
    has_gpu = subprocess.run(["bash", "-c", "ls /dev/kfd >/dev/null 2>&1"]).returncode == 0
    if not has_gpu:
        assert lib.ggml_backend_score() == 0
        assert lib.ggml_backend_init() is None
        try:
            L.Backend(0)
            assert False, "Backend() must raise without a device"
        except RuntimeError:
            pass


def test_no_product_file_references_the_oracle():
    """oracle/ is test infrastructure: nothing under llama_box_amd/ may include, link or import it."""
    bad = []
    for root, _, files in os.walk(os.path.join(L.REPO, "llama_box_amd")):
        if "build" in root:
            continue
        for f in files:
            if f.endswith((".cpp", ".h", ".hip", ".py", "Makefile")):
                s = open(os.path.join(root, f), errors="ignore").read()
                if re.search(r'#include\s+"[^"]*oracle|liboracle|import\s+harness|from\s+harness', s):
                    bad.append(os.path.join(root, f))
    assert not bad, bad


def test_tensor_struct_layout(H):
    ctx = H.ggml_init(L.InitParams(0, None, True))
    t = H.ggml_new_tensor_3d(ctx, L.Q4_K, 512, 7, 3)
    tt = t.contents
    assert list(tt.ne) == [512, 7, 3, 1] and list(tt.nb) == [144, 288, 288 * 7, 288 * 21]
    assert H.ggml_nbytes(t) == 288 * 21
    v = H.ggml_view_2d(ctx, t, 256, 7, 288, 144)
    assert v.contents.view_offs == 144 and C.addressof(v.contents.view_src.contents) == C.addressof(tt)
    p = H.ggml_permute(ctx, H.ggml_new_tensor_4d(ctx, L.F32, 2, 3, 4, 5), 0, 2, 1, 3)
    assert list(p.contents.ne) == [2, 4, 3, 5] and list(p.contents.nb) == [4, 24, 8, 96]
    q = H.ggml_new_tensor_3d(ctx, L.F32, 64, 5, 8)
    k = H.ggml_new_tensor_3d(ctx, L.F16, 64, 32, 2)
    fa = H.ggml_flash_attn_ext(ctx, q, k, k, None, 1.0, 0.0, 0.0)
    assert list(fa.contents.ne) == [64, 8, 5, 1]
    H.ggml_free(ctx)


def test_graph_allocator_never_overlaps_live_tensors(H):
    """ggml_gallocr mirror: tensors whose lifetimes overlap must not share bytes (except an in-place op and the parent it
    replaces); same graph -> same addresses."""
    rng = np.random.default_rng(0)

    def build():
        ctx = H.ggml_init(L.InitParams(0, None, True))
        x = H.ggml_new_tensor_2d(ctx, L.F32, 256, 4)
        H.ggml_set_input(x)
        live = [x]
        for i in range(40):
            a = live[int(rng.integers(len(live)))]
            b = live[int(rng.integers(len(live)))]
            live.append(H.ggml_add(ctx, a, b) if i % 3 else H.ggml_scale(ctx, a, 0.5))
        gf = H.ggml_new_graph(ctx)
        H.ggml_set_output(live[-1])
        H.ggml_build_forward_expand(gf, live[-1])
        return ctx, gf

    ga = H.ggml_gallocr_new(H.ggml_backend_cpu_buffer_type())
    rng = np.random.default_rng(0)
    ctx, gf = build()
    assert H.ggml_gallocr_alloc_graph(ga, gf)
    n = gf.contents.n_nodes
    nodes = [gf.contents.nodes[i].contents for i in range(n)]
    last_use = {}
    for i, nd in enumerate(nodes):
        for s in range(10):
            if nd.src[s]:
                last_use[C.addressof(nd.src[s].contents)] = i
    for i, a in enumerate(nodes):
        ea = last_use.get(C.addressof(a), n if (a.flags & 2) else i)
        for j in range(i + 1, n):
            b = nodes[j]
            if j < ea:  # b is written while a is still needed
                assert a.data + 4096 <= b.data or b.data + 4096 <= a.data, (i, j)
            elif j == ea and not (a.data + 4096 <= b.data or b.data + 4096 <= a.data):
                # b is a's LAST reader: it may run in place (ggml_op_can_inplace) — then it takes exactly a's block
                assert b.data == a.data and any(b.src[s] and C.addressof(b.src[s].contents) == C.addressof(a) for s in range(10)), (i, j)
    addrs = [nd.data for nd in nodes]
    rng = np.random.default_rng(0)
    ctx2, gf2 = build()
    assert H.ggml_gallocr_alloc_graph(ga, gf2)
    assert addrs == [gf2.contents.nodes[i].contents.data for i in range(n)]
    assert H.ggml_gallocr_get_buffer_size(ga, 0) < 41 * 4096  # re-use happened
    H.ggml_free(ctx); H.ggml_free(ctx2); H.ggml_gallocr_free(ga)


def test_gguf_roundtrip_and_loader_errors(H):
    hp = preset("test-qwen2")
    cpu = H.ggml_backend_cpu_buffer_type()
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "m.gguf")
        assert H.llm_synth_gguf(hp, 77, path.encode()) == 0
        raw = open(path, "rb").read()
        assert raw[:4] == b"GGUF" and int.from_bytes(raw[4:8], "little") == 3
        m1 = Model(path=path, buft=cpu)
        m2 = Model(hp, 77, cpu)
        for name in ("token_embd.weight", "blk.1.attn_q.weight", "blk.0.attn_k.bias", "blk.1.ffn_down.weight", "output.weight", "output_norm.weight"):
            a, b = H.llm_model_tensor(m1.m, name.encode()), H.llm_model_tensor(m2.m, name.encode())
            assert a and b and a.contents.type == b.contents.type and list(a.contents.ne) == list(b.contents.ne)
            n = H.ggml_nbytes(a)
            assert C.string_at(a.contents.data, n) == C.string_at(b.contents.data, n), name
        assert m1.hp.n_head_kv == hp.n_head_kv and abs(m1.hp.rope_freq_base - hp.rope_freq_base) < 1 and m1.hp.rope_type == 2
        m1.free(); m2.free()
        open(path, "wb").write(raw[:100])
        assert not H.llm_model_load(path.encode(), cpu)  # truncated
        open(path, "wb").write(b"XXXX" + raw[4:])
        assert not H.llm_model_load(path.encode(), cpu)  # bad magic


def test_driver_on_oracle_determinism_reuse_and_batching(H):
    hp = preset("test-llama")
    m = Model(hp, 1234, H.ggml_backend_cpu_buffer_type())
    fn = T.oracle_compute_fn()
    prompt = [1, 5, 9, 300, 17, 42, 99, 7]
    try:
        c1 = Context(m, compute=fn, graph_reuse=0)
        c2 = Context(m, compute=fn, graph_reuse=1)
        ids1, rows1 = greedy(c1, prompt, 6)
        ids2, rows2 = greedy(c2, prompt, 6)
        assert ids1 == ids2 and all(np.array_equal(a, b) for a, b in zip(rows1, rows2))
        # token-by-token prefill == batched prefill (every column of a mat-mul is computed independently)
        c1.clear()
        rows = []
        for i, t in enumerate(prompt):
            rc, lg = c1.decode([t], [i])
            assert rc == 0
            rows.append(lg[0])
        c2.clear()
        rc, lg = c2.decode(prompt, range(len(prompt)))
        T.compare("sequential vs batched prefill (oracle)", np.stack(rows), lg, max_nmse=1e-12)
        # flash path stays close to the soft-max path (f16 accumulation noise only)
        c3 = Context(m, compute=fn, flash_attn=1)
        rc, lf = c3.decode(prompt, range(len(prompt)))
        assert T.nmse(lf, lg) < 5e-3
        # quantised KV cache (-ctk / -ctv q8_0): Q8_0 rounding of K, V and of the query moves the logits a little, not a lot;
        # decode after prefill works on the quantised rows; a quantised cache without flash attention is refused
        c4 = Context(m, compute=fn, flash_attn=1, type_k=L.Q8_0, type_v=L.Q8_0)
        rc, lq = c4.decode(prompt, range(len(prompt)))
        assert rc == 0 and not np.array_equal(lq, lf) and T.nmse(lq, lf) < 2e-2
        rc, lq1 = c4.decode([11], [len(prompt)])
        rc2, lf1 = c3.decode([11], [len(prompt)])
        assert rc == 0 and rc2 == 0 and T.nmse(lq1, lf1) < 2e-2
        c5 = Context(m, compute=fn, flash_attn=1, type_k=L.Q8_0)  # K only
        rc, lk = c5.decode(prompt, range(len(prompt)))
        assert rc == 0 and T.nmse(lk, lf) < 2e-2
        with pytest.raises(RuntimeError):
            Context(m, compute=fn, flash_attn=0, type_v=L.Q8_0)
        # a prompt in micro-batches, logits wanted for its last token only: the chunks before the last have ZERO outputs — out_ids is empty, the
        # last layer's tensors behind get_rows and the output head have no rows and are skipped (ggml_is_empty), the cache is filled all the same
        long_prompt = [(7 * i + 3) % hp.n_vocab for i in range(20)]
        want = [0] * 19 + [1]
        c6 = Context(m, compute=fn, flash_attn=1, n_ubatch=8)
        c7 = Context(m, compute=fn, flash_attn=1, n_ubatch=32)
        rc6, l6 = c6.decode(long_prompt, range(20), want=want)
        rc7, l7 = c7.decode(long_prompt, range(20), want=want)
        assert rc6 == 0 and rc7 == 0 and l6.shape == (1, hp.n_vocab) and np.array_equal(l6, l7)
        # return codes
        assert c3.decode([hp.n_vocab], [0])[0] == -1
        for c in (c1, c2, c3, c4, c5, c6, c7):
            c.free()
    finally:
        m.free()


@pytest.mark.parametrize("name,type_k", [("test-llama", 0), ("test-qwen2", 0), ("test-llama", L.Q8_0), ("test-llama", L.Q5_1), ("test-llama", L.Q4_0), ("test-llama", L.IQ4_NL), ("test-llama", -1)])
def test_context_shift_k_shift_oracle(H, name, type_k):
    """llama-box context shift (httpserver.hpp:3453-3537): seq_rm of [n_keep, n_keep + n_discard) + seq_add of the rest by
    -n_discard re-rotates the cached K rows.  In a ONE-layer model a K row depends only on its own token and position, so the
    shifted cache must equal a fresh prefill of the shortened sequence (up to the extra f16 / q8_0 rounding of the stored
    rows) — that pins direction and magnitude of the shift on the oracle; the GPU is then compared with the oracle."""
    over = dict(n_layer=1)
    if type_k:
        over.update(n_head=2, n_head_kv=1, n_embd_head=128)
    hp = preset(name, **over)
    m = Model(hp, 77, H.ggml_backend_cpu_buffer_type())
    fn = T.oracle_compute_fn()
    try:
        prompt = [3, 9, 27, 81, 243, 217, 139, 411, 210, 118, 354, 40, 120, 360, 58, 174]
        n_keep, n_discard = 3, 5
        a = Context(m, compute=fn, flash_attn=1, type_k=type_k, type_v=type_k)
        assert a.decode(prompt, range(len(prompt)))[0] == 0
        assert a.seq_rm(0, n_keep, n_keep + n_discard) == 1
        assert a.seq_add(0, n_keep + n_discard, len(prompt), -n_discard) == 0
        rc, la = a.decode([11], [len(prompt) - n_discard])
        assert rc == 0
        b = Context(m, compute=fn, flash_attn=1, type_k=type_k, type_v=type_k)
        short = prompt[:n_keep] + prompt[n_keep + n_discard:]
        assert b.decode(short, range(len(short)))[0] == 0
        rc, lb = b.decode([11], [len(short)])
        assert rc == 0
        # and the unshifted cache (positions left as they were) must NOT agree: the shift did something
        c = Context(m, compute=fn, flash_attn=1, type_k=type_k, type_v=type_k)
        assert c.decode(prompt, range(len(prompt)))[0] == 0
        c.seq_rm(0, n_keep, n_keep + n_discard)
        rc, lc = c.decode([11], [len(prompt)])
        e_shift, e_noshift = T.nmse(la, lb), T.nmse(lc, lb)
        print(f"K-shift on the oracle, cache type {type_k}: shifted vs fresh prefill {e_shift:.3e}, unshifted vs fresh {e_noshift:.3e}")
        # (a shifted row of a block-format cache has been quantised twice — stored, read, rotated, stored again — the fresh prefill once; measured: q8_0 6e-5, q5_1 1.3e-4,
        # q4_0 2.2e-4, iq4_nl 1.3e-4, against 5e-3 .. 6e-3 when the rows are NOT rotated)
        assert e_shift <= (1e-5 if type_k in (0, -1) else 2e-3), e_shift
        assert e_noshift > 20 * e_shift
        for x in (a, b, c):
            x.free()
    finally:
        m.free()


def test_kv_cache_type_rules_of_the_driver(H):
    """-ctk / -ctv as the harness takes them (llama-box/engine_param.hpp:51-54; llama.cpp's own rules): every listed type for K with or without flash attention, a V cache
    other than f16 only with it; a type outside the list is refused; a bf16 K cache cannot be context-shifted (ROPE has no bf16 form and llama.cpp ropes a non-quantised
    cache in place)."""
    hp = preset("test-llama", n_head=2, n_head_kv=1, n_embd_head=128, n_layer=1)
    m = Model(hp, 5, H.ggml_backend_cpu_buffer_type())
    fn = T.oracle_compute_fn()
    try:
        for tk in (0, -1, L.BF16, L.Q8_0, L.Q4_0, L.Q4_1, L.Q5_0, L.Q5_1, L.IQ4_NL):
            for fa in (0, 1):
                c = Context(m, compute=fn, flash_attn=fa, type_k=tk, type_v=0)
                rc, lg = c.decode([1, 2, 3, 4], range(4))
                assert rc == 0 and np.all(np.isfinite(lg))
                c.free()
        for tv in (L.Q8_0, L.Q4_0, L.BF16, -1):
            with pytest.raises(Exception):
                Context(m, compute=fn, flash_attn=0, type_k=0, type_v=tv)
            c = Context(m, compute=fn, flash_attn=1, type_k=0, type_v=tv)
            assert c.decode([1, 2, 3, 4], range(4))[0] == 0
            c.free()
        with pytest.raises(Exception):
            Context(m, compute=fn, flash_attn=1, type_k=L.Q6_K, type_v=0)  # (a K-quant is not a cache type)
        c = Context(m, compute=fn, flash_attn=1, type_k=L.BF16, type_v=L.BF16)
        assert c.decode([1, 2, 3, 4, 5, 6], range(6))[0] == 0
        assert c.seq_rm(0, 1, 3) == 1
        assert c.seq_add(0, 3, 6, -2) == -2
        c.free()
    finally:
        m.free()


def test_stream_bytes_matches_survey_for_llama3_8b(H):
    """Llama-3-8B Q4_K_M streams 4 617 398 528 B per decoded token (weights + norms + 1 embd row)."""
    hp = preset("llama3-8b-q4_k_m")
    E, FF, V, HD = hp.n_embd, hp.n_ff, hp.n_vocab, hp.n_embd_head
    q4, q6 = 144 / 256, 210 / 256
    total = 0
    for il in range(hp.n_layer):
        more = il < 4 or il >= 28 or (il - 4) % 3 == 2
        total += (E * E * 2 + E * 1024 + E * FF * 2) * q4 + (E * 1024 + FF * E) * (q6 if more else q4) + 2 * E * 4
    total += V * E * q6 + E * 4 + E * q4
    assert int(total) == 4617398528


def _rope_multi_oracle(H, x, pos4, n_dims, sections, mode, dt=None, freq_base=10000.0):
    dt = dt or L.F32

    def build(g):
        ne = list(x.shape[::-1])
        sec = (C.c_int * 4)(*sections)
        return H.ggml_rope_multi(g.ctx, g.new(dt, ne, x), g.new(L.I32, [pos4.size], pos4), None, n_dims, sec, mode, 32768, freq_base, 1.0, 0.0, 1.0, 32.0, 1.0)

    return T.run_case(build, "oracle")[0]


def test_rope_multi_oracle_against_closed_form(H):
    """ggml_rope_multi (mrope.patch touches exactly this routine): the oracle's recurrence against the closed form
    theta(pair) = pos[stream(pair)] * base^(-2 * k / n_dims) in float64, k = pair index (mrope) or index within the section
    (vision); and mrope with four identical position streams must reproduce NeoX rope bit for bit."""
    rng = np.random.default_rng(31)
    HD, NH, NT = 128, 3, 5
    x = rng.standard_normal((NT, NH, HD)).astype(np.float32)
    pos4 = rng.integers(0, 4000, 4 * NT).astype(np.int32)
    sections = [16, 24, 24, 0]
    got = _rope_multi_oracle(H, x, pos4, HD, sections, L.ROPE_MROPE, freq_base=1000000.0).reshape(NT, NH, HD)
    half = HD // 2
    ref = np.zeros_like(x, dtype=np.float64)
    for t in range(NT):
        for ic in range(half):
            sector = ic % sum(sections)
            stream = 0 if sector < 16 else (1 if sector < 40 else 2)
            th = float(pos4[t + stream * NT]) * 1000000.0 ** (-2.0 * ic / HD)
            c, s_ = np.cos(th), np.sin(th)
            x0, x1 = x[t, :, ic].astype(np.float64), x[t, :, ic + half].astype(np.float64)
            ref[t, :, ic], ref[t, :, ic + half] = x0 * c - x1 * s_, x0 * s_ + x1 * c
    # the f32 recurrence theta *= theta_scale drifts from the float64 power by ~1e-6 relative per step; angles reach 4e3 rad
    assert T.nmse(got, ref) < 1e-5
    # identical streams == NeoX
    p1 = rng.integers(0, 4000, NT).astype(np.int32)
    same = _rope_multi_oracle(H, x, np.tile(p1, 4), HD, sections, L.ROPE_MROPE)

    def build_neox(g):
        return H.ggml_rope_ext(g.ctx, g.new(L.F32, [HD, NH, NT], x), g.new(L.I32, [NT], p1), None, HD, L.ROPE_NEOX, 32768, 10000.0, 1.0, 0.0, 1.0, 32.0, 1.0)
    neox = T.run_case(build_neox, "oracle")[0]
    assert np.array_equal(same.view(np.uint32), neox.view(np.uint32))
    # vision mode: n_dims = head_dim / 2, sections of head_dim / 4 pairs, pairs (ic, ic + n_dims), each section's angle restarts
    HDv = 80
    xv = rng.standard_normal((NT, NH, HDv)).astype(np.float32)
    sec_v = [HDv // 4] * 4
    gv = _rope_multi_oracle(H, xv, pos4, HDv // 2, sec_v, L.ROPE_VISION).reshape(NT, NH, HDv)
    refv = np.zeros_like(xv, dtype=np.float64)
    nd = HDv // 2
    for t in range(NT):
        for ic in range(nd):
            stream, k = (0, ic) if ic < HDv // 4 else (1, ic - HDv // 4)
            th = float(pos4[t + stream * NT]) * 10000.0 ** (-2.0 * k / nd)
            c, s_ = np.cos(th), np.sin(th)
            x0, x1 = xv[t, :, ic].astype(np.float64), xv[t, :, ic + nd].astype(np.float64)
            refv[t, :, ic], refv[t, :, ic + nd] = x0 * c - x1 * s_, x0 * s_ + x1 * c
    assert T.nmse(gv, refv) < 1e-5
    # all-zero sections: upstream divides by zero there (llama-box only removed the assertion) -> refused, not guessed
    with pytest.raises(RuntimeError):
        _rope_multi_oracle(H, x, pos4, HD, [0, 0, 0, 0], L.ROPE_MROPE)


def test_logits_ith_and_host_sampler_mirrors():
    """llama_get_logits_ith semantics (batch position / negative = from the end / NULL without logits) and the two host-side
    consumers the reference runs on those rows: common_sampler_sample2 with a greedy chain (sampling.patch:58-81, called at
    httpserver.hpp:4294) and get_token_probabilities (httpserver.hpp:440-467) — here over the CPU oracle."""
    import ctypes as C

    import harness as T
    from model_util import Context, Model, preset

    H = L.host()
    hp = preset("test-llama")
    m = Model(hp, 77, H.ggml_backend_cpu_buffer_type())
    c = Context(m, compute=T.oracle_compute_fn())
    try:
        toks = [3, 9, 27, 81, 243, 5]
        rc, lg = c.decode(toks, range(6), want=[0, 1, 0, 0, 1, 1])
        assert rc == 0 and lg.shape[0] == 3
        nv = hp.n_vocab

        def row(i):
            p = H.llm_get_logits_ith(c.c, i)
            return None if not p else np.ctypeslib.as_array(p, shape=(nv,)).copy()

        assert row(0) is None and row(2) is None and row(6) is None and row(-4) is None  # no logits there / out of range
        assert np.array_equal(row(1), lg[0]) and np.array_equal(row(4), lg[1]) and np.array_equal(row(5), lg[2])
        assert np.array_equal(row(-1), lg[2]) and np.array_equal(row(-3), lg[0])
        assert H.llm_sample_greedy(c.c, 0) == -1
        for pos, r in ((1, 0), (4, 1), (-1, 2)):
            assert H.llm_sample_greedy(c.c, pos) == int(np.argmax(lg[r]))  # first maximal logit
        ids = (C.c_int32 * 8)()
        pr = (C.c_float * 8)()
        assert H.llm_token_probabilities(c.c, 5, 8, ids, pr) == 8
        order = np.argsort(-lg[2], kind="stable")[:8]
        p_ref = np.exp(lg[2].astype(np.float32) - lg[2].max())
        p_ref = (p_ref / p_ref.sum(dtype=np.float32))[order]
        assert list(ids) == order.tolist()
        assert np.allclose(np.array(list(pr)), p_ref, rtol=1e-5)
    finally:
        c.free()
        m.free()


# ------------------------------------------------------------------------------------------------ bench.py's multi-GPU entry point
def _run_bench(args, env_extra=None, timeout=300):
    import subprocess
    import sys as _sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([_sys.executable, os.path.join(repo, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=timeout)


def test_bench_refuses_a_rank_count_other_than_gpus():
    """`--gpus N` under a launcher that started another number of ranks prints NO line and exits non-zero."""
    r = _run_bench(["--gpus", "4", "--steps", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}, timeout=120)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "{" not in r.stdout


def test_bench_gpus_n_starts_n_ranks_itself_and_fails_without_gpus():
    """`python bench.py --gpus 2` with no launcher around it re-executes itself under torch.distributed.run with two ranks; on a box
    without two GPUs every rank refuses, the exit code is non-zero and no JSON line appears (it used to run one rank and print n_gpus 1)."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs here: the real run is the driver's SCALE leg")
    r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--prefill", "0", "--layers", "1"], timeout=300)
    assert r.returncode != 0, r.stdout[-500:]
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln], r.stdout[-500:]
    assert "refusing to run" in r.stderr  # came from the ranks torch.distributed.run started, not from the parent


def test_graph_key_decides_what_a_captured_graph_stands_for():
    """graph.cpp: a captured hipGraph is replayed for a graph whose KEY — per node {type, op, ne, nb, op_params, flags, buffer, data, which sources,
    mask-content hint}, per source {type, buffer, ne, nb, data} — equals the one it was captured for; since round 5 the key of the graph replayed
    last is compared in place (no hash: a collision cannot replay another graph, and no 240 KB byte-serial FNV before every step).  Host arithmetic,
    probed through the registration's proc address without a device: a rebuilt identical graph matches; one more node, another op parameter, another
    shape, another data address or another source do not."""
    import ctypes as C

    import numpy as np

    import llama_box_amd as L

    H = L.host()
    lib = C.CDLL(L.BACKEND_SO)
    lib.ggml_backend_mi355x_reg.restype = C.c_void_p
    reg = lib.ggml_backend_mi355x_reg()
    addr = H.ggml_backend_reg_get_proc_address(reg, b"ggml_backend_mi355x_graph_key_probe")
    assert addr
    probe = C.CFUNCTYPE(C.c_int, C.POINTER(L.CGraph), C.POINTER(L.CGraph), C.POINTER(C.c_int64))(addr)
    keep = []

    def build(scale=0.5, rows=8, extra=False, swap=False, shift=0):
        ctx = H.ggml_init(L.InitParams(0, None, True))
        keep.append(ctx)
        x = H.ggml_new_tensor_4d(ctx, L.F32, 64, rows, 1, 1)
        y = H.ggml_new_tensor_4d(ctx, L.F32, 64, rows, 1, 1)
        w = H.ggml_new_tensor_4d(ctx, L.F32, 64, 1, 1, 1)
        # (addresses are part of the key: the tensors get fixed fake ones — nothing is computed here)
        for i, t in enumerate((x, y, w)):
            t.contents.data = 0x10000 + 0x4000 * i + shift
        a = H.ggml_add(ctx, x, y if not swap else x)
        b = H.ggml_mul(ctx, H.ggml_rms_norm(ctx, a, 1e-5), w)
        c = H.ggml_scale(ctx, b, scale)
        if extra:
            c = H.ggml_scale(ctx, c, 2.0)
        gf = H.ggml_new_graph_custom(ctx, 64, False)
        H.ggml_build_forward_expand(gf, c)
        for i in range(gf.contents.n_nodes):
            gf.contents.nodes[i].contents.data = 0x80000 + 0x4000 * i
        return gf

    n_words = C.c_int64(0)
    g0 = build()
    assert probe(g0, build(), C.byref(n_words)) == 1 and n_words.value > 0          # the same step, rebuilt by the host
    assert n_words.value == 1 + 4 * 20 + 6 * 11                                       # 4 nodes x 20 words + 6 sources x 11 words + the node count
    assert probe(g0, build(extra=True), None) == 0                                    # one more node
    assert probe(g0, build(scale=0.25), None) == 0                                    # another op parameter
    assert probe(g0, build(rows=16), None) == 0                                       # another shape (a grown batch / cache view)
    assert probe(g0, build(shift=256), None) == 0                                     # another input address (the allocator moved a tensor)
    assert probe(g0, build(swap=True), None) == 0                                     # another source tensor in one slot
    for ctx in keep:
        H.ggml_free(ctx)
