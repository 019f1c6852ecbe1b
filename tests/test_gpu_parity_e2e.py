"""End-to-end parity evidence (VERDICT r01 "Next round" #1 d, e): what the logits of a whole model do, and WHY.

north_star: logits within 1e-3 of the CPU backend, greedy ids identical.  Per op and per real-shape layer group that bar is
met with orders of magnitude to spare (test_gpu_ops.py, test_gpu_baseline_shapes.py: NMSE <= 1e-10 on identical inputs).
End to end a network that re-quantises its activations to 8 bits before every mat-mul is DISCONTINUOUS in its inputs: a
one-ulp difference in f32 summation order now and then moves a value across a rounding boundary, and the flip is worth 1/127
of a block's range.  This file turns that explanation into tested facts:

  1. test_e2e_deviation_matches_oracle_order_sensitivity — 20 seeds x {GPU, oracle with reversed block order, oracle with an
     f32 RMS_NORM sum, oracle with expf one ulp off}: the GPU's distance from the oracle is distributed like the oracle's
     distance from ITSELF under those one-ulp changes, flip-free runs are <= 1e-10, and the greedy id equals the oracle's
     wherever the top-2 margin exceeds the oracle-vs-oracle deviation (a yardstick the GPU has no part in).
  2. every GPU run above 1e-10 is CLASSIFIED: the step's graph is re-run with every intermediate kept, the first node whose
     output deviates while its inputs agree is located, and the deviation is attributed — for a quantised MUL_MAT by counting
     the Q8 activation codes that differ between the two inputs and by checking that the oracle, fed the GPU's input, lands
     on the GPU's output (<= 1e-10); for an f16 cache store by an f16 rounding.  Anything else fails the test.
  3. test_config1_tinyllama_greedy_128 — BASELINE config 1 as written: TinyLlama-1.1B-shaped Q8_0 model, 16-token prompt, 128
     greedy tokens; first divergence and its margin are logged; teacher-forced logits are gated like (1)."""
import ctypes as C
import os

import numpy as np
import pytest

import harness as T
import llama_box_amd as L
from model_util import Context, Model, preset

pytestmark = pytest.mark.gpu

PROMPT = [1, 5, 9, 300, 17, 42, 99, 7]
N_DEC = 16
VIEW_OPS = None


def _rows(ctx, prompt, forced):
    """Logits of the last prompt position, then of every teacher-forced step (the tokens are given, not sampled)."""
    rc, lg = ctx.decode(prompt, range(len(prompt)), want=[0] * (len(prompt) - 1) + [1])
    assert rc == 0
    rows = [lg[-1]]
    for i, t in enumerate(forced):
        rc, l1 = ctx.decode([t], [len(prompt) + i])
        assert rc == 0
        rows.append(l1[0])
    return np.stack(rows)


def _oracle_rows(mc, variant, fa, forced, **kw):
    T.oracle().oracle_set_variant(variant)
    try:
        c = Context(mc, compute=T.oracle_compute_fn(T.host_threads(16)), flash_attn=fa, **kw)
        r = _rows(c, PROMPT, forced)
        c.free()
        return r
    finally:
        T.oracle().oracle_set_variant(0)


def _greedy_forced(mc, fa, n):
    """The oracle's own greedy continuation (variant 0): the token sequence every other run is forced through."""
    c = Context(mc, compute=T.oracle_compute_fn(T.host_threads(16)), flash_attn=fa)
    rc, lg = c.decode(PROMPT, range(len(PROMPT)), want=[0] * (len(PROMPT) - 1) + [1])
    toks, row = [], lg[-1]
    for i in range(n):
        toks.append(int(np.argmax(row)))
        rc, l1 = c.decode([toks[-1]], [len(PROMPT) + i])
        row = l1[0]
    c.free()
    return toks


# ------------------------------------------------------------------------------------------------ classification of a deviation
def _tensor_bytes(H, t):
    n = H.ggml_nbytes(t)
    raw = np.empty(n, np.uint8)
    H.ggml_backend_tensor_get(t, raw.ctypes.data_as(C.c_void_p), 0, n)
    return raw


def _as_float(H, t):
    tt = t.contents
    raw = _tensor_bytes(H, t)
    if tt.type == L.F32:
        return raw.view(np.float32).astype(np.float64)
    if tt.type == L.F16:
        return raw.view(np.float16).astype(np.float64)
    return None


def _is_view(op_name):
    return op_name in ("NONE", "RESHAPE", "VIEW", "PERMUTE", "TRANSPOSE")


def _classify_first_deviation(H, gc, gg, plog, tag):
    """gc / gg: the same graph evaluated by the oracle and by the backend with every intermediate kept (no allocator reuse,
    no fusion).  Returns a verdict string; raises AssertionError when the first deviating node is not explained by a rounding flip."""
    lib = T.oracle()
    n = gc.n_nodes
    assert n == gg.n_nodes
    dev = {}

    def nm(i, which):
        a, b = gc.nodes[i], gg.nodes[i]
        va, vb = _as_float(H, a), _as_float(H, b)
        if va is None:
            return 0.0
        ok = np.isfinite(va) & np.isfinite(vb)
        return T.nmse(vb[ok], va[ok])

    for i in range(n):
        tc = gc.nodes[i].contents
        op = H.ggml_op_name(tc.op).decode()
        if _is_view(op) or tc.type not in (L.F32, L.F16):
            continue
        e = nm(i, "out")
        dev[i] = e
        if e <= 1e-11:  # (flip-free nodes sit at 1e-14; a flip in a small block can be worth as little as 1e-11)
            continue
        name = tc.name.decode()
        # inputs of this node on both sides
        src_dev = []
        for s in range(10):
            if not tc.src[s]:
                continue
            sa, sb = tc.src[s], gg.nodes[i].contents.src[s]
            va, vb = _as_float(H, sa), _as_float(H, sb)
            if va is not None and va.shape == vb.shape:
                ok = np.isfinite(va) & np.isfinite(vb)
                src_dev.append(T.nmse(vb[ok], va[ok]))
        worst_in = max(src_dev) if src_dev else 0.0
        if worst_in * 30.0 > e:
            continue  # this node only carries a deviation that entered earlier; the JUMP is what gets classified
        if op == "MUL_MAT" and tc.src[0].contents.type in (L.Q4_K, L.Q5_K, L.Q6_K, L.Q8_0):
            w = tc.src[0].contents
            K = int(w.ne[0])
            xa = _tensor_bytes(H, tc.src[1]).view(np.float32)
            xb = _tensor_bytes(H, gg.nodes[i].contents.src[1]).view(np.float32)
            if w.type == L.Q8_0:
                qa, qb = np.zeros(xa.size // 32 * 34, np.uint8), np.zeros(xa.size // 32 * 34, np.uint8)
                lib.oracle_quantize_row_q8_0(xa.ctypes.data_as(C.c_void_p), qa.ctypes.data_as(C.c_void_p), xa.size)
                lib.oracle_quantize_row_q8_0(np.ascontiguousarray(xb).ctypes.data_as(C.c_void_p), qb.ctypes.data_as(C.c_void_p), xb.size)
                codes_a, codes_b = qa.reshape(-1, 34)[:, 2:], qb.reshape(-1, 34)[:, 2:]
            else:
                qa, qb = np.zeros(xa.size // 256 * 292, np.uint8), np.zeros(xa.size // 256 * 292, np.uint8)
                lib.oracle_quantize_row_q8_K(xa.ctypes.data_as(C.c_void_p), qa.ctypes.data_as(C.c_void_p), xa.size)
                lib.oracle_quantize_row_q8_K(np.ascontiguousarray(xb).ctypes.data_as(C.c_void_p), qb.ctypes.data_as(C.c_void_p), xb.size)
                codes_a, codes_b = qa.reshape(-1, 292)[:, 4:260], qb.reshape(-1, 292)[:, 4:260]
            flips = int(np.count_nonzero(codes_a != codes_b))
            # the oracle, fed the GPU's activations, must land on the GPU's output: the op itself is exact on its own input
            wb = _tensor_bytes(H, tc.src[0])
            N, M = int(w.ne[1]), xb.size // K

            def build(g):
                return H.ggml_mul_mat(g.ctx, g.new(w.type, [K, N], wb), g.new(L.F32, [K, M], xb.reshape(M, K)))

            redo = T.run_case(build, "oracle", T.host_threads(16))[0].ravel().astype(np.float64)
            got = _as_float(H, gg.nodes[i])
            e_own = T.nmse(got, redo)
            verdict = (f"{tag}: first deviation at node {i} '{name}' (MUL_MAT {w.type}): inputs nmse {worst_in:.1e}, output nmse {e:.1e}; "
                       f"{flips} Q8 activation code(s) differ between the two inputs; oracle on the GPU's input vs GPU output nmse {e_own:.1e}")
            plog("[parity-classify] " + verdict)
            assert flips >= 1 and e_own <= 1e-10, "unexplained deviation: " + verdict
            return "q8-flip"
        if op == "MUL_MAT" and tc.src[0].contents.type == L.F16:
            # f16 weights (the K / V cache on the soft-max path): src1 is rounded to f16 first (vec_dot_type), an ulp apart in f32
            # can be a whole f16 step apart after the rounding
            xa = _tensor_bytes(H, tc.src[1]).view(np.float32).astype(np.float16)
            xb = _tensor_bytes(H, gg.nodes[i].contents.src[1]).view(np.float32).astype(np.float16)
            flips = int(np.count_nonzero(xa.view(np.uint16) != xb.view(np.uint16)))
            verdict = f"{tag}: first deviation at node {i} '{name}' (MUL_MAT f16): inputs nmse {worst_in:.1e}, output nmse {e:.1e}; {flips} f16 activation code(s) differ"
            plog("[parity-classify] " + verdict)
            assert flips >= 1, "unexplained deviation: " + verdict
            return "f16-flip"
        if op in ("SET_ROWS", "CPY", "CONT", "DUP") and tc.type == L.F16:
            verdict = f"{tag}: first deviation at node {i} '{name}' ({op} -> f16): inputs nmse {worst_in:.1e}, output nmse {e:.1e}: f16 rounding of a value that differs by an ulp"
            plog("[parity-classify] " + verdict)
            return "f16-flip"
        verdict = f"{tag}: first deviation at node {i} '{name}' ({op}): inputs nmse {worst_in:.1e}, output nmse {e:.1e}"
        plog("[parity-classify] UNEXPLAINED " + verdict)
        raise AssertionError("unexplained deviation: " + verdict)
    return "none"


def _classify_run(H, backend, hp, seed, forced, fused_rows, plog, tag):
    """Re-runs the sequence on both sides with every intermediate kept and fusion off, checks that the unfused GPU run reproduces the
    fused one bit for bit, and classifies the first step whose logits deviate."""
    H.ggml_lite_set_no_reuse(1)
    backend.set_option("fusion", 0)
    backend.set_option("graphs", 0)
    mc = Model(hp, seed, H.ggml_backend_cpu_buffer_type())
    mg = Model(hp, seed, backend.buft)
    cc = Context(mc, compute=T.oracle_compute_fn(T.host_threads(16)), flash_attn=0, graph_reuse=0)
    cg = Context(mg, backend=backend, flash_attn=0, graph_reuse=0)
    try:
        steps = [(PROMPT, list(range(len(PROMPT))), [0] * (len(PROMPT) - 1) + [1])] + [([t], [len(PROMPT) + i], None) for i, t in enumerate(forced)]
        for si, (toks, pos, want) in enumerate(steps):
            rc, a = cc.decode(toks, pos, want=want)
            rc2, b = cg.decode(toks, pos, want=want)
            assert rc == 0 and rc2 == 0
            same = np.array_equal(b[-1].view(np.uint32), fused_rows[si].view(np.uint32))
            if T.nmse(b[-1], a[-1]) > 1e-10:
                v = _classify_first_deviation(H, H.llm_last_graph(cc.c).contents, H.llm_last_graph(cg.c).contents, plog, f"{tag} step {si}")
                plog(f"[parity-classify] {tag} step {si}: the node-by-node run " + ("reproduces the fused run bit for bit" if same else "differs from the fused run (another summation order: merged launches / K splits)"))
                return v
        # the node-by-node run never left the 1e-10 band although the fused one did: the two differ in summation order only
        plog(f"[parity-classify] {tag}: node-by-node run is flip-free; the fused run's deviation depends on its summation order")
        return "order-dependent"
    finally:
        backend.set_option("fusion", 1)
        backend.set_option("graphs", 1)
        H.ggml_lite_set_no_reuse(0)
        for o in (cc, cg, mc, mg):
            o.free()


# ------------------------------------------------------------------------------------------------ (e) seeds x variants
@pytest.mark.parametrize("name", ["test-llama", "test-qwen2"])
def test_e2e_deviation_matches_oracle_order_sensitivity(backend, H, plog, name):
    hp = preset(name)
    n_seeds = 20
    e_gpu, e_var, dev_var, per_seed = [], [], [], []
    for seed in range(n_seeds):
        mc = Model(hp, 5000 + seed, H.ggml_backend_cpu_buffer_type())
        mg = Model(hp, 5000 + seed, backend.buft)
        try:
            forced = _greedy_forced(mc, 0, N_DEC)
            ref = _oracle_rows(mc, 0, 0, forced)
            var = [_oracle_rows(mc, v, 0, forced) for v in (1, 2, 3)]
            cg = Context(mg, backend=backend, flash_attn=0)
            got = _rows(cg, PROMPT, forced)
            cg.free()
        finally:
            mc.free(); mg.free()
        ev = [T.nmse(v, ref) for v in var]
        eg = T.nmse(got, ref)
        e_var += ev
        e_gpu.append(eg)
        dev_var.append(max(float(np.max(np.abs(v - ref))) for v in var))
        top2 = np.sort(ref, axis=1)
        per_seed.append((seed, forced, got, ref, top2[:, -1] - top2[:, -2], eg))
        plog(f"[parity-e2e] {name} seed {seed}: nmse gpu={eg:.3e} | oracle variants rev-blocks={ev[0]:.3e} f32-norm-sum={ev[1]:.3e} expf+1ulp={ev[2]:.3e} | max|d| gpu={np.max(np.abs(got - ref)):.3e} variants={dev_var[-1]:.3e}")
    e_gpu_a, e_var_a = np.array(e_gpu), np.array(e_var)
    flip_free_gpu, flip_free_var = float(np.mean(e_gpu_a <= 1e-10)), float(np.mean(e_var_a <= 1e-10))
    q = lambda a, p: float(np.quantile(a, p))
    plog(f"[parity-e2e] {name}: {n_seeds} seeds — GPU nmse median {q(e_gpu_a, .5):.2e} p90 {q(e_gpu_a, .9):.2e} max {e_gpu_a.max():.2e}, flip-free {flip_free_gpu:.0%}; "
         f"oracle-vs-oracle ({len(e_var)} runs) median {q(e_var_a, .5):.2e} p90 {q(e_var_a, .9):.2e} max {e_var_a.max():.2e}, flip-free {flip_free_var:.0%}")
    # 1. the GPU is not further from the oracle than the oracle is from itself under one-ulp changes
    assert e_gpu_a.max() <= max(10.0 * e_var_a.max(), 1e-10)
    assert q(e_gpu_a, .5) <= max(10.0 * q(e_var_a, .75), 1e-10)
    assert e_gpu_a.max() <= 1e-3  # north_star's absolute band, for the record
    # 2. greedy ids: equal wherever the oracle's top-2 margin exceeds the oracle-vs-oracle deviation of this model (pooled over seeds)
    yard = 2.0 * max(dev_var)
    n_checked = n_total = 0
    for seed, forced, got, ref, margin, eg in per_seed:
        decisive = margin > yard
        agree = np.argmax(got, axis=1) == np.argmax(ref, axis=1)
        n_checked += int(decisive.sum())
        n_total += len(margin)
        assert bool(np.all(agree[decisive])), f"{name} seed {seed}: greedy id differs where the margin ({margin[~agree].min():.3e}) exceeds the oracle-vs-oracle deviation ({yard:.3e})"
    plog(f"[parity-e2e] {name}: greedy ids identical at all {n_checked}/{n_total} positions whose top-2 margin exceeds the oracle-vs-oracle deviation {yard:.3e}")
    # 3. every run above 1e-10 is classified (up to 6 per model: each costs a full re-run with all intermediates kept)
    classified = {}
    for seed, forced, got, ref, margin, eg in per_seed:
        if eg > 1e-10 and len(classified) < 6:
            classified[seed] = _classify_run(H, backend, hp, 5000 + seed, forced, got, plog, f"{name} seed {seed}")
            assert classified[seed] in ("q8-flip", "f16-flip", "earlier-cache-flip", "order-dependent"), classified
    plog(f"[parity-e2e] {name}: classified deviations {classified}")


# ------------------------------------------------------------------------------------------------ (d) BASELINE config 1 as written
@pytest.mark.parametrize("name", ["tinyllama-1.1b-q8_0", "tinyllama-1.1b-q8_0-peaked"])
def test_config1_tinyllama_greedy_128(backend, H, plog, name):
    """TinyLlama-1.1B Q8_0 shape (22 layers, 2048, 32/4 heads of 64, ffn 5632, vocab 32000), 16-token prompt, 128 greedy tokens:
    the reference's own CPU-runnable configuration (BASELINE.json configs[0]) — here with the GPU against the CPU oracle.

    Two weight sets of that shape.  Independent random weights give near-flat logits (top-2 margins down to 1e-4, below the
    distance between two correct implementations), so the id gate can only cover the decisive positions.  The "-peaked" set
    (llm_hparams::peaked: 3 of every 8 blocks of output.weight row r repeat token_embd row (7919 r + 13) mod n_vocab) has the
    logit shape of a trained model — one row well ahead — while the layers still move every logit: there north_star's bar is
    applied as written: ALL 128 greedy ids identical, teacher-forced and free-running."""
    hp = preset(name)
    peaked = hp.peaked > 0
    rng = np.random.default_rng(16)
    prompt = rng.integers(3, hp.n_vocab, 16).tolist()
    n_gen = 128
    nt = T.host_threads(128)
    mc = Model(hp, 1, H.ggml_backend_cpu_buffer_type())
    mg = Model(hp, 1, backend.buft)
    fa = 0  # llama-box's default (engine_param.hpp:772-779: flash attention only with -fa)
    cc = Context(mc, compute=T.oracle_compute_fn(nt), flash_attn=fa, n_ctx=256)
    cg = Context(mg, backend=backend, flash_attn=fa, n_ctx=256)
    cf = Context(mg, backend=backend, flash_attn=fa, n_ctx=256)
    try:
        # oracle: free-running greedy
        rc, lg = cc.decode(prompt, range(16), want=[0] * 15 + [1])
        assert rc == 0
        ids_ref, rows_ref, row = [], [], lg[-1]
        for i in range(n_gen):
            ids_ref.append(int(np.argmax(row)))
            rows_ref.append(row)
            rc, l1 = cc.decode([ids_ref[-1]], [16 + i])
            assert rc == 0
            row = l1[0]
        rows_ref = np.stack(rows_ref)

        def forced_rows(ctx):
            rc, lg = ctx.decode(prompt, range(16), want=[0] * 15 + [1])
            assert rc == 0
            rows = [lg[-1]]
            for i, t in enumerate(ids_ref[:-1]):
                rc, l1 = ctx.decode([t], [16 + i])
                assert rc == 0
                rows.append(l1[0])
            return np.stack(rows)

        # the yardstick: the oracle against itself with its block dots summed in reverse order, same forced tokens
        T.oracle().oracle_set_variant(1)
        try:
            cv = Context(mc, compute=T.oracle_compute_fn(nt), flash_attn=fa, n_ctx=256)
            rows_var = forced_rows(cv)
            cv.free()
        finally:
            T.oracle().oracle_set_variant(0)
        # GPU: teacher-forced through the oracle's tokens (row-by-row comparison stays meaningful after a near-tie) ...
        rows_got = forced_rows(cg)
        # ... and free-running, as llama-box would drive it
        rc, lg = cf.decode(prompt, range(16), want=[0] * 15 + [1])
        ids_free, row = [], lg[-1]
        for i in range(n_gen):
            ids_free.append(int(np.argmax(row)))
            rc, l1 = cf.decode([ids_free[-1]], [16 + i])
            row = l1[0]
        e, e_var = T.nmse(rows_got, rows_ref), T.nmse(rows_var, rows_ref)
        dmax, dvar = np.max(np.abs(rows_got - rows_ref), axis=1), np.max(np.abs(rows_var - rows_ref), axis=1)
        top2 = np.sort(rows_ref, axis=1)
        margin = top2[:, -1] - top2[:, -2]
        agree = np.argmax(rows_got, axis=1) == np.array(ids_ref)
        agree_var = np.argmax(rows_var, axis=1) == np.array(ids_ref)
        first = next((i for i, (a, b) in enumerate(zip(ids_ref, ids_free)) if a != b), None)
        plog(f"[parity-e2e] config 1 ({name}, 16-token prompt, 128 greedy, flash_attn off): teacher-forced logits nmse gpu={e:.3e} "
             f"(oracle reversed-blocks {e_var:.3e}) max|d| gpu={dmax.max():.3e} (oracle variant {dvar.max():.3e}); argmax agreement gpu {int(agree.sum())}/{n_gen}, "
             f"oracle variant {int(agree_var.sum())}/{n_gen}; min margin {margin.min():.3e}; free-running ids "
             + ("identical for all 128 tokens" if first is None else f"first differ at token {first}: oracle margin there {margin[first]:.3e}, |d| there {dmax[first]:.3e}"))
        assert e <= 1e-3 and e <= max(10.0 * e_var, 1e-10)
        # greedy ids: identical wherever the margin exceeds the oracle-vs-oracle deviation (a yardstick the GPU has no part in)
        yard = 2.0 * float(dvar.max())
        decisive = margin > yard
        assert bool(np.all(agree[decisive])), f"greedy id differs at a decisive margin (> {yard:.3e})"
        plog(f"[parity-e2e] config 1 ({name}): greedy ids identical at all {int(decisive.sum())}/{n_gen} positions whose margin exceeds the oracle-vs-oracle deviation {yard:.3e} "
             f"({decisive.mean():.0%} of the positions gated; margin >= 10 x the deviation at {np.mean(margin > 5.0 * yard):.0%})")
        if peaked:
            # the bar as north_star writes it: every one of the 128 ids, on logits whose margins dwarf the distance between two correct implementations
            assert np.mean(margin > 5.0 * yard) >= 0.95, f"the peaked weight set is not peaked: margin >= 10 x deviation at only {np.mean(margin > 5.0 * yard):.0%} of the positions"
            assert decisive.mean() >= 0.90
            assert bool(np.all(agree)), f"teacher-forced greedy ids differ at positions {np.nonzero(~agree)[0].tolist()}"
            assert first is None, f"free-running greedy ids diverge at token {first}"
            assert len(set(ids_ref)) > n_gen // 2  # (a real sequence, not a fixed point)
        if first is not None:
            assert margin[first] <= yard, f"free-running ids diverge at token {first} with a decisive margin {margin[first]:.3e} > {yard:.3e}"
    finally:
        for o in (cc, cg, cf, mc, mg):
            o.free()
