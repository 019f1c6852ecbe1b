"""Full-depth end-to-end parity on the models the bench lines are quoted on (VERDICT r04 "Next round" #1).

Rounds 1-4 compared whole models at TinyLlama depth only; Llama-3-8B Q4_K_M (BASELINE config 2 / 3, the timed headline), Qwen2-7B Q5_K_M
(config 5) and Llama-3-70B Q4_K_M (config 4) were validated per layer (teacher-forced per node) and end to end at n_layer = 4.  Here every
layer is there: a prompt through one llama_decode, then teacher-forced decode steps (the oracle's own greedy continuation is fed to every
run, so a near-tie cannot derail the row-by-row comparison), flash attention on and off, logits of every step against the CPU oracle.

The yardstick is north_star's bar read the only way a second implementation can meet it: activations are re-quantised to 8
bits before every mat-mul, so two CORRECT implementations that differ in f32 summation order land a rounding flip apart.  The oracle is
therefore also run as ggml-cpu's x86 kernels sum (oracle_set_fast: the same integer block sums, eight f32 lanes accumulated with FMA —
what llama-box's CPU path executes on an AVX2 host, against the generic scalar order of the reference run) and with an f32 RMS_NORM sum on
top; the GPU's distance from the reference must not exceed 10 x the oracle's distance from itself, must stay inside north_star's absolute
NMSE band, and the greedy id must equal the reference's wherever the top-2 margin exceeds the oracle-vs-oracle deviation.
Reference call site of what is compared: llama_decode at /root/reference/llama-box/httpserver.hpp:3591, logits consumed at :442 / :4294.

Round 6: on those untrained weight sets the oracle sits NMSE 8.7e-4 / max|d| 0.2 from ITSELF at 32 layers — north_star's bar (1e-3 on the logits,
ids exact) is not testable as written there, and almost no position is decisive (7 / 1 / 1 / 1 / 0 in round 5).  The "-damped" weight sets (llama_lite.cpp:
llm_preset — residual branches at gain 0.08 / sqrt(2 n_layer), output rows tied to the embeddings across the Q4_K / Q5_K -> Q6_K formats) have a trained
network's two relevant properties; on them the oracle-vs-oracle distance is <= 1e-5 NMSE, >= 90 % of the positions are decisive, and the gate is the bar AS
WRITTEN: max|d| <= 1e-3 of the logit range, greedy id equal at every decisive position, >= 80 positions per model, flash attention on and off, batch-1 and a
32-sequence continuous batch.  What that buys and what it does not: an error in the embedding / output norm / output matrix path shows at 1e-3; the layers
contribute ~8 % of a damped model's logit spread, so an error INSIDE the layers must reach ~10 % to trip this gate — the per-op (<= 1e-10) and per-layer
teacher-forced gates (tests/test_gpu_ops.py, test_gpu_baseline_shapes.py) and the chaotic sets below (GPU no further from the oracle than 1.5 x the oracle from
itself) carry that.  The chaotic sets stay as the second, looser case."""
import ctypes as C
import os
import time

import numpy as np
import pytest

import harness as T
from model_util import Context, Model, preset

pytestmark = pytest.mark.gpu

NT = T.host_threads(128)


def _run(ctx, toks_prompt, n_par, n_prompt, n_dec, forced=None):
    """One llama_decode over the prompts of n_par sequences (logits at EVERY position), then n_dec decode steps of n_par tokens — free-running greedy when
    `forced` is None (returns the tokens it took), else teacher-forced.  Rows: [n_par * n_prompt + n_dec * n_par, n_vocab]."""
    pos = [i for _ in range(n_par) for i in range(n_prompt)]
    seq = [k for k in range(n_par) for _ in range(n_prompt)]
    rc, lg = ctx.decode(toks_prompt, pos, seq=seq, want=[1] * len(toks_prompt))
    assert rc == 0
    rows, taken = [lg], []
    last = lg[[k * n_prompt + n_prompt - 1 for k in range(n_par)]]
    for i in range(n_dec):
        t = [int(x) for x in np.argmax(last, axis=1)] if forced is None else forced[i]
        taken.append(t)
        rc, last = ctx.decode(t, [n_prompt + i] * n_par, seq=list(range(n_par)))
        assert rc == 0
        rows.append(last)
    return np.concatenate(rows), taken


def _oracle_rows(mc, toks_prompt, n_par, n_prompt, n_dec, fa, n_ctx, fast, variant, forced=None, kv=(0, 0)):
    lib = T.oracle()
    lib.oracle_set_fast.restype = C.c_int
    have = int(lib.oracle_set_fast(fast))
    lib.oracle_set_variant(variant)
    try:
        c = Context(mc, compute=T.oracle_compute_fn(NT), flash_attn=fa, n_ctx=n_ctx, n_threads=NT, type_k=kv[0], type_v=kv[1])
        try:
            rows, taken = _run(c, toks_prompt, n_par, n_prompt, n_dec, forced)
            return rows, taken, have
        finally:
            c.free()
    finally:
        lib.oracle_set_variant(0)
        lib.oracle_set_fast(0)


def full_depth_parity(backend, H, plog, name, fa, n_prompt, n_dec, n_par=1, seed=11, strict=None, ref_fast=False, n_var_dec=0, kv=(0, 0), check=True, n_var_prompt=0):
    """Returns the numbers bench.py's `parity` object is made of too.  `strict` (default: the weight set is a "-damped" one) gates north_star's bar as written.
    ref_fast: the REFERENCE run is the oracle in ggml-cpu's x86 lane order (what llama-box's CPU path executes on an AVX2 host; 3 x cheaper than the generic scalar
    order) and the generic order is the second opinion; otherwise the other way round.  The second opinion runs the prompt batch + n_var_dec steps — or, with
    n_var_prompt > 0 (one sequence), only the prompt's first n_var_prompt tokens: causal attention makes those rows the same computation as the full prompt's first
    rows, and the yardstick (how far the oracle moves under another f32 summation order) does not need all of them."""
    hp = preset(name)
    strict = name.endswith("-damped") if strict is None else strict
    rng = np.random.default_rng(1000 + seed)
    toks_prompt = rng.integers(3, hp.n_vocab, n_par * n_prompt).tolist()
    n_ctx = (n_par * (n_prompt + n_dec) + 255) // 256 * 256
    t0 = time.time()
    mc = Model(hp, seed, H.ggml_backend_cpu_buffer_type())
    mg = Model(hp, seed, backend.buft)
    try:
        t1 = time.time()
        ref, forced, have_fast = _oracle_rows(mc, toks_prompt, n_par, n_prompt, n_dec, fa, n_ctx, 1 if ref_fast else 0, 0, kv=kv)
        if ref_fast and not have_fast:
            ref_fast = False  # (a build without AVX2: the reference was the generic code; its block dots summed in reverse order are the second opinion)
        t2 = time.time()
        vp, vtoks, vdec = n_prompt, toks_prompt, n_var_dec
        if n_var_prompt > 0 and n_par == 1 and n_var_prompt < n_prompt:
            vp, vtoks, vdec = n_var_prompt, toks_prompt[:n_var_prompt], 0
        if ref_fast:
            var, _, _ = _oracle_rows(mc, vtoks, n_par, vp, vdec, fa, n_ctx, 0, 0, forced, kv=kv)
        elif have_fast:
            var, _, _ = _oracle_rows(mc, vtoks, n_par, vp, vdec, fa, n_ctx, 1, 2, forced, kv=kv)  # x86 lane order + f32 RMS_NORM sum
        else:
            var, _, _ = _oracle_rows(mc, vtoks, n_par, vp, vdec, fa, n_ctx, 0, 1, forced, kv=kv)
        t3 = time.time()
        g0 = {k: backend.stat(k) for k in ("graph_launches", "kernel_launches")}
        cg = Context(mg, backend=backend, flash_attn=fa, n_ctx=n_ctx, type_k=kv[0], type_v=kv[1])
        try:
            got, _ = _run(cg, toks_prompt, n_par, n_prompt, n_dec, forced)
        finally:
            cg.free()
        launched = {k: backend.stat(k) - v for k, v in g0.items()}
    finally:
        mc.free()
        mg.free()
    nv = len(var)
    e_gpu, e_var = T.nmse(got, ref), T.nmse(var, ref[:nv])
    d_gpu, d_var = float(np.max(np.abs(got - ref))), float(np.max(np.abs(var - ref[:nv])))
    span = float(ref.max() - ref.min())
    e_rows = [T.nmse(got[i], ref[i]) for i in range(len(ref))]
    top2 = np.sort(ref, axis=1)
    margin = top2[:, -1] - top2[:, -2]
    agree = np.argmax(got, axis=1) == np.argmax(ref, axis=1)
    decisive = margin > 2.0 * d_var
    n_pos = len(ref)
    tag = f"{name} ({hp.n_layer} layers) fa={fa}: {n_par} x {n_prompt}-token prompt (logits at every position) + {n_dec} teacher-forced steps of {n_par}"
    plog(f"[parity-full-depth] {tag}: logits nmse gpu={e_gpu:.3e} (worst row {max(e_rows):.3e}) | oracle-vs-oracle ({'generic scalar order vs the x86 lane order reference' if ref_fast else 'x86 lane order + f32 norm sum vs the generic reference'}, "
         f"{nv} rows) {e_var:.3e} | max|d| gpu={d_gpu:.3e} = {d_gpu / span:.2e} of the logit range {span:.1f}; oracle-vs-oracle={d_var:.3e} = {d_var / span:.2e} | argmax agreement {int(agree.sum())}/{n_pos}, "
         f"{int(decisive.sum())} positions with margin > 2 x oracle-vs-oracle, all equal there: {bool(np.all(agree[decisive]))} | min margin {margin.min():.3e} median {np.median(margin):.3e} | {launched} | "
         f"seconds: models {t1 - t0:.1f}, reference {t2 - t1:.1f}, second opinion {t3 - t2:.1f}")
    assert launched["kernel_launches"] > 0
    out = {"nmse": e_gpu, "max_abs": d_gpu, "max_abs_rel_to_logit_range": d_gpu / span, "argmax_agree": f"{int(agree.sum())}/{n_pos}", "decisive_positions": int(decisive.sum()),
           "argmax_agree_at_decisive_positions": f"{int((agree & decisive).sum())}/{int(decisive.sum())}", "oracle_vs_oracle_nmse": e_var, "oracle_vs_oracle_max_abs": d_var, "positions": n_pos,
           "what": tag + "; reference = the CPU oracle in " + ("ggml-cpu's x86 lane order" if ref_fast else "the generic scalar order") + f", oracle-vs-oracle over its first {nv} rows"}
    if not check:  # (bench.py reports, the tests gate)
        return out
    if strict:
        # north_star as written: logits within 1e-3 (of the logit range), greedy ids exact — on a weight set where that is a property of the implementation, not of chance
        assert e_var <= 2e-5 and int(decisive.sum()) >= 0.9 * n_pos, f"{tag}: the weight set is not damped / peaked enough to carry the bar (oracle-vs-oracle {e_var:.3e}, {int(decisive.sum())}/{n_pos} decisive)"
        assert d_gpu <= 1e-3 * span, f"{tag}: max|d| {d_gpu:.3e} exceeds 1e-3 of the logit range ({span:.2f})"
        assert e_gpu <= max(3.0 * e_var, 1e-10) and e_gpu <= 1e-4, f"{tag}: NMSE {e_gpu:.3e} (oracle from itself: {e_var:.3e})"
        assert bool(np.all(agree[decisive])), f"{tag}: greedy id differs at a decisive position"
        return out
    # the chaotic sets — north_star's absolute band (1e-3) or, where the ORACLE is further than that from itself (measured: 8.7e-4 at 32 layers, 1.8e-3 at 80 layers of
    # these untrained synthetic weights: the Q8 re-quantisation noise of DESIGN.md §2 accumulated over the depth), no further than the oracle is
    assert e_gpu <= max(1e-3, 1.5 * e_var), f"{tag}: beyond north_star's band and beyond 1.5 x the oracle's distance from itself ({e_var:.3e})"
    assert e_gpu <= max(3.0 * e_var, 1e-10), f"{tag}: the GPU is further from the oracle ({e_gpu:.3e}) than 3 x the oracle from itself ({e_var:.3e})"
    assert bool(np.all(agree[decisive])), f"{tag}: greedy id differs at a position whose margin exceeds the oracle-vs-oracle deviation"
    return out


# (name, fa, n_par, n_prompt, n_dec, reference in x86 lane order): >= 80 positions per case; the headline configuration keeps the generic scalar reference
DAMPED = [("llama3-8b-q4_k_m-damped", 1, 1, 64, 32, False), ("llama3-8b-q4_k_m-damped", 0, 1, 64, 32, True), ("llama3-8b-q4_k_m-damped", 1, 32, 2, 2, True),
          ("llama3-8b-q4_k_m-damped", 0, 32, 2, 2, True), ("qwen2-7b-q5_k_m-damped", 1, 1, 64, 32, True), ("qwen2-7b-q5_k_m-damped", 0, 1, 64, 32, True),
          ("qwen2-7b-q5_k_m-damped", 1, 32, 2, 2, True)]


@pytest.mark.parametrize("name,fa,n_par,n_prompt,n_dec,ref_fast", DAMPED)
def test_full_depth_bar_as_written_on_damped_weights(backend, H, plog, name, fa, n_par, n_prompt, n_dec, ref_fast):
    """BASELINE configs 2 / 3 (Llama-3-8B Q4_K_M: batch-1 and -np 32) and 5 (Qwen2-7B Q5_K_M: Q5_K + Q6_K, biases, NeoX rope) at full depth on the damped + peaked weight
    sets: max|d| <= 1e-3 of the logit range, ids equal at every decisive position, 96 / 128 positions per case (prompt positions through the prompt kernels, decode steps
    through the mat-vec / 32-column kernels), flash attention on and off."""
    r = full_depth_parity(backend, H, plog, name, fa, n_prompt=n_prompt, n_dec=n_dec, n_par=n_par, ref_fast=ref_fast, n_var_dec=2 if n_par == 1 else 1, n_var_prompt=32 if ref_fast else 0)
    assert r["decisive_positions"] >= 80 and r["positions"] >= 96


def test_full_depth_bar_as_written_llama3_70b_damped(backend, H, plog):
    """BASELINE config 4's model, all 80 layers on one GPU (42.5 GB of weights on each side), damped + peaked: 64 prompt positions + 24 steps against the oracle in
    x86 lane order (the generic scalar order costs ~5 s per position at this size: it is the second opinion on the prompt's first rows)."""
    n_prompt, n_dec = int(os.environ.get("FULL_DEPTH_70B_PROMPT", "64")), int(os.environ.get("FULL_DEPTH_70B_STEPS", "24"))
    r = full_depth_parity(backend, H, plog, "llama3-70b-q4_k_m-damped", 1, n_prompt=n_prompt, n_dec=n_dec, ref_fast=True, n_var_dec=0, n_var_prompt=16)
    assert r["decisive_positions"] >= min(80, int(0.9 * (n_prompt + n_dec)))


@pytest.mark.parametrize("name,fa,n_prompt,n_dec", [("llama3-8b-q4_k_m", 1, 32, 8), ("qwen2-7b-q5_k_m", 0, 32, 8)])
def test_full_depth_logits_and_ids(backend, H, plog, name, fa, n_prompt, n_dec):
    """The untrained (chaotic) weight sets the bench lines of rounds 1-5 were quoted on, as the second, looser case: the GPU no further from the oracle than 1.5 x
    the oracle from itself (FULL_DEPTH_LONG=1: 64 + 16 positions)."""
    if os.environ.get("FULL_DEPTH_LONG") == "1":
        n_prompt, n_dec = 64, 16
    full_depth_parity(backend, H, plog, name, fa, n_prompt=n_prompt, n_dec=n_dec, n_var_dec=n_dec)


@pytest.mark.skipif(os.environ.get("FULL_DEPTH_LONG") != "1", reason="the chaotic 70B case (45 s: two more 42 GB models) runs with FULL_DEPTH_LONG=1; the damped 70B case above runs always")
def test_full_depth_llama3_70b(backend, H, plog):
    """Config 4's model on the chaotic set: a short prompt and a few steps."""
    full_depth_parity(backend, H, plog, "llama3-70b-q4_k_m", 1, n_prompt=6, n_dec=3, ref_fast=True, n_var_dec=3)
