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
Reference call site of what is compared: llama_decode at /root/reference/llama-box/httpserver.hpp:3591, logits consumed at :442 / :4294."""
import ctypes as C
import os
import time

import numpy as np
import pytest

import harness as T
from model_util import Context, Model, preset

pytestmark = pytest.mark.gpu

NT = T.host_threads(128)


def _forced_rows(ctx, prompt, forced):
    rc, lg = ctx.decode(prompt, range(len(prompt)), want=[0] * (len(prompt) - 1) + [1])
    assert rc == 0
    rows = [lg[-1]]
    for i, t in enumerate(forced):
        rc, l1 = ctx.decode([t], [len(prompt) + i])
        assert rc == 0
        rows.append(l1[0])
    return np.stack(rows)


def _greedy_reference(mc, prompt, n_dec, fa, n_ctx):
    """The reference run: generic scalar oracle, free-running greedy — its logits rows and the tokens every other run is forced through."""
    c = Context(mc, compute=T.oracle_compute_fn(NT), flash_attn=fa, n_ctx=n_ctx, n_threads=NT)
    try:
        rc, lg = c.decode(prompt, range(len(prompt)), want=[0] * (len(prompt) - 1) + [1])
        assert rc == 0
        rows, toks, row = [], [], lg[-1]
        for i in range(n_dec):
            rows.append(row)
            toks.append(int(np.argmax(row)))
            rc, l1 = c.decode([toks[-1]], [len(prompt) + i])
            assert rc == 0
            row = l1[0]
        rows.append(row)
        return np.stack(rows), toks
    finally:
        c.free()


def _variant_rows(mc, prompt, forced, fa, n_ctx, fast, variant):
    lib = T.oracle()
    lib.oracle_set_fast.restype = C.c_int
    have = int(lib.oracle_set_fast(fast))
    lib.oracle_set_variant(variant)
    try:
        c = Context(mc, compute=T.oracle_compute_fn(NT), flash_attn=fa, n_ctx=n_ctx, n_threads=NT)
        try:
            return _forced_rows(c, prompt, forced), have
        finally:
            c.free()
    finally:
        lib.oracle_set_variant(0)
        lib.oracle_set_fast(0)


def full_depth_parity(backend, H, plog, name, fa, n_prompt, n_dec, seed=11):
    """Returns the numbers bench.py's `parity` object is made of too (same procedure on the bench's own model)."""
    hp = preset(name)
    rng = np.random.default_rng(1000 + seed)
    prompt = rng.integers(3, hp.n_vocab, n_prompt).tolist()
    n_ctx = 256
    t0 = time.time()
    mc = Model(hp, seed, H.ggml_backend_cpu_buffer_type())
    mg = Model(hp, seed, backend.buft)
    try:
        t1 = time.time()
        ref, forced = _greedy_reference(mc, prompt, n_dec, fa, n_ctx)
        t2 = time.time()
        v_fast, have_fast = _variant_rows(mc, prompt, forced, fa, n_ctx, 1, 0)
        v_norm, _ = _variant_rows(mc, prompt, forced, fa, n_ctx, 1, 2)
        if not have_fast:  # (a build without AVX2: the generic code with its block dots summed in reverse order is the second opinion)
            v_fast, _ = _variant_rows(mc, prompt, forced, fa, n_ctx, 0, 1)
        t3 = time.time()
        g0 = {k: backend.stat(k) for k in ("graph_launches", "kernel_launches")}
        cg = Context(mg, backend=backend, flash_attn=fa, n_ctx=n_ctx)
        try:
            got = _forced_rows(cg, prompt, forced)
        finally:
            cg.free()
        launched = {k: backend.stat(k) - v for k, v in g0.items()}
    finally:
        mc.free()
        mg.free()
    e_gpu = T.nmse(got, ref)
    e_var = max(T.nmse(v_fast, ref), T.nmse(v_norm, ref))
    d_gpu = float(np.max(np.abs(got - ref)))
    d_var = max(float(np.max(np.abs(v_fast - ref))), float(np.max(np.abs(v_norm - ref))))
    e_rows = [T.nmse(got[i], ref[i]) for i in range(len(ref))]
    top2 = np.sort(ref, axis=1)
    margin = top2[:, -1] - top2[:, -2]
    agree = np.argmax(got, axis=1) == np.argmax(ref, axis=1)
    decisive = margin > 2.0 * d_var
    tag = f"{name} ({hp.n_layer} layers) fa={fa}: {n_prompt}-token prompt + {n_dec} teacher-forced steps"
    plog(f"[parity-full-depth] {tag}: logits nmse gpu={e_gpu:.3e} (worst row {max(e_rows):.3e}) | oracle-vs-oracle (x86 lane order / + f32 norm sum) {e_var:.3e} | max|d| gpu={d_gpu:.3e} "
         f"oracle-vs-oracle={d_var:.3e} | argmax agreement {int(agree.sum())}/{len(agree)}, {int(decisive.sum())} positions with margin > 2 x oracle-vs-oracle, all equal there: "
         f"{bool(np.all(agree[decisive]))} | min margin {margin.min():.3e} | {launched} | seconds: models {t1 - t0:.1f}, reference {t2 - t1:.1f}, variants {t3 - t2:.1f}")
    assert launched["kernel_launches"] > 0
    # north_star's absolute band (1e-3) — or, where the ORACLE is further than that from itself (measured: 8.7e-4 at 32 layers, 1.8e-3 at 80 layers of
    # these untrained synthetic weights: the Q8 re-quantisation noise of DESIGN.md §2 accumulated over the depth), no further than the oracle is
    assert e_gpu <= max(1e-3, 1.5 * e_var), f"{tag}: beyond north_star's band and beyond 1.5 x the oracle's distance from itself ({e_var:.3e})"
    assert e_gpu <= max(3.0 * e_var, 1e-10), f"{tag}: the GPU is further from the oracle ({e_gpu:.3e}) than 3 x the oracle from itself ({e_var:.3e})"
    assert bool(np.all(agree[decisive])), f"{tag}: greedy id differs at a position whose margin exceeds the oracle-vs-oracle deviation"
    return {"nmse": e_gpu, "max_abs": d_gpu, "argmax_agree": f"{int(agree.sum())}/{len(agree)}", "oracle_vs_oracle_nmse": e_var, "oracle_vs_oracle_max_abs": d_var}


@pytest.mark.parametrize("name,fa,n_prompt,n_dec", [("llama3-8b-q4_k_m", 1, 64, 16), ("llama3-8b-q4_k_m", 0, 32, 8), ("qwen2-7b-q5_k_m", 1, 32, 8), ("qwen2-7b-q5_k_m", 0, 32, 8)])
def test_full_depth_logits_and_ids(backend, H, plog, name, fa, n_prompt, n_dec):
    """BASELINE configs 2 / 3 (Llama-3-8B Q4_K_M, 32 layers) and 5 (Qwen2-7B Q5_K_M: Q5_K + Q6_K, biases, NeoX rope, 28 layers) at full depth.  The
    headline configuration (flash attention on) runs the 64 + 16 positions VERDICT r04 asks for (~60 s, of which the scalar reference and its two
    variants are 55); the other three run 32 + 8 (FULL_DEPTH_LONG=1: 64 + 16 everywhere)."""
    if os.environ.get("FULL_DEPTH_LONG") == "1":
        n_prompt, n_dec = 64, 16
    full_depth_parity(backend, H, plog, name, fa, n_prompt=n_prompt, n_dec=n_dec)


def test_full_depth_llama3_70b(backend, H, plog):
    """BASELINE config 4's model, all 80 layers on one GPU (42.5 GB of weights on each side): a short prompt and a few steps — the scalar
    reference costs ~10 x the 8B's per token."""
    full_depth_parity(backend, H, plog, "llama3-70b-q4_k_m", 1, n_prompt=int(os.environ.get("FULL_DEPTH_70B_PROMPT", "6")), n_dec=int(os.environ.get("FULL_DEPTH_70B_STEPS", "3")))
