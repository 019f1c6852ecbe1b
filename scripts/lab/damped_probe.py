"""CPU-only probe (oracle vs oracle): how far the oracle sits from ITSELF under its lane-order / f32-norm variants on a weight set, and how many
positions are decisive.  Usage: python scripts/lab/damped_probe.py <preset> [n_prompt] [n_dec] [branch_gain] [peaked]"""
import ctypes as C
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import harness as T
import llama_box_amd as L
from model_util import Context, Model, preset

name = sys.argv[1]
n_prompt = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n_dec = int(sys.argv[3]) if len(sys.argv) > 3 else 4
over = {}
if len(sys.argv) > 4:
    over["branch_gain"] = float(sys.argv[4])
if len(sys.argv) > 5:
    over["peaked"] = int(sys.argv[5])
fa = int(os.environ.get("FA", "1"))
NT = T.host_threads(128)
H = L.host()
hp = preset(name, **over)
rng = np.random.default_rng(1011)
prompt = rng.integers(3, hp.n_vocab, n_prompt).tolist()
t0 = time.time()
mc = Model(hp, 11, H.ggml_backend_cpu_buffer_type())
print(f"model {time.time() - t0:.1f}s peaked={hp.peaked} branch_gain={hp.branch_gain:.4f}", flush=True)


def rows_of(fast, variant, forced=None):
    lib = T.oracle()
    lib.oracle_set_fast.restype = C.c_int
    lib.oracle_set_fast(fast)
    lib.oracle_set_variant(variant)
    try:
        c = Context(mc, compute=T.oracle_compute_fn(NT), flash_attn=fa, n_ctx=256, n_threads=NT)
        rc, lg = c.decode(prompt, range(n_prompt), want=[1] * n_prompt)
        assert rc == 0
        rows, toks = list(lg), []
        for i in range(n_dec):
            t = int(np.argmax(rows[-1])) if forced is None else forced[i]
            toks.append(t)
            rc, l1 = c.decode([t], [n_prompt + i])
            assert rc == 0
            rows.append(l1[0])
        c.free()
        return np.stack(rows), toks
    finally:
        lib.oracle_set_variant(0)
        lib.oracle_set_fast(0)


t0 = time.time()
ref, forced = rows_of(0, 0)
print(f"reference {time.time() - t0:.1f}s", flush=True)
t0 = time.time()
v1, _ = rows_of(1, 0, forced)
v2, _ = rows_of(1, 2, forced)
print(f"variants {time.time() - t0:.1f}s", flush=True)
e = max(T.nmse(v1, ref), T.nmse(v2, ref))
d = max(float(np.max(np.abs(v1 - ref))), float(np.max(np.abs(v2 - ref))))
srt = np.sort(ref, axis=1)
margin = srt[:, -1] - srt[:, -2]
rng_l = float(ref.max() - ref.min())
print(f"{name} {over}: oracle-vs-oracle nmse {e:.3e} max|d| {d:.3e} logit range {rng_l:.2f} rel {d / rng_l:.2e} | decisive (margin > 2 d) {int((margin > 2 * d).sum())}/{len(margin)} "
      f"| margins min {margin.min():.3e} median {np.median(margin):.3e} | std of logits {ref.std():.3f} | ids agree v1 {int((v1.argmax(1) == ref.argmax(1)).sum())} v2 {int((v2.argmax(1) == ref.argmax(1)).sum())}")
for nm, v in (("v1", v1), ("v2", v2)):
    pr = np.max(np.abs(v - ref), axis=1)
    print(nm, "per-row max|d|:", " ".join(f"{x:.1e}" for x in pr))
