"""LAB: in-kernel timeline of the decode attention split kernel (csrc/fattn.hip built with -DFA_STAMP=1 as llama_box_amd/ab/fa_stamp.so).
   GGML_BACKEND_PATH=llama_box_amd/ab/fa_stamp.so python scripts/lab/fa_stamps.py [n_past]"""
import ctypes as C
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import llama_box_amd as L
from model_util import Context, Model, preset

n_past = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
be = L.Backend(0)
hp = preset("llama3-8b-q4_k_m-damped", n_layer=4)
m = Model(hp, 1, be.buft)
n_ctx = (n_past + 64 + 255) // 256 * 256
c = Context(m, backend=be, n_ctx=n_ctx, flash_attn=1)
rng = np.random.default_rng(0)
toks = rng.integers(3, hp.n_vocab, n_past + 32).tolist()
for p0 in range(0, n_past, 2048):
    n = min(2048, n_past - p0)
    rc, _ = c.decode(toks[p0:p0 + n], range(p0, p0 + n), want=[0] * (n - 1) + [1])
    assert rc == 0
assert c.decode_steps([[t] for t in toks[n_past:n_past + 24]], 1, n_past) == 0
be.synchronize()
lib = C.CDLL(os.environ.get("GGML_BACKEND_PATH", L.BACKEND_SO))
lib.mi355x_fa_stamps_dump.argtypes = [C.c_int]
splits = int(os.environ.get("N_WG", "192"))
print(f"n_past {n_past + 24}: stamps of the last launch ({splits} workgroups assumed; 100 ticks = 1 us at the 100 MHz constant clock if s_memtime counts REFCLK, else shader clock)")
lib.mi355x_fa_stamps_dump(splits)
c.free(); m.free(); be.close()
