"""Helpers shared by the model-level tests, smoke() and bench.py: a thin object wrapper over llama_lite's C API."""
import ctypes as C

import numpy as np

import llama_box_amd as L


def preset(name, **over):
    hp = L.HParams()
    if L.host().llm_preset(name.encode(), C.byref(hp)) != 0:
        raise ValueError(name)
    for k, v in over.items():
        setattr(hp, k, v)
    return hp


class Model:
    def __init__(self, hp=None, seed=1234, buft=None, path=None, tp_rank=0, tp_size=1, rowpar_buft=None, split_buft=None, layer_bufts=None):
        H = L.host()
        self.H = H
        if layer_bufts is not None:  # -sm layer: device d's buffer type holds a contiguous range of layers
            arr = (C.c_void_p * len(layer_bufts))(*layer_bufts)
            self.m = H.llm_model_synth_layer_split(C.byref(hp), seed, arr, len(layer_bufts))
        elif path is not None:
            self.m = H.llm_model_load(path.encode(), buft)
        elif split_buft is not None:  # -sm row: mat-mul weights in the split buffer type
            self.m = H.llm_model_synth_split(C.byref(hp), seed, buft, split_buft)
        else:
            self.m = H.llm_model_synth(C.byref(hp), seed, buft, tp_rank, tp_size, rowpar_buft)
        if not self.m:
            raise RuntimeError("model creation failed")
        self.hp = H.llm_model_hparams(self.m).contents
        self.n_vocab_local = self.hp.n_vocab // tp_size

    def stream_bytes(self):
        return int(self.H.llm_model_stream_bytes(self.m))

    def free(self):
        if self.m:
            self.H.llm_model_free(self.m)
            self.m = None


class Context:
    def __init__(self, model, backend=None, compute=None, n_ctx=512, n_ubatch=512, flash_attn=0, n_threads=0, graph_reuse=1, type_k=0, type_v=0, backends=None):
        H = L.host()
        self.H = H
        self.model = model
        self._compute = compute if compute is not None else L.COMPUTE_FN()
        cp = L.ContextParams(n_ctx, n_ubatch, flash_attn, n_threads, graph_reuse, type_k, type_v)
        if backends is not None:  # -sm layer: one backend per device, the graph cut at the device boundaries
            arr = (C.c_void_p * len(backends))(*[b.backend for b in backends])
            self.c = H.llm_context_new_layer_split(model.m, arr, len(backends), C.byref(cp))
        else:
            self.c = H.llm_context_new(model.m, backend.backend if backend is not None else None, self._compute, C.byref(cp))
        if not self.c:
            raise RuntimeError("context creation failed")

    def decode(self, tokens, pos, seq=None, want=None, copy_logits=True):
        """copy_logits=False returns a VIEW of the context's logits buffer (valid until the next decode): what an engine
        loop that samples in place would use; the default copies (tests keep results across calls)."""
        n = len(tokens)
        tk = (C.c_int32 * n)(*[int(t) for t in tokens])
        ps = (C.c_int32 * n)(*[int(p) for p in pos])
        sq = (C.c_int32 * n)(*[int(s) for s in seq]) if seq is not None else None
        wl = (C.c_int8 * n)(*[int(w) for w in want]) if want is not None else None
        rc = self.H.llm_decode(self.c, n, tk, ps, sq, wl)
        if rc != 0:
            return rc, None
        no = self.H.llm_n_outputs(self.c)
        nv = self.model.n_vocab_local
        lg = np.ctypeslib.as_array(self.H.llm_get_logits(self.c), shape=(max(no, 1), nv))[:no]
        return 0, (lg.copy() if copy_logits else lg)

    def decode_steps(self, tokens, n_par, pos0):
        """n_steps decode steps of n_par sequences inside the host library (tokens: [n_steps][n_par] ints); returns rc."""
        flat = [int(t) for row in tokens for t in row]
        arr = (C.c_int32 * len(flat))(*flat)
        return self.H.llm_decode_steps(self.c, len(tokens), n_par, arr, int(pos0))

    def verify_steps(self, tokens, n_par, n_draft, pos0):
        """n_steps speculative-decoding steps inside the host library (tokens: [n_steps][n_par * (1 + n_draft)] ints): every step
        decodes each sequence's sampled token + n_draft drafts with logits at every position, then drops the drafts' cells; returns rc."""
        flat = [int(t) for row in tokens for t in row]
        arr = (C.c_int32 * len(flat))(*flat)
        return self.H.llm_verify_steps(self.c, len(tokens), n_par, n_draft, arr, int(pos0))

    def clear(self):
        self.H.llm_kv_clear(self.c)

    def seq_rm(self, seq_id, p0, p1):
        return self.H.llm_kv_seq_rm(self.c, seq_id, p0, p1)

    def seq_add(self, seq_id, p0, p1, delta):
        """llama_memory_seq_add + K-shift (context shift)."""
        return self.H.llm_kv_seq_add(self.c, seq_id, p0, p1, delta)

    def layer_split_stats(self):
        out = (C.c_int64 * 4)()
        self.H.llm_layer_split_stats(self.c, out)
        return dict(zip(("cpy_tensor_async", "blocking_input_copies", "events_recorded", "events_waited"), [int(x) for x in out]))

    def timings(self):
        out = (C.c_double * 4)()
        self.H.llm_last_timings(self.c, out)
        return list(out)

    def free(self):
        if self.c:
            self.H.llm_context_free(self.c)
            self.c = None


def greedy(ctx, prompt, n_gen):
    """Prefill `prompt`, then n_gen greedy steps. Returns (token ids, list of logits rows used for each pick)."""
    rc, lg = ctx.decode(prompt, range(len(prompt)), want=[0] * (len(prompt) - 1) + [1])
    assert rc == 0, rc
    ids, rows = [], []
    pos = len(prompt)
    row = lg[-1]
    for _ in range(n_gen):
        t = int(np.argmax(row))
        ids.append(t)
        rows.append(row)
        rc, lg = ctx.decode([t], [pos])
        assert rc == 0, rc
        row = lg[0]
        pos += 1
    return ids, rows
