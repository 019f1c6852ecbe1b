"""llama_box_amd — MI355X-native ggml backend for the llama-box hot path, plus its harness bindings.

Layout
------
csrc/   the product: libggml-mi355x.so (hand-written gfx950 HIP kernels behind the ggml backend C-ABI,
        include/ggml_mi355x.h).  This is what a llama-box / llama.cpp host dlopen()s.
host/   harness: libmi355x_host.so, our own driver that plays ggml's / llama.cpp's role (graph construction,
        allocators, GGUF, a llama_decode-shaped loop) because neither is vendored in the reference snapshot.

This module only binds the two libraries with ctypes (plain pointers and sizes at every boundary).  The GPU
path fails loudly when the backend library or a gfx950 device is missing: there is no CPU fallback here.
The CPU oracle (oracle/) is bound by tests and bench.py's cpu_baseline leg only — never from this package.
"""
import ctypes as C
import os
import subprocess

_DIR = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(_DIR)
# (GGML_BACKEND_PATH: the variable the reference's loader honours for an out-of-tree backend library, pure_cpu.patch:101-102 — here it
# lets the harness run an A/B build of the same library)
BACKEND_SO = os.environ.get("GGML_BACKEND_PATH") or os.path.join(_DIR, "libggml-mi355x.so")
HOST_SO = os.path.join(_DIR, "libmi355x_host.so")

# ggml enums (include/ggml_abi.h)
F32, F16, Q8_0, Q4_K, Q5_K, Q6_K, Q8_K, I32, I64 = 0, 1, 8, 12, 13, 14, 15, 26, 27
Q4_0, Q4_1, Q5_0, Q5_1, Q8_1, IQ4_NL, BF16 = 2, 3, 6, 7, 9, 20, 30  # the other types a KV cache may be kept in (-ctk / -ctv) and the Q8_1 activation block
TYPE_BLCK = {F32: 1, F16: 1, Q8_0: 32, Q4_K: 256, Q5_K: 256, Q6_K: 256, Q8_K: 256, I32: 1, I64: 1, Q4_0: 32, Q4_1: 32, Q5_0: 32, Q5_1: 32, Q8_1: 32, IQ4_NL: 32, BF16: 1}
TYPE_SIZE = {F32: 4, F16: 2, Q8_0: 34, Q4_K: 144, Q5_K: 176, Q6_K: 210, Q8_K: 292, I32: 4, I64: 8, Q4_0: 18, Q4_1: 20, Q5_0: 22, Q5_1: 24, Q8_1: 36, IQ4_NL: 18, BF16: 2}
TYPE_NAME = {F32: "f32", F16: "f16", BF16: "bf16", Q8_0: "q8_0", Q4_0: "q4_0", Q4_1: "q4_1", Q5_0: "q5_0", Q5_1: "q5_1", IQ4_NL: "iq4_nl"}
GGML_MAX_NAME = 128
ROPE_NEOX = 2
ROPE_MROPE = 8    # ggml_rope_multi: four position streams over sections of the rotation pairs (Qwen2-VL)
ROPE_VISION = 24  # mrope with independent sections, pairs (i, i + n_dims) over the whole row (vision towers)


def build(verbose=False):
    """Compile both libraries for gfx950 (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", _DIR, "-j8"], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:])
        print(r.stderr[-4000:])
    if r.returncode != 0:
        raise RuntimeError("llama_box_amd: build failed")
    return BACKEND_SO, HOST_SO


class Tensor(C.Structure):
    """struct ggml_tensor (include/ggml_abi.h)."""
    pass


Tensor._fields_ = [
    ("type", C.c_int),
    ("buffer", C.c_void_p),
    ("ne", C.c_int64 * 4),
    ("nb", C.c_size_t * 4),
    ("op", C.c_int),
    ("op_params", C.c_int32 * 16),
    ("flags", C.c_int32),
    ("src", C.POINTER(Tensor) * 10),
    ("view_src", C.POINTER(Tensor)),
    ("view_offs", C.c_size_t),
    ("data", C.c_void_p),
    ("name", C.c_char * GGML_MAX_NAME),
    ("extra", C.c_void_p),
    ("padding", C.c_char * 8),
]
assert C.sizeof(Tensor) == 272 + GGML_MAX_NAME
TP = C.POINTER(Tensor)


class HashSet(C.Structure):
    _fields_ = [("size", C.c_size_t), ("used", C.POINTER(C.c_uint32)), ("keys", C.POINTER(TP))]


class CGraph(C.Structure):
    """struct ggml_cgraph (include/ggml_abi.h), complete: ggml_graph_view returns it by value."""
    _fields_ = [("size", C.c_int), ("n_nodes", C.c_int), ("n_leafs", C.c_int), ("nodes", C.POINTER(TP)),
                ("grads", C.c_void_p), ("grad_accs", C.c_void_p), ("leafs", C.POINTER(TP)),
                ("use_counts", C.POINTER(C.c_int32)), ("visited_hash_set", HashSet), ("order", C.c_int)]


class HParams(C.Structure):
    _fields_ = [("arch", C.c_char * 32), ("n_layer", C.c_int32), ("n_embd", C.c_int32), ("n_head", C.c_int32),
                ("n_head_kv", C.c_int32), ("n_embd_head", C.c_int32), ("n_ff", C.c_int32), ("n_vocab", C.c_int32),
                ("n_ctx_train", C.c_int32), ("rope_freq_base", C.c_float), ("rms_eps", C.c_float),
                ("rope_type", C.c_int32), ("qkv_bias", C.c_int32), ("ftype", C.c_int32), ("attn_v_q5k_70b", C.c_int32), ("peaked", C.c_int32), ("branch_gain", C.c_float)]


class ContextParams(C.Structure):
    _fields_ = [("n_ctx", C.c_int32), ("n_ubatch", C.c_int32), ("flash_attn", C.c_int32), ("n_threads", C.c_int32),
                ("graph_reuse", C.c_int32), ("type_k", C.c_int32), ("type_v", C.c_int32)]


class InitParams(C.Structure):
    _fields_ = [("mem_size", C.c_size_t), ("mem_buffer", C.c_void_p), ("no_alloc", C.c_bool)]


COMPUTE_FN = C.CFUNCTYPE(C.c_int, C.POINTER(CGraph), C.c_int)

_P, _I, _I64, _F, _SZ, _S, _B = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t, C.c_char_p, C.c_bool
_SIGS = {
    # ggml.h mirror
    "ggml_init": (_P, [InitParams]), "ggml_free": (None, [_P]),
    "ggml_new_tensor_1d": (TP, [_P, _I, _I64]), "ggml_new_tensor_2d": (TP, [_P, _I, _I64, _I64]),
    "ggml_new_tensor_3d": (TP, [_P, _I, _I64, _I64, _I64]), "ggml_new_tensor_4d": (TP, [_P, _I, _I64, _I64, _I64, _I64]),
    "ggml_set_name": (TP, [TP, _S]), "ggml_set_input": (None, [TP]), "ggml_set_output": (None, [TP]),
    "ggml_op_name": (_S, [_I]), "ggml_nbytes": (_SZ, [TP]), "ggml_nelements": (_I64, [TP]), "ggml_row_size": (_SZ, [_I, _I64]),
    "ggml_view_1d": (TP, [_P, TP, _I64, _SZ]), "ggml_view_2d": (TP, [_P, TP, _I64, _I64, _SZ, _SZ]),
    "ggml_view_3d": (TP, [_P, TP, _I64, _I64, _I64, _SZ, _SZ, _SZ]),
    "ggml_view_4d": (TP, [_P, TP, _I64, _I64, _I64, _I64, _SZ, _SZ, _SZ, _SZ]),
    "ggml_reshape_1d": (TP, [_P, TP, _I64]), "ggml_reshape_2d": (TP, [_P, TP, _I64, _I64]),
    "ggml_reshape_3d": (TP, [_P, TP, _I64, _I64, _I64]), "ggml_reshape_4d": (TP, [_P, TP, _I64, _I64, _I64, _I64]),
    "ggml_permute": (TP, [_P, TP, _I, _I, _I, _I]), "ggml_transpose": (TP, [_P, TP]),
    "ggml_dup": (TP, [_P, TP]), "ggml_cont": (TP, [_P, TP]), "ggml_cont_2d": (TP, [_P, TP, _I64, _I64]),
    "ggml_cpy": (TP, [_P, TP, TP]), "ggml_cast": (TP, [_P, TP, _I]),
    "ggml_add": (TP, [_P, TP, TP]), "ggml_sub": (TP, [_P, TP, TP]), "ggml_mul": (TP, [_P, TP, TP]), "ggml_div": (TP, [_P, TP, TP]),
    "ggml_scale": (TP, [_P, TP, _F]), "ggml_scale_bias": (TP, [_P, TP, _F, _F]), "ggml_rms_norm": (TP, [_P, TP, _F]),
    "ggml_mul_mat": (TP, [_P, TP, TP]), "ggml_get_rows": (TP, [_P, TP, TP]), "ggml_set_rows": (TP, [_P, TP, TP, TP]),
    "ggml_silu": (TP, [_P, TP]), "ggml_unary": (TP, [_P, TP, _I]), "ggml_swiglu": (TP, [_P, TP]), "ggml_swiglu_split": (TP, [_P, TP, TP]),
    "ggml_soft_max": (TP, [_P, TP]), "ggml_soft_max_ext": (TP, [_P, TP, TP, _F, _F]), "ggml_soft_max_add_sinks": (None, [TP, TP]),
    "ggml_rope_ext": (TP, [_P, TP, TP, TP, _I, _I, _I, _F, _F, _F, _F, _F, _F]),
    "ggml_rope_multi": (TP, [_P, TP, TP, TP, _I, C.POINTER(C.c_int), _I, _I, _F, _F, _F, _F, _F, _F]),
    "ggml_rope_ext_inplace": (TP, [_P, TP, TP, TP, _I, _I, _I, _F, _F, _F, _F, _F, _F]),
    "ggml_flash_attn_ext": (TP, [_P, TP, TP, TP, TP, _F, _F, _F]), "ggml_flash_attn_ext_set_prec": (None, [TP, _I]),
    "ggml_flash_attn_ext_add_sinks": (None, [TP, TP]), "ggml_argmax": (TP, [_P, TP]),
    "ggml_new_graph": (C.POINTER(CGraph), [_P]), "ggml_new_graph_custom": (C.POINTER(CGraph), [_P, _SZ, _B]),
    "ggml_build_forward_expand": (None, [C.POINTER(CGraph), TP]), "ggml_graph_n_nodes": (_I, [C.POINTER(CGraph)]),
    "ggml_graph_node": (TP, [C.POINTER(CGraph), _I]), "ggml_graph_view": (CGraph, [C.POINTER(CGraph), _I, _I]),
    "ggml_lite_set_no_reuse": (None, [_I]),
    # ggml-backend.h mirror
    "ggml_backend_load": (_P, [_S]), "ggml_backend_reg_name": (_S, [_P]), "ggml_backend_reg_dev_count": (_SZ, [_P]),
    "ggml_backend_reg_dev_get": (_P, [_P, _SZ]), "ggml_backend_reg_get_proc_address": (_P, [_P, _S]),
    "ggml_backend_dev_name": (_S, [_P]), "ggml_backend_dev_description": (_S, [_P]),
    "ggml_backend_dev_memory": (None, [_P, C.POINTER(_SZ), C.POINTER(_SZ)]), "ggml_backend_dev_type": (_I, [_P]),
    "ggml_backend_dev_init": (_P, [_P, _S]), "ggml_backend_dev_buffer_type": (_P, [_P]), "ggml_backend_dev_host_buffer_type": (_P, [_P]),
    "ggml_backend_dev_supports_op": (_B, [_P, TP]), "ggml_backend_dev_supports_buft": (_B, [_P, _P]),
    "ggml_backend_buft_name": (_S, [_P]), "ggml_backend_buft_alloc_buffer": (_P, [_P, _SZ]),
    "ggml_backend_buft_get_alignment": (_SZ, [_P]), "ggml_backend_buft_is_host": (_B, [_P]),
    "ggml_backend_buffer_free": (None, [_P]), "ggml_backend_buffer_get_base": (_P, [_P]), "ggml_backend_buffer_get_size": (_SZ, [_P]),
    "ggml_backend_buffer_clear": (None, [_P, C.c_uint8]), "ggml_backend_buffer_set_usage": (None, [_P, _I]),
    "ggml_backend_name": (_S, [_P]), "ggml_backend_free": (None, [_P]),
    "ggml_backend_tensor_set": (None, [TP, _P, _SZ, _SZ]), "ggml_backend_tensor_get": (None, [TP, _P, _SZ, _SZ]),
    "ggml_backend_tensor_memset": (None, [TP, C.c_uint8, _SZ, _SZ]),
    "ggml_backend_synchronize": (None, [_P]), "ggml_backend_graph_compute": (_I, [_P, C.POINTER(CGraph)]),
    "ggml_backend_supports_op": (_B, [_P, TP]), "ggml_backend_cpu_buffer_type": (_P, []),
    "ggml_backend_alloc_ctx_tensors_from_buft": (_P, [_P, _P]),
    "ggml_gallocr_new": (_P, [_P]), "ggml_gallocr_free": (None, [_P]), "ggml_gallocr_reserve": (_B, [_P, C.POINTER(CGraph)]),
    "ggml_gallocr_alloc_graph": (_B, [_P, C.POINTER(CGraph)]), "ggml_gallocr_get_buffer_size": (_SZ, [_P, _I]),
    # llama_lite
    "llm_preset": (_I, [_S, C.POINTER(HParams)]),
    "llm_model_synth": (_P, [C.POINTER(HParams), C.c_uint64, _P, _I, _I, _P]), "llm_model_synth_split": (_P, [C.POINTER(HParams), C.c_uint64, _P, _P]), "llm_synth_gguf": (_I, [C.POINTER(HParams), C.c_uint64, _S]),
    "llm_model_synth_layer_split": (_P, [C.POINTER(HParams), C.c_uint64, C.POINTER(_P), _I]),
    "llm_context_new_layer_split": (_P, [_P, C.POINTER(_P), _I, C.POINTER(ContextParams)]), "llm_layer_split_stats": (None, [_P, C.POINTER(C.c_int64)]),
    "llm_model_load": (_P, [_S, _P]), "llm_model_free": (None, [_P]), "llm_model_hparams": (C.POINTER(HParams), [_P]),
    "llm_model_stream_bytes": (C.c_uint64, [_P]), "llm_model_total_bytes": (C.c_uint64, [_P]), "llm_model_tensor": (TP, [_P, _S]),
    "llm_context_new": (_P, [_P, _P, COMPUTE_FN, C.POINTER(ContextParams)]), "llm_context_free": (None, [_P]),
    "llm_decode": (_I, [_P, _I, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int8)]),
    "llm_decode_steps": (_I, [_P, _I, _I, C.POINTER(C.c_int32), _I]),
    "llm_verify_steps": (_I, [_P, _I, _I, _I, C.POINTER(C.c_int32), _I]),
    "llm_n_outputs": (_I, [_P]), "llm_get_logits": (C.POINTER(C.c_float), [_P]), "llm_get_logits_ith": (C.POINTER(C.c_float), [_P, _I]),
    "llm_kv_clear": (None, [_P]), "llm_kv_seq_rm": (_I, [_P, _I, _I, _I]), "llm_kv_seq_add": (_I, [_P, _I, _I, _I, _I]), "llm_last_graph": (C.POINTER(CGraph), [_P]), "llm_context_cache_tensor": (TP, [_P, _I, _I]),
    "llm_sample_greedy": (C.c_int32, [_P, _I]), "llm_token_probabilities": (_I, [_P, _I, _I, C.POINTER(C.c_int32), C.POINTER(C.c_float)]),
    "llm_last_timings": (None, [_P, C.POINTER(C.c_double)]),
}

_host = None


def host():
    """The harness library (RTLD_GLOBAL so the backend finds ggml_backend_buffer_init the way it would in a ggml host)."""
    global _host
    if _host is None:
        if not os.path.exists(HOST_SO):
            raise RuntimeError(f"{HOST_SO} is missing: run __graft_entry__.build() first")
        lib = C.CDLL(HOST_SO, mode=C.RTLD_GLOBAL)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _host = lib
    return _host


class Backend:
    """One initialised MI355X backend instance reached ONLY through the C-ABI (ggml_backend_init + vtables)."""

    def __init__(self, device_index=0):
        H = host()
        if not os.path.exists(BACKEND_SO):
            raise RuntimeError(f"{BACKEND_SO} is missing: run __graft_entry__.build() first (no CPU fallback exists)")
        self.reg = H.ggml_backend_load(BACKEND_SO.encode())
        if not self.reg:
            raise RuntimeError("ggml-mi355x: backend failed to load/register (needs a gfx950 device; there is no fallback path)")
        n = H.ggml_backend_reg_dev_count(self.reg)
        if device_index >= n:
            raise RuntimeError(f"ggml-mi355x: device {device_index} requested, {n} present")
        self.dev = H.ggml_backend_reg_dev_get(self.reg, device_index)
        self.backend = H.ggml_backend_dev_init(self.dev, None)
        if not self.backend:
            raise RuntimeError("ggml-mi355x: init_backend failed")
        self.buft = H.ggml_backend_dev_buffer_type(self.dev)
        self.device_index = device_index
        self._procs = {}

    def proc(self, name, restype, argtypes):
        if name not in self._procs:
            addr = host().ggml_backend_reg_get_proc_address(self.reg, name.encode())
            if not addr:
                raise RuntimeError(f"ggml-mi355x: proc address {name} not exported")
            self._procs[name] = C.CFUNCTYPE(restype, *argtypes)(addr)
        return self._procs[name]

    def set_option(self, key, value):
        fn = self.proc("ggml_backend_mi355x_set_option", C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p])
        rc = fn(self.backend, key.encode(), str(value).encode())
        if rc != 0:
            raise ValueError(f"unknown backend option {key}" if rc == -1 else f"backend option {key}={value} refused ({rc})")

    def stat(self, key):
        fn = self.proc("ggml_backend_mi355x_get_stat", C.c_int64, [C.c_void_p, C.c_char_p])
        return fn(self.backend, key.encode())

    def timing_report(self, reset=True):
        """{class: (count, total_ms, bytes)} of the hipEvent-bracketed kernel classes (option timing=1)."""
        fn = self.proc("ggml_backend_mi355x_timing_report", C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int])
        buf = C.create_string_buffer(1 << 16)
        fn(self.backend, buf, len(buf), 1 if reset else 0)
        out = {}
        for line in buf.value.decode().splitlines():
            cls, cnt, ms, nbytes = line.split()
            out[cls] = (int(cnt), float(ms), float(nbytes))
        return out

    def rowpar_buft(self):
        fn = self.proc("ggml_backend_mi355x_tp_rowpar_buffer_type", C.c_void_p, [C.c_int])
        return fn(self.device_index)

    def tp_unique_id(self):
        fn = self.proc("ggml_backend_mi355x_tp_get_unique_id", C.c_int, [C.c_void_p, C.c_size_t])
        buf = C.create_string_buffer(128)
        if fn(buf, 128) != 0:
            raise RuntimeError("ggml-mi355x: ncclGetUniqueId failed")
        return buf.raw

    def tp_init(self, rank, world, uid):
        fn = self.proc("ggml_backend_mi355x_tp_init", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_size_t])
        rc = fn(self.backend, rank, world, uid, len(uid))
        if rc != 0:
            raise RuntimeError(f"ggml-mi355x: tp_init failed ({rc})")

    def tp_p2p_export(self, rank, world):
        """Step 1 of the one-shot peer-to-peer all-reduce (csrc/tp_p2p.hip): this rank's mailbox as a 64-byte IPC handle."""
        fn = self.proc("ggml_backend_mi355x_tp_p2p_export", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t])
        buf = C.create_string_buffer(64)
        rc = fn(self.backend, rank, world, buf, 64)
        if rc != 0:
            raise RuntimeError(f"ggml-mi355x: tp_p2p_export failed ({rc})")
        return buf.raw

    def tp_p2p_attach(self, handles):
        """Step 2: `handles` = the world's 64-byte handles in rank order (bytes, or a list of bytes)."""
        blob = b"".join(handles) if isinstance(handles, (list, tuple)) else handles
        fn = self.proc("ggml_backend_mi355x_tp_p2p_attach", C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t])
        rc = fn(self.backend, blob, len(blob))
        if rc != 0:
            raise RuntimeError(f"ggml-mi355x: tp_p2p_attach failed ({rc})")

    def tp_all_reduce(self, device_ptr, n):
        """In-place sum over the ranks of n f32 values at device_ptr, in the backend's stream (synchronize() to wait)."""
        fn = self.proc("ggml_backend_mi355x_tp_all_reduce", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t])
        rc = fn(self.backend, C.c_void_p(device_ptr), n)
        if rc != 0:
            raise RuntimeError(f"ggml-mi355x: tp_all_reduce failed ({rc})")

    def synchronize(self):
        host().ggml_backend_synchronize(self.backend)

    def close(self):
        if self.backend:
            host().ggml_backend_free(self.backend)
            self.backend = None
