"""Shared test harness: builds one ggml graph per target and runs it on the CPU oracle or on the MI355X backend.

The shape follows upstream's test-backend-ops: every case is a `build(G)` function that creates
inputs and ops through the ggml API mirror (llama_box_amd/host/ggml_lite.h); the harness evaluates the SAME
function once against the oracle (host memory, oracle_graph_compute) and once against the candidate backend
(device memory, reached only through ggml_backend_init + the vtables) and compares.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/ (see oracle/oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

import llama_box_amd as L

REPO = L.REPO
ORACLE_DIR = os.path.join(REPO, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
_oracle = None


def build_oracle():
    r = subprocess.run(["make", "-C", ORACLE_DIR], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout[-2000:] + r.stderr[-2000:])
    return ORACLE_SO


def oracle():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        lib = C.CDLL(ORACLE_SO)
        lib.oracle_graph_compute.restype = C.c_int
        lib.oracle_graph_compute.argtypes = [C.POINTER(L.CGraph), C.c_int]
        lib.oracle_compute_node.restype = C.c_int
        lib.oracle_compute_node.argtypes = [L.TP, C.c_int]
        lib.oracle_dequantize_row.restype = None
        lib.oracle_dequantize_row.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
        lib.oracle_quantize_row_q8_K.restype = None
        lib.oracle_quantize_row_q8_K.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        lib.oracle_quantize_row_q8_0.restype = None
        lib.oracle_quantize_row_q8_0.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        for n in ("q4_K_q8_K", "q5_K_q8_K", "q6_K_q8_K", "q8_0_q8_0"):
            f = getattr(lib, "oracle_vec_dot_" + n)
            f.restype = C.c_float
            f.argtypes = [C.c_int64, C.c_void_p, C.c_void_p]
        lib.oracle_fp16_to_fp32.restype = C.c_float
        lib.oracle_fp16_to_fp32.argtypes = [C.c_uint16]
        lib.oracle_fp32_to_fp16.restype = C.c_uint16
        lib.oracle_fp32_to_fp16.argtypes = [C.c_float]
        lib.oracle_max_threads.restype = C.c_int
        lib.oracle_quantize_row.restype = C.c_int
        lib.oracle_quantize_row.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
        lib.oracle_kq_dot.restype = C.c_float
        lib.oracle_kq_dot.argtypes = [C.c_int, C.c_int64, C.c_void_p, C.c_void_p]
        lib.oracle_fp32_to_bf16.restype = C.c_uint16
        lib.oracle_fp32_to_bf16.argtypes = [C.c_float]
        lib.oracle_bf16_to_fp32.restype = C.c_float
        lib.oracle_bf16_to_fp32.argtypes = [C.c_uint16]
        _oracle = lib
    return _oracle


def oracle_compute_fn(n_threads=4):
    """llm_compute_fn wrapping the oracle (keep the returned object alive while the context lives)."""
    lib = oracle()

    def fn(graph, nth):
        return lib.oracle_graph_compute(graph, nth if nth > 0 else n_threads)

    return L.COMPUTE_FN(fn)


# ------------------------------------------------------------------------------------------ random quant data
def rand_blocks(qtype, n_blocks, K, rng):
    """Directly sampled quant blocks with sane scales (same recipe as llama_lite.cpp's synth_block)."""
    bs = L.TYPE_SIZE[qtype]
    out = np.zeros((n_blocks, bs), dtype=np.uint8)
    s = rng.uniform(0.5, 1.5, n_blocks).astype(np.float32) / np.sqrt(K)
    if qtype == L.Q8_0:
        q = rng.integers(-127, 128, (n_blocks, 32), dtype=np.int8)
        out[:, 0:2] = (s / 73.0).astype(np.float16).view(np.uint8).reshape(n_blocks, 2)
        out[:, 2:] = q.view(np.uint8)
    elif qtype in (L.Q4_K, L.Q5_K):
        q5 = qtype == L.Q5_K
        sc = rng.integers(0, 64, (n_blocks, 8), dtype=np.uint8)
        mn = rng.integers(0, 64, (n_blocks, 8), dtype=np.uint8)
        d = s / (36.0 * (9.2 if q5 else 4.6))
        out[:, 0:2] = d.astype(np.float16).view(np.uint8).reshape(n_blocks, 2)
        out[:, 2:4] = (d * (15.5 if q5 else 7.5) * 0.5).astype(np.float16).view(np.uint8).reshape(n_blocks, 2)
        for j in range(4):
            out[:, 4 + j] = (sc[:, j] & 63) | ((sc[:, j + 4] >> 4) << 6)
            out[:, 8 + j] = (mn[:, j] & 63) | ((mn[:, j + 4] >> 4) << 6)
            out[:, 12 + j] = (sc[:, j + 4] & 0xF) | ((mn[:, j + 4] & 0xF) << 4)
        out[:, 16:] = rng.integers(0, 256, (n_blocks, bs - 16), dtype=np.uint8)
    elif qtype == L.Q6_K:
        out[:, 0:192] = rng.integers(0, 256, (n_blocks, 192), dtype=np.uint8)
        out[:, 192:208] = rng.integers(-128, 128, (n_blocks, 16), dtype=np.int8).view(np.uint8)
        out[:, 208:210] = (s / (70.0 * 18.5)).astype(np.float16).view(np.uint8).reshape(n_blocks, 2)
    else:
        raise ValueError(qtype)
    return out


def rand_weight(qtype, K, N, rng):
    """Raw bytes of a [K, N] weight tensor of the given ggml type."""
    if qtype == L.F32:
        return (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    if qtype == L.F16:
        return (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float16)
    nb = K // L.TYPE_BLCK[qtype]
    return rand_blocks(qtype, N * nb, K, rng).reshape(N, nb * L.TYPE_SIZE[qtype])


# ------------------------------------------------------------------------------------------ graph builder
_NP_OF = {L.F32: np.float32, L.F16: np.float16, L.I32: np.int32, L.I64: np.int64}


class G:
    """One graph instance bound to a target ('oracle' or a llama_box_amd.Backend)."""

    def __init__(self, target):
        self.H = L.host()
        self.target = target
        self.ctx = self.H.ggml_init(L.InitParams(0, None, True))
        self.inputs = []  # (tensor, bytes)
        self.keep = []
        self.buf = None

    def new(self, qtype, ne, data=None, name=None):
        H = self.H
        ne = list(ne) + [1] * (4 - len(ne))
        t = H.ggml_new_tensor_4d(self.ctx, qtype, *ne)
        if name:
            H.ggml_set_name(t, name.encode())
        if data is not None:
            raw = np.ascontiguousarray(data)
            assert raw.nbytes == H.ggml_nbytes(t), (raw.nbytes, H.ggml_nbytes(t), ne, qtype)
            self.inputs.append((t, raw))
        return t

    def compute(self, outs, n_threads=4, expand_first=(), cuts=None):
        """Allocates every tensor, uploads inputs, evaluates the graph for `outs`, returns numpy copies of `outs`.
        expand_first: tensors added to the graph before the outputs, unflagged (llama.cpp's build_attn expands q, k, v together
        before the cache stores); cuts: node indices at which the graph is handed to the backend in pieces — ggml_graph_view of
        the full graph, what ggml_backend_sched gives a backend for each of its splits."""
        H = self.H
        gf = H.ggml_new_graph_custom(self.ctx, 4096, False)
        for o in expand_first:
            H.ggml_build_forward_expand(gf, o)
        for o in outs:
            H.ggml_set_output(o)
            H.ggml_build_forward_expand(gf, o)
        if self.target == "oracle":
            buft = H.ggml_backend_cpu_buffer_type()
        else:
            buft = self.target.buft
            for i in range(gf.contents.n_nodes):
                node = gf.contents.nodes[i]
                if not H.ggml_backend_dev_supports_op(self.target.dev, node):
                    raise RuntimeError(f"backend reports supports_op=false for node {i} op={node.contents.op} '{node.contents.name.decode()}'")
        self.buf = H.ggml_backend_alloc_ctx_tensors_from_buft(self.ctx, buft)
        assert self.buf, "buffer allocation failed"
        H.ggml_backend_buffer_clear(self.buf, 0)
        for t, raw in self.inputs:
            H.ggml_backend_tensor_set(t, raw.ctypes.data_as(C.c_void_p), 0, raw.nbytes)
        if self.target == "oracle":
            st = oracle().oracle_graph_compute(gf, n_threads)
        elif cuts:
            st = 0
            edges = [0] + [c if c >= 0 else gf.contents.n_nodes + c for c in cuts] + [gf.contents.n_nodes]
            for i0, i1 in zip(edges[:-1], edges[1:]):
                if st == 0 and i1 > i0:
                    view = H.ggml_graph_view(gf, i0, i1)
                    st = H.ggml_backend_graph_compute(self.target.backend, C.byref(view))
        else:
            st = H.ggml_backend_graph_compute(self.target.backend, gf)
        if st != 0:
            raise RuntimeError(f"graph compute failed with status {st} on {self.target}")
        self.gf = gf
        return [self.read(o) for o in outs]

    def read(self, t):
        H = self.H
        tt = t.contents
        n = H.ggml_nbytes(t)
        raw = np.empty(n, dtype=np.uint8)
        H.ggml_backend_tensor_get(t, raw.ctypes.data_as(C.c_void_p), 0, n)
        if tt.type in _NP_OF:
            dt = np.dtype(_NP_OF[tt.type])
            contiguous = True
            exp = dt.itemsize
            for i in range(4):
                if tt.ne[i] != 1 and tt.nb[i] != exp:
                    contiguous = False
                exp *= tt.ne[i]
            if contiguous:
                return raw.view(dt).reshape([tt.ne[3], tt.ne[2], tt.ne[1], tt.ne[0]]).copy()
        return raw

    def free(self):
        H = self.H
        if self.buf:
            H.ggml_backend_buffer_free(self.buf)
            self.buf = None
        if self.ctx:
            H.ggml_free(self.ctx)
            self.ctx = None


def run_case(build, target, n_threads=4, **kw):
    g = G(target)
    try:
        outs = build(g)
        if not isinstance(outs, (list, tuple)):
            outs = [outs]
        return g.compute(list(outs), n_threads, **kw)
    finally:
        g.free()


def cpu_quota():
    """CPUs this container may actually use: the cgroup's CFS quota (cpu.max: "1600000 100000" = 16 CPUs on the GPU boxes of round 5, whose
    host shows 256 logical CPUs) or, without one, the logical CPU count.  An OpenMP team larger than the quota is throttled by the kernel — the
    round-4 oracle runs with 64 / 128 threads spent most of their wall time descheduled (profiles/r05_cpu_scaling.txt: 8 threads 125 GB/s,
    64 threads 42, 128 threads 10)."""
    n = os.cpu_count() or 4
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, per = int(f.read()), int(g.read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def host_threads(cap=64):
    """Threads for oracle runs at real model shapes: what the host offers THIS container (cpu_quota), minus one for the driving thread."""
    q = cpu_quota()
    return max(2, min(cap, q - 1 if q > 4 else q))


def nmse(a, b):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    den = float(np.sum(b * b))
    num = float(np.sum((a - b) ** 2))
    if den == 0.0:
        return 0.0 if num == 0.0 else float("inf")
    return num / den


def compare(name, got, ref, max_nmse, max_abs=None, log=None):
    """Records the comparison (for gpurun_out logs) and asserts the gates."""
    got = np.asarray(got)
    ref = np.asarray(ref)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    if got.dtype == np.uint8 or np.issubdtype(got.dtype, np.integer):
        nbad = int(np.count_nonzero(got != ref))
        msg = f"{name}: integer/byte compare, mismatches={nbad}/{got.size}"
        if log is not None:
            log(msg)
        assert nbad == 0, msg
        return
    g64 = got.astype(np.float64)
    r64 = ref.astype(np.float64)
    both_nan = np.isnan(g64) & np.isnan(r64)
    g64 = np.where(both_nan, 0.0, g64)
    r64 = np.where(both_nan, 0.0, r64)
    e = nmse(g64, r64)
    mabs = float(np.max(np.abs(g64 - r64))) if got.size else 0.0
    nexact = int(np.count_nonzero(g64 == r64))
    msg = f"{name}: nmse={e:.3e} max_abs={mabs:.3e} bit_equal={nexact}/{got.size} ref_absmax={float(np.max(np.abs(r64))) if got.size else 0:.3e}"
    if log is not None:
        log(msg)
    assert np.isfinite(e) and e <= max_nmse, msg
    if max_abs is not None:
        assert mabs <= max_abs, msg
