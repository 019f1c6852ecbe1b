"""Subprocess body of tests/test_gpu_split.py: the row-split buffer type (-sm row) exercised on ONE GPU by registering several logical
devices on it (GGML_MI355X_FAKE_DEVICES, set by the parent before this process loads the backend).  Prints one JSON line."""
import ctypes as C
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import harness as T  # noqa: E402
import llama_box_amd as L  # noqa: E402


def main():
    H = L.host()
    be = L.Backend(0)
    n_dev = H.ggml_backend_reg_dev_count(be.reg)
    out = {"n_dev": int(n_dev), "cases": []}
    split_fn = be.proc("ggml_backend_split_buffer_type", C.c_void_p, [C.c_int, C.POINTER(C.c_float)])
    rows_fn = be.proc("ggml_backend_mi355x_split_rows", None, [C.c_int64, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int64)])
    rng = np.random.default_rng(7)
    for ts in ([3.0, 1.0, 2.0, 2.0], [0.0] * 4, [1.0, 0.0, 1.0, 0.0]):
        arr = (C.c_float * 16)(*(ts + [0.0] * (16 - len(ts))))
        buft = split_fn(0, arr)
        assert buft, "ggml_backend_split_buffer_type returned NULL"
        sup = [bool(H.ggml_backend_dev_supports_buft(H.ggml_backend_reg_dev_get(be.reg, d), buft)) for d in range(n_dev)]
        for qt, K, N, M in ((L.Q4_K, 4096, 1024, 1), (L.Q6_K, 2048, 640, 1), (L.Q4_K, 1024, 512, 5), (L.Q8_0, 2048, 1100, 1), (L.Q5_K, 512, 384, 9)):
            w = T.rand_weight(qt, K, N, rng)
            x = rng.standard_normal((M, K)).astype(np.float32)
            row0 = (C.c_int64 * 17)()
            rows_fn(N, arr, n_dev, row0)
            # weights in the split buffer type, everything else in the main device's buffer type
            ctx_w = H.ggml_init(L.InitParams(0, None, True))
            tw = H.ggml_new_tensor_4d(ctx_w, qt, K, N, 1, 1)
            buf_w = H.ggml_backend_alloc_ctx_tensors_from_buft(ctx_w, buft)
            assert buf_w
            raw = np.ascontiguousarray(w)
            H.ggml_backend_tensor_set(tw, raw.ctypes.data_as(C.c_void_p), 0, raw.nbytes)
            back = np.empty(raw.nbytes, np.uint8)
            H.ggml_backend_tensor_get(tw, back.ctypes.data_as(C.c_void_p), 0, raw.nbytes)
            g = T.G(be)
            tx = g.new(L.F32, [K, M], x)
            y = H.ggml_mul_mat(g.ctx, tw, tx)
            got = g.compute([y])[0]
            g.free()

            def build(go):
                return H.ggml_mul_mat(go.ctx, go.new(qt, [K, N], w), go.new(L.F32, [K, M], x))

            ref = T.run_case(build, "oracle")[0]
            plain = T.run_case(build, be)[0]
            out["cases"].append({"ts": ts, "qt": int(qt), "K": K, "N": N, "M": M, "supports_buft": sup, "row0": [int(row0[d]) for d in range(n_dev + 1)],
                                 "roundtrip_equal": bool(np.array_equal(back, raw.view(np.uint8).ravel())), "nmse_vs_oracle": float(T.nmse(got, ref)),
                                 "equal_to_unsplit_gpu": bool(np.array_equal(np.asarray(got).view(np.uint32), np.asarray(plain).view(np.uint32))) if M == 1 else None,
                                 "nmse_vs_unsplit_gpu": float(T.nmse(got, plain))})
            H.ggml_backend_buffer_free(buf_w)
            H.ggml_free(ctx_w)
    # llama.cpp's loader asks, weight by weight, whether the device can run the op that will read the weight with the weight living in a
    # (dummy, zero-size) buffer of the split type — weight_buft_supported.  Only MUL_MAT may say yes: anything else would dereference the
    # split buffer's fake base address (ADVICE r02 / graph.cpp supports_op)
    arr = (C.c_float * 16)(*([1.0] * 4 + [0.0] * 12))
    buft = split_fn(0, arr)
    dummy = H.ggml_backend_buft_alloc_buffer(buft, 0)
    ctx_p = H.ggml_init(L.InitParams(0, None, True))
    K, N, M, HD, NH = 1024, 512, 3, 64, 4
    x = H.ggml_new_tensor_4d(ctx_p, L.F32, K, M, 1, 1)
    x3 = H.ggml_new_tensor_4d(ctx_p, L.F32, HD, NH, M, 1)
    pos = H.ggml_new_tensor_4d(ctx_p, L.I32, M, 1, 1, 1)
    ids = H.ggml_new_tensor_4d(ctx_p, L.I32, M, 1, 1, 1)

    def weight(qt, *ne):
        w = H.ggml_new_tensor_4d(ctx_p, qt, *(list(ne) + [1] * (4 - len(ne))))
        w.contents.buffer = dummy
        return w

    probes = {
        "MUL_MAT": H.ggml_mul_mat(ctx_p, weight(L.Q4_K, K, N), x),
        "MUL": H.ggml_mul(ctx_p, x, weight(L.F32, K)),
        "ADD": H.ggml_add(ctx_p, x, weight(L.F32, K)),
        "ROPE": H.ggml_rope_ext(ctx_p, x3, pos, weight(L.F32, HD // 2), HD, 0, 8192, 10000.0, 1.0, 0.0, 1.0, 32.0, 1.0),
        "GET_ROWS": H.ggml_get_rows(ctx_p, weight(L.Q4_K, K, N), ids),
    }
    out["weight_probes"] = {k: bool(H.ggml_backend_dev_supports_op(be.dev, v)) for k, v in probes.items()}
    H.ggml_free(ctx_p)
    H.ggml_backend_buffer_free(dummy)
    print("SPLIT_JSON " + json.dumps(out))


def model_main():
    """A whole model under -sm row on two logical devices: mat-mul weights in the split buffer type (attn_output / ffn_down cut along K, the
    rest by rows), everything else on the main device — prompt batch + decode steps against the CPU oracle and against the same model on
    one device; the reductions per graph are counted."""
    from model_util import Context, Model, preset

    H = L.host()
    be = L.Backend(0)
    split_fn = be.proc("ggml_backend_split_buffer_type", C.c_void_p, [C.c_int, C.POINTER(C.c_float)])
    out = {"n_dev": int(H.ggml_backend_reg_dev_count(be.reg)), "graphs_env": os.environ.get("GGML_MI355X_SPLIT_GRAPHS", "0"), "cases": []}
    prompt = [1, 5, 9, 300, 17, 42, 99, 7, 250, 3]
    cases = (("test-llama-tp", 1, [1.0, 1.0], 0), ("test-llama-tp", 5, [1.0, 1.0], 0), ("test-llama-tp", 1, [3.0, 1.0], 0), ("test-qwen2", 5, [1.0, 1.0], 0))
    if len(sys.argv) > 2 and sys.argv[2] == "70b":
        # BASELINE config 4 at its real shard shapes: Llama-3-70B's layer (8192 / 28672, 64 heads on 8 KV heads, Q4_K_M type
        # map with Q5_K attn_v) cut to two layers, --tensor-split 1,1,1,1,1,1,1,1 over eight logical devices: every device holds 1024 / 128 /
        # 128 rows of wq / wk / wv, the 1024-value K slice of attn_output, 3584 rows of gate / up and the 3584-value K slice of ffn_down
        cases = (("llama3-70b-q4_k_m", 1, [1.0] * int(os.environ.get("GGML_MI355X_FAKE_DEVICES", "8")), int(os.environ.get("SPLIT_LAYERS", "2"))),)
    for name, ftype, ts, n_layer in cases:
        hp = preset(name)
        if n_layer:
            hp.n_layer = n_layer
        hp.ftype = ftype  # 1 = Q4_K_M type map (gate / up share a type: the sharded FFN chain), 5 = mixed (every kernel; gate / up differ: separate paths)
        arr = (C.c_float * 16)(*(ts + [0.0] * (16 - len(ts))))
        buft = split_fn(0, arr)
        ms = Model(hp, 77, be.buft, split_buft=buft)
        mg = Model(hp, 77, be.buft)
        mc = Model(hp, 77, H.ggml_backend_cpu_buffer_type())
        fa = int(os.environ.get("SPLIT_FA", "1"))  # 0: llama-box's default attention path (K.q -> SOFT_MAX -> V^T.p over a transposed V cache)
        # SPLIT_KV=<type>: the KV cache in another of llama-box's -ctk / -ctv types (csrc/kv_types.hip) under the sharded graphs
        kvt = {"": 0, "f16": 0, "q8_0": L.Q8_0, "q4_0": L.Q4_0, "q4_1": L.Q4_1, "q5_0": L.Q5_0, "q5_1": L.Q5_1, "iq4_nl": L.IQ4_NL, "bf16": L.BF16}[os.environ.get("SPLIT_KV", "")]
        kv = dict(type_k=kvt, type_v=kvt if fa else 0)
        cs = Context(ms, backend=be, flash_attn=fa, **kv)
        cg = Context(mg, backend=be, flash_attn=fa, **kv)
        cc = Context(mc, compute=T.oracle_compute_fn(), flash_attn=fa, **kv)

        def run(ctx, count=False):
            rows, reds = [], []
            a0 = be.stat("allreduces")
            rc, lg = ctx.decode(prompt, range(len(prompt)), want=[0] * (len(prompt) - 1) + [1])
            assert rc == 0
            reds.append(be.stat("allreduces") - a0)
            rows.append(lg[-1])
            for i, t in enumerate((11, 12, 13, 14, 15, 16)):
                a0 = be.stat("allreduces")
                rc, l1 = ctx.decode([t], [len(prompt) + i])
                assert rc == 0
                reds.append(be.stat("allreduces") - a0)
                rows.append(l1[0])
            return np.stack(rows), reds

        g0 = be.stat("graph_launches")
        ip0 = {k: be.stat(k) for k in ("ip_graphs", "ip_declined", "ip_input_copies", "ip_output_copies", "ip_kv_gathers", "ip_kv_scatters", "ip_worker_kernel_launches", "p2p_allreduces")}
        r_s, reds = run(cs, True)
        replays = be.stat("graph_launches") - g0
        ip = {k: int(be.stat(k) - v) for k, v in ip0.items()}
        ip["devices"] = int(be.stat("ip_devices"))
        ip["p2p_timeouts"] = int(be.stat("p2p_timeouts")) + int(be.stat("ip_worker_p2p_timeouts"))
        ip["timeouts_by_device"] = [int(be.stat(f"ip_dbg_timeouts_{d}")) for d in range(int(be.stat("ip_devices")))]
        # the host reads a cache tensor back (a slot save): the shards come home — compare K of layer 0 with the one-device run's below
        def cache_bytes(ctx_model, ctx):
            t = H.llm_context_cache_tensor(ctx.c, 0, 0 if fa else 1)  # (without flash attention: the TRANSPOSED V cache, sharded by rows)
            n = H.ggml_nbytes(t)
            raw = np.empty(n, np.uint8)
            H.ggml_backend_tensor_get(t, raw.ctypes.data_as(C.c_void_p), 0, n)
            return raw
        k_split = cache_bytes(ms, cs)
        ip["kv_gathers_after_get_tensor"] = int(be.stat("ip_kv_gathers") - ip0["ip_kv_gathers"])
        r_g, _ = run(cg)
        k_one = cache_bytes(mg, cg)
        ip["cache_equal_to_one_device"] = bool(np.array_equal(k_split, k_one))
        # (another mat-mul tiling over the narrower shards may sum in another order: an ulp in f32, now and then an f16 rounding of a cached value)
        if kvt == 0 or not fa:  # (f16 cells; without flash attention the tensor read back is the transposed f16 V cache)
            ip["cache_nmse_vs_one_device"] = float(T.nmse(k_split.view(np.float16).astype(np.float64), k_one.view(np.float16).astype(np.float64)))
        else:  # block formats: a value that differs in the last bit before the store may land on the neighbouring level — count the bytes
            ip["cache_nmse_vs_one_device"] = 0.0
            ip["cache_bytes_equal_fraction"] = float(np.mean(k_split == k_one))
        # ... and decoding goes on after the host looked (and after it WROTE: the same bytes back — the shards are re-scattered)
        H.ggml_backend_tensor_set(H.llm_context_cache_tensor(cs.c, 0, 0 if fa else 1), k_split.ctypes.data_as(C.c_void_p), 0, k_split.nbytes)
        rc_a, la = cs.decode([21], [len(prompt) + 6])
        rc_b, lb = cg.decode([21], [len(prompt) + 6])
        ip["nmse_after_host_write_vs_one_device"] = float(T.nmse(la[0], lb[0])) if rc_a == 0 and rc_b == 0 else -1.0
        ip["kv_scatters_total"] = int(be.stat("ip_kv_scatters") - ip0["ip_kv_scatters"])
        r_c, _ = run(cc)
        out["cases"].append({"model": name, "ftype": ftype, "ts": ts, "n_layer": int(hp.n_layer), "reductions_per_graph": [int(r) for r in reds], "graph_replays": int(replays),
                             "nmse_vs_oracle": float(T.nmse(r_s, r_c)), "nmse_one_device_vs_oracle": float(T.nmse(r_g, r_c)), "nmse_vs_one_device": float(T.nmse(r_s, r_g)),
                             "argmax_equal_one_device": bool(np.array_equal(np.argmax(r_s, 1), np.argmax(r_g, 1))), "ip": ip,
                             "nmse_rows_vs_one_device": [float(T.nmse(r_s[i], r_g[i])) for i in range(len(r_s))],
                             "nmse_row0_by_vocab_eighth": [float(T.nmse(x, y)) for x, y in zip(np.array_split(r_s[0], 8), np.array_split(r_g[0], 8))],
                             "nmse_row1_by_vocab_eighth": [float(T.nmse(x, y)) for x, y in zip(np.array_split(r_s[1], 8), np.array_split(r_g[1], 8))],
                             "row0_absmean_by_vocab_eighth": [float(np.mean(np.abs(x))) for x in np.array_split(r_s[0], 8)],
                             "row0_zero_fraction_by_vocab_eighth": [float(np.mean(x == 0)) for x in np.array_split(r_s[0], 8)]})
        if os.environ.get("SPLIT_TIME"):  # host view of a step: every decode reads its logits back (the sampler's synchronisation)
            import time
            for tag, ctx in (("split", cs), ("one_device", cg)):
                p0 = len(prompt) + 7
                for i in range(8):
                    ctx.decode([31 + i], [p0 + i])
                h0, t0 = be.stat("graph_compute_host_ns"), time.perf_counter()
                n_t = int(os.environ["SPLIT_TIME"])
                for i in range(n_t):
                    ctx.decode([41 + i], [p0 + 8 + i])
                out["cases"][-1][f"timed_{tag}"] = {"ms_per_step": (time.perf_counter() - t0) / n_t * 1e3, "graph_compute_host_us_per_step": (be.stat("graph_compute_host_ns") - h0) / n_t / 1e3}
        for o in (cs, cg, cc, ms, mg, mc):
            o.free()
    print("SPLIT_JSON " + json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "model":
        model_main()
    else:
        main()
