"""-sm row --tensor-split through the reference's own interface: reg.get_proc_address("ggml_backend_split_buffer_type")(main_gpu,
tensor_split) -> a buffer type whose tensors are cut by rows over the devices; MUL_MAT on them = broadcast, per-device rows, gather
(csrc/split.cpp; /root/reference/llama-box/engine_param.hpp:821-842, :902-916).  The GPU box has ONE device, so the worker process
registers four logical devices on it (GGML_MI355X_FAKE_DEVICES): placement, scatter/gather and the multi-stream execution are the
real code; only the peer copies degenerate to same-device copies."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_row_split_buffer_type_on_four_logical_devices(plog):
    env = dict(os.environ, GGML_MI355X_FAKE_DEVICES="4")
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "split_worker.py")], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("SPLIT_JSON ")][-1]
    res = json.loads(line[len("SPLIT_JSON "):])
    assert res["n_dev"] == 4
    # only mat-mul weights may be placed in the split buffer type (llama.cpp's weight_buft_supported probes)
    assert res["weight_probes"] == {"MUL_MAT": True, "MUL": False, "ADD": False, "ROPE": False, "GET_ROWS": False}, res["weight_probes"]
    for c in res["cases"]:
        plog(f"[split] ts={c['ts']} type={c['qt']} K={c['K']} N={c['N']} M={c['M']}: rows {c['row0']} nmse vs oracle {c['nmse_vs_oracle']:.2e} vs unsplit gpu {c['nmse_vs_unsplit_gpu']:.2e}")
        assert c["supports_buft"] == [True, False, False, False]  # only the main device's backend computes on split weights
        assert c["roundtrip_equal"]  # set_tensor scatters, get_tensor gathers: the host sees GGUF bytes
        assert c["nmse_vs_oracle"] <= 1e-10
        if c["M"] == 1:
            assert c["equal_to_unsplit_gpu"]  # rows are independent: the same kernel on a slice gives the same bits
        r0, n = c["row0"], c["N"]
        assert r0[0] == 0 and r0[-1] == n and all(a <= b for a, b in zip(r0[:-1], r0[1:]))
        ts = c["ts"]
        tot = sum(ts)
        for d in range(4):  # proportions, up to the 64-row granule
            want = n * (sum(ts[:d]) / tot if tot > 0 else d / 4)
            gran = 256 if n % 256 == 0 else 64  # (rows that can become another weight's K range are cut at super-block granules)
            assert abs(r0[d] - want) < gran, (d, r0, want)
            assert r0[d] % gran == 0 or r0[d] == n


@pytest.mark.gpu
@pytest.mark.parametrize("graphs,fa", [("0", "1"), ("1", "1"), ("1", "0")])
def test_tensor_parallel_model_through_the_split_buffer_type(plog, graphs, fa):
    """-sm row as a tensor-parallel layout (llama-box/engine_param.hpp:821-842, :902-916): attn_output / ffn_down cut along K,
    the FFN sharded end to end, TWO in-stream reductions per layer.  graphs = 1: the multi-stream step captured and replayed as a hipGraph."""
    # (GPU_MAX_HW_QUEUES: the logical devices' streams must not share a hardware queue of the one GPU — a device's all-reduce polls for its peers)
    # fa = 0: llama-box's DEFAULT attention path (engine_param.hpp:772-779) — K.q -> SOFT_MAX -> V^T.p over a transposed V cache that the host indexes per element:
    # the engine shards that cache by rows and gives every device its own cut-down, rebased copy of the index tensor
    env = dict(os.environ, GGML_MI355X_FAKE_DEVICES="2", GGML_MI355X_SPLIT_GRAPHS=graphs, GPU_MAX_HW_QUEUES="8", SPLIT_FA=fa)
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "split_worker.py"), "model"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("SPLIT_JSON ")][-1][len("SPLIT_JSON "):])
    assert res["n_dev"] == 2
    for c in res["cases"]:
        plog(f"[split-tp] graphs={graphs} {c['model']} ftype={c['ftype']} ts={c['ts']}: reductions per graph {c['reductions_per_graph']} (2 x {c['n_layer']} layers), graph replays {c['graph_replays']}, "
             f"logits nmse vs oracle {c['nmse_vs_oracle']:.2e} (one device {c['nmse_one_device_vs_oracle']:.2e}), vs one device {c['nmse_vs_one_device']:.2e}")
        assert all(r_ == 2 * c["n_layer"] for r_ in c["reductions_per_graph"]), c  # wo and ffn_down: one sum each, nothing else crosses devices as a sum
        # the gates of tests/test_tp_gloo.py: the sharded sums differ from the one-device run in f32 summation order only
        assert c["nmse_vs_oracle"] <= 1e-3 and c["nmse_vs_oracle"] <= 10.0 * max(c["nmse_one_device_vs_oracle"], 1e-7)
        assert c["nmse_vs_one_device"] <= 1e-3
        if graphs == "1":
            assert c["graph_replays"] >= 3, "the decode steps of the split model were not replayed as hipGraphs"
        ip = c["ip"]
        plog(f"[split-tp] graphs={graphs} fa={fa} {c['model']} ftype={c['ftype']} ts={c['ts']}: in-process tensor parallel {ip}")
        assert ip["p2p_timeouts"] == 0, ip
        if c["ts"][0] == c["ts"][1] and c["model"] == "test-llama-tp":
            # an even split of a model whose heads divide: ALL seven graphs run as tensor parallelism over the two devices (round 5) — attention and the
            # KV cache sharded by heads, the sums on the peer-to-peer all-reduce; per graph 5 input tensors copied to the other device and one
            # vocab shard written back, NOTHING per layer
            assert ip["devices"] == 2 and ip["ip_graphs"] == 7 and ip["ip_declined"] == 0, ip
            assert ip["ip_input_copies"] <= 7 * 6 and ip["ip_output_copies"] == 7 * 2, ip
            assert ip["ip_worker_kernel_launches"] > 0 and ip["p2p_allreduces"] > 0, ip
            # the host's cache tensors: gathered when the host reads, re-scattered after it wrote
            assert ip["kv_gathers_after_get_tensor"] == 1 and ip["cache_equal_to_one_device"], ip
            assert ip["kv_scatters_total"] == 2 and 0.0 <= ip["nmse_after_host_write_vs_one_device"] <= 1e-3, ip


@pytest.mark.gpu
@pytest.mark.parametrize("kv,fa", [("q4_0", "1"), ("q5_1", "1"), ("q8_0", "0")])
def test_tensor_parallel_model_with_a_kv_cache_in_another_type(plog, kv, fa):
    """-sm row together with -ctk / -ctv <type> (csrc/kv_types.hip under csrc/tp_inproc.cpp): every device stores its heads' rows into its shard of the
    cache in the block format and reads them through its own f16 image; the graphs still run as tensor parallelism (nothing declined), the logits
    against the same model and cache type on one device."""
    env = dict(os.environ, GGML_MI355X_FAKE_DEVICES="2", GGML_MI355X_SPLIT_GRAPHS="1", GPU_MAX_HW_QUEUES="8", SPLIT_FA=fa, SPLIT_KV=kv)
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "split_worker.py"), "model"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("SPLIT_JSON ")][-1][len("SPLIT_JSON "):])
    for c in res["cases"]:
        ip = c["ip"]
        plog(f"[split-tp kv={kv} fa={fa}] {c['model']} ftype={c['ftype']} ts={c['ts']}: logits nmse vs oracle {c['nmse_vs_oracle']:.2e} (one device {c['nmse_one_device_vs_oracle']:.2e}), vs one device "
             f"{c['nmse_vs_one_device']:.2e}; graphs as tensor parallelism {ip['ip_graphs']}, declined {ip['ip_declined']}, cache bytes equal to one device {ip.get('cache_bytes_equal_fraction', 1.0):.4f}")
        assert all(r_ == 2 * c["n_layer"] for r_ in c["reductions_per_graph"]), c
        assert c["nmse_vs_oracle"] <= 2e-3 and c["nmse_vs_one_device"] <= 2e-3, c  # (a 4-bit cache re-quantises values that differ in the last bit: level flips)
        assert ip["p2p_timeouts"] == 0, ip
        if c["ts"][0] == c["ts"][1] and c["model"] == "test-llama-tp":
            assert ip["devices"] == 2 and ip["ip_graphs"] == 7 and ip["ip_declined"] == 0, ip
            assert ip.get("cache_bytes_equal_fraction", 1.0) >= 0.97, ip


@pytest.mark.gpu
@pytest.mark.parametrize("fa,threads,order", [("1", "1", ""), ("0", "1", ""), ("1", "0", ""), ("1", "1", "desc")])
def test_llama3_70b_shard_shapes_on_eight_logical_devices(plog, fa, threads, order):
    """BASELINE config 4 (Llama-3-70B Q4_K_M, --tensor-split 1,1,1,1,1,1,1,1; llama-box/engine_param.hpp:821-842, :902-916) at its REAL per-device
    shard shapes, through "ggml_backend_split_buffer_type" on eight logical devices of the one GPU: two layers of the 70B layer shape, prompt
    batch + decode steps; reductions == 2 x n_layer per graph, logits against the CPU oracle and against the same model on one device.
    threads = 1: one launcher thread per device (the default); 0: the main thread submits the devices in ascending order.  order = desc: the launcher
    threads pass a turnstile so that device 7 submits FIRST, eagerly — the order that exposed half-copied weight replicas (profiles/r05_inproc_tp_submit_order.txt):
    a device must not depend on the main device having started before it."""
    env = dict(os.environ, GGML_MI355X_FAKE_DEVICES="8", GGML_MI355X_SPLIT_GRAPHS=os.environ.get("TEST_70B_SPLIT_GRAPHS", "0" if order else "1"), GPU_MAX_HW_QUEUES="16", SPLIT_FA=fa,
               GGML_MI355X_SPLIT_THREADS=threads)
    if order:
        env["GGML_MI355X_DBG_SUBMIT_ORDER"] = order
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "split_worker.py"), "model", "70b"], capture_output=True, text=True, env=env, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("SPLIT_JSON ")][-1][len("SPLIT_JSON "):])
    assert res["n_dev"] == 8
    (c,) = res["cases"]
    plog(f"[split-tp 70b shards] {c['model']} n_layer={c['n_layer']} ts={c['ts']}: reductions per graph {c['reductions_per_graph']}, "
         f"logits nmse vs oracle {c['nmse_vs_oracle']:.2e} (one device {c['nmse_one_device_vs_oracle']:.2e}), vs one device {c['nmse_vs_one_device']:.2e}")
    assert all(r_ == 2 * c["n_layer"] for r_ in c["reductions_per_graph"]), c
    assert c["nmse_vs_oracle"] <= 1e-3 and c["nmse_vs_oracle"] <= 10.0 * max(c["nmse_one_device_vs_oracle"], 1e-7)
    assert c["nmse_vs_one_device"] <= 1e-3
    # round 5 (VERDICT r04 "next" #5): the 70B shard shapes run as in-process tensor parallelism over the eight devices — one KV head and eight query
    # heads each, reductions == 2 x n_layer on the one-shot all-reduce, and NO gather / broadcast per layer: per graph the inputs go out (<= 6
    # tensors x 7 devices) and the vocab shards come back (8), whatever the number of layers
    ip = c["ip"]
    plog(f"[split-tp 70b shards] fa={fa} threads={threads} order={order or 'free'} in-process tensor parallel: {ip}; graph replays {c['graph_replays']}")
    assert ip["devices"] == 8 and ip["ip_graphs"] == 7 and ip["ip_declined"] == 0 and ip["p2p_timeouts"] == 0, ip
    assert ip["ip_input_copies"] <= 7 * 6 * 7 and ip["ip_output_copies"] == 7 * 8, ip
    assert ip["kv_gathers_after_get_tensor"] == 1 and ip["cache_nmse_vs_one_device"] <= 1e-6, ip  # (gathered from eight devices' shards into the host's tensor)
    assert 0.0 <= ip["nmse_after_host_write_vs_one_device"] <= 1e-3 and ip["kv_scatters_total"] == 2, ip


def test_split_rows_planning_without_a_device():
    """The row plan is host arithmetic: reachable through the registration's proc address on a box without any GPU."""
    import ctypes as C

    import llama_box_amd as L

    H = L.host()
    lib = C.CDLL(L.BACKEND_SO)
    lib.ggml_backend_mi355x_reg.restype = C.c_void_p
    reg = lib.ggml_backend_mi355x_reg()
    assert reg
    addr = H.ggml_backend_reg_get_proc_address(reg, b"ggml_backend_mi355x_split_rows")
    assert addr and H.ggml_backend_reg_get_proc_address(reg, b"ggml_backend_split_buffer_type")
    fn = C.CFUNCTYPE(None, C.c_int64, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int64))(addr)
    row0 = (C.c_int64 * 17)()
    # Llama-3-70B ffn rows over eight devices, even split: 28672 / 8 = 3584 rows each
    fn(28672, (C.c_float * 16)(*([1.0] * 8 + [0.0] * 8)), 8, row0)
    assert [row0[d] for d in range(9)] == [3584 * d for d in range(9)]
    # proportions 3:1 over two devices
    fn(4096, (C.c_float * 16)(3.0, 1.0), 2, row0)
    assert [row0[d] for d in range(3)] == [0, 3072, 4096]
    # all-zero proportions = even; 64-row granules when the row count is not a multiple of 256
    fn(1000, (C.c_float * 16)(), 4, row0)
    assert [row0[d] for d in range(5)] == [0, 192, 448, 704, 1000]
    # a device with proportion 0 gets no rows; 256-row granules otherwise (a device's rows of ffn_gate / ffn_up ARE its K range of ffn_down)
    fn(1024, (C.c_float * 16)(1.0, 0.0, 1.0), 3, row0)
    assert [row0[d] for d in range(4)] == [0, 512, 512, 1024]
    fn(14336, (C.c_float * 16)(*([1.0] * 8)), 8, row0)
    assert [row0[d] for d in range(9)] == [1792 * d for d in range(9)]


@pytest.mark.gpu
@pytest.mark.parametrize("n_dev", [2, 4])
def test_layer_split_across_logical_devices(plog, n_dev):
    """-sm layer, llama.cpp's DEFAULT multi-device mode (/root/reference/llama-box/engine_param.hpp:900-916; patches/llama.cpp/max_devices.patch:5-10;
    VERDICT r04 row e'): one backend per device, each holding a range of layers and their KV cache; the graph is cut at the device boundaries and driven
    as ggml_backend_sched drives its splits — blocking input copies through the buffer's cpy_tensor, the residual stream through the destination
    backend's cpy_tensor_async, slot re-use behind event_record / event_wait / event_synchronize — with hipGraph replay off and on.  The arithmetic
    is the one-device model's, kernel for kernel: the logits must be BIT-EQUAL to the one-device run."""
    env = dict(os.environ, GGML_MI355X_FAKE_DEVICES=str(n_dev))
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "layer_split_worker.py")], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("LAYER_SPLIT_JSON ")][-1][len("LAYER_SPLIT_JSON "):])
    assert res["n_dev"] == n_dev
    for c in res["cases"]:
        if c["n_layer"] < n_dev:
            continue
        for graphs in (0, 1):
            g = c[f"graphs{graphs}"]
            plog(f"[layer-split] {n_dev} devices, {c['model']} fa={c['fa']} n_layer={c['n_layer']} prompt {c['n_prompt']} in micro-batches of {c['n_ubatch']}, graphs={graphs}: "
                 f"nmse vs oracle {g['nmse_vs_oracle']:.2e} (one device {c['nmse_one_device_vs_oracle']:.2e}), bit-equal to one device: {g['bit_equal_to_one_device']}; {g['stats']}; "
                 f"graph replays per device {g['graph_replays_per_device']}, kernel launches per device {g['kernel_launches_per_device']}")
            assert g["bit_equal_to_one_device"], (c, graphs)
            assert g["nmse_vs_oracle"] <= 1e-3
            # every graph crosses every device boundary once with cpy_tensor_async; every device records one event per graph
            assert g["stats"]["cpy_tensor_async"] == g["n_graphs"] * (n_dev - 1), g
            assert g["stats"]["events_recorded"] == g["n_graphs"] * n_dev, g
            assert g["stats"]["blocking_input_copies"] >= 3 * g["n_graphs"] * (n_dev - 1), g
            assert all(k > 0 for k in g["kernel_launches_per_device"]), g  # every device computed its layers
            if graphs == 1:
                assert all(r_ >= 3 for r_ in g["graph_replays_per_device"]), g  # the decode steps were replayed on every device
