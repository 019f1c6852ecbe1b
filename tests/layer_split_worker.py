"""Subprocess body of tests/test_gpu_split.py::test_layer_split_*: -sm layer (llama.cpp's DEFAULT split mode; /root/reference/llama-box/engine_param.hpp:900-916,
patches/llama.cpp/max_devices.patch:5-10) on logical devices of the one GPU (GGML_MI355X_FAKE_DEVICES, set by the parent).  Every device gets its own backend
instance, a contiguous range of layers with their KV cache in ITS buffer type, and the graph is cut at the device boundaries: user inputs reach the other
devices by blocking ggml_backend_tensor_copy (-> the buffer's cpy_tensor), the residual stream by the destination backend's cpy_tensor_async, slots are
re-used behind event_record / event_wait / event_synchronize — the calls ggml_backend_sched issues, in its order (host/llama_lite.cpp: decode_ubatch_ls).
Prints one JSON line."""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import harness as T  # noqa: E402
import llama_box_amd as L  # noqa: E402
from model_util import Context, Model, preset  # noqa: E402


def rows_of(ctx, prompt, forced, n_ubatch_prompt=True):
    rc, lg = ctx.decode(prompt, range(len(prompt)), want=[0] * (len(prompt) - 1) + [1])
    assert rc == 0, rc
    rows = [lg[-1]]
    for i, t in enumerate(forced):
        rc, l1 = ctx.decode([t], [len(prompt) + i])
        assert rc == 0, rc
        rows.append(l1[0])
    return np.stack(rows)


def main():
    H = L.host()
    n_dev = int(os.environ.get("GGML_MI355X_FAKE_DEVICES", "2"))
    bes = [L.Backend(d) for d in range(n_dev)]
    assert int(H.ggml_backend_reg_dev_count(bes[0].reg)) == n_dev
    out = {"n_dev": n_dev, "cases": []}
    nt = T.host_threads(32)
    D128 = dict(n_head=4, n_head_kv=2, n_embd=512, n_embd_head=128)
    # (the last two: head_dim 128 and a 40-token prompt whose causal mask is sparse in a 512-cell cache — the mask statistics that choose the attention kernel must
    # follow the mask through cpy_tensor to the second device, or that device picks the dense kernel where device 0 walked position lists: round 6)
    for name, fa, n_layer, n_prompt, n_ubatch, kw in (("test-llama", 1, 2, 40, 512, {}), ("test-llama", 0, 2, 40, 512, {}), ("test-qwen2", 1, 4, 150, 64, {}), ("test-llama", 1, 4, 150, 64, {}),
                                                      ("test-llama", 1, 2, 40, 512, D128), ("test-llama", 1, 4, 150, 64, D128)):
        if n_layer < n_dev:
            continue  # (a device without a layer: llama.cpp never builds that split)
        hp = preset(name, n_layer=n_layer, **kw)
        rng = np.random.default_rng(17)
        prompt = rng.integers(1, hp.n_vocab, n_prompt).tolist()
        forced = rng.integers(1, hp.n_vocab, 6).tolist()
        mc = Model(hp, 77, H.ggml_backend_cpu_buffer_type())
        cc = Context(mc, compute=T.oracle_compute_fn(nt), flash_attn=fa, n_ctx=512, n_ubatch=n_ubatch, n_threads=nt)
        ref = rows_of(cc, prompt, forced)
        cc.free(); mc.free()
        m1 = Model(hp, 77, bes[0].buft)
        c1 = Context(m1, backend=bes[0], flash_attn=fa, n_ctx=512, n_ubatch=n_ubatch)
        one = rows_of(c1, prompt, forced)
        c1.free(); m1.free()
        res = {"model": name + (" (head_dim 128)" if kw else ""), "fa": fa, "n_layer": n_layer, "n_prompt": n_prompt, "n_ubatch": n_ubatch, "nmse_one_device_vs_oracle": float(T.nmse(one, ref))}
        for graphs in (0, 1):
            for b in bes:
                b.set_option("graphs", graphs)
            ml = Model(hp, 77, layer_bufts=[b.buft for b in bes])
            g0 = [b.stat("graph_launches") for b in bes]
            k0 = [b.stat("kernel_launches") for b in bes]
            cl = Context(ml, backends=bes, flash_attn=fa, n_ctx=512, n_ubatch=n_ubatch)
            got = rows_of(cl, prompt, forced)
            st = cl.layer_split_stats()
            cl.free(); ml.free()
            n_graphs = -(-n_prompt // n_ubatch) + len(forced)
            res[f"graphs{graphs}"] = {"nmse_vs_oracle": float(T.nmse(got, ref)), "nmse_vs_one_device": float(T.nmse(got, one)),
                                      "bit_equal_to_one_device": bool(np.array_equal(got.view(np.uint32), one.view(np.uint32))),
                                      "argmax_equal_to_one_device": bool(np.array_equal(np.argmax(got, 1), np.argmax(one, 1))),
                                      "stats": st, "n_graphs": n_graphs,
                                      "graph_replays_per_device": [int(b.stat("graph_launches") - g) for b, g in zip(bes, g0)],
                                      "kernel_launches_per_device": [int(b.stat("kernel_launches") - k) for b, k in zip(bes, k0)]}
        out["cases"].append(res)
    for b in bes:
        b.close()
    print("LAYER_SPLIT_JSON " + json.dumps(out))


if __name__ == "__main__":
    main()
