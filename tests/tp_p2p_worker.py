"""Rank body of tests/test_gpu_tp_p2p.py: one process per rank, BOTH on GPU 0 (the box has one), rendezvous over gloo on 127.0.0.1.
Every rank shards the model (column-parallel wq / wk / wv / gate / up + vocab-parallel output, row-parallel wo / down: llm_model_synth's
tp_rank / tp_size), attaches the one-shot peer-to-peer all-reduce (csrc/tp_p2p.hip) — NO RCCL communicator: RCCL refuses two ranks on one
device — and decodes through the backend's C-ABI; rank 0 gathers the vocab shards and compares with the CPU oracle on the unsharded model."""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import harness as T  # noqa: E402
import llama_box_amd as L  # noqa: E402
from model_util import Context, Model, preset  # noqa: E402


def drill(be, rank, world):
    """(1) the only transport of a group cannot be switched off, (2) a time-out of the one-shot all-reduce is REPORTED — every later
    graph_compute of the rank fails (llama_decode rc -2) instead of summing stale mailboxes for good — and (3) the host can reset the group."""
    res = {"rank": rank}
    try:
        be.set_option("tp_p2p", 0)
        res["off_refused"] = False
    except ValueError:
        res["off_refused"] = True
    hp = preset("test-llama-tp")
    m = Model(hp, 2024, be.buft, tp_rank=rank, tp_size=world, rowpar_buft=be.rowpar_buft())
    c = Context(m, backend=be, flash_attn=1)
    dist.barrier()
    rc, _ = c.decode([1, 5, 9], range(3))
    res["rc_healthy"] = rc
    dist.barrier()
    n = 4096
    if rank == 0:  # a sum nobody else joins: the kernel's bounded spins run out (GGML_MI355X_P2P_MAX_SPINS is small in this process)
        x = torch.ones(n, dtype=torch.float32, device="cuda:0")
        be.tp_all_reduce(x.data_ptr(), n)
        be.synchronize()
        res["timeouts_after_lone_sum"] = int(be.stat("p2p_timeouts"))
        rc, _ = c.decode([7], [3])
        res["rc_after_timeout"] = rc
    dist.barrier()
    be.set_option("tp_p2p_reset", 1)  # every rank, all idle
    dist.barrier()
    x = torch.full((n,), float(rank + 1), dtype=torch.float32, device="cuda:0")
    be.tp_all_reduce(x.data_ptr(), n)
    be.synchronize()
    res["sum_after_reset_ok"] = bool(torch.equal(x.cpu(), torch.full((n,), world * (world + 1) / 2.0)))
    c.free()
    c = Context(m, backend=be, flash_attn=1)
    dist.barrier()
    rc, _ = c.decode([1, 5, 9], range(3))
    res["rc_after_reset"] = rc
    res["timeouts_after_reset"] = int(be.stat("p2p_timeouts"))
    c.free()
    m.free()
    allres = [None] * world
    dist.all_gather_object(allres, res)
    if rank == 0:
        print("TP_P2P_JSON " + json.dumps({"drill": allres}), flush=True)
    be.close()
    dist.barrier()
    dist.destroy_process_group()


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    H = L.host()
    be = L.Backend(0)
    # ---- the group: handles go round over the control plane
    mine = be.tp_p2p_export(rank, world)
    handles = [None] * world
    dist.all_gather_object(handles, mine)
    be.tp_p2p_attach(handles)
    graphs = int(os.environ.get("TP_GRAPHS", "0"))
    be.set_option("graphs", graphs)
    out = {"cases": []}
    if os.environ.get("TP_DRILL") == "1":
        return drill(be, rank, world)
    prompt = [1, 5, 9, 300, 17, 42, 99, 7, 250]
    cases = (("test-llama-tp", 1), ("test-llama-tp", 5)) if world <= 2 else (("test-llama-tp4", 1), ("test-llama-tp4", 5))
    if os.environ.get("TP_PRESET"):  # e.g. llama3-70b-q4_k_m cut to TP_LAYERS layers: a rank's real shard shapes
        cases = ((os.environ["TP_PRESET"], 1),)
    for name, ftype in cases:
        hp = preset(name)
        hp.ftype = ftype
        if os.environ.get("TP_LAYERS"):
            hp.n_layer = int(os.environ["TP_LAYERS"])
        m = Model(hp, 2024, be.buft, tp_rank=rank, tp_size=world, rowpar_buft=be.rowpar_buft())
        c = Context(m, backend=be, flash_attn=1)
        dist.barrier()  # (every rank has its shard: nobody's first sum waits for a peer that is still loading)
        a0, p0, k0, g0, h0 = be.stat("allreduces"), be.stat("p2p_allreduces"), be.stat("kernel_launches"), be.stat("graph_launches"), be.stat("ss_handoffs")
        rc, lg = c.decode(prompt, range(len(prompt)))
        assert rc == 0
        rows = [lg]
        per_step = []
        for i in range(8):
            a1 = be.stat("allreduces")
            rc, l1 = c.decode([11 + i], [len(prompt) + i])
            assert rc == 0
            rows.append(l1)
            per_step.append(int(be.stat("allreduces") - a1))
        local = np.concatenate(rows)  # [9 + 8, n_vocab / world]
        n_ar, n_p2p = int(be.stat("allreduces") - a0), int(be.stat("p2p_allreduces") - p0)
        replays = int(be.stat("graph_launches") - g0)
        handoffs = int(be.stat("ss_handoffs") - h0)
        gathered = [torch.zeros(local.shape, dtype=torch.float32) for _ in range(world)]
        dist.all_gather(gathered, torch.from_numpy(np.ascontiguousarray(local)))
        c.free()
        m.free()
        if rank == 0:
            full = np.concatenate([g.numpy() for g in gathered], axis=1)
            cpu = H.ggml_backend_cpu_buffer_type()
            mf = Model(hp, 2024, cpu)
            cc = Context(mf, compute=T.oracle_compute_fn(2), flash_attn=1)
            rc, lg = cc.decode(prompt, range(len(prompt)))
            ref = [lg]
            for i in range(8):
                rc, l1 = cc.decode([11 + i], [len(prompt) + i])
                ref.append(l1)
            ref = np.concatenate(ref)
            cc.free()
            # the same model unsharded on the GPU (one rank, no collective): the yardstick for "as far from the oracle as one device is"
            mg = Model(hp, 2024, be.buft)
            cg = Context(mg, backend=be, flash_attn=1)
            rc, lg = cg.decode(prompt, range(len(prompt)))
            one = [lg]
            for i in range(8):
                rc, l1 = cg.decode([11 + i], [len(prompt) + i])
                one.append(l1)
            one = np.concatenate(one)
            cg.free()
            mg.free()
            mf.free()
            out["cases"].append({"model": name, "ftype": ftype, "n_layer": int(hp.n_layer), "allreduces": n_ar, "p2p_allreduces": n_p2p, "per_decode_step": per_step,
                                 "graph_replays": replays, "ss_handoffs": handoffs, "nmse_vs_oracle": float(T.nmse(full, ref)), "nmse_one_device_vs_oracle": float(T.nmse(one, ref)),
                                 "nmse_vs_one_device": float(T.nmse(full, one)), "argmax_equal": bool(np.array_equal(np.argmax(full, 1), np.argmax(ref, 1))),
                                 "nmse_rows_vs_one_device": [float(T.nmse(full[i], one[i])) for i in range(len(full))]})
        dist.barrier()
    out["p2p_timeouts"] = int(be.stat("p2p_timeouts"))
    tmo = torch.tensor([out["p2p_timeouts"]], dtype=torch.int64)
    dist.all_reduce(tmo)
    out["p2p_timeouts_all_ranks"] = int(tmo[0])
    if rank == 0:
        print("TP_P2P_JSON " + json.dumps(out), flush=True)
    be.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
