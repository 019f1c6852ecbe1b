"""CPU, world_size 2 over gloo: the tensor-split sharding of the driver (column-parallel wq/wk/wv/gate/up + vocab-
parallel output, row-parallel wo/down with an all-reduce of the f32 partials) reproduces the unsharded model.
The GPU path performs the same all-reduce with RCCL inside the backend (csrc/tp.cpp); here the collective is gloo and
the kernels are the CPU oracle, so what is tested is the partitioning + exchange logic that both share."""
import ctypes as C
import os
import sys
import tempfile

import numpy as np
import pytest

WORKER = r'''
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.environ["REPO"]); sys.path.insert(0, os.path.join(os.environ["REPO"], "tests"))
import torch, torch.distributed as dist
import harness as T, llama_box_amd as L
from model_util import Context, Model, preset
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
H = L.host(); o = T.oracle()
hp = preset("test-llama-tp")
hp.ftype = int(os.environ["FTYPE"])
cpu = H.ggml_backend_cpu_buffer_type()
n_allreduce = [0]
def compute(graph, nth):
    g = graph.contents
    for i in range(g.n_nodes):
        node = g.nodes[i]
        st = o.oracle_compute_node(node, 1)
        if st != 0: return st
        name = node.contents.name.decode()
        if name.startswith("attn_out-") or name.startswith("ffn_out-"):   # row-parallel mat-muls: sum the partials
            n = int(H.ggml_nelements(node))
            arr = np.ctypeslib.as_array(C.cast(node.contents.data, C.POINTER(C.c_float)), shape=(n,))
            t = torch.from_numpy(arr)
            dist.all_reduce(t)
            n_allreduce[0] += 1
    return 0
fn = L.COMPUTE_FN(compute)
m = Model(hp, 2024, cpu, tp_rank=rank, tp_size=world)
prompt = [1, 5, 9, 300, 17, 42, 99, 7, 250]
outs = []
for fa in (0, 1):
    c = Context(m, compute=fn, flash_attn=fa)
    rc, lg = c.decode(prompt, range(len(prompt)))
    assert rc == 0
    rc, l2 = c.decode([11], [len(prompt)])
    outs.append(np.concatenate([lg, l2]))
    c.free()
local = np.stack(outs)                       # [2, 10, n_vocab / world]
gathered = [torch.zeros(local.shape, dtype=torch.float32) for _ in range(world)]
dist.all_gather(gathered, torch.from_numpy(local))
if rank == 0:
    full = np.concatenate([g.numpy() for g in gathered], axis=2)
    mf = Model(hp, 2024, cpu)
    ref = []
    for fa in (0, 1):
        c = Context(mf, compute=T.oracle_compute_fn(1), flash_attn=fa)
        rc, lg = c.decode(prompt, range(len(prompt)))
        rc, l2 = c.decode([11], [len(prompt)])
        ref.append(np.concatenate([lg, l2])); c.free()
    ref = np.stack(ref)
    np.savez(os.environ["OUT"], full=full, ref=ref, n_allreduce=n_allreduce[0], n_layer=hp.n_layer)
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("ftype,gate_sm,gate_fa", [(4, 1e-9, 1e-4), (5, 1e-4, 1e-3)])
def test_tensor_split_world2_gloo(built, ftype, gate_sm, gate_fa):
    """ftype 4 = F16 weights: no 8-bit activation rounding, so the sharded sum must agree to f32 noise (tight gate:
    this is the check of the partitioning logic).  ftype 5 = MIXED K-quants: the K-sliced partial sums are added in a
    different order, and a one-ulp change can flip a Q8 activation rounding downstream, which the random network
    amplifies (same mechanism as tests/test_gpu_ops.py::test_fused_chains_equal_unfused) — looser gate."""
    import subprocess
    import llama_box_amd as L
    with tempfile.TemporaryDirectory() as d:
        script = os.path.join(d, "worker.py")
        open(script, "w").write(WORKER)
        out = os.path.join(d, "out.npz")
        env = dict(os.environ, REPO=L.REPO, OUT=out, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", WORLD_SIZE="2", OMP_NUM_THREADS="2", FTYPE=str(ftype))
        procs = [subprocess.Popen([sys.executable, script], env=dict(env, RANK=str(r))) for r in range(2)]
        rcs = [p.wait(timeout=600) for p in procs]
        assert rcs == [0, 0], rcs
        z = np.load(out)
        import harness as T
        T.compare("tp2 logits vs unsharded (soft-max path)", z["full"][0], z["ref"][0], max_nmse=gate_sm)
        T.compare("tp2 logits vs unsharded (flash path)", z["full"][1], z["ref"][1], max_nmse=gate_fa)
        assert int(z["n_allreduce"]) == 2 * int(z["n_layer"]) * 2 * 2  # 2 per layer x (prefill + decode) x (fa 0/1)
