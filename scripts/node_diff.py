"""Debug aid: per-node comparison of one llama_lite graph between the MI355X backend and the CPU oracle.
Run on a GPU box with GGML_LITE_NO_REUSE=1 so that every intermediate survives the graph."""
import ctypes as C
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
os.environ["GGML_LITE_NO_REUSE"] = "1"
import harness as T  # noqa: E402
import llama_box_amd as L  # noqa: E402
from model_util import Context, Model, preset  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "test-llama"
    fa = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    ntok = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    H = L.host()
    be = L.Backend(0)
    be.set_option("graphs", 0)
    if len(sys.argv) > 4:
        be.set_option("fusion", int(sys.argv[4]))
    hp = preset(name)
    mc = Model(hp, 1234, H.ggml_backend_cpu_buffer_type())
    mg = Model(hp, 1234, be.buft)
    cc = Context(mc, compute=T.oracle_compute_fn(), flash_attn=fa, graph_reuse=0)
    cg = Context(mg, backend=be, flash_attn=fa, graph_reuse=0)
    prompt = [1, 5, 9, 300, 17, 42, 99, 7, 256, 31, 3, 77, 101, 480, 2, 64, 200, 11, 19, 23][:ntok]
    cc.decode(prompt, range(len(prompt)))
    cg.decode(prompt, range(len(prompt)))
    gc, gg = H.llm_last_graph(cc.c).contents, H.llm_last_graph(cg.c).contents
    assert gc.n_nodes == gg.n_nodes
    VIEWS = (33, 34, 35, 36, 37)  # reshape/view/permute/transpose are numbered after CPY/CONT; skip by name instead
    bad = 0
    for i in range(gc.n_nodes):
        a, b = gc.nodes[i], gg.nodes[i]
        ta, tb = a.contents, b.contents
        if ta.type not in (L.F32, L.F16):
            continue
        n = H.ggml_nbytes(a)
        ra, rb = np.empty(n, np.uint8), np.empty(n, np.uint8)
        try:
            H.ggml_backend_tensor_get(a, ra.ctypes.data_as(C.c_void_p), 0, n)
            H.ggml_backend_tensor_get(b, rb.ctypes.data_as(C.c_void_p), 0, n)
        except Exception as e:  # noqa
            continue
        dt = np.float32 if ta.type == L.F32 else np.float16
        va, vb = ra.view(dt).astype(np.float64), rb.view(dt).astype(np.float64)
        ok = np.isfinite(va) & np.isfinite(vb)
        e = T.nmse(vb[ok], va[ok])
        flag = "" if e < 1e-9 else "   <<<<<<"
        if flag:
            bad += 1
        print(f"{i:4d} op={ta.op:3d} {ta.name.decode():28s} ne={list(ta.ne)} nmse={e:.3e} max|d|={np.max(np.abs(va[ok] - vb[ok])) if ok.any() else 0:.3e}{flag}")
        if bad > 12:
            break


if __name__ == "__main__":
    main()
