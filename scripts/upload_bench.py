"""Loader path throughput: buffer.set_tensor of weight-sized pieces from PAGEABLE host memory, as llama.cpp's loader
issues them (one call per tensor; --no-mmap reads land in an ordinary buffer: /root/reference/llama-box/engine_param.hpp:403).  Prints one
JSON line: GB/s and what a 42.5 GB (Llama-3-70B Q4_K_M) load would take.  GGML_MI355X_STAGED_UPLOAD=0 selects the plain hipMemcpy path."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import llama_box_amd as L  # noqa: E402

H = L.host()
be = L.Backend(0)
total = int(os.environ.get("UPLOAD_BENCH_GIB", "6")) << 30
piece = int(os.environ.get("UPLOAD_BENCH_PIECE_MIB", "256")) << 20
ctx = H.ggml_init(L.InitParams(0, None, True))
ts = [H.ggml_new_tensor_4d(ctx, L.I32, piece // 4, 1, 1, 1) for _ in range(total // piece)]
buf = H.ggml_backend_alloc_ctx_tensors_from_buft(ctx, be.buft)
assert buf
src = np.random.default_rng(0).integers(0, 256, piece * 2, dtype=np.uint8)  # pageable, touched
rates = []
for rep in range(3):
    t0 = time.perf_counter()
    for i, t in enumerate(ts):
        H.ggml_backend_tensor_set(t, src[(i & 1) * piece:].ctypes.data_as(C.c_void_p), 0, piece)
    tail = np.empty(4, np.uint8)
    H.ggml_backend_tensor_get(ts[-1], tail.ctypes.data_as(C.c_void_p), piece - 4, 4)  # (waits for everything in flight)
    dt = time.perf_counter() - t0
    rates.append(total / dt / 1e9)
    assert np.array_equal(tail, src[((len(ts) - 1) & 1) * piece + piece - 4:][:4])
best = max(rates)
print(json.dumps({"staged": os.environ.get("GGML_MI355X_STAGED_UPLOAD", "1") != "0", "slot_mib": os.environ.get("GGML_MI355X_UPLOAD_SLOT_MIB", "32"),
                  "threads": os.environ.get("GGML_MI355X_UPLOAD_THREADS", "4"), "bytes": total, "piece_mib": piece >> 20, "GBps_runs": [round(r, 2) for r in rates],
                  "GBps": round(best, 2), "llama3_70b_q4_k_m_load_s": round(42.5e9 / (best * 1e9), 1), "staged_bytes": be.stat("staged_upload_bytes")}))
