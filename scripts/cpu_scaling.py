"""Thread scaling of the CPU port (oracle, AVX2 block dots) on this host: tok/s and GB/s of weights for a few thread counts, on a model cut to
N layers of the Llama-3-8B shape (the per-layer arithmetic is the full model's; fewer layers keep the probe at seconds).  VERDICT r04 #9."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import harness as T, llama_box_amd as L
from model_util import Context, Model, preset
H = L.host(); lib = T.oracle(); lib.oracle_set_fast.restype = C.c_int; lib.oracle_set_fast(1)
hp = preset("llama3-8b-q4_k_m", n_layer=int(os.environ.get("LAYERS", "8")))
mc = Model(hp, 1, H.ggml_backend_cpu_buffer_type())
nmax = lib.oracle_max_threads()
for nth in [int(x) for x in os.environ.get("THREADS", "1,2,4,8,16,32,48,64,96,128,192,256").split(",") if int(x) <= nmax]:
    c = Context(mc, compute=T.oracle_compute_fn(nth), n_ctx=256, flash_attn=1, n_threads=nth)
    c.decode([1], [0]); c.decode([2], [1])
    t0 = time.perf_counter(); n = 6
    for i in range(n): c.decode([3 + i], [2 + i])
    dt = (time.perf_counter() - t0) / n
    print(f"threads={nth:4d}: {dt * 1e3:8.2f} ms/token  {mc.stream_bytes() / dt / 1e9:7.1f} GB/s  ({mc.stream_bytes() / dt / 1e9 / nth:5.2f} per thread)", flush=True)
    c.free()
