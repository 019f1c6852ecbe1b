// tp.cpp — tensor parallelism, one process per GPU: RCCL all-reduce of row-parallel partial sums over xGMI.
//
// The reference has no collective at all (multi-GPU in llama.cpp is layer split or the row-split
// buffer with peer copies; no NCCL/RCCL).  `north_star` asks for --tensor-split re-implemented as sharding with an
// all-reduce of the residual, so this is new design, MI355X-first: each rank owns a column-parallel slice of
// wq/wk/wv/gate/up and a row-parallel (K-sliced) slice of wo/down; weights placed in the "RowPar" buffer type mark
// the two mat-muls per layer whose f32 partial results are summed across ranks, in-stream, right after the kernel
// (graph.cpp).  The message is n_embd x M floats (16 KiB at batch 1 for 8B, 32 KiB for 70B): latency-bound, so it
// is issued on the compute stream itself (no extra stream hop) and is capturable into the decode hipGraph.
// RCCL is dlopen()ed on first use: single-GPU runs never load the 570 MB library.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "common.h"

namespace mi355x {

struct rccl_api {
    void * handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char * (*GetErrorString)(ncclResult_t) = nullptr;
};

static rccl_api * load_rccl() {
    static rccl_api api;
    static bool tried = false;
    if (tried) return api.handle ? &api : nullptr;
    tried = true;
    const char * names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char * n : names) {
        api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (api.handle) break;
    }
    if (!api.handle) {
        MI_ERR("cannot load RCCL: %s", dlerror());
        return nullptr;
    }
    api.GetUniqueId = (decltype(api.GetUniqueId)) dlsym(api.handle, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank)) dlsym(api.handle, "ncclCommInitRank");
    api.AllReduce = (decltype(api.AllReduce)) dlsym(api.handle, "ncclAllReduce");
    api.CommDestroy = (decltype(api.CommDestroy)) dlsym(api.handle, "ncclCommDestroy");
    api.GetErrorString = (decltype(api.GetErrorString)) dlsym(api.handle, "ncclGetErrorString");
    if (!api.GetUniqueId || !api.CommInitRank || !api.AllReduce || !api.CommDestroy) {
        MI_ERR("RCCL library lacks a required symbol");
        api.handle = nullptr;
        return nullptr;
    }
    return &api;
}

struct tp_state {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

int tp_get_unique_id(void * out, size_t size) {
    rccl_api * api = load_rccl();
    if (!api || size < sizeof(ncclUniqueId)) return -1;
    ncclUniqueId id;
    if (api->GetUniqueId(&id) != ncclSuccess) return -2;
    memcpy(out, &id, sizeof(id));
    return 0;
}

int tp_init(backend_ctx * c, int rank, int world, const void * uid, size_t uid_size) {
    if (world <= 1) return 0;
    rccl_api * api = load_rccl();
    if (!api || uid_size < sizeof(ncclUniqueId)) return -1;
    ncclUniqueId id;
    memcpy(&id, uid, sizeof(id));
    tp_state * t = new tp_state();
    t->rank = rank;
    t->world = world;
    HIP_CHECK(hipSetDevice(c->device));
    ncclResult_t r = api->CommInitRank(&t->comm, world, id, rank);
    if (r != ncclSuccess) {
        MI_ERR("ncclCommInitRank failed: %s", api->GetErrorString ? api->GetErrorString(r) : "?");
        delete t;
        return -2;
    }
    c->tp = t;
    // The all-reduces are enqueued on the backend's stream like every kernel, so a decode step with a communicator attached CAN be
    // captured and replayed as one hipGraph like a single-GPU step (RCCL supports stream capture; the first sighting of a topology runs
    // eagerly, which lets RCCL finish its lazy connection set-up outside a capture, and a failed capture falls back to eager execution:
    // graph.cpp).  A replay that hangs has no fallback, though, and this path has not run on multi-GPU hardware yet: with a communicator
    // attached, graphs are OPT-IN — GGML_MI355X_TP_GRAPHS=1, or the host's explicit set_option("graphs", 1) after tp_init (bench.py does
    // that for its second, watchdog-guarded leg).  ADVICE r02.
    // Only ever LOWERED here: a host that switched graphs off before tp_init keeps them off whatever the environment says.
    const char * e = getenv("GGML_MI355X_TP_GRAPHS");
    const bool env_on = e != nullptr && atoi(e) != 0;
    c->opt.graphs = c->opt.graphs && env_on;
    MI_INFO("tensor parallel rank %d / %d: hipGraph replay of steps with all-reduces is %s (GGML_MI355X_TP_GRAPHS=%s; set_option(\"graphs\", 1) after tp_init turns it on)",
            rank, world, c->opt.graphs ? "ON" : "off", e ? e : "unset");
    return 0;
}

bool tp_active(const backend_ctx * c) { return c->tp != nullptr && c->tp->world > 1; }

bool tp_all_reduce(backend_ctx * c, float * ptr, size_t n) {
    if (!tp_active(c)) return true;
    rccl_api * api = load_rccl();
    ncclResult_t r = api->AllReduce(ptr, ptr, n, ncclFloat32, ncclSum, c->tp->comm, c->stream);
    if (r != ncclSuccess) {
        MI_ERR("ncclAllReduce failed: %s", api->GetErrorString ? api->GetErrorString(r) : "?");
        return false;
    }
    return true;
}

void tp_free(backend_ctx * c) {
    if (!c->tp) return;
    rccl_api * api = load_rccl();
    if (api && c->tp->comm) api->CommDestroy(c->tp->comm);
    delete c->tp;
    c->tp = nullptr;
}

}  // namespace mi355x
