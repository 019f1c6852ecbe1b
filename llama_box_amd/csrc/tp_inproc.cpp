// tp_inproc.cpp — `-sm row --tensor-split ...` served as TENSOR PARALLELISM inside one process (round 5; VERDICT r04 "missing" #3, "next" #5).
//
// llama-box is ONE process (/root/reference/llama-box/engine.cpp:87-95).  With -sm row (engine_param.hpp:902-916, -ts :821-842) llama.cpp puts
// the mat-mul weights into the buffer type our registry returns for "ggml_backend_split_buffer_type" (split.cpp), everything else — norms,
// biases, embeddings, the KV cache, every activation — into the MAIN device's buffer, and hands every graph to the main device's backend.
// Rounds 3-4 executed such a graph node by node on the main device with one broadcast / gather / scatter / reduction around each sharded
// mat-mul (split.cpp: 5 cross-device hops per layer, attention and the KV cache on the main device only): correct, not fast.  The fast form
// existed only behind a launcher the reference does not have (one process per GPU, tp.cpp).  This file makes the one-process path BE that form:
//
//   * the host's graph is REWRITTEN, once per graph key, into one graph per device: every tensor is either replicated or sharded along one
//     dimension, and the shard a device owns follows from the split weights by a small set of propagation rules (rows of wq / wk / wv -> heads
//     -> that device's heads of the KV cache -> its heads of the attention result -> the K range of attn_output it holds; rows of gate / up ->
//     its K range of ffn_down).  A row-parallel mat-mul ends in a sum over the devices; nothing else crosses a device boundary inside a layer;
//   * each device executes ITS graph with the ordinary engine (graph.cpp: the fused Q/K/V launch, the attention kernels, the mat-vec prologues,
//     hipGraph replay — everything the one-device path has) on its own stream, from its own backend context;
//   * the two sums per layer are the one-shot peer-to-peer all-reduce of tp_p2p.hip between the process's own devices (plain peer-mapped
//     mailboxes: tp_attach_local), with the residual ADD and the next norm's sum of squares in the same launch;
//   * the KV cache is sharded with the heads: every device keeps its heads' rows in a SHADOW of the host's cache buffer (same offsets, narrower
//     rows).  The host's cache tensors on the main device stay authoritative for everything that is not such a graph — get_tensor (state save),
//     copies, the K-shift graph: the shards are gathered into them lazily when the host touches the buffer, and scattered back before the next
//     sharded graph if the host wrote (ip_host_access);
//   * per graph (not per layer) the inputs the host uploaded to the main device are copied to the other devices, and the vocab shards of the
//     logits are written into the main device's result tensor.
//
//   * graphs WITHOUT flash attention (llama-box's default, engine_param.hpp:772-779) store V transposed, one element per host-computed row index: the
//     transposed cache is sharded by rows, and every device gets its own cut-down copy of the index tensor rebased onto its shard (k_rebase_row_index, once
//     per graph); K.q, SOFT_MAX and V^T.p are batches over the heads a device owns.
//
// What it declines (the caller then runs the graph as before, split.cpp): shard boundaries that cut a head, a device without a share, ops outside the rules below.
// GGML_MI355X_SPLIT_TP=0 switches the engine off.  On ONE physical GPU (GGML_MI355X_FAKE_DEVICES) the devices' streams must not share a hardware
// queue — a device's all-reduce polls for its peers' contributions — so the tests run with GPU_MAX_HW_QUEUES >= the number of logical devices.
#include <algorithm>
#include <condition_variable>
#include <deque>
#include <list>
#include <mutex>
#include <thread>
#include <unordered_map>

#include "common.h"
#include "kernels.h"

namespace mi355x {

static constexpr int MAXD = GGML_MI355X_MAX_DEVICES;

struct sdesc {  // how a tensor of the host's graph is laid out over the devices
    int kind = 0;  // 0: replicated (every device holds / computes the whole tensor), 1: sharded along `dim`
    int dim = 0;
    int64_t off[MAXD + 1] = {0};  // device d owns [off[d], off[d + 1]) of dimension `dim`
};
static bool same_off(const sdesc & a, const sdesc & b, int n) {
    for (int i = 0; i <= n; ++i)
        if (a.off[i] != b.off[i]) return false;
    return true;
}

struct ip_mirror {  // a buffer of the host (compute buffer, KV cache buffer) and where its bytes live on every device
    ggml_backend_buffer_t host = nullptr;
    char * host_base = nullptr;
    size_t size = 0;
    char * base[MAXD] = {nullptr};  // non-KV: the main device's entry is the host's own memory
    bool kv = false;                // holds sharded cache tensors: every device (the main one too) works on a shadow
    int state = 0;                  // kv: 0 host memory and shadows agree, 1 the shadows are newer, 2 the host's memory is newer
};
struct ip_kv_tensor {  // a cache tensor of the host: [ne0, rows] sharded along ne0 (K, V: a device's heads of every cell), or — the TRANSPOSED V cache of graphs
                       // without flash attention, [n_ctx, n_embd_kv] — sharded along dimension 1 (dim == 1: a device's rows are one contiguous block)
    char * host_ptr = nullptr;
    int dim = 0;
    int type = 0;
    int64_t ne0 = 0, rows = 0;
    size_t host_row_bytes = 0;
    int64_t off[MAXD + 1] = {0};
    int mirror = 0;
};
struct ip_input { int mirror; size_t off, bytes; };  // a graph input in a mirrored buffer: copied from the main device to the others before every run
struct ip_output {                                    // a sharded result the host reads on the main device: [ne0 (sharded), cols]
    int node;          // its index in the graph
    char * host_ptr;
    int64_t ne0, cols;
    int64_t off[MAXD + 1];
    size_t stage_off;  // where every device's shard sits in its staging area
};
// graphs without flash attention store V transposed, one ELEMENT per row index: v_idxs[i * full + j] = j * n_ctx + slot_i (a host-computed input).  A device
// stores its rows [off[d], off[d + 1]) of every token into its shard of the cache through its own copy of that tensor, cut down and rebased (k_rebase_row_index)
struct ip_rebase {
    int mirror;        // where the host's index tensor lives
    size_t moff;
    int64_t n_tok, full, n_ctx;
    int64_t off[MAXD + 1];
    size_t stage_off;  // the device's tensor in its staging area
    std::vector<ggml_tensor *> clones[MAXD];
};
struct ip_plan {
    std::vector<uint64_t> key;
    bool declined = false;
    std::vector<ip_rebase> rebases;
    std::deque<ggml_tensor> store[MAXD];
    std::vector<ggml_tensor *> nodes[MAXD];
    ggml_cgraph graph[MAXD];
    std::vector<ip_input> inputs;
    std::vector<ip_output> outputs;
    size_t stage_bytes = 0;
    uint64_t last_use = 0;
};
struct ip_plan;
// one launcher thread per device other than the main one: a step's graphs are submitted to all devices AT ONCE (a replay is ~100 us of host time per
// device — key comparison + hipGraphLaunch —: eight of them in a row would hold the last device back by most of a millisecond of a ~3 ms step)
struct ip_worker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    ip_plan * job = nullptr;  // set by the main thread, cleared by the worker when the submission is done
    std::atomic<uint64_t> prepared{0};  // the engine epoch whose wait + input copies this device has issued
    bool quit = false;
    enum ggml_status result = GGML_STATUS_SUCCESS;
};
struct ip_engine {
    backend_ctx * main = nullptr;
    int n_dev = 0, main_dev = 0;
    backend_ctx * ctx[MAXD] = {nullptr};
    ggml_backend_t owned[MAXD] = {nullptr};
    int ordinal[MAXD] = {0};
    ggml_backend_buffer buf_plain[MAXD], buf_rowpar[MAXD];
    buffer_ctx bctx_plain[MAXD], bctx_rowpar[MAXD];
    std::vector<ip_mirror> mirrors;
    std::vector<ip_kv_tensor> kv;
    std::unordered_map<const void *, void *> replicas[MAXD];
    char * stage[MAXD] = {nullptr};
    size_t stage_size = 0;
    std::list<ip_plan> plans;
    ip_plan * last = nullptr;
    hipEvent_t ev_in = nullptr, ev_done[MAXD] = {nullptr};
    // "the inputs are on the main device" (ev_in) is an event of the main stream: every other device's stream must have been told to wait for it
    // BEFORE the main stream starts capturing its graph — HIP refuses a wait on an event of a capturing stream from outside the capture (error 905) —
    // so the main thread holds its own graph back until all devices have issued their waits and input copies (`prepared` below)
    uint64_t epoch = 0;
    uint64_t tick = 0;
    options opt_seen;
    bool dead = false;
    bool running = false;
    ip_worker * workers[MAXD] = {nullptr};
    bool threaded = false;
    std::mutex turn_m;
    std::condition_variable turn_cv;
    int turn = 0;
    // counters (ip_stat)
    int64_t graphs = 0, declined = 0, plans_built = 0, input_copies = 0, output_copies = 0, kv_gathers = 0, kv_scatters = 0, replica_bytes = 0;
};

static std::mutex g_ip_mtx;
static std::vector<ip_engine *> g_engines;
bool ip_any() { return !g_engines.empty(); }

static size_t type_row_bytes(int type, int64_t n) { return (size_t) (n / ggml_abi_blck_size((ggml_type) type)) * ggml_abi_type_size((ggml_type) type); }

// ------------------------------------------------------------------------------------------------ engine set-up
static void worker_main(ip_engine * E, int d);
static ip_engine * engine_new(backend_ctx * c, int n_dev, int main_dev) {
    if (n_dev < 2 || n_dev > MAXD || n_dev > P2P_MAX_RANKS) return nullptr;
    ip_engine * E = new ip_engine();
    E->main = c;
    E->n_dev = n_dev;
    E->main_dev = main_dev;
    bool ok = true;
    for (int d = 0; d < n_dev && ok; ++d) {
        E->ordinal[d] = logical_device_ordinal(d);
        if (d == main_dev) {
            E->ctx[d] = c;
        } else {
            E->owned[d] = internal_backend(d);
            ok = E->owned[d] != nullptr;
            if (ok) E->ctx[d] = (backend_ctx *) E->owned[d]->context;
        }
        make_internal_buffer(&E->buf_plain[d], &E->bctx_plain[d], E->ordinal[d], false);
        make_internal_buffer(&E->buf_rowpar[d], &E->bctx_rowpar[d], E->ordinal[d], true);
    }
    if (ok) {
        // every device reads the inputs / weights it replicates from the main device and writes its shard of the result there
        for (int d = 0; d < n_dev; ++d) {
            if (E->ordinal[d] == c->device) continue;
            int can = 0;
            if (hipSetDevice(E->ordinal[d]) == hipSuccess && hipDeviceCanAccessPeer(&can, E->ordinal[d], c->device) == hipSuccess && can) {
                if (hipDeviceEnablePeerAccess(c->device, 0) != hipSuccess) (void) hipGetLastError();
            }
            if (hipSetDevice(c->device) == hipSuccess && hipDeviceCanAccessPeer(&can, c->device, E->ordinal[d]) == hipSuccess && can) {
                if (hipDeviceEnablePeerAccess(E->ordinal[d], 0) != hipSuccess) (void) hipGetLastError();
            }
        }
        (void) hipGetLastError();
        ok = tp_attach_local(E->ctx, n_dev) == 0;
    }
    for (int d = 0; d < n_dev && ok; ++d) {
        ok = hipSetDevice(E->ordinal[d]) == hipSuccess && hipEventCreateWithFlags(&E->ev_done[d], hipEventDisableTiming) == hipSuccess;
        // the workers start from the main backend's options (a host that switched graphs or fusion off means all of it)
        if (ok && d != main_dev) E->ctx[d]->opt = c->opt;
    }
    ok = ok && hipSetDevice(c->device) == hipSuccess && hipEventCreateWithFlags(&E->ev_in, hipEventDisableTiming) == hipSuccess;
    (void) hipSetDevice(c->device);
    if (!ok) {
        (void) hipGetLastError();
        MI_ERR("in-process tensor parallel: set-up over %d devices failed; -sm row graphs run node by node on the main device (split.cpp)", n_dev);
        E->dead = true;
    }
    // One launcher thread per device (GGML_MI355X_SPLIT_THREADS=0: the main thread submits the devices one after the other, ascending).  Until late in
    // round 5 this was opt-in: with the threads — or with any order in which a device other than the main one submits first — the first graph came out wrong
    // on most boxes.  The cause was replica_of(): non-split weights were copied with a device-to-device hipMemcpy, which returns before the copy has run,
    // and the early device read half-copied norm weights.  Same box, same binary, five alternating runs each: the copy as it was 8 of 10 wrong, the
    // stream-ordered copy 10 of 10 equal to the one-device logits (profiles/r05_inproc_tp_submit_order.txt).
    static const bool threads_on = !(getenv("GGML_MI355X_SPLIT_THREADS") && atoi(getenv("GGML_MI355X_SPLIT_THREADS")) == 0);
    if (ok && threads_on) {
        for (int d = 0; d < n_dev; ++d) {
            if (d == main_dev) continue;
            E->workers[d] = new ip_worker();
            E->workers[d]->th = std::thread(worker_main, E, d);
        }
        E->threaded = true;
    }
    E->opt_seen = c->opt;
    std::lock_guard<std::mutex> lk(g_ip_mtx);
    g_engines.push_back(E);
    return E;
}

void ip_free(backend_ctx * c) {
    ip_engine * E = c->ip;
    if (!E) return;
    {
        std::lock_guard<std::mutex> lk(g_ip_mtx);
        g_engines.erase(std::remove(g_engines.begin(), g_engines.end(), E), g_engines.end());
    }
    for (int d = 0; d < E->n_dev; ++d) {
        if (ip_worker * W = E->workers[d]) {
            {
                std::lock_guard<std::mutex> lk(W->m);
                W->quit = true;
                W->cv.notify_all();
            }
            W->th.join();
            delete W;
            E->workers[d] = nullptr;
        }
    }
    for (int d = 0; d < E->n_dev; ++d) {
        if (hipSetDevice(E->ordinal[d]) != hipSuccess) { (void) hipGetLastError(); continue; }
        if (E->ctx[d] && E->ctx[d]->stream) (void) hipStreamSynchronize(E->ctx[d]->stream);
    }
    for (int d = 0; d < E->n_dev; ++d) {
        if (hipSetDevice(E->ordinal[d]) != hipSuccess) { (void) hipGetLastError(); continue; }
        for (ip_mirror & m : E->mirrors)
            if (m.base[d] && m.base[d] != m.host_base) (void) hipFree(m.base[d]);
        for (auto & kv : E->replicas[d]) (void) hipFree(kv.second);
        if (E->stage[d]) (void) hipFree(E->stage[d]);
        if (E->ev_done[d]) (void) hipEventDestroy(E->ev_done[d]);
        if (E->owned[d]) E->owned[d]->iface.free(E->owned[d]);  // (be_free: frees the worker's own tp_state and scratch)
    }
    (void) hipSetDevice(c->device);
    if (E->ev_in) (void) hipEventDestroy(E->ev_in);
    (void) hipGetLastError();
    delete E;
    c->ip = nullptr;
}

int64_t ip_stat(const backend_ctx * c, const char * key) {
    const ip_engine * E = c->ip;
    const std::string k = key;
    if (k == "ip_graphs") return E ? E->graphs : 0;
    if (k == "ip_declined") return E ? E->declined : 0;
    if (k == "ip_plans") return E ? E->plans_built : 0;
    if (k == "ip_input_copies") return E ? E->input_copies : 0;
    if (k == "ip_output_copies") return E ? E->output_copies : 0;
    if (k == "ip_kv_gathers") return E ? E->kv_gathers : 0;
    if (k == "ip_kv_scatters") return E ? E->kv_scatters : 0;
    if (k == "ip_devices") return E && !E->dead ? E->n_dev : 0;
    if (k == "ip_worker_kernel_launches") {  // launches issued by the OTHER devices' contexts (the main one counts its own)
        int64_t n = 0;
        for (int d = 0; E && d < E->n_dev; ++d)
            if (d != E->main_dev && E->ctx[d]) n += E->ctx[d]->st.kernel_launches;
        return n;
    }
    if (k == "ip_worker_p2p_timeouts") {  // time-outs of the all-reduce on the OTHER devices (stat "p2p_timeouts" is the main device's)
        int64_t n = 0;
        for (int d = 0; E && d < E->n_dev; ++d)
            if (d != E->main_dev && E->ctx[d]) n += std::max<int64_t>(0, tp_p2p_timeouts(E->ctx[d]));
        return n;
    }
    if (k.rfind("ip_dbg_timeouts_", 0) == 0) {  // ip_dbg_timeouts_<d>: the all-reduce time-outs of device d's context
        const int d = atoi(key + 16);
        return E && d >= 0 && d < E->n_dev && E->ctx[d] ? tp_p2p_timeouts(E->ctx[d]) : -2;
    }
    if (k == "ip_worker_graph_launches") {
        int64_t n = 0;
        for (int d = 0; E && d < E->n_dev; ++d)
            if (d != E->main_dev && E->ctx[d]) n += E->ctx[d]->st.graph_launches;
        return n;
    }
    return -1;
}

// ------------------------------------------------------------------------------------------------ memory: mirrors, replicas, KV shadows
static ggml_backend_buffer_t root_buffer(const ggml_tensor * t) { return t->view_src ? t->view_src->buffer : t->buffer; }

static int mirror_of(ip_engine * E, ggml_backend_buffer_t b) {
    for (size_t i = 0; i < E->mirrors.size(); ++i)
        if (E->mirrors[i].host == b) return (int) i;
    if (!buffer_is_ours(b)) return -1;
    ip_mirror m;
    m.host = b;
    m.host_base = (char *) b->iface.get_base(b);
    m.size = b->size;
    E->mirrors.push_back(m);
    return (int) E->mirrors.size() - 1;
}
// the device copies of mirror `mi` exist (kv: on every device, zero-filled like a cleared cache; otherwise on the devices other than the main one)
static bool mirror_ready(ip_engine * E, int mi, bool kv) {
    ip_mirror & m = E->mirrors[(size_t) mi];
    if (kv && !m.kv) {
        m.kv = true;
        m.state = 2;  // whatever the host's cache holds right now is what the shards must start from
        if (m.base[E->main_dev] == m.host_base) m.base[E->main_dev] = nullptr;
    }
    for (int d = 0; d < E->n_dev; ++d) {
        if (m.base[d]) continue;
        if (!m.kv && d == E->main_dev) { m.base[d] = m.host_base; continue; }
        if (hipSetDevice(E->ordinal[d]) != hipSuccess || hipMalloc((void **) &m.base[d], m.size + 256) != hipSuccess) {
            (void) hipGetLastError();
            MI_ERR("in-process tensor parallel: %.1f MiB for a buffer mirror on device %d failed", m.size / 1048576.0, E->ordinal[d]);
            (void) hipSetDevice(E->main->device);
            return false;
        }
        // (on the device's OWN stream: a plain hipMemset runs on the null stream, which the devices' non-blocking streams do not wait for — it zeroed
        // a device's mirror in the middle of its first graph whenever that device got going before the others)
        if (hipMemsetAsync(m.base[d], 0, m.size, E->ctx[d]->stream) != hipSuccess) (void) hipGetLastError();
    }
    (void) hipSetDevice(E->main->device);
    return true;
}
static void * replica_of(ip_engine * E, int d, const ggml_tensor * t) {
    if (d == E->main_dev) return t->data;
    auto it = E->replicas[d].find(t->data);
    if (it != E->replicas[d].end()) return it->second;
    const size_t n = ggml_abi_nbytes(t);
    void * p = nullptr;
    if (hipSetDevice(E->ordinal[d]) != hipSuccess || hipMalloc(&p, n + 256) != hipSuccess) { (void) hipGetLastError(); (void) hipSetDevice(E->main->device); return nullptr; }
    // ON THE DEVICE'S OWN STREAM: a device-to-device hipMemcpy returns before the copy has run (it goes to the null stream, which the devices' non-blocking
    // streams do not wait for) — a device that got going before the others read half-copied norm weights and embeddings in its first graph (the open question
    // of profiles/r05_inproc_tp_submit_order.txt, answered late in round 5).  The main device's memory is read over peer access by a copy kernel.
    static const bool dbg_null_stream = getenv("GGML_MI355X_DBG_REPLICA_MEMCPY") != nullptr;  // (the A/B of the evidence file: the copy as it was)
    if (dbg_null_stream) (void) hipMemcpyPeer(p, E->ordinal[d], t->data, E->main->device, n);
    else launch_copy2d(E->ctx[d]->stream, p, n, t->data, n, n, 1);
    const hipError_t e = hipGetLastError();
    (void) hipSetDevice(E->main->device);
    if (e != hipSuccess) return nullptr;
    E->replicas[d][t->data] = p;
    E->replica_bytes += (int64_t) n;
    return p;
}

// gather (to_host) / scatter of every registered cache tensor of mirror `mi`; blocking (rare: state save, context shift, the first sharded graph)
static bool kv_move(ip_engine * E, int mi, bool to_host) {
    ip_mirror & m = E->mirrors[(size_t) mi];
    bool ok = true;
    for (int d = 0; d < E->n_dev && ok; ++d) {
        ok = hipSetDevice(E->ordinal[d]) == hipSuccess;
        hipStream_t s = E->ctx[d]->stream;
        for (const ip_kv_tensor & t : E->kv) {
            if (t.mirror != mi || !ok) continue;
            const int64_t ext = t.off[d + 1] - t.off[d];
            if (ext <= 0) continue;
            if (t.dim == 1) {  // rows [off[d], off[d + 1]) of the transposed cache: one contiguous block on either side
                char * hostb = t.host_ptr + (size_t) t.off[d] * t.host_row_bytes;
                char * shardb = m.base[d] + (t.host_ptr - m.host_base);
                const size_t nb = (size_t) ext * t.host_row_bytes;
                if (to_host) launch_copy2d(s, hostb, nb, shardb, nb, nb, 1);
                else launch_copy2d(s, shardb, nb, hostb, nb, nb, 1);
                continue;
            }
            char * host = t.host_ptr + type_row_bytes(t.type, t.off[d]);
            char * shard = m.base[d] + (t.host_ptr - m.host_base);
            const size_t w = type_row_bytes(t.type, ext);
            if (to_host) launch_copy2d(s, host, t.host_row_bytes, shard, w, w, (size_t) t.rows);
            else launch_copy2d(s, shard, w, host, t.host_row_bytes, w, (size_t) t.rows);
        }
        ok = ok && hipStreamSynchronize(s) == hipSuccess;
    }
    (void) hipSetDevice(E->main->device);
    if (!ok) { (void) hipGetLastError(); MI_ERR("in-process tensor parallel: moving the KV shards %s failed", to_host ? "into the host's cache" : "out of the host's cache"); }
    (to_host ? E->kv_gathers : E->kv_scatters)++;
    return ok;
}

void ip_host_access(ggml_backend_buffer_t b, bool write) {
    if (g_engines.empty()) return;
    std::lock_guard<std::mutex> lk(g_ip_mtx);
    for (ip_engine * E : g_engines) {
        for (size_t i = 0; i < E->mirrors.size(); ++i) {
            ip_mirror & m = E->mirrors[i];
            if (m.host != b || !m.kv) continue;
            if (m.state == 1) {  // the shards are newer than the host's tensors: bring them home first (a write may cover only part of them)
                (void) hipSetDevice(E->main->device);
                (void) hipStreamSynchronize(E->main->stream);
                if (!kv_move(E, (int) i, true)) {  // the host's tensors stay stale: the shards remain the newer copy, and everything after this fails loudly
                    note_hip_failure();
                    continue;
                }
                m.state = 0;
            }
            if (write) m.state = 2;
        }
    }
}

// the host frees one of its buffers (a context goes away: its compute buffer, its KV cache): the device copies, the cache tensors registered in it and
// every plan (they hold addresses inside mirrors) go with it — the next graph over a new buffer at the same address starts from nothing
void ip_host_buffer_freed(ggml_backend_buffer_t b) {
    if (g_engines.empty()) return;
    std::lock_guard<std::mutex> lk(g_ip_mtx);
    const char * b0 = buffer_is_ours(b) ? (const char *) b->iface.get_base(b) : nullptr;
    for (ip_engine * E : g_engines) {
        // replicas of weights that lived in this buffer (norm weights, biases, embeddings of a model that is being freed)
        bool purged = false;
        for (int d = 0; d < E->n_dev && b0; ++d) {
            for (auto it = E->replicas[d].begin(); it != E->replicas[d].end();) {
                if ((const char *) it->first >= b0 && (const char *) it->first < b0 + b->size) {
                    if (hipSetDevice(E->ordinal[d]) == hipSuccess) {
                        if (E->ctx[d] && E->ctx[d]->stream) (void) hipStreamSynchronize(E->ctx[d]->stream);
                        (void) hipFree(it->second);
                    }
                    it = E->replicas[d].erase(it);
                    purged = true;
                } else {
                    ++it;
                }
            }
        }
        if (purged) {
            for (int d = 0; d < E->n_dev; ++d)
                if (hipSetDevice(E->ordinal[d]) == hipSuccess) free_graph_cache(E->ctx[d]);
            (void) hipSetDevice(E->main->device);
            (void) hipGetLastError();
            E->plans.clear();
            E->last = nullptr;
        }
        int mi = -1;
        for (size_t i = 0; i < E->mirrors.size(); ++i)
            if (E->mirrors[i].host == b) mi = (int) i;
        if (mi < 0) continue;
        for (int d = 0; d < E->n_dev; ++d) {
            if (hipSetDevice(E->ordinal[d]) != hipSuccess) { (void) hipGetLastError(); continue; }
            if (E->ctx[d] && E->ctx[d]->stream) (void) hipStreamSynchronize(E->ctx[d]->stream);
            free_graph_cache(E->ctx[d]);  // (captured graphs of the devices' contexts hold mirror addresses)
            char * p = E->mirrors[(size_t) mi].base[d];
            if (p && p != E->mirrors[(size_t) mi].host_base) (void) hipFree(p);
        }
        (void) hipSetDevice(E->main->device);
        (void) hipGetLastError();
        E->plans.clear();
        E->last = nullptr;
        // (indices into `mirrors` are stored in the cache-tensor registry: rebuild it without the freed buffer's entries)
        std::vector<ip_kv_tensor> keep;
        for (const ip_kv_tensor & k : E->kv) {
            if (k.mirror == mi) continue;
            ip_kv_tensor k2 = k;
            if (k2.mirror > mi) k2.mirror--;
            keep.push_back(k2);
        }
        E->kv.swap(keep);
        E->mirrors.erase(E->mirrors.begin() + mi);
    }
}

// ------------------------------------------------------------------------------------------------ analysis: who owns what
struct analysis {
    ip_engine * E;
    const ggml_cgraph * g;
    int n;
    std::unordered_map<const ggml_tensor *, sdesc> d;
    std::unordered_map<const ggml_tensor *, int> node_index;
    struct vidx_t { int64_t n_tok, full, n_ctx; int64_t off[MAXD + 1]; };
    std::unordered_map<const ggml_tensor *, vidx_t> vidx;  // per-element row-index inputs of the transposed V cache (see ip_rebase)
    const char * why = "";
    const ggml_tensor * where = nullptr;
    bool fail(const char * w, const ggml_tensor * t) { why = w; where = t; return false; }
};
static bool is_weight(const ggml_tensor * t) {
    ggml_backend_buffer_t b = root_buffer(t);
    return b != nullptr && b->usage == GGML_BACKEND_BUFFER_USAGE_WEIGHTS;
}
static bool leaf_desc(analysis & A, const ggml_tensor * t, sdesc & o) {
    o = sdesc();
    if (buffer_is_split(t->buffer)) {
        const split_tensor_info * info = split_info(t);
        if (!info || info->n_dev != A.n) return A.fail("split weight of another device count", t);
        o.kind = 1;
        o.dim = info->kind == 1 ? 0 : 1;
        for (int i = 0; i <= A.n; ++i) o.off[i] = info->row0[i];
        for (int i = 0; i < A.n; ++i)
            if (o.off[i + 1] <= o.off[i]) return A.fail("a device without a share of a split weight", t);
        return true;
    }
    for (const ip_kv_tensor & k : A.E->kv)  // a cache tensor sharded by an earlier graph stays sharded the same way
        if (k.host_ptr == (char *) t->data && t->view_src == nullptr) {
            o.kind = 1;
            o.dim = k.dim;
            for (int i = 0; i <= A.n; ++i) o.off[i] = k.off[i];
            return true;
        }
    return true;  // replicated: an input, a norm weight, a bias, the embeddings
}
static bool get_desc(analysis & A, const ggml_tensor * t, sdesc & o) {
    auto it = A.d.find(t);
    if (it != A.d.end()) { o = it->second; return true; }
    if (t->op != GGML_OP_NONE) return A.fail("a source that is neither a leaf nor an earlier node", t);
    if (!leaf_desc(A, t, o)) return false;
    A.d[t] = o;
    return true;
}
// dst has the elements of src in the same order (RESHAPE): the shard of src along s.dim becomes a shard of dst along some dimension, or nothing does
static bool reshape_shard(const ggml_tensor * src, const sdesc & s, const ggml_tensor * dst, sdesc & o, int n) {
    int64_t inner = 1;
    for (int i = 0; i < s.dim; ++i) inner *= src->ne[i];
    const int64_t slab = inner * src->ne[s.dim];
    int64_t acc = 1;
    for (int k = 0; k < 4; ++k) {
        if (acc * dst->ne[k] == slab && dst->ne[k] > 1) {
            bool ok = true;
            for (int i = 0; i <= n && ok; ++i) ok = (s.off[i] * inner) % acc == 0;
            if (ok) {
                o.kind = 1;
                o.dim = k;
                for (int i = 0; i <= n; ++i) o.off[i] = s.off[i] * inner / acc;
                return true;
            }
        }
        acc *= dst->ne[k];
        if (acc > slab) break;
    }
    return false;
}

static bool analyse_node(analysis & A, const ggml_tensor * t) {
    sdesc a, b, o;
    const ggml_tensor * s0 = t->src[0];
    const ggml_tensor * s1 = t->src[1];
    auto all_replicated = [&]() {
        for (int s = 0; s < GGML_MAX_SRC; ++s) {
            if (!t->src[s] || t->src[s] == t) continue;  // (ggml_cast: the CPY node names itself as src[1])
            sdesc x;
            if (!get_desc(A, t->src[s], x)) return false;
            if (x.kind != 0) return A.fail("an op without a sharding rule reads a sharded tensor", t);
        }
        A.d[t] = sdesc();
        return true;
    };
    switch (t->op) {
        case GGML_OP_MUL_MAT: {
            if (!get_desc(A, s0, a) || !get_desc(A, s1, b)) return false;
            if (buffer_is_split(s0->buffer)) {
                const split_tensor_info * info = split_info(s0);
                if (info->kind == 0) {  // column-parallel: every device computes its rows of the result from the whole activation
                    if (b.kind != 0) return A.fail("a row-split weight multiplies a sharded activation", t);
                    o.kind = 1;
                    o.dim = 0;
                    for (int i = 0; i <= A.n; ++i) o.off[i] = a.off[i];
                } else {  // row-parallel: the activation is sharded along K exactly as the weight is; the partial products are summed over the devices
                    if (b.kind != 1 || b.dim != 0 || !same_off(a, b, A.n)) return A.fail("the K ranges of a row-parallel weight and of its activation differ", t);
                    o = sdesc();
                }
                A.d[t] = o;
                return true;
            }
            if (a.kind == 1 && b.kind == 1 && a.dim == 2 && b.dim == 2) {
                // graphs without flash attention: K.q and V^T.p — a batch of mat-muls over the head dimension, both operands sharded by heads
                for (int i = 0; i <= A.n; ++i)
                    if (b.off[i] * s0->ne[2] != a.off[i] * s1->ne[2]) return A.fail("query heads and KV heads are cut at different places", t);
                o.kind = 1;
                o.dim = 2;
                for (int i = 0; i <= A.n; ++i) o.off[i] = b.off[i];
                A.d[t] = o;
                return true;
            }
            if (a.kind != 0 || b.kind != 0) return A.fail("a mat-mul of sharded operands that is not a split weight's", t);
            A.d[t] = sdesc();
            return true;
        }
        case GGML_OP_ADD: case GGML_OP_SUB: case GGML_OP_MUL: case GGML_OP_DIV: {
            if (!get_desc(A, s0, a) || !get_desc(A, s1, b)) return false;
            if (a.kind == 0 && b.kind == 0) { A.d[t] = sdesc(); return true; }
            if (a.kind == 1 && b.kind == 1 && a.dim == b.dim && same_off(a, b, A.n) && ggml_abi_nelements(s0) == ggml_abi_nelements(s1)) { A.d[t] = a; return true; }
            // a sharded activation + a replicated row (bias): the row is sliced
            if (a.kind == 1 && a.dim == 0 && b.kind == 0 && s1->op == GGML_OP_NONE && s1->ne[0] == s0->ne[0] && s1->ne[1] == 1 && s1->ne[2] == 1 && s1->ne[3] == 1 &&
                s1->type == GGML_TYPE_F32 && ggml_abi_is_contiguous(s1)) {
                A.d[t] = a;
                return true;
            }
            return A.fail("an element-wise op over differently sharded operands", t);
        }
        case GGML_OP_GLU: {
            if (!get_desc(A, s0, a)) return false;
            if (s1) {
                if (!get_desc(A, s1, b)) return false;
                if (a.kind != b.kind || (a.kind == 1 && (a.dim != b.dim || !same_off(a, b, A.n)))) return A.fail("GLU halves sharded differently", t);
                A.d[t] = a;
                return true;
            }
            if (a.kind != 0) return A.fail("a fused-halves GLU of a sharded tensor", t);
            A.d[t] = sdesc();
            return true;
        }
        case GGML_OP_RESHAPE: {
            if (!get_desc(A, s0, a)) return false;
            if (a.kind == 0) { A.d[t] = sdesc(); return true; }
            if (!reshape_shard(s0, a, t, o, A.n)) {
                // [head_dim, n_head_kv, n_tokens] -> [1, n_tokens * n_embd_kv] (the source rows of the transposed V store): in the HOST's order a device's
                // elements are strided, in the device's own packed tensor they are simply all of it — a share of the rows, consumed only by SET_ROWS below
                if (!(t->ne[0] == 1 && t->ne[2] == 1 && t->ne[3] == 1 && ggml_abi_nelements(t) % s0->ne[a.dim] == 0)) return A.fail("a reshape that does not keep the shard in one dimension", t);
                const int64_t per = ggml_abi_nelements(t) / s0->ne[a.dim];
                o.kind = 1;
                o.dim = 1;
                for (int i = 0; i <= A.n; ++i) o.off[i] = a.off[i] * per;
            }
            A.d[t] = o;
            return true;
        }
        case GGML_OP_PERMUTE: {
            if (!get_desc(A, s0, a)) return false;
            if (a.kind == 1) a.dim = t->op_params[a.dim];
            A.d[t] = a;
            return true;
        }
        case GGML_OP_TRANSPOSE: {
            if (!get_desc(A, s0, a)) return false;
            if (a.kind == 1 && a.dim < 2) a.dim = 1 - a.dim;
            A.d[t] = a;
            return true;
        }
        case GGML_OP_VIEW: {
            if (!get_desc(A, s0, a)) return false;
            if (a.kind == 0) { A.d[t] = sdesc(); return true; }
            // the attention's window on a sharded cache tensor [n_embd_kv, n_ctx]: [head_dim, n_kv, n_head_kv] with the cache's row stride and one
            // head between neighbours of dimension 2 — the heads of dimension 2 are the shard
            if (s0->op == GGML_OP_NONE && a.dim == 0 && t->view_offs == 0 && t->nb[1] == s0->nb[1] && t->ne[0] * t->ne[2] == s0->ne[0] &&
                t->nb[2] == type_row_bytes(t->type, t->ne[0]) && t->ne[3] == 1) {
                o.kind = 1;
                o.dim = 2;
                for (int i = 0; i <= A.n; ++i) {
                    if (a.off[i] % t->ne[0] != 0) return A.fail("a cache shard boundary inside a head", t);
                    o.off[i] = a.off[i] / t->ne[0];
                }
                A.d[t] = o;
                return true;
            }
            // ... on the TRANSPOSED V cache [n_ctx, n_embd_kv] (no flash attention): [n_kv, head_dim, n_head_kv], rows of the cache per dimension 1, one head of rows per step of dimension 2
            if (s0->op == GGML_OP_NONE && a.dim == 1 && t->view_offs == 0 && t->nb[1] == s0->nb[1] && t->ne[1] * t->ne[2] == s0->ne[1] && t->nb[2] == t->nb[1] * (size_t) t->ne[1] && t->ne[3] == 1) {
                o.kind = 1;
                o.dim = 2;
                for (int i = 0; i <= A.n; ++i) {
                    if (a.off[i] % t->ne[1] != 0) return A.fail("a cache shard boundary inside a head", t);
                    o.off[i] = a.off[i] / t->ne[1];
                }
                A.d[t] = o;
                return true;
            }
            return A.fail("a view of a sharded tensor that is not the attention's cache window", t);
        }
        case GGML_OP_ROPE: {
            if (!get_desc(A, s0, a) || !get_desc(A, s1, b)) return false;
            if (b.kind != 0) return A.fail("sharded positions", t);
            if (t->src[2]) { sdesc f; if (!get_desc(A, t->src[2], f) || f.kind != 0) return A.fail("sharded frequency factors", t); }
            if (a.kind == 1 && a.dim == 0) return A.fail("a rotation across a shard boundary", t);
            A.d[t] = a;
            return true;
        }
        case GGML_OP_SET_ROWS: {
            const ggml_tensor * dst = t->view_src;
            if (!dst || dst->op != GGML_OP_NONE || dst->view_src != nullptr) return A.fail("SET_ROWS into something that is not a plain cache tensor", t);
            if (!get_desc(A, s0, a) || !get_desc(A, s1, b)) return false;
            if (b.kind != 0) return A.fail("sharded row indices", t);
            sdesc cd;
            if (!get_desc(A, dst, cd)) return false;
            if (a.kind == 0) {
                if (cd.kind != 0) return A.fail("replicated rows into a sharded cache", t);
                A.d[t] = sdesc();
                return true;
            }
            if (t->ne[0] == 1 && a.dim == 1 && s0->ne[0] == 1 && dst->ne[2] == 1 && dst->ne[3] == 1 && s1->op == GGML_OP_NONE && s1->type == GGML_TYPE_I64 && ggml_abi_is_contiguous(s1) &&
                ggml_abi_nelements(s1) == s0->ne[1] && s0->ne[1] % dst->ne[1] == 0) {
                // the transposed V cache [n_ctx, n_embd_kv], one element per row index: a device's rows of the cache are its share of every token's values
                const int64_t n_tok = s0->ne[1] / dst->ne[1];
                sdesc leaf;
                leaf.kind = 1;
                leaf.dim = 1;
                for (int i = 0; i <= A.n; ++i) {
                    if (a.off[i] % n_tok != 0) return A.fail("the transposed V store's share is not whole rows per token", t);
                    leaf.off[i] = a.off[i] / n_tok;
                }
                if (cd.kind == 1 && (cd.dim != 1 || !same_off(cd, leaf, A.n))) return A.fail("a cache tensor sharded differently by another graph", t);
                analysis::vidx_t v{n_tok, dst->ne[1], dst->ne[0], {0}};
                for (int i = 0; i <= A.n; ++i) v.off[i] = leaf.off[i];
                A.vidx[s1] = v;
                A.d[dst] = leaf;
                o.kind = 1;
                o.dim = 1;
                for (int i = 0; i <= A.n; ++i) o.off[i] = leaf.off[i] * dst->ne[0];
                A.d[t] = o;
                return true;
            }
            if (a.dim != 0 || dst->ne[0] != s0->ne[0] || dst->ne[2] != 1 || dst->ne[3] != 1) return A.fail("rows sharded along another dimension than the cache row", t);
            if (cd.kind == 1 && !same_off(cd, a, A.n)) return A.fail("a cache tensor sharded differently by another graph", t);
            const int64_t blck = ggml_abi_blck_size(dst->type);
            for (int i = 0; i <= A.n; ++i)
                if (a.off[i] % blck != 0) return A.fail("a cache shard boundary inside a quantised block", t);
            a.dim = 0;
            A.d[dst] = a;  // the cache tensor itself is sharded from here on (its shadow holds this device's heads)
            A.d[t] = a;
            return true;
        }
        case GGML_OP_FLASH_ATTN_EXT: {
            sdesc k, v, m;
            if (!get_desc(A, s0, a) || !get_desc(A, s1, k) || !get_desc(A, t->src[2], v)) return false;
            if (t->src[3] && (!get_desc(A, t->src[3], m) || m.kind != 0)) return A.fail("a sharded mask", t);
            if (t->src[4]) { sdesc sk; if (!get_desc(A, t->src[4], sk) || sk.kind != 0) return A.fail("sharded sinks", t); }
            if (a.kind == 0 && k.kind == 0 && v.kind == 0) { A.d[t] = sdesc(); return true; }
            if (a.kind != 1 || k.kind != 1 || v.kind != 1 || a.dim != 2 || k.dim != 2 || v.dim != 2 || !same_off(k, v, A.n)) return A.fail("attention over operands not sharded by heads", t);
            for (int i = 0; i <= A.n; ++i)  // a device's query heads are exactly the groups of its KV heads
                if (a.off[i] * s1->ne[2] != k.off[i] * s0->ne[2]) return A.fail("query heads and KV heads are cut at different places", t);
            // the kernels derive ALiBi slopes (n_head_log2, m0 / m1) and sinks[h] from the head count and head index of the tensor they are given: a
            // device's share would get the slopes / sinks of heads 0..n_local-1.  Declined — such graphs run node by node (attention on the main device)
            if (ggml_abi_op_param_f32(t, 1) != 0.0f) return A.fail("attention with ALiBi slopes (max_bias) sharded by heads", t);
            if (t->src[4]) return A.fail("attention with per-head sinks sharded by heads", t);
            o.kind = 1;
            o.dim = 1;  // result: [head_dim, n_head, n_tokens]
            for (int i = 0; i <= A.n; ++i) o.off[i] = a.off[i];
            A.d[t] = o;
            return true;
        }
        case GGML_OP_SOFT_MAX: {
            if (!get_desc(A, s0, a)) return false;
            if (a.kind == 0) return all_replicated();
            if (a.dim == 0) return A.fail("a soft-max across a shard boundary", t);
            if (s1 && (!get_desc(A, s1, b) || b.kind != 0)) return A.fail("a sharded mask", t);
            if (t->src[2]) { sdesc sk; if (!get_desc(A, t->src[2], sk) || sk.kind != 0) return A.fail("sharded sinks", t); }
            if (a.dim >= 2) {  // rows of [n_kv, n_tokens, n_head, ...] sharded by heads: the same two per-head quantities as in FLASH_ATTN_EXT above
                if (ggml_abi_op_param_f32(t, 1) != 0.0f) return A.fail("a soft-max with ALiBi slopes (max_bias) sharded by heads", t);
                if (t->src[2]) return A.fail("a soft-max with per-head sinks sharded by heads", t);
            }
            A.d[t] = a;
            return true;
        }
        case GGML_OP_CONT: {
            if (!get_desc(A, s0, a)) return false;
            if (a.kind == 0) return all_replicated();
            if (!reshape_shard(s0, a, t, o, A.n)) return A.fail("a copy that does not keep the shard in one dimension", t);  // (cont_2d of the permuted attention result)
            A.d[t] = o;
            return true;
        }
        case GGML_OP_GET_ROWS: case GGML_OP_RMS_NORM: case GGML_OP_SCALE: case GGML_OP_UNARY: case GGML_OP_CPY: case GGML_OP_DUP: case GGML_OP_ARGMAX:
            return all_replicated();
        default:
            return A.fail("an op the engine has no rule for", t);
    }
}

// ------------------------------------------------------------------------------------------------ rewriting: one graph per device
struct rewriter {
    ip_engine * E;
    ip_plan * P;
    analysis * A;
    int dev;
    std::unordered_map<const ggml_tensor *, ggml_tensor *> done;
    std::unordered_map<const ggml_tensor *, ggml_tensor *> vidx_done;  // this device's cut-down copies of per-element row-index inputs (ip_rebase)
    bool ok = true;
};
static void contiguous_strides(ggml_tensor * c) {
    c->nb[0] = ggml_abi_type_size(c->type);
    c->nb[1] = c->nb[0] * (size_t) (c->ne[0] / ggml_abi_blck_size(c->type));
    c->nb[2] = c->nb[1] * (size_t) c->ne[1];
    c->nb[3] = c->nb[2] * (size_t) c->ne[2];
}
static char * mirror_addr(rewriter & R, const ggml_tensor * t, bool kv) {
    const int mi = mirror_of(R.E, root_buffer(t));
    if (mi < 0 || !mirror_ready(R.E, mi, kv)) { R.ok = false; return nullptr; }
    const ip_mirror & m = R.E->mirrors[(size_t) mi];
    return m.base[R.dev] + ((char *) t->data - m.host_base);
}
static ggml_tensor * clone_of(rewriter & R, const ggml_tensor * t);
static ggml_tensor * new_clone(rewriter & R, const ggml_tensor * t) {
    R.P->store[R.dev].push_back(*t);
    ggml_tensor * c = &R.P->store[R.dev].back();
    c->extra = nullptr;
    c->buffer = &R.E->buf_plain[R.dev];
    for (int s = 0; s < GGML_MAX_SRC; ++s) c->src[s] = nullptr;
    c->view_src = nullptr;
    return c;
}
static ggml_tensor * clone_leaf(rewriter & R, const ggml_tensor * t) {
    const sdesc & sd = R.A->d[t];
    ggml_tensor * c = new_clone(R, t);
    const int d = R.dev;
    if (buffer_is_split(t->buffer)) {
        const split_tensor_info * info = split_info(t);
        const int64_t ext = sd.off[d + 1] - sd.off[d];
        if (info->kind == 0) {
            c->ne[1] = ext;
            c->nb[1] = info->row_bytes;
        } else {
            c->ne[0] = ext;
            c->nb[1] = info->slice_row_bytes[d];
            c->buffer = &R.E->buf_rowpar[d];  // graph.cpp: the product of this weight is a partial sum -> tp_all_reduce
        }
        c->nb[2] = c->nb[1] * (size_t) c->ne[1];
        c->nb[3] = c->nb[2];
        c->data = info->slice[d];
        return c;
    }
    if (sd.kind == 1) {  // a cache tensor: this device's heads of every cell (or its rows of the transposed V cache), in the shadow of the host's cache buffer
        const int64_t ext = sd.off[d + 1] - sd.off[d];
        c->ne[sd.dim] = ext;
        contiguous_strides(c);
        c->data = mirror_addr(R, t, true);
        return c;
    }
    if (is_weight(t)) {
        c->data = replica_of(R.E, d, t);
        if (!c->data) R.ok = false;
        return c;
    }
    c->data = mirror_addr(R, t, false);  // an input of the graph (or any other tensor the host keeps in a buffer of ours)
    return c;
}
static ggml_tensor * clone_of(rewriter & R, const ggml_tensor * t) {
    auto it = R.done.find(t);
    if (it != R.done.end()) return it->second;
    ggml_tensor * c = nullptr;
    if (t->op == GGML_OP_NONE) {
        c = clone_leaf(R, t);
        R.done[t] = c;
        return c;
    }
    const sdesc & sd = R.A->d[t];
    const int d = R.dev;
    c = new_clone(R, t);
    R.done[t] = c;
    for (int s = 0; s < GGML_MAX_SRC; ++s)
        if (t->src[s]) c->src[s] = clone_of(R, t->src[s]);
    if (t->view_src) c->view_src = clone_of(R, t->view_src);
    const int64_t ext = sd.kind == 1 ? sd.off[d + 1] - sd.off[d] : 0;
    switch (t->op) {
        case GGML_OP_RESHAPE:
            if (sd.kind == 1) { c->ne[sd.dim] = ext; contiguous_strides(c); }
            c->data = c->src[0]->data;
            break;
        case GGML_OP_PERMUTE:
            for (int i = 0; i < 4; ++i) { c->ne[t->op_params[i]] = c->src[0]->ne[i]; c->nb[t->op_params[i]] = c->src[0]->nb[i]; }
            c->data = c->src[0]->data;
            break;
        case GGML_OP_TRANSPOSE:
            c->ne[0] = c->src[0]->ne[1]; c->ne[1] = c->src[0]->ne[0]; c->nb[0] = c->src[0]->nb[1]; c->nb[1] = c->src[0]->nb[0];
            c->ne[2] = c->src[0]->ne[2]; c->ne[3] = c->src[0]->ne[3]; c->nb[2] = c->src[0]->nb[2]; c->nb[3] = c->src[0]->nb[3];
            c->data = c->src[0]->data;
            break;
        case GGML_OP_VIEW:
            if (sd.kind == 1) {  // the attention's window on the sharded cache: this device's heads, the shard's row stride (the transposed cache keeps n_ctx-long rows)
                c->ne[2] = ext;
                c->nb[1] = c->src[0]->nb[1];
                c->nb[3] = c->nb[2] * (size_t) c->ne[2];
            }
            c->data = (char *) (c->view_src ? c->view_src->data : c->src[0]->data) + t->view_offs;
            break;
        case GGML_OP_SET_ROWS:
            if (t->ne[0] == 1 && sd.kind == 1) {  // the transposed V cache seen as rows of ONE element each: this device's [n_ctx, rows] block of it, flat
                c->ne[1] = ext;
                contiguous_strides(c);
                auto vi = R.A->vidx.find(t->src[1]);
                if (vi != R.A->vidx.end()) {
                    ggml_tensor *& ix = R.vidx_done[t->src[1]];
                    if (!ix) {
                        ix = new_clone(R, t->src[1]);
                        ix->ne[0] = vi->second.n_tok * (vi->second.off[d + 1] - vi->second.off[d]);
                        ix->ne[1] = ix->ne[2] = ix->ne[3] = 1;
                        contiguous_strides(ix);
                        ix->data = nullptr;  // (its place in the staging area is set before every run)
                    }
                    c->src[1] = ix;
                }
            } else {
                for (int i = 0; i < 4; ++i) { c->ne[i] = c->view_src->ne[i]; c->nb[i] = c->view_src->nb[i]; }
            }
            c->data = c->view_src->data;
            break;
        default:
            if (sd.kind == 1) { c->ne[sd.dim] = ext; contiguous_strides(c); }
            c->data = mirror_addr(R, t, false);
            break;
    }
    // a replicated row added to a sharded activation (bias): this device's slice of it
    if ((t->op == GGML_OP_ADD || t->op == GGML_OP_SUB || t->op == GGML_OP_MUL || t->op == GGML_OP_DIV) && sd.kind == 1 && R.A->d[t->src[1]].kind == 0) {
        ggml_tensor * b = new_clone(R, t->src[1]);
        b->ne[0] = ext;
        contiguous_strides(b);
        b->data = (char *) c->src[1]->data + (size_t) sd.off[d] * sizeof(float);
        c->src[1] = b;
    }
    return c;
}

static ip_plan * build_plan(ip_engine * E, const ggml_cgraph * g) {
    E->plans.emplace_back();
    ip_plan * P = &E->plans.back();
    graph_key_build(g, P->key);
    E->plans_built++;
    analysis A;
    A.E = E;
    A.g = g;
    A.n = E->n_dev;
    bool any_split = false;
    for (int i = 0; i < g->n_nodes; ++i) {
        const ggml_tensor * t = g->nodes[i];
        if (!analyse_node(A, t)) {
            MI_INFO("in-process tensor parallel declines a graph of %d nodes: %s (node %d '%s'); it runs node by node on the main device", g->n_nodes, A.why, i, A.where ? A.where->name : "?");
            P->declined = true;
            return P;
        }
        any_split = any_split || (t->op == GGML_OP_MUL_MAT && buffer_is_split(t->src[0]->buffer));
    }
    if (!any_split) { P->declined = true; return P; }
    // register the cache tensors this graph shards (and refuse a non-flash graph: its V cache is transposed and indexed per element by the host)
    for (auto & kv : A.d) {
        const ggml_tensor * t = kv.first;
        if (t->op != GGML_OP_NONE || kv.second.kind != 1 || buffer_is_split(t->buffer)) continue;
        bool known = false;
        for (const ip_kv_tensor & k : E->kv) known = known || k.host_ptr == (char *) t->data;
        if (known) continue;
        const int mi = mirror_of(E, root_buffer(t));
        if (mi < 0) { P->declined = true; return P; }
        ip_kv_tensor k;
        k.host_ptr = (char *) t->data;
        k.dim = kv.second.dim;
        k.type = t->type;
        k.ne0 = t->ne[0];
        k.rows = t->ne[1];
        k.host_row_bytes = t->nb[1];
        for (int i = 0; i <= E->n_dev; ++i) k.off[i] = kv.second.off[i];
        k.mirror = mi;
        E->kv.push_back(k);
        if (E->mirrors[(size_t) mi].kv && E->mirrors[(size_t) mi].state == 0) E->mirrors[(size_t) mi].state = 2;  // (a tensor joins a live shadow: its rows come from the host's copy)
    }
    for (int d = 0; d < E->n_dev; ++d) {
        rewriter R;
        R.E = E;
        R.P = P;
        R.A = &A;
        R.dev = d;
        for (int i = 0; i < g->n_nodes && R.ok; ++i) P->nodes[d].push_back(clone_of(R, g->nodes[i]));
        if (!R.ok) {
            MI_ERR("in-process tensor parallel: building device %d's graph failed (memory for a mirror or a replica?)", E->ordinal[d]);
            P->declined = true;
            return P;
        }
        for (auto & vd : R.vidx_done) {  // this device's cut-down index tensors: one rebase per distinct host tensor
            const analysis::vidx_t & v = A.vidx[vd.first];
            ip_rebase * rb = nullptr;
            const int mi = mirror_of(E, root_buffer(vd.first));
            if (mi < 0) { P->declined = true; return P; }
            const size_t moff = (size_t) ((char *) vd.first->data - E->mirrors[(size_t) mi].host_base);
            for (ip_rebase & x : P->rebases)
                if (x.mirror == mi && x.moff == moff) rb = &x;
            if (!rb) {
                P->rebases.emplace_back();
                rb = &P->rebases.back();
                rb->mirror = mi;
                rb->moff = moff;
                rb->n_tok = v.n_tok;
                rb->full = v.full;
                rb->n_ctx = v.n_ctx;
                for (int k = 0; k <= E->n_dev; ++k) rb->off[k] = v.off[k];
                int64_t widest = 0;
                for (int k = 0; k < E->n_dev; ++k) widest = std::max(widest, v.off[k + 1] - v.off[k]);
                rb->stage_off = P->stage_bytes;
                P->stage_bytes += ((size_t) (v.n_tok * widest) * 8 + 255) & ~(size_t) 255;
            }
            rb->clones[d].push_back(vd.second);
        }
        ggml_cgraph & cg = P->graph[d];
        memset(&cg, 0, sizeof(cg));
        cg.size = (int) P->nodes[d].size();
        cg.n_nodes = (int) P->nodes[d].size();
        cg.nodes = P->nodes[d].data();
        if (d != 0) continue;
        // what the main device must hand to the others before a run: the leaves in mirrored (non-cache) buffers — the graph's inputs
        for (auto & kv : R.done) {
            const ggml_tensor * t = kv.first;
            if (t->op != GGML_OP_NONE || buffer_is_split(t->buffer) || is_weight(t) || A.d[t].kind != 0) continue;
            const int mi = mirror_of(E, root_buffer(t));
            if (mi < 0 || E->mirrors[(size_t) mi].kv) continue;
            P->inputs.push_back({mi, (size_t) ((char *) t->data - E->mirrors[(size_t) mi].host_base), ggml_abi_nbytes(t)});
        }
    }
    // sharded results the host reads on the main device (the vocab shards of the logits): every device computes into a staging area of its own
    for (int i = 0; i < g->n_nodes; ++i) {
        const ggml_tensor * t = g->nodes[i];
        const sdesc & sd = A.d[t];
        if (!(t->flags & GGML_TENSOR_FLAG_OUTPUT) || sd.kind == 0) continue;
        if (sd.dim != 0 || t->type != GGML_TYPE_F32 || !ggml_abi_is_contiguous(t)) { P->declined = true; return P; }
        ip_output o;
        o.node = i;
        o.host_ptr = (char *) t->data;
        o.ne0 = t->ne[0];
        o.cols = ggml_abi_nelements(t) / t->ne[0];
        for (int k = 0; k <= E->n_dev; ++k) o.off[k] = sd.off[k];
        o.stage_off = P->stage_bytes;
        int64_t widest = 0;
        for (int k = 0; k < E->n_dev; ++k) widest = std::max(widest, sd.off[k + 1] - sd.off[k]);
        P->stage_bytes += ((size_t) widest * (size_t) o.cols * 4 + 255) & ~(size_t) 255;
        P->outputs.push_back(o);
        for (int d = 0; d < E->n_dev; ++d) P->nodes[d][(size_t) i]->data = nullptr;  // (set per run: the staging areas may still grow)
    }
    return P;
}

// ------------------------------------------------------------------------------------------------ execution
static bool same_options(const options & a, const options & b) {
    return a.graphs == b.graphs && a.fusion == b.fusion && a.prologue == b.prologue && a.qkv == b.qkv && a.mmvq_max_cols == b.mmvq_max_cols && a.mmq_min_cols == b.mmq_min_cols &&
           a.mmq_i8 == b.mmq_i8 && a.mm_merge == b.mm_merge && a.ss_partials == b.ss_partials && a.fa_self_merge == b.fa_self_merge && a.attn_nf == b.attn_nf &&
           a.softmax_mm == b.softmax_mm && a.skinny_rope == b.skinny_rope && a.skinny_mix == b.skinny_mix && a.mmq_skinny == b.mmq_skinny && a.mmq_bn == b.mmq_bn &&
           a.fa_splits == b.fa_splits && a.fa_wo == b.fa_wo && a.small_uploads == b.small_uploads && a.small_downloads == b.small_downloads && a.timing == b.timing;
}
static bool ensure_stage(ip_engine * E, size_t need) {
    if (need <= E->stage_size) return true;
    for (int d = 0; d < E->n_dev; ++d) {
        if (hipSetDevice(E->ordinal[d]) != hipSuccess) { (void) hipGetLastError(); return false; }
        (void) hipStreamSynchronize(E->ctx[d]->stream);
        if (E->stage[d]) (void) hipFree(E->stage[d]);
        E->stage[d] = nullptr;
        if (hipMalloc((void **) &E->stage[d], need + (1u << 20)) != hipSuccess) { (void) hipGetLastError(); (void) hipSetDevice(E->main->device); return false; }
        free_graph_cache(E->ctx[d]);  // (captured graphs hold the old staging address)
    }
    E->stage_size = need + (1u << 20);
    (void) hipSetDevice(E->main->device);
    return true;
}

// device d's part of a step, first half: its stream waits for the inputs and copies them over (leaves device d current)
static enum ggml_status prepare_device(ip_engine * E, ip_plan * P, int d) {
    backend_ctx * w = E->ctx[d];
    HIP_TRY(hipSetDevice(E->ordinal[d]), GGML_STATUS_FAILED);
    if (d != E->main_dev) {
        HIP_TRY(hipStreamWaitEvent(w->stream, E->ev_in, 0), GGML_STATUS_FAILED);
        for (const ip_input & in : P->inputs) {
            const ip_mirror & m = E->mirrors[(size_t) in.mirror];
            // a copy KERNEL of this device reading the main device's memory over peer access: an ordinary launch of this stream, behind the event wait
            // above by construction (an asynchronous device-to-device memcpy may take a DMA engine with an ordering of its own)
            launch_copy2d(w->stream, m.base[d] + in.off, in.bytes, m.host_base + in.off, in.bytes, in.bytes, 1);
        }
    }
    // graphs without flash attention: this device's share of the host's per-element V row indices, rebased onto its shard of the transposed cache
    for (const ip_rebase & rb : P->rebases) {
        const ip_mirror & m = E->mirrors[(size_t) rb.mirror];
        launch_rebase_row_index(w->stream, (int64_t *) (E->stage[d] + rb.stage_off), (const int64_t *) (m.base[d] + rb.moff), rb.n_tok, rb.full, rb.off[d + 1] - rb.off[d], rb.off[d], rb.n_ctx);
    }
    return hipGetLastError() == hipSuccess ? GGML_STATUS_SUCCESS : GGML_STATUS_FAILED;
}
// ... second half: its graph
static enum ggml_status launch_device(ip_engine * E, ip_plan * P, int d) {
    backend_ctx * w = E->ctx[d];
    HIP_TRY(hipSetDevice(E->ordinal[d]), GGML_STATUS_FAILED);
    // replayed graphs with cross-device waits: on unless GGML_MI355X_SPLIT_GRAPHS=0 (the kernels' spins are bounded: a device that does not show
    // up costs a failed llama_decode, not a hang)
    static const bool graphs_ok = !getenv("GGML_MI355X_SPLIT_GRAPHS") || atoi(getenv("GGML_MI355X_SPLIT_GRAPHS")) != 0;
    const bool graphs_was = w->opt.graphs;
    w->opt.graphs = graphs_was && graphs_ok;
    // debugging aid: GGML_MI355X_DBG_SUBMIT_ORDER=asc | desc | mdesc — the devices' graphs are submitted one after the other in that order
    static const char * dbg_order = getenv("GGML_MI355X_DBG_SUBMIT_ORDER");
    if (dbg_order && E->threaded) {
        const int my = dbg_order[0] == 'd' ? E->n_dev - 1 - d : (dbg_order[0] == 'm' ? (d == E->main_dev ? 0 : E->n_dev - d) : d);
        std::unique_lock<std::mutex> lk(E->turn_m);
        E->turn_cv.wait(lk, [&] { return E->turn == my; });
    }
    const enum ggml_status st = graph_compute(w, &P->graph[d]);
    if (dbg_order && E->threaded) {
        std::lock_guard<std::mutex> lk(E->turn_m);
        E->turn++;
        E->turn_cv.notify_all();
    }
    w->opt.graphs = graphs_was;
    return st;
}
static void worker_main(ip_engine * E, int d) {
    ip_worker * W = E->workers[d];
    std::unique_lock<std::mutex> lk(W->m);
    for (;;) {
        W->cv.wait(lk, [&] { return W->job != nullptr || W->quit; });
        if (W->quit) return;
        ip_plan * P = W->job;
        lk.unlock();
        enum ggml_status st = prepare_device(E, P, d);
        W->prepared.store(E->epoch, std::memory_order_release);  // (the main thread starts its own graph — possibly a capture — only after this)
        if (st == GGML_STATUS_SUCCESS) st = launch_device(E, P, d);
        lk.lock();
        W->result = st;
        W->job = nullptr;
        W->cv.notify_all();
    }
}

static enum ggml_status run_plan(ip_engine * E, ip_plan * P, const ggml_cgraph * g) {
    backend_ctx * c = E->main;
    struct guard { int dev; ~guard() { if (hipSetDevice(dev) != hipSuccess) (void) hipGetLastError(); } } restore{c->device};
    // options follow the main backend (tests and hosts flip graphs / fusion there)
    if (!same_options(E->opt_seen, c->opt)) {
        for (int d = 0; d < E->n_dev; ++d) {
            if (d == E->main_dev) continue;
            if (hipSetDevice(E->ordinal[d]) == hipSuccess) (void) hipStreamSynchronize(E->ctx[d]->stream);
            E->ctx[d]->opt = c->opt;
            free_graph_cache(E->ctx[d]);
        }
        E->opt_seen = c->opt;
        (void) hipSetDevice(c->device);
    }
    if (!ensure_stage(E, P->stage_bytes)) return GGML_STATUS_ALLOC_FAILED;
    (void) g;
    for (const ip_output & o : P->outputs)
        for (int d = 0; d < E->n_dev; ++d) P->nodes[d][(size_t) o.node]->data = E->stage[d] + o.stage_off;
    for (const ip_rebase & rb : P->rebases)
        for (int d = 0; d < E->n_dev; ++d)
            for (ggml_tensor * ix : rb.clones[d]) ix->data = E->stage[d] + rb.stage_off;
    // the host's cache tensors were written behind the engine's back (or never sharded yet): hand every device its heads
    for (size_t mi = 0; mi < E->mirrors.size(); ++mi) {
        ip_mirror & m = E->mirrors[mi];
        if (!m.kv) continue;
        if (m.state == 2) {
            HIP_TRY(hipStreamSynchronize(c->stream), GGML_STATUS_FAILED);
            if (!kv_move(E, (int) mi, false)) return GGML_STATUS_FAILED;
        }
        m.state = 1;
    }
    // the inputs exist on the main stream from here on
    HIP_TRY(hipSetDevice(c->device), GGML_STATUS_FAILED);
    HIP_TRY(hipEventRecord(E->ev_in, c->stream), GGML_STATUS_FAILED);
    enum ggml_status st = GGML_STATUS_SUCCESS;
    E->epoch++;
    E->input_copies += (int64_t) P->inputs.size() * (E->n_dev - 1);
    E->turn = 0;
    if (E->threaded) {
        for (int d = 0; d < E->n_dev; ++d) {
            if (d == E->main_dev) continue;
            ip_worker * W = E->workers[d];
            std::lock_guard<std::mutex> lk(W->m);
            W->job = P;
            W->cv.notify_all();
        }
    }
    // every device's stream is told to wait for the inputs before the main stream may start a capture
    for (int d = 0; d < E->n_dev; ++d) {
        if (d == E->main_dev) continue;
        if (E->threaded) {
            while (E->workers[d]->prepared.load(std::memory_order_acquire) != E->epoch) std::this_thread::yield();
        } else if (st == GGML_STATUS_SUCCESS) {
            st = prepare_device(E, P, d);
        }
    }
    if (st == GGML_STATUS_SUCCESS) st = prepare_device(E, P, E->main_dev);
    if (st == GGML_STATUS_SUCCESS) {
        E->running = true;  // (the main device's context re-enters graph_compute with ITS graph: not a graph for the engine)
        st = launch_device(E, P, E->main_dev);
        E->running = false;
    }
    if (st == GGML_STATUS_SUCCESS) HIP_TRY(hipEventRecord(E->ev_done[E->main_dev], c->stream), GGML_STATUS_FAILED);  // "the main device is through with its graph"
    for (int d = 0; d < E->n_dev; ++d) {
        if (d == E->main_dev) continue;
        if (E->threaded) {
            ip_worker * W = E->workers[d];
            std::unique_lock<std::mutex> lk(W->m);
            W->cv.wait(lk, [&] { return W->job == nullptr; });
            if (W->result != GGML_STATUS_SUCCESS) st = W->result;
        } else if (st == GGML_STATUS_SUCCESS) {
            st = launch_device(E, P, d);
        }
    }
    // Every device's rows of a sharded result go straight into the host's tensor on the main device — but only once the MAIN device's graph has
    // ended: the host's allocator gave that tensor a block its graph may use for intermediates until the last node (a device that finishes first
    // would write its logits over them: random 4e-2 deviations in the first 70B-shape runs)
    for (int d = 0; d < E->n_dev && st == GGML_STATUS_SUCCESS; ++d) {
        backend_ctx * w = E->ctx[d];
        HIP_TRY(hipSetDevice(E->ordinal[d]), GGML_STATUS_FAILED);
        if (!P->outputs.empty() && d != E->main_dev) HIP_TRY(hipStreamWaitEvent(w->stream, E->ev_done[E->main_dev], 0), GGML_STATUS_FAILED);
        for (const ip_output & o : P->outputs) {
            const int64_t ext = o.off[d + 1] - o.off[d];
            launch_copy2d(w->stream, o.host_ptr + (size_t) o.off[d] * 4, (size_t) o.ne0 * 4, E->stage[d] + o.stage_off, (size_t) ext * 4, (size_t) ext * 4, (size_t) o.cols);
            E->output_copies++;
        }
        if (d != E->main_dev) HIP_TRY(hipEventRecord(E->ev_done[d], w->stream), GGML_STATUS_FAILED);
    }
    HIP_TRY(hipSetDevice(c->device), GGML_STATUS_FAILED);
    if (st == GGML_STATUS_SUCCESS)
        for (int d = 0; d < E->n_dev; ++d)
            if (d != E->main_dev) HIP_TRY(hipStreamWaitEvent(c->stream, E->ev_done[d], 0), GGML_STATUS_FAILED);
    if (st != GGML_STATUS_SUCCESS) {
        // a device failed in the middle of a step: the others may be polling for its sums — let every stream run out (the spins are bounded)
        for (int d = 0; d < E->n_dev; ++d)
            if (hipSetDevice(E->ordinal[d]) == hipSuccess) (void) hipStreamSynchronize(E->ctx[d]->stream);
        (void) hipGetLastError();
    }
    E->graphs++;
    return st;
}

enum ggml_status ip_graph_compute(backend_ctx * c, ggml_cgraph * g, bool * handled) {
    *handled = false;
    static const bool on = !getenv("GGML_MI355X_SPLIT_TP") || atoi(getenv("GGML_MI355X_SPLIT_TP")) != 0;
    if (!on || c->capturing) return GGML_STATUS_SUCCESS;
    ip_engine * E = c->ip;
    if (E && (E->dead || E->running)) return GGML_STATUS_SUCCESS;
    ip_plan * P = nullptr;
    if (E && E->last && graph_key_equals(g, E->last->key)) {
        P = E->last;
    } else {
        // a graph with row-split weights?  (the scan is the price of every graph of a backend that has an engine, and of none otherwise)
        const ggml_tensor * w = nullptr;
        for (int i = 0; i < g->n_nodes && !w; ++i)
            if (g->nodes[i]->op == GGML_OP_MUL_MAT && buffer_is_split(g->nodes[i]->src[0]->buffer)) w = g->nodes[i]->src[0];
        if (!w) {
            if (E) {  // a graph the engine does not take (the K-shift, a state copy): it may touch the host's cache tensors
                for (int i = 0; i < g->n_nodes; ++i) {
                    const ggml_tensor * t = g->nodes[i];
                    if (root_buffer(t)) ip_host_access(root_buffer(t), true);
                    for (int s = 0; s < GGML_MAX_SRC; ++s)
                        if (t->src[s] && root_buffer(t->src[s])) ip_host_access(root_buffer(t->src[s]), false);
                }
            }
            return GGML_STATUS_SUCCESS;
        }
        if (!E) {
            const split_tensor_info * info = split_info(w);
            const int main_dev = split_buft_main_device(w->buffer->buft);
            if (!info || main_dev < 0 || main_dev >= info->n_dev || logical_device_ordinal(main_dev) != c->device) return GGML_STATUS_SUCCESS;
            E = c->ip = engine_new(c, info->n_dev, main_dev);
            if (!E || E->dead) return GGML_STATUS_SUCCESS;
        }
        std::vector<uint64_t> key;
        graph_key_build(g, key);
        for (ip_plan & p : E->plans)
            if (p.key == key) { P = &p; break; }
        if (!P) {
            if (E->plans.size() > 48) {  // bound the cache: forget the plans that have not run for a while
                for (auto it = E->plans.begin(); it != E->plans.end();) {
                    if (&*it != E->last && it->last_use + 64 < E->tick) it = E->plans.erase(it);
                    else ++it;
                }
            }
            P = build_plan(E, g);
        }
    }
    E->tick++;
    P->last_use = E->tick;
    if (P->declined) {
        E->declined++;
        // the old path works on the host's cache tensors on the main device
        for (size_t mi = 0; mi < E->mirrors.size(); ++mi)
            if (E->mirrors[mi].kv) ip_host_access(E->mirrors[mi].host, true);
        return GGML_STATUS_SUCCESS;
    }
    E->last = P;
    *handled = true;
    return run_plan(E, P, g);
}

}  // namespace mi355x
