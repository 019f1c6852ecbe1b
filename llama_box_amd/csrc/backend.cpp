// backend.cpp — registration, device, buffer-type, buffer and backend vtables of the MI355X ggml backend.
//
// The shapes of these tables are the drop-in boundary (include/ggml_abi.h cites the in-tree
// evidence for each signature).  Design is MI355X-first: one HIP stream per backend instance (llama-box forces
// GPU_MAX_HW_QUEUES=1, /root/reference/llama-box/engine.cpp:16-20, so nothing here relies on queue-level
// overlap), buffers are plain hipMalloc arenas sized for 288 GB parts, repeated graphs are replayed as hipGraphs.
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <mutex>
#include <thread>

#include "common.h"
#include "kernels.h"

namespace mi355x {

int log_level() {
    static int lvl = [] {
        const char * e = getenv("GGML_MI355X_LOG");
        return e ? atoi(e) : 0;
    }();
    return lvl;
}

// ------------------------------------------------------------------------------------------------ devices
struct device_ctx {
    int device = 0;  // HIP ordinal
    std::string name, description;
    ggml_backend_buffer_type buft{};
    ggml_backend_buffer_type buft_rowpar{};
    ggml_backend_buffer_type buft_host{};
};

struct reg_ctx {
    std::vector<ggml_backend_device *> devices;
};

static ggml_backend_reg g_reg;
static reg_ctx g_reg_ctx;
static std::once_flag g_reg_once;
static ggml_guid g_guid = {0x4d, 0x49, 0x33, 0x35, 0x35, 0x58, 0x2d, 0x67, 0x67, 0x6d, 0x6c, 0x2d, 0x62, 0x6b, 0x6e, 0x64};

static device_ctx * dctx(ggml_backend_dev_t dev) { return (device_ctx *) dev->context; }
int logical_device_count() { return (int) g_reg_ctx.devices.size(); }
int logical_device_ordinal(int i) { return dctx(g_reg_ctx.devices[(size_t) i])->device; }
ggml_backend_dev_t logical_device(int i) { return g_reg_ctx.devices[(size_t) i]; }

// ------------------------------------------------------------------------------------------------ buffers
typedef ggml_backend_buffer_t (*buffer_init_fn)(ggml_backend_buffer_type_t, struct ggml_backend_buffer_i, void *, size_t);

// libggml-base's ggml_backend_buffer_init when we are loaded by a ggml host; own allocation otherwise (the host
// frees the object with `delete`, both sides share libstdc++'s allocator)
ggml_backend_buffer_t make_backend_buffer(ggml_backend_buffer_type_t buft, const ggml_backend_buffer_i & iface, void * context, size_t size);
static ggml_backend_buffer_t make_buffer(ggml_backend_buffer_type_t buft, const ggml_backend_buffer_i & iface, void * context, size_t size) {
    return make_backend_buffer(buft, iface, context, size);
}
ggml_backend_buffer_t make_backend_buffer(ggml_backend_buffer_type_t buft, const ggml_backend_buffer_i & iface, void * context, size_t size) {
    static buffer_init_fn host_fn = (buffer_init_fn) dlsym(RTLD_DEFAULT, "ggml_backend_buffer_init");
    if (host_fn) return host_fn(buft, iface, context, size);
    return new ggml_backend_buffer{iface, buft, context, size, GGML_BACKEND_BUFFER_USAGE_ANY};
}

static void uploader_drain(int device);
// ---- the decode copy's bookkeeping (common.h: buffer_ctx::shadow)
static std::atomic<uint64_t> g_decode_epoch{1};
uint64_t decode_copy_epoch() { return g_decode_epoch.load(std::memory_order_acquire); }
// bytes [off, off + n) of the buffer are about to change: the copies of tensors in that range are stale from now on
static void decode_copy_drop(buffer_ctx * c, size_t off, size_t n) {
    if (c->shadow == nullptr) return;
    std::lock_guard<std::mutex> lk(c->sh_mtx);
    if (c->sh_valid.empty()) return;
    bool any = false;
    for (auto it = c->sh_valid.begin(); it != c->sh_valid.end();) {
        if (it->first < off + n && off < it->first + it->second) { it = c->sh_valid.erase(it); any = true; }
        else ++it;
    }
    if (any) g_decode_epoch.fetch_add(1, std::memory_order_acq_rel);
}
static void decode_copy_drop(ggml_backend_buffer_t b, const ggml_tensor * t, size_t offset, size_t size) {
    buffer_ctx * c = (buffer_ctx *) b->context;
    if (c->shadow != nullptr && t->data != nullptr) decode_copy_drop(c, (size_t) ((const char *) t->data - (const char *) c->base) + offset, size);
}
// (q80_panels: the OTHER second copy — Q8_0 matrices regrouped into the 32-row x 4-block tiles the 9 .. 32-column matrix-core kernel loads with whole-line
// wave-instructions, repack.hip: k_repack_q80_panels.  A tensor has one kind of copy: the K-quants planes, Q8_0 panels; same shadow allocation, same invalidation)
static const uint8_t * second_copy(backend_ctx * c, const ggml_tensor * w, const bool q80_panels) {
    if (!c->opt.decode_copy || w == nullptr || w->view_src != nullptr || w->data == nullptr || w->ne[2] != 1 || w->ne[3] != 1) return nullptr;
    ggml_backend_buffer_t b = w->buffer;
    if (b == nullptr || !buffer_is_ours(b) || b->usage != GGML_BACKEND_BUFFER_USAGE_WEIGHTS || buffer_is_split(b) || buffer_is_rowpar(b)) return nullptr;
    if (q80_panels ? !repack_q80_supported(w->type, w->ne[0], w->ne[1], (int64_t) w->nb[1]) : !repack_supported(w->type, w->ne[0], (int64_t) w->nb[1])) return nullptr;
    buffer_ctx * bc = (buffer_ctx *) b->context;
    if (bc->device != c->device || bc->shadow_failed) return nullptr;
    const size_t off = (size_t) ((const char *) w->data - (const char *) bc->base), bytes = (size_t) w->nb[1] * (size_t) w->ne[1];
    if (off + bytes > bc->size) return nullptr;
    std::lock_guard<std::mutex> lk(bc->sh_mtx);
    auto it = bc->sh_valid.find(off);
    if (it != bc->sh_valid.end() && it->second == bytes) return (const uint8_t *) bc->shadow + off;
    if (c->capturing) return nullptr;  // (a capture is the SECOND run of a graph: its weights were repacked by the first; anything else keeps the block layout)
    if (bc->shadow == nullptr) {
        // 288 GB of HBM: a second copy of the weights one GPU serves always fits beside the first — unless the user filled the device on purpose; then the
        // mat-vecs keep reading the block layout, as before round 6
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < bc->size + ((size_t) std::max(0, c->opt.decode_copy_headroom_gib) << 30) || hipMalloc(&bc->shadow, bc->size) != hipSuccess) {
            (void) hipGetLastError();
            bc->shadow = nullptr;
            bc->shadow_failed = true;
            MI_INFO("decode copy: no room for a second copy of a %.1f GiB weights buffer on device %d — the mat-vec kernels read the block layout", (double) bc->size / (1 << 30), bc->device);
            return nullptr;
        }
    }
    if (q80_panels) launch_repack_q80_panels(c->stream, w->data, (char *) bc->shadow + off, w->ne[0], w->ne[1], (int64_t) w->nb[1]);
    else launch_repack_planes(c->stream, w->type, w->data, (char *) bc->shadow + off, w->ne[0], (int64_t) w->nb[1], 0, w->ne[1]);
    // (another backend instance of this device — its own stream — may find the entry valid a moment later: the copy is complete before it is announced)
    if (hipStreamSynchronize(c->stream) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
    bc->sh_valid[off] = bytes;
    c->st.decode_copy_tensors++;
    c->st.decode_copy_bytes += (int64_t) bytes;
    return (const uint8_t *) bc->shadow + off;
}
const uint8_t * decode_copy(backend_ctx * c, const ggml_tensor * w) { return second_copy(c, w, false); }
const uint8_t * q80_panel_copy(backend_ctx * c, const ggml_tensor * w) { return second_copy(c, w, true); }
static void buf_free(ggml_backend_buffer_t b) {
    buffer_ctx * c = (buffer_ctx *) b->context;
    if (ip_any()) ip_host_buffer_freed(b);  // (tp_inproc.cpp mirrors host buffers on the other devices)
    HIP_SOFT(hipSetDevice(c->device));
    uploader_drain(c->device);  // a staged upload may still be writing into this arena
    if (c->shadow) {  // (graphs captured over it must never be replayed: a new buffer may get the same addresses)
        HIP_SOFT(hipDeviceSynchronize());
        HIP_NOTE(hipFree(c->shadow));
        g_decode_epoch.fetch_add(1, std::memory_order_acq_rel);
    }
    HIP_NOTE(hipFree(c->base));
    delete c;
}
static void * buf_get_base(ggml_backend_buffer_t b) { return ((buffer_ctx *) b->context)->base; }
static enum ggml_status buf_init_tensor(ggml_backend_buffer_t, ggml_tensor *) { return GGML_STATUS_SUCCESS; }
static void buf_memset_tensor(ggml_backend_buffer_t b, ggml_tensor * t, uint8_t value, size_t offset, size_t size) {
    buffer_ctx * c = (buffer_ctx *) b->context;
    if (ip_any()) ip_host_access(b, true);
    HIP_SOFT(hipSetDevice(c->device));
    decode_copy_drop(b, t, offset, size);
    uploader_drain(c->device);
    HIP_SOFT(hipMemset((char *) t->data + offset, value, size));
    HIP_SOFT(hipDeviceSynchronize());
}
// ---- the loader's fast path: set_tensor of a weight
// llama.cpp's loader calls buffer.set_tensor once per tensor with pageable memory (the mmap'd GGUF, or a read buffer with --no-mmap:
// /root/reference/llama-box/engine_param.hpp:403); 42 GB for the 70B model.  A hipMemcpy from pageable memory is staged by the runtime
// through its own bounce buffer on the calling thread, copy and DMA taking turns.  Here uploads above 1 MiB go through a per-device
// engine: a ring of pinned slots, the host copy into a slot spread over a few threads, hipMemcpyAsync out of the slot on a dedicated
// stream — the DMA of slot i overlaps the host copy of slot i+1, ACROSS set_tensor calls too: the call returns once its bytes sit in
// pinned memory (the caller may reuse `data`), the last DMAs still in flight.  Everything else that can observe the device memory
// waits for the engine first: the buffer vtable through uploader_drain(), a backend's stream through an event (uploader_join()).
// No repack: the bytes land verbatim.  Any failure inside falls back to the plain synchronous copy.
struct uploader {
    // device / failed / in_flight are read without the mutex on the fast paths (every decode step passes uploader_join): atomics.
    // Initialisation happens once, under mtx (two RPC connections may issue their first set_tensor concurrently — ADVICE r03).
    std::atomic<int> device{-1};
    std::atomic<bool> failed{false};
    hipStream_t stream = nullptr;
    static constexpr int NSLOT = 3;
    char * slot[NSLOT] = {nullptr, nullptr, nullptr};
    hipEvent_t ev[NSLOT] = {nullptr, nullptr, nullptr};
    bool busy[NSLOT] = {false, false, false};
    size_t slot_bytes = 0;
    int next = 0;
    hipEvent_t last = nullptr;  // recorded behind the newest DMA
    std::atomic<bool> in_flight{false};
    int n_threads = 1;
    std::mutex mtx;  // set_tensor may come from several host threads (RPC connections share the backend: SURVEY.md §8b "Threading")
    uint64_t bytes = 0;
    double seconds = 0;
};
static uploader g_uploaders[GGML_MI355X_MAX_DEVICES];
// OFF by default.  Measured on the MI355X box (profiles/r03_upload_bench.jsonl, 6 GiB in 256 MiB tensors from pageable memory): plain hipMemcpy
// 56.5 GB/s, this engine 55.5 (4 or 8 copy threads; 33 with one) — ROCm 7.2's own pageable path already runs at the PCIe Gen5 rate, a
// 42.5 GB model is 0.8 s of copying either way.  What the engine still buys is the early return (the loader's next read overlaps the DMA);
// GGML_MI355X_STAGED_UPLOAD=1 / set_option("staged_upload", 1) turns it on.
static std::atomic<int> g_staged_upload{[] { const char * e = getenv("GGML_MI355X_STAGED_UPLOAD"); return e ? atoi(e) : 0; }()};

static void par_memcpy(char * dst, const char * src, size_t n, int n_threads) {
    if (n_threads <= 1 || n < ((size_t) 4 << 20)) { memcpy(dst, src, n); return; }
    std::vector<std::thread> th;
    const size_t per = ((n / (size_t) n_threads) + 4095) & ~(size_t) 4095;
    for (int t = 1; t < n_threads; ++t) {
        const size_t o = (size_t) t * per;
        if (o >= n) break;
        th.emplace_back([=] { memcpy(dst + o, src + o, std::min(per, n - o)); });
    }
    memcpy(dst, src, std::min(per, n));
    for (auto & x : th) x.join();
}
static uploader * uploader_for(int device) {
    if (device < 0 || device >= GGML_MI355X_MAX_DEVICES) return nullptr;
    uploader * u = &g_uploaders[device];
    if (u->failed) return nullptr;
    if (u->device == device) return u;
    if (!g_staged_upload.load(std::memory_order_relaxed)) return nullptr;
    std::lock_guard<std::mutex> init_lock(u->mtx);
    if (u->failed) return nullptr;
    if (u->device == device) return u;  // another thread initialised it while this one waited
    const char * e_mb = getenv("GGML_MI355X_UPLOAD_SLOT_MIB");
    const char * e_th = getenv("GGML_MI355X_UPLOAD_THREADS");
    u->slot_bytes = (size_t) std::max(1, e_mb ? atoi(e_mb) : 32) << 20;
    u->n_threads = std::max(1, std::min(16, e_th ? atoi(e_th) : 4));
    bool ok = hipStreamCreateWithFlags(&u->stream, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&u->last, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < uploader::NSLOT && ok; ++i)
        ok = hipHostMalloc((void **) &u->slot[i], u->slot_bytes, hipHostMallocDefault) == hipSuccess && hipEventCreateWithFlags(&u->ev[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) {  // (pinned memory is a limited resource: the loader keeps working on the synchronous path)
        (void) hipGetLastError();
        MI_INFO("staged uploads unavailable on device %d (pinned allocation failed): set_tensor falls back to hipMemcpy", device);
        for (int i = 0; i < uploader::NSLOT; ++i) {
            if (u->slot[i]) (void) hipHostFree(u->slot[i]);
            if (u->ev[i]) (void) hipEventDestroy(u->ev[i]);
            u->slot[i] = nullptr;
            u->ev[i] = nullptr;
        }
        if (u->last) (void) hipEventDestroy(u->last);
        if (u->stream) (void) hipStreamDestroy(u->stream);
        u->failed = true;
        return nullptr;
    }
    u->device = device;
    return u;
}
// host-side wait for every staged upload to this device (callers hold no lock)
static void uploader_drain(int device) {
    if (device < 0 || device >= GGML_MI355X_MAX_DEVICES) return;
    uploader * u = &g_uploaders[device];
    if (u->device != device) return;
    std::lock_guard<std::mutex> lock(u->mtx);
    if (!u->in_flight) return;
    HIP_SOFT(hipStreamSynchronize(u->stream));
    for (bool & b : u->busy) b = false;
    u->in_flight = false;
}
// device-side: `s` (a backend's stream on this device) waits for the newest staged upload
static void uploader_join(int device, hipStream_t s) {
    if (device < 0 || device >= GGML_MI355X_MAX_DEVICES) return;
    uploader * u = &g_uploaders[device];
    if (u->device != device || !u->in_flight) return;
    std::lock_guard<std::mutex> lock(u->mtx);
    if (!u->in_flight) return;
    if (hipEventQuery(u->last) == hipSuccess) {  // long done (every decode step passes here): nothing to wait for any more
        for (bool & b : u->busy) b = false;
        u->in_flight = false;
        return;
    }
    (void) hipGetLastError();  // (hipErrorNotReady)
    HIP_SOFT(hipStreamWaitEvent(s, u->last, 0));
}
static bool staged_upload(int device, char * dst, const char * src, size_t size) {
    uploader * u = uploader_for(device);
    if (!u) return false;
    std::lock_guard<std::mutex> lock(u->mtx);
    const auto t0 = std::chrono::steady_clock::now();
    for (size_t o = 0; o < size; o += u->slot_bytes) {
        const size_t n = std::min(u->slot_bytes, size - o);
        const int k = u->next;
        u->next = (k + 1) % uploader::NSLOT;
        if (u->busy[k] && hipEventSynchronize(u->ev[k]) != hipSuccess) return false;
        par_memcpy(u->slot[k], src + o, n, u->n_threads);
        if (hipMemcpyAsync(dst + o, u->slot[k], n, hipMemcpyHostToDevice, u->stream) != hipSuccess || hipEventRecord(u->ev[k], u->stream) != hipSuccess) {
            (void) hipGetLastError();
            (void) hipStreamSynchronize(u->stream);
            u->failed = true;  // the caller repeats the whole tensor on the synchronous path
            return false;
        }
        u->busy[k] = true;
        u->in_flight = true;
    }
    if (hipEventRecord(u->last, u->stream) != hipSuccess) { (void) hipGetLastError(); (void) hipStreamSynchronize(u->stream); }
    u->bytes += size;
    u->seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return true;
}

static std::atomic<int> g_hip_failed{0};
void note_hip_failure() { g_hip_failed.store(1, std::memory_order_relaxed); }
bool hip_failed() { return g_hip_failed.load(std::memory_order_relaxed) != 0; }
void clear_hip_failure() { g_hip_failed.store(0, std::memory_order_relaxed); }

// ------------------------------------------------------------------------------------------------ mask statistics (common.h)
namespace {
struct mask_entry { const void * dev = nullptr; mask_stats st; uint64_t stamp = 0; };
std::mutex g_mask_mtx;
mask_entry g_masks[8];
std::atomic<uint64_t> g_mask_clock{0};
}  // namespace
void note_mask_upload(const ggml_tensor * t, const void * host, size_t offset, size_t size) {
    // candidates: a whole 2-D f16 / f32 tensor of up to 256 rows whose name says mask ("KQ_mask" in llama.cpp's graphs).  Larger ones are prompt
    // micro-batches: dense by construction, and scanning 2 MB per micro-batch would cost more than the decision is worth
    // anything else written to a tensor on record makes the record stale (a partial update, another tensor recycling the address)
    auto forget = [&]() {
        std::lock_guard<std::mutex> lock(g_mask_mtx);
        for (mask_entry & e : g_masks)
            if (e.dev == t->data) { e.dev = nullptr; e.stamp = 0; }
    };
    bool named = false;
    for (const char * p = t->name; *p && !named; ++p) named = (p[0] == 'm' || p[0] == 'M') && (p[1] == 'a' || p[1] == 'A') && (p[2] == 's' || p[2] == 'S') && (p[3] == 'k' || p[3] == 'K');
    // whole rows from the first one on: the tensor, or its first n_tokens rows (the rows behind them are GGML_KQ_MASK_PAD padding that no kernel reads:
    // a host that re-uses its graph uploads only the live rows every step — llama_lite.cpp does)
    const bool whole = offset == 0 && t->ne[2] == 1 && t->ne[3] == 1 && (t->type == GGML_TYPE_F16 || t->type == GGML_TYPE_F32) && t->nb[1] > 0 && size % (size_t) t->nb[1] == 0 &&
                       size <= (size_t) t->nb[1] * (size_t) t->ne[1] && t->nb[0] == (t->type == GGML_TYPE_F16 ? 2u : 4u);
    // (16 MiB: a 160-token draft batch — 192 rows with padding — over a 40 k-cell cache; the 2 MiB of round 4's first cut lost the 10 k-cell cache of the
    // bench's own `--np 32 --draft 4` line: 10.3 -> 12.1 ms per step on the dense kernel.  A sparse mask is read in full, ~0.1 ms per MiB on one host
    // core; a dense one is given up as soon as a quarter of its cells have been seen visible)
    if (!named || !whole || t->ne[1] < 2 || t->ne[1] > 256 || size > ((size_t) 16 << 20)) { forget(); return; }
    const int64_t n = t->ne[0], rows = (int64_t) (size / (size_t) t->nb[1]);
    if (rows < 2) { forget(); return; }
    mask_stats st;
    int64_t total = 0;
    const int64_t dense_at = rows * n / 4 + 1;
    for (int64_t r = 0; r < rows; ++r) {
        if (total >= dense_at) {  // dense whatever follows: density > 0.25 is all its readers ask (the longest row so far stands in for the rest)
            st.rows = (int) rows;
            st.max_visible = std::max(st.max_visible, (int) n);
            total = rows * n;
            break;
        }
        int64_t vis = 0;
        if (t->type == GGML_TYPE_F16) {
            const uint16_t * m = (const uint16_t *) ((const char *) host + r * t->nb[1]);
            for (int64_t i = 0; i < n; ++i) vis += m[i] != 0xFC00u;  // (-inf; anything else — 0 or an ALiBi slope — is a visible cell)
        } else {
            const uint32_t * m = (const uint32_t *) ((const char *) host + r * t->nb[1]);
            for (int64_t i = 0; i < n; ++i) vis += m[i] != 0xFF800000u;
        }
        total += vis;
        st.rows += vis > 0;
        st.max_visible = std::max(st.max_visible, (int) vis);
    }
    st.density = st.rows > 0 ? (float) ((double) total / ((double) st.rows * (double) n)) : 0.0f;
    std::lock_guard<std::mutex> lock(g_mask_mtx);
    mask_entry * slot = &g_masks[0];
    for (mask_entry & e : g_masks) {
        if (e.dev == t->data) { slot = &e; break; }
        if (e.stamp < slot->stamp) slot = &e;
    }
    slot->dev = t->data;
    slot->st = st;
    slot->stamp = ++g_mask_clock;
}
// a device-side write (a graph node's output, a tensor copy) into memory on record makes the record stale: the statistics describe bytes that
// came through set_tensor, nothing else
void forget_mask_stats(const void * dev_ptr, size_t nbytes) {
    if (g_mask_clock.load(std::memory_order_relaxed) == 0 || !dev_ptr) return;  // (nothing was ever recorded: the common case costs one load)
    std::lock_guard<std::mutex> lock(g_mask_mtx);
    for (mask_entry & e : g_masks)
        if (e.stamp != 0 && (const char *) e.dev >= (const char *) dev_ptr && (const char *) e.dev < (const char *) dev_ptr + nbytes) { e.dev = nullptr; e.stamp = 0; }
}
// a WHOLE tensor copied between two of this library's buffers (ggml_backend_sched / -sm layer hand the graph's inputs — the mask among them — from device to
// device through cpy_tensor): the statistics describe the same bytes at their new address, so the second device takes the kernel decisions the first one took
// (without them a layer-split run chose the dense kernel where the one-device run walked position lists: same values to 1e-5, not the same bits)
void copy_mask_stats(const ggml_tensor * src, const ggml_tensor * dst, size_t nbytes) {
    if (g_mask_clock.load(std::memory_order_relaxed) == 0 || !src->data || !dst->data) return;
    std::lock_guard<std::mutex> lock(g_mask_mtx);
    const mask_entry * from = nullptr;
    for (const mask_entry & e : g_masks)
        if (e.stamp != 0 && e.dev == src->data) from = &e;
    for (mask_entry & e : g_masks)  // whatever was on record inside the destination range is stale now
        if (e.stamp != 0 && (const char *) e.dev >= (const char *) dst->data && (const char *) e.dev < (const char *) dst->data + nbytes) { e.dev = nullptr; e.stamp = 0; }
    if (!from || src->view_src || dst->view_src || src->type != dst->type || src->ne[0] != dst->ne[0] || src->ne[1] != dst->ne[1] || src->ne[2] != dst->ne[2] || src->ne[3] != dst->ne[3] ||
        src->nb[1] != dst->nb[1])
        return;
    const mask_stats st = from->st;
    mask_entry * slot = &g_masks[0];
    for (mask_entry & e : g_masks) {
        if (e.dev == dst->data) { slot = &e; break; }
        if (e.stamp < slot->stamp) slot = &e;
    }
    if (slot == from) return;  // (eight slots, the source is the oldest: leave it)
    slot->dev = dst->data;
    slot->st = st;
    slot->stamp = ++g_mask_clock;
}
bool lookup_mask_stats(const void * dev_ptr, mask_stats * out) {
    std::lock_guard<std::mutex> lock(g_mask_mtx);
    for (const mask_entry & e : g_masks)
        if (e.dev == dev_ptr && e.stamp != 0) { *out = e.st; return true; }
    return false;
}
int mask_sparse_hint(const void * dev_ptr) {
    mask_stats st;
    return dev_ptr != nullptr && lookup_mask_stats(dev_ptr, &st) && st.rows > 0 && st.density <= 0.25f ? 1 : 0;
}

static void buf_set_tensor(ggml_backend_buffer_t b, ggml_tensor * t, const void * data, size_t offset, size_t size) {
    buffer_ctx * c = (buffer_ctx *) b->context;
    if (ip_any()) ip_host_access(b, true);  // (a cache buffer whose heads live sharded on several devices: tp_inproc.cpp)
    HIP_SOFT(hipSetDevice(c->device));
    note_mask_upload(t, data, offset, size);
    decode_copy_drop(b, t, offset, size);
    if (size > ((size_t) 1 << 20) && staged_upload(c->device, (char *) t->data + offset, (const char *) data, size)) return;
    uploader_drain(c->device);  // (keeps the writes of one tensor ordered: a small piece behind a staged one)
    HIP_SOFT(hipMemcpy((char *) t->data + offset, data, size, hipMemcpyHostToDevice));
}
static void buf_get_tensor(ggml_backend_buffer_t b, const ggml_tensor * t, void * data, size_t offset, size_t size) {
    buffer_ctx * c = (buffer_ctx *) b->context;
    if (ip_any()) ip_host_access(b, false);
    HIP_SOFT(hipSetDevice(c->device));
    uploader_drain(c->device);
    HIP_SOFT(hipMemcpy(data, (const char *) t->data + offset, size, hipMemcpyDeviceToHost));
}
static bool buf_cpy_tensor(ggml_backend_buffer_t b, const ggml_tensor * src, ggml_tensor * dst) {
    ggml_backend_buffer_t sb = src->view_src ? src->view_src->buffer : src->buffer;
    if (!buffer_is_ours(sb)) return false;
    buffer_ctx * sc = (buffer_ctx *) sb->context;
    buffer_ctx * dc = (buffer_ctx *) b->context;
    const size_t n = ggml_abi_nbytes(src);
    copy_mask_stats(src, dst, n);
    decode_copy_drop(b, dst, 0, n);
    if (ip_any()) { ip_host_access(sb, false); ip_host_access(b, true); }
    uploader_drain(sc->device);
    if (dc->device != sc->device) uploader_drain(dc->device);
    if (sc->device == dc->device) {
        HIP_SOFT(hipSetDevice(dc->device));
        HIP_SOFT(hipMemcpy(dst->data, src->data, n, hipMemcpyDeviceToDevice));
    } else {
        HIP_SOFT(hipMemcpyPeer(dst->data, dc->device, src->data, sc->device, n));
    }
    HIP_SOFT(hipDeviceSynchronize());
    return true;
}
static void buf_clear(ggml_backend_buffer_t b, uint8_t value) {
    buffer_ctx * c = (buffer_ctx *) b->context;
    if (ip_any()) ip_host_access(b, true);
    HIP_SOFT(hipSetDevice(c->device));
    decode_copy_drop(c, 0, c->size);
    uploader_drain(c->device);
    HIP_SOFT(hipMemset(c->base, value, c->size));
    HIP_SOFT(hipDeviceSynchronize());
}
static const ggml_backend_buffer_i k_buffer_iface = {buf_free, buf_get_base, buf_init_tensor, buf_memset_tensor, buf_set_tensor,
                                                     buf_get_tensor, buf_cpy_tensor, buf_clear, nullptr};

bool buffer_is_ours(ggml_backend_buffer_t b) { return b != nullptr && b->iface.free_buffer == buf_free; }
// a buffer object that only TAGS tensors the backend creates itself (tp_inproc.cpp: the per-device shards of a rewritten graph): it owns no memory
void make_internal_buffer(ggml_backend_buffer * out, buffer_ctx * bc, int device_ordinal, bool rowpar) {
    bc->device = device_ordinal;
    bc->base = nullptr;
    bc->size = 0;
    bc->rowpar = rowpar;
    *out = ggml_backend_buffer{k_buffer_iface, nullptr, bc, 0, GGML_BACKEND_BUFFER_USAGE_COMPUTE};
}
bool buffer_is_rowpar(ggml_backend_buffer_t b) { return buffer_is_ours(b) && ((buffer_ctx *) b->context)->rowpar; }

// ---- device buffer type
struct buft_ctx {
    int device;
    bool rowpar;
    std::string name;
};
static const char * buft_get_name(ggml_backend_buffer_type_t buft) { return ((buft_ctx *) buft->context)->name.c_str(); }
static ggml_backend_buffer_t buft_alloc(ggml_backend_buffer_type_t buft, size_t size) {
    buft_ctx * bc = (buft_ctx *) buft->context;
    HIP_SOFT(hipSetDevice(bc->device));
    void * p = nullptr;
    // (+256: the skinny Q6_K fetch reads whole 256-byte windows of 210-byte blocks — up to 46 bytes past the last block of a tensor)
    hipError_t err = hipMalloc(&p, size + 256);
    if (err != hipSuccess) {
        (void) hipGetLastError();
        MI_ERR("allocating %.2f MiB on device %d: hipMalloc failed: %s", size / 1024.0 / 1024.0, bc->device, hipGetErrorString(err));
        return nullptr;  // handled by hosts: llama-box/rpcserver.hpp:1068-1080
    }
    buffer_ctx * c = new buffer_ctx{bc->device, p, size, bc->rowpar};
    return make_buffer(buft, k_buffer_iface, c, size);
}
static size_t buft_alignment(ggml_backend_buffer_type_t) { return 256; }
static size_t buft_max_size(ggml_backend_buffer_type_t buft) {
    buft_ctx * bc = (buft_ctx *) buft->context;
    hipDeviceProp_t prop{};
    if (hipGetDeviceProperties(&prop, bc->device) != hipSuccess) {
        (void) hipGetLastError();
        return SIZE_MAX;  // (what ggml assumes for a buffer type without get_max_size)
    }
    return prop.totalGlobalMem;
}
static size_t buft_alloc_size(ggml_backend_buffer_type_t, const ggml_tensor * t) {
    // kernels never read past the last block of a row, so no row padding is needed; keep 16-byte granularity
    return (ggml_abi_nbytes(t) + 15) / 16 * 16;
}
static bool buft_is_host(ggml_backend_buffer_type_t) { return false; }
static const ggml_backend_buffer_type_i k_buft_iface = {buft_get_name, buft_alloc, buft_alignment, buft_max_size, buft_alloc_size, buft_is_host};

// ---- pinned host buffer type (uploads at PCIe rate; llama.cpp asks for it via get_host_buffer_type)
// The ranges handed out are remembered: memory of this type is mapped into the device's address space (hipHostMalloc), so a kernel can write
// it — get_tensor_async uses that for the logits rows of a decode step (be_get_tensor_async).
static std::mutex g_pinned_mtx;
static std::vector<std::pair<const char *, size_t>> g_pinned;
static bool is_pinned_range(const void * p, size_t n) {
    std::lock_guard<std::mutex> lock(g_pinned_mtx);
    for (const auto & r : g_pinned)
        if ((const char *) p >= r.first && (const char *) p + n <= r.first + r.second) return true;
    return false;
}
static void hbuf_free(ggml_backend_buffer_t b) {
    {
        std::lock_guard<std::mutex> lock(g_pinned_mtx);
        for (size_t i = 0; i < g_pinned.size(); ++i)
            if (g_pinned[i].first == (const char *) b->context) { g_pinned.erase(g_pinned.begin() + (long) i); break; }
    }
    HIP_NOTE(hipHostFree(b->context));
}
static void * hbuf_base(ggml_backend_buffer_t b) { return b->context; }
static void hbuf_memset(ggml_backend_buffer_t, ggml_tensor * t, uint8_t v, size_t off, size_t sz) { memset((char *) t->data + off, v, sz); }
static void hbuf_set(ggml_backend_buffer_t, ggml_tensor * t, const void * d, size_t off, size_t sz) { memcpy((char *) t->data + off, d, sz); }
static void hbuf_get(ggml_backend_buffer_t, const ggml_tensor * t, void * d, size_t off, size_t sz) { memcpy(d, (const char *) t->data + off, sz); }
static void hbuf_clear(ggml_backend_buffer_t b, uint8_t v) { memset(b->context, v, b->size); }
static const char * hbuft_name(ggml_backend_buffer_type_t) { return GGML_MI355X_NAME "_Host"; }
static ggml_backend_buffer_t hbuft_alloc(ggml_backend_buffer_type_t buft, size_t size) {
    void * p = nullptr;
    hipError_t err = hipHostMalloc(&p, size > 0 ? size : 1, hipHostMallocPortable | hipHostMallocMapped);  // (every device of the process may read and write it)
    if (err != hipSuccess) {
        (void) hipGetLastError();
        return nullptr;
    }
    ggml_backend_buffer_i iface = {hbuf_free, hbuf_base, nullptr, hbuf_memset, hbuf_set, hbuf_get, nullptr, hbuf_clear, nullptr};
    {
        std::lock_guard<std::mutex> lock(g_pinned_mtx);
        g_pinned.emplace_back((const char *) p, size > 0 ? size : 1);
    }
    return make_buffer(buft, iface, p, size);
}
static size_t hbuft_alignment(ggml_backend_buffer_type_t) { return 64; }
static bool hbuft_is_host(ggml_backend_buffer_type_t) { return true; }

// ------------------------------------------------------------------------------------------------ backend
static const char * be_get_name(ggml_backend_t be) { return ((backend_ctx *) be->context)->name.c_str(); }
static void be_free(ggml_backend_t be) {
    backend_ctx * c = (backend_ctx *) be->context;
    HIP_SOFT(hipSetDevice(c->device));
    HIP_SOFT(hipStreamSynchronize(c->stream));
    free_graph_cache(c);
    ip_free(c);
    tp_free(c);
    free_split_helpers(c);
    if (c->ws) HIP_NOTE(hipFree(c->ws));
    if (c->up_ring) HIP_NOTE(hipHostFree(c->up_ring));
    if (c->fa_lists) HIP_NOTE(hipFree(c->fa_lists));
    if (c->rope_tab) HIP_NOTE(hipFree(c->rope_tab));
    if (c->fa_arrive) HIP_NOTE(hipFree(c->fa_arrive));
    if (c->ss_buf) HIP_NOTE(hipFree(c->ss_buf));
    HIP_SOFT(hipStreamDestroy(c->stream));
    delete c;
    delete be;
}
// small uploads wait in the pinned ring until something else is about to enter the stream (or the host asks to synchronise):
// every entry point that touches the stream calls this first
void flush_uploads(backend_ctx * c) {
    if (c->up_pending.n == 0) return;
    launch_upload_multi(c->stream, c->up_pending);
    c->up_pending.n = 0;
}
static void be_set_tensor_async(ggml_backend_t be, ggml_tensor * t, const void * data, size_t offset, size_t size) {
    backend_ctx * c = (backend_ctx *) be->context;
    if (ip_any()) ip_host_access(t->view_src ? t->view_src->buffer : t->buffer, true);
    HIP_SOFT(hipSetDevice(c->device));
    note_mask_upload(t, data, offset, size);
    {
        ggml_backend_buffer_t tb = t->view_src ? t->view_src->buffer : t->buffer;
        if (tb && buffer_is_ours(tb) && ((buffer_ctx *) tb->context)->shadow) decode_copy_drop(tb, t, offset, size);
    }
    constexpr size_t SMALL = 64 * 1024, RING = 4u << 20;
    if (size > 0 && size <= SMALL && c->opt.small_uploads) {
        if (!c->up_ring) {
            if (hipHostMalloc((void **) &c->up_ring, RING, hipHostMallocDefault) == hipSuccess) c->up_cap = RING;
            else { (void) hipGetLastError(); c->up_ring = nullptr; c->opt.small_uploads = false; }
        }
        if (c->up_ring) {
            size_t at = (c->up_head + 255) & ~(size_t) 255;
            if (at + size > c->up_cap) {  // wrap: everything staged so far must have been consumed
                flush_uploads(c);
                HIP_SOFT(hipStreamSynchronize(c->stream));
                at = 0;
            }
            memcpy(c->up_ring + at, data, size);
            c->up_head = at + size;
            if (c->up_pending.n == 8) flush_uploads(c);
            c->up_pending.seg[c->up_pending.n++] = {(char *) t->data + offset, c->up_ring + at, size};
            return;
        }
    }
    flush_uploads(c);
    uploader_join(c->device, c->stream);
    HIP_SOFT(hipMemcpyAsync((char *) t->data + offset, data, size, hipMemcpyHostToDevice, c->stream));
}
static void be_get_tensor_async(ggml_backend_t be, const ggml_tensor * t, void * data, size_t offset, size_t size) {
    backend_ctx * c = (backend_ctx *) be->context;
    if (ip_any()) ip_host_access(t->view_src ? t->view_src->buffer : t->buffer, false);
    HIP_SOFT(hipSetDevice(c->device));
    flush_uploads(c);
    uploader_join(c->device, c->stream);
    // The logits of a decode step (513 KB per sequence, llama.cpp's pinned output buffer): a blit through hipMemcpyAsync starts ~20 us after
    // the graph's last kernel and takes 10 us (profiles/r05_decode_gaps_*.txt); a copy kernel writing the mapped pinned memory is an ordinary
    // launch behind that kernel.  Only into memory of OUR host buffer type (known to be device-mapped), up to 8 MiB.
    if (c->opt.small_downloads && size > 0 && size <= ((size_t) 8 << 20) && is_pinned_range(data, size)) {
        launch_upload_small(c->stream, data, (const char *) t->data + offset, size);
        c->st.kernel_downloads++;
        return;
    }
    HIP_SOFT(hipMemcpyAsync(data, (const char *) t->data + offset, size, hipMemcpyDeviceToHost, c->stream));
}
static bool be_is_ours(ggml_backend_t be);
static bool be_cpy_tensor_async(ggml_backend_t be_src, ggml_backend_t be_dst, const ggml_tensor * src, ggml_tensor * dst) {
    if (!be_is_ours(be_src) || !be_is_ours(be_dst)) return false;
    ggml_backend_buffer_t sb = src->view_src ? src->view_src->buffer : src->buffer;
    ggml_backend_buffer_t db = dst->view_src ? dst->view_src->buffer : dst->buffer;
    if (!buffer_is_ours(sb) || !buffer_is_ours(db)) return false;
    backend_ctx * cs = (backend_ctx *) be_src->context;
    backend_ctx * cd = (backend_ctx *) be_dst->context;
    flush_uploads(cs);
    if (cd != cs) flush_uploads(cd);
    const size_t n = ggml_abi_nbytes(src);
    copy_mask_stats(src, dst, n);
    if (((buffer_ctx *) db->context)->shadow) decode_copy_drop(db, dst, 0, n);
    if (ip_any()) { ip_host_access(sb, false); ip_host_access(db, true); }
    uploader_join(cs->device, cs->stream);
    if (cd->device != cs->device) uploader_drain(cd->device);
    if (cs->device == cd->device) {
        HIP_SOFT(hipSetDevice(cs->device));
        HIP_SOFT(hipMemcpyAsync(dst->data, src->data, n, hipMemcpyDeviceToDevice, cs->stream));
    } else {
        HIP_SOFT(hipSetDevice(cs->device));
        HIP_SOFT(hipMemcpyPeerAsync(dst->data, cd->device, src->data, cs->device, n, cs->stream));
    }
    if (be_src != be_dst) {  // make the destination stream wait for the copy
        hipEvent_t ev;
        HIP_NOTE(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        HIP_SOFT(hipEventRecord(ev, cs->stream));
        HIP_SOFT(hipSetDevice(cd->device));
        HIP_SOFT(hipStreamWaitEvent(cd->stream, ev, 0));
        HIP_NOTE(hipEventDestroy(ev));
    }
    return true;
}
static void be_synchronize(ggml_backend_t be) {
    backend_ctx * c = (backend_ctx *) be->context;
    HIP_SOFT(hipSetDevice(c->device));
    flush_uploads(c);
    uploader_drain(c->device);
    HIP_SOFT(hipStreamSynchronize(c->stream));
}
static enum ggml_status be_graph_compute(ggml_backend_t be, ggml_cgraph * g) {
    backend_ctx * c = (backend_ctx *) be->context;
    if (hipSetDevice(c->device) != hipSuccess) return GGML_STATUS_FAILED;
    flush_uploads(c);
    uploader_join(c->device, c->stream);  // weights staged by set_tensor are complete before the first kernel reads them
    return graph_compute(c, g);
}
static void be_event_record(ggml_backend_t be, ggml_backend_event_t ev) {
    backend_ctx * c = (backend_ctx *) be->context;
    flush_uploads(c);
    HIP_SOFT(hipEventRecord((hipEvent_t) ev->context, c->stream));
}
static void be_event_wait(ggml_backend_t be, ggml_backend_event_t ev) {
    backend_ctx * c = (backend_ctx *) be->context;
    flush_uploads(c);
    HIP_SOFT(hipStreamWaitEvent(c->stream, (hipEvent_t) ev->context, 0));
}
static const ggml_backend_i k_backend_iface = {
    be_get_name, be_free, be_set_tensor_async, be_get_tensor_async, be_cpy_tensor_async, be_synchronize,
    /* graph_plan_create */ nullptr, /* graph_plan_free */ nullptr, /* graph_plan_update */ nullptr, /* graph_plan_compute */ nullptr,
    be_graph_compute, be_event_record, be_event_wait,
#if GGML_ABI_HAS_GRAPH_OPTIMIZE
    nullptr,
#endif
};
static bool be_is_ours(ggml_backend_t be) { return be != nullptr && be->iface.get_name == be_get_name; }
static ggml_backend_t dev_init_backend(ggml_backend_dev_t dev, const char *);
ggml_backend_t internal_backend(int logical) {
    if (logical < 0 || logical >= logical_device_count()) return nullptr;
    return dev_init_backend(logical_device(logical), nullptr);
}

// ------------------------------------------------------------------------------------------------ device iface
static const char * dev_get_name(ggml_backend_dev_t dev) { return dctx(dev)->name.c_str(); }
static const char * dev_get_description(ggml_backend_dev_t dev) { return dctx(dev)->description.c_str(); }
static void dev_get_memory(ggml_backend_dev_t dev, size_t * free, size_t * total) {
    *free = 0;  // (what a failed query reports: llama-box's /v1/models and --list-devices print these)
    *total = 0;
    HIP_NOTE(hipSetDevice(dctx(dev)->device));
    HIP_NOTE(hipMemGetInfo(free, total));
}
static enum ggml_backend_dev_type dev_get_type(ggml_backend_dev_t) { return GGML_BACKEND_DEVICE_TYPE_GPU; }
static void dev_get_props(ggml_backend_dev_t dev, ggml_backend_dev_props * props) {
    props->name = dev_get_name(dev);
    props->description = dev_get_description(dev);
    props->type = GGML_BACKEND_DEVICE_TYPE_GPU;
    dev_get_memory(dev, &props->memory_free, &props->memory_total);
    props->caps = {/* async */ true, /* host_buffer */ true, /* buffer_from_host_ptr */ false, /* events */ true};
}
static ggml_backend_t dev_init_backend(ggml_backend_dev_t dev, const char *) {
    device_ctx * d = dctx(dev);
    // (init_backend HAS a way to say no: NULL — ggml_backend_dev_init's callers check it)
    if (hipSetDevice(d->device) != hipSuccess) { (void) hipGetLastError(); MI_ERR("init_backend: device %d cannot be made current", d->device); return nullptr; }
    backend_ctx * c = new backend_ctx();
    c->device = d->device;
    c->name = d->name;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        (void) hipGetLastError();
        MI_ERR("init_backend: no stream on device %d", d->device);
        delete c;
        return nullptr;
    }
    {   // every kernel file's code object onto this device now, not in front of the first prompt (kernels.h: MI_TU_TOUCH)
        static const bool preload = !getenv("GGML_MI355X_PRELOAD") || atoi(getenv("GGML_MI355X_PRELOAD")) != 0;
        static std::mutex pre_mtx;
        static bool preloaded[64] = {false};
        std::lock_guard<std::mutex> lk(pre_mtx);
        if (preload && d->device >= 0 && d->device < 64 && !preloaded[d->device]) {
            preload_kernel_files(c->stream);
            if (hipStreamSynchronize(c->stream) != hipSuccess) { (void) hipGetLastError(); MI_ERR("init_backend: kernel preload failed on device %d", d->device); }
            preloaded[d->device] = true;
        }
    }
    c->fa_lists_bytes = (size_t) 8 << 20;  // visible-position lists of a small batch: (n_kv + 1) ints per query token
    if (hipMalloc((void **) &c->fa_lists, c->fa_lists_bytes) != hipSuccess) { (void) hipGetLastError(); c->fa_lists = nullptr; c->fa_lists_bytes = 0; }
    if (hipMalloc((void **) &c->rope_tab, backend_ctx::rope_tab_floats * sizeof(float)) != hipSuccess) { (void) hipGetLastError(); c->rope_tab = nullptr; }
    if (hipMalloc((void **) &c->fa_arrive, backend_ctx::fa_arrive_slots * sizeof(unsigned)) != hipSuccess || hipMemset(c->fa_arrive, 0, backend_ctx::fa_arrive_slots * sizeof(unsigned)) != hipSuccess) {
        (void) hipGetLastError();
        c->fa_arrive = nullptr;
    }
    if (const char * e = getenv("GGML_MI355X_GRAPHS")) c->opt.graphs = atoi(e) != 0;
    if (const char * e = getenv("GGML_MI355X_FUSION")) c->opt.fusion = atoi(e) != 0;
    if (const char * e = getenv("GGML_MI355X_PROLOGUE")) c->opt.prologue = atoi(e) != 0;
    if (const char * e = getenv("GGML_MI355X_QKV")) c->opt.qkv = atoi(e) != 0;
    if (const char * e = getenv("GGML_MI355X_MMQ_MIN_COLS")) c->opt.mmq_min_cols = atoi(e);
    if (const char * e = getenv("GGML_MI355X_MMVQ_MAX_COLS")) c->opt.mmvq_max_cols = atoi(e);
    if (const char * e = getenv("GGML_MI355X_FA_SPLITS")) c->opt.fa_splits = atoi(e);
    if (const char * e = getenv("GGML_MI355X_FA_WO")) c->opt.fa_wo = atoi(e) != 0;
    if (const char * e = getenv("GGML_MI355X_SMALL_UPLOADS")) c->opt.small_uploads = atoi(e) != 0;
    if (const char * e = getenv("GGML_MI355X_SMALL_DOWNLOADS")) c->opt.small_downloads = atoi(e) != 0;
    if (const char * e = getenv("GGML_MI355X_MMQ_I8")) c->opt.mmq_i8 = atoi(e) != 0;
    if (const char * e = getenv("GGML_MI355X_MM_MERGE")) c->opt.mm_merge = atoi(e) != 0;
    if (const char * e = getenv("GGML_MI355X_MMQ_BN")) c->opt.mmq_bn = atoi(e);
    if (const char * e = getenv("GGML_MI355X_MMQ_SKINNY")) c->opt.mmq_skinny = atoi(e) != 0;
    if (const char * e = getenv("GGML_MI355X_SKINNY_ROPE")) c->opt.skinny_rope = atoi(e) != 0;
    if (const char * e = getenv("GGML_MI355X_SKINNY_MIX")) c->opt.skinny_mix = atoi(e) != 0;
    if (const char * e = getenv("GGML_MI355X_SOFTMAX_MM")) c->opt.softmax_mm = atoi(e) != 0;
    if (const char * e = getenv("GGML_MI355X_ATTN_NF")) c->opt.attn_nf = atoi(e) != 0;
    if (const char * e = getenv("GGML_MI355X_FA_SELF_MERGE")) c->opt.fa_self_merge = atoi(e) != 0;
    if (const char * e = getenv("GGML_MI355X_SS_PARTIALS")) c->opt.ss_partials = atoi(e) != 0;
    if (hipMalloc((void **) &c->ss_buf, 256 * sizeof(double)) != hipSuccess) {  // (optional: without it every norm prologue sums its own row)
        (void) hipGetLastError();
        c->ss_buf = nullptr;
    }
    return new ggml_backend{&g_guid, k_backend_iface, dev, c};
}
static ggml_backend_buffer_type_t dev_get_buffer_type(ggml_backend_dev_t dev) { return &dctx(dev)->buft; }
static ggml_backend_buffer_type_t dev_get_host_buffer_type(ggml_backend_dev_t dev) { return &dctx(dev)->buft_host; }
static bool dev_supports_op(ggml_backend_dev_t, const ggml_tensor * op) { return supports_op(op); }
static bool dev_supports_buft(ggml_backend_dev_t dev, ggml_backend_buffer_type_t buft) {
    if (buft_is_split(buft)) return logical_device(split_buft_main_device(buft)) == dev;  // row-split weights: the MAIN device's backend computes on them
    if (buft->iface.get_name != buft_get_name) return false;
    return ((buft_ctx *) buft->context)->device == dctx(dev)->device;
}
static bool dev_offload_op(ggml_backend_dev_t, const ggml_tensor * op) {
    // same policy as the stock GPU backends: weights left on the host are worth uploading for batches >= 32
    const int64_t batch = op->op == GGML_OP_MUL_MAT ? op->ne[1] : 0;
    return batch >= 32;
}
static ggml_backend_event_t dev_event_new(ggml_backend_dev_t dev) {
    HIP_SOFT(hipSetDevice(dctx(dev)->device));
    hipEvent_t ev;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {  // (event_new may return NULL: the scheduler then runs without events)
        (void) hipGetLastError();
        return nullptr;
    }
    return new ggml_backend_event{dev, ev};
}
static void dev_event_free(ggml_backend_dev_t, ggml_backend_event_t ev) {
    HIP_NOTE(hipEventDestroy((hipEvent_t) ev->context));
    delete ev;
}
static void dev_event_synchronize(ggml_backend_dev_t, ggml_backend_event_t ev) { HIP_SOFT(hipEventSynchronize((hipEvent_t) ev->context)); }
static const ggml_backend_device_i k_device_iface = {
    dev_get_name, dev_get_description, dev_get_memory, dev_get_type, dev_get_props, dev_init_backend, dev_get_buffer_type,
    dev_get_host_buffer_type, /* buffer_from_host_ptr */ nullptr, dev_supports_op, dev_supports_buft, dev_offload_op,
    dev_event_new, dev_event_free, dev_event_synchronize,
};

// ------------------------------------------------------------------------------------------------ proc addresses
static ggml_backend_feature g_features[] = {{"ARCH", "gfx950"}, {"WAVE", "64"}, {"GRAPHS", "1"}, {"RCCL_TP", "1"}, {nullptr, nullptr}};
static ggml_backend_feature * get_features(ggml_backend_reg_t) { return g_features; }

static int api_tp_init(ggml_backend_t be, int rank, int world, const void * uid, size_t n) {
    if (!be_is_ours(be)) return -1;
    return tp_init((backend_ctx *) be->context, rank, world, uid, n);
}
static int api_tp_get_unique_id(void * out, size_t n) { return tp_get_unique_id(out, n); }
static int api_tp_p2p_export(ggml_backend_t be, int rank, int world, void * handle_out, size_t n) {
    if (!be_is_ours(be)) return -1;
    return tp_p2p_export((backend_ctx *) be->context, rank, world, handle_out, n);
}
// in-place sum over the ranks of n f32 values in device memory, in the backend's stream (what graph_compute issues behind a row-parallel mat-mul);
// exported so that a launcher can check its group before it trusts it
static int api_tp_all_reduce(ggml_backend_t be, float * device_ptr, size_t n) {
    if (!be_is_ours(be)) return -1;
    backend_ctx * c = (backend_ctx *) be->context;
    if (!tp_active(c)) return -2;
    HIP_TRY(hipSetDevice(c->device), -3);
    return tp_all_reduce(c, device_ptr, n) ? 0 : -4;
}
static int api_tp_p2p_attach(ggml_backend_t be, const void * handles, size_t n) {
    if (!be_is_ours(be)) return -1;
    return tp_p2p_attach((backend_ctx *) be->context, handles, n);
}
static ggml_backend_buffer_type_t api_split_buffer_type(int main_device, const float * tensor_split) { return split_buffer_type(main_device, tensor_split); }
static void api_split_rows(int64_t nrows, const float * tensor_split, int n_dev, int64_t * row0) { split_rows(nrows, tensor_split, n_dev, nrows % 256 == 0 ? 256 : 64, row0); }  // (the granule sbuf_init_tensor uses)
static ggml_backend_buffer_type_t api_tp_rowpar_buft(int device) {
    if (device < 0 || device >= (int) g_reg_ctx.devices.size()) return nullptr;
    return &dctx(g_reg_ctx.devices[device])->buft_rowpar;
}
// the graph key of `a` compared in place with graph `b` (what decides "replay the captured hipGraph"): 1 equal, 0 different; *n_words = key length.
// Host arithmetic only — tests reach it without a device.
static int api_graph_key_probe(const ggml_cgraph * a, const ggml_cgraph * b, int64_t * n_words) {
    std::vector<uint64_t> key;
    graph_key_build(a, key);
    if (n_words) *n_words = (int64_t) key.size();
    return graph_key_equals(b, key) ? 1 : 0;
}
// tests: the decode copy of weight matrix t (repacking it first if need be) into `out`; returns its bytes, 0 when the tensor has / gets no copy, -1 on error
static int64_t api_decode_copy_read(ggml_backend_t be, const ggml_tensor * t, void * out, size_t size) {
    if (!be_is_ours(be) || t == nullptr) return -1;
    backend_ctx * c = (backend_ctx *) be->context;
    if (hipSetDevice(c->device) != hipSuccess) return -1;
    uploader_join(c->device, c->stream);
    const uint8_t * p = t->type == GGML_TYPE_Q8_0 ? q80_panel_copy(c, t) : decode_copy(c, t);
    if (!p) return 0;
    const size_t bytes = (size_t) t->nb[1] * (size_t) t->ne[1];
    if (out == nullptr || size < bytes) return (int64_t) bytes;
    if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(out, p, bytes, hipMemcpyDeviceToHost) != hipSuccess) { (void) hipGetLastError(); return -1; }
    return (int64_t) bytes;
}
static int api_set_option(ggml_backend_t be, const char * key, const char * value) {
    if (!be_is_ours(be)) return -1;
    backend_ctx * c = (backend_ctx *) be->context;
    const std::string k = key;
    const int v = atoi(value);
    if (k == "graphs") c->opt.graphs = v != 0;
    else if (k == "fusion") c->opt.fusion = v != 0;
    else if (k == "prologue") c->opt.prologue = v != 0;
    else if (k == "qkv") c->opt.qkv = v != 0;
    else if (k == "mmvq_max_cols") c->opt.mmvq_max_cols = v;
    else if (k == "mmq_min_cols") c->opt.mmq_min_cols = v;
    else if (k == "mmq_i8") c->opt.mmq_i8 = v != 0;
    else if (k == "mm_merge") c->opt.mm_merge = v != 0;
    else if (k == "mmq_bn") c->opt.mmq_bn = v;
    else if (k == "mmq_skinny") c->opt.mmq_skinny = v != 0;
    else if (k == "skinny_rope") c->opt.skinny_rope = v != 0;
    else if (k == "skinny_mix") c->opt.skinny_mix = v != 0;
    else if (k == "softmax_mm") c->opt.softmax_mm = v != 0;
    else if (k == "attn_nf") c->opt.attn_nf = v != 0;
    else if (k == "fa_self_merge") c->opt.fa_self_merge = v != 0;
    else if (k == "ss_partials") c->opt.ss_partials = v != 0;
    else if (k == "tp_p2p") { if (!tp_p2p_enable(c, v != 0)) return -2; }
    else if (k == "tp_p2p_reset") { if (!tp_p2p_reset(c)) return -2; }
    else if (k == "fa_splits") c->opt.fa_splits = v;
    else if (k == "fa_wo") c->opt.fa_wo = v != 0;
    else if (k == "small_uploads") c->opt.small_uploads = v != 0;
    else if (k == "small_downloads") c->opt.small_downloads = v != 0;
    else if (k == "timing") c->opt.timing = v != 0;
    else if (k == "exec_update") c->opt.exec_update = v;
    else if (k == "shadow_capture") c->opt.shadow_capture = v;
    else if (k == "q80_min_cols") c->opt.q80_min_cols = v;
    else if (k == "decode_copy") c->opt.decode_copy = v != 0;
    else if (k == "decode_copy_headroom_gib") c->opt.decode_copy_headroom_gib = v;
    else if (k == "clear_failure") { if (v) clear_hip_failure(); }
    else if (k == "staged_upload") g_staged_upload.store(v != 0);
    else return -1;
    HIP_SOFT(hipStreamSynchronize(c->stream));
    free_graph_cache(c);
    return 0;
}
static int64_t api_get_stat(ggml_backend_t be, const char * key) {
    if (!be_is_ours(be)) return -1;
    backend_ctx * c = (backend_ctx *) be->context;
    const std::string k = key;
    if (k == "graph_launches") return c->st.graph_launches;
    if (k == "graph_captures") return c->st.graph_captures;
    if (k == "eager_graphs") return c->st.eager_graphs;
    if (k == "kernel_launches") return c->st.kernel_launches;
    if (k == "fused_nodes") return c->st.fused_nodes;
    if (k == "allreduces") return c->st.allreduces;
    if (k == "ss_handoffs") return c->st.ss_handoffs;
    if (k == "fa_list_launches") return c->st.fa_list_launches;
    if (k == "nf_mma_chains") return c->st.nf_mma_chains;
    if (k == "p2p_allreduces") return c->st.p2p_allreduces;
    if (k == "p2p_timeouts") return tp_p2p_timeouts(c);
    if (k == "step_heads") return c->st.step_heads;
    if (k == "elided_conts") return c->st.elided_conts;
    if (k == "decode_copy_tensors") return c->st.decode_copy_tensors;
    if (k == "decode_copy_bytes") return c->st.decode_copy_bytes;
    if (k == "decode_copy_launches") return c->st.decode_copy_launches;
    if (k == "graph_exec_update_failures") return c->st.graph_exec_update_failures;
    if (k == "graph_evictions") return c->st.graph_evictions;
    if (k == "graph_cache_size") return (int64_t) c->graphs.size();
    if (k == "graph_launch_host_ns") return c->st.graph_launch_host_ns;
    if (k == "kernel_downloads") return c->st.kernel_downloads;
    if (k == "kv_image_nodes") return c->st.kv_image_nodes;
    if (k == "kv_native_nodes") return c->st.kv_native_nodes;
    if (k == "graph_key_host_ns") return c->st.graph_key_host_ns;
    if (k == "graph_compute_host_ns") return c->st.graph_compute_host_ns;
    if (k == "graph_key_fast_hits") return c->st.graph_key_fast_hits;
    if (k == "graph_key_collisions") return c->st.graph_key_collisions;
    if (k == "graph_early_captures") return c->st.graph_early_captures;
    if (k == "graph_shadow_captures") return c->st.graph_shadow_captures;
    if (k == "graph_capture_walk_ns") return c->st.graph_capture_walk_ns;
    if (k == "graph_exec_update_ns") return c->st.graph_exec_update_ns;
    if (k == "graph_shadow_eager_ns") return c->st.graph_shadow_eager_ns;
    if (k == "graph_exec_updates") return c->st.graph_exec_updates;
    if (k == "skinny_launches") return c->st.skinny_launches;
    if (k == "wide_launches") return c->st.wide_launches;
    if (k == "tiled_launches") return c->st.tiled_launches;
    if (k == "rope_epilogues") return c->st.rope_epilogues;
    if (k.rfind("ip_", 0) == 0) return ip_stat(c, key);
    if (k == "staged_upload_bytes") return (int64_t) g_uploaders[c->device].bytes;
    if (k == "staged_upload_us") return (int64_t) (g_uploaders[c->device].seconds * 1e6);
    return -1;
}
static int api_timing_report(ggml_backend_t be, char * buf, size_t size, int reset) {
    if (!be_is_ours(be) || !buf || size == 0) return -1;
    backend_ctx * c = (backend_ctx *) be->context;
    HIP_SOFT(hipStreamSynchronize(c->stream));
    for (auto & pe : c->pending_events) {
        float ms = 0;
        HIP_NOTE(hipEventElapsedTime(&ms, pe.second.first, pe.second.second));
        c->timing[pe.first].total_ms += ms;
        c->timing[pe.first].count += 1;
        HIP_NOTE(hipEventDestroy(pe.second.first));
        HIP_NOTE(hipEventDestroy(pe.second.second));
    }
    c->pending_events.clear();
    std::string out;
    for (auto & kv : c->timing) {
        if (kv.first.rfind("bytes:", 0) == 0) continue;
        auto itb = c->timing.find("bytes:" + kv.first);
        char line[256];
        snprintf(line, sizeof(line), "%s %lld %.6f %.0f\n", kv.first.c_str(), (long long) kv.second.count, kv.second.total_ms,
                 itb == c->timing.end() ? 0.0 : itb->second.total_ms);
        out += line;
    }
    snprintf(buf, size, "%s", out.c_str());
    if (reset) c->timing.clear();
    return 0;
}

static const char * reg_get_name(ggml_backend_reg_t) { return GGML_MI355X_NAME; }
static size_t reg_get_device_count(ggml_backend_reg_t) { return g_reg_ctx.devices.size(); }
static ggml_backend_dev_t reg_get_device(ggml_backend_reg_t, size_t i) { return i < g_reg_ctx.devices.size() ? g_reg_ctx.devices[i] : nullptr; }
static void * reg_get_proc_address(ggml_backend_reg_t, const char * name) {
    const std::string n = name;
    if (n == "ggml_backend_get_features") return (void *) get_features;
    if (n == "ggml_backend_mi355x_tp_init") return (void *) api_tp_init;
    if (n == "ggml_backend_mi355x_tp_get_unique_id") return (void *) api_tp_get_unique_id;
    if (n == "ggml_backend_mi355x_tp_p2p_export") return (void *) api_tp_p2p_export;
    if (n == "ggml_backend_mi355x_tp_p2p_attach") return (void *) api_tp_p2p_attach;
    if (n == "ggml_backend_mi355x_tp_all_reduce") return (void *) api_tp_all_reduce;
    if (n == "ggml_backend_mi355x_tp_rowpar_buffer_type") return (void *) api_tp_rowpar_buft;
    if (n == "ggml_backend_mi355x_set_option") return (void *) api_set_option;
    if (n == "ggml_backend_mi355x_decode_copy_read") return (void *) api_decode_copy_read;
    if (n == "ggml_backend_mi355x_get_stat") return (void *) api_get_stat;
    if (n == "ggml_backend_mi355x_timing_report") return (void *) api_timing_report;
    if (n == "ggml_backend_split_buffer_type") return (void *) api_split_buffer_type;  // -sm row (split.cpp)
    if (n == "ggml_backend_mi355x_split_rows") return (void *) api_split_rows;
    if (n == "ggml_backend_mi355x_graph_key_probe") return (void *) api_graph_key_probe;
    return nullptr;
}
static const ggml_backend_reg_i k_reg_iface = {reg_get_name, reg_get_device_count, reg_get_device, reg_get_proc_address};

static int count_gfx950() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void) hipGetLastError();
        return 0;
    }
    int ok = 0;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, i) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) ok++;
    }
    return ok;
}

static void init_reg() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void) hipGetLastError();
        n = 0;
    }
    // GGML_MI355X_FAKE_DEVICES=N: N logical devices per physical one (tests of the multi-device paths on a single-GPU box)
    const int fake = getenv("GGML_MI355X_FAKE_DEVICES") ? std::max(1, atoi(getenv("GGML_MI355X_FAKE_DEVICES"))) : 1;
    for (int ii = 0; ii < n * fake && (int) g_reg_ctx.devices.size() < GGML_MI355X_MAX_DEVICES; ++ii) {
        const int i = ii / fake;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, i) != hipSuccess) continue;
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            MI_INFO("skipping HIP device %d (%s): not gfx950", i, prop.gcnArchName);
            continue;
        }
        device_ctx * d = new device_ctx();
        d->device = i;
        d->name = std::string(GGML_MI355X_NAME) + std::to_string(g_reg_ctx.devices.size());
        d->description = prop.name;
        ggml_backend_device * dev = new ggml_backend_device{k_device_iface, &g_reg, d};
        d->buft = {k_buft_iface, dev, new buft_ctx{i, false, d->name}};
        d->buft_rowpar = {k_buft_iface, dev, new buft_ctx{i, true, d->name + "_RowPar"}};
        d->buft_host = {{hbuft_name, hbuft_alloc, hbuft_alignment, nullptr, nullptr, hbuft_is_host}, dev, nullptr};
        g_reg_ctx.devices.push_back(dev);
    }
    // (no run-time override of api_version: a host of another version also has other struct layouts — ggml_backend_dev_props
    // grew a field between 1 and 2 — so registering under its number would be a lie; rebuild with -DGGML_BACKEND_API_VERSION=N)
    g_reg = {GGML_BACKEND_API_VERSION, k_reg_iface, &g_reg_ctx};
}

// include/ggml_abi.h restates enum ggml_op from memory (the headers are un-vendored).  If the host was built from a
// tree that numbers the operators differently — upstream inserts ops now and then — every dispatch after the insertion point would
// silently run the wrong kernel.  The host's own name table settles it: ggml_op_name() lives in libggml-base, which a
// GGML_BACKEND_DL host has loaded before it dlopen()s a backend.  A symbol that cannot be resolved (fully static host) is
// logged and tolerated; a NAME MISMATCH refuses registration.
static bool op_numbering_matches_host() {
    typedef const char * (*op_name_fn)(int);
    op_name_fn host_name = (op_name_fn) dlsym(RTLD_DEFAULT, "ggml_op_name");
    if (!host_name) {
        MI_INFO("ggml_op_name is not resolvable in this process: operator numbering of include/ggml_abi.h not verified against the host");
        return true;
    }
    static const struct { int op; const char * name; } k_expect[] = {
        {GGML_OP_NONE, "NONE"}, {GGML_OP_DUP, "DUP"}, {GGML_OP_ADD, "ADD"}, {GGML_OP_SUB, "SUB"}, {GGML_OP_MUL, "MUL"}, {GGML_OP_DIV, "DIV"},
        {GGML_OP_ARGMAX, "ARGMAX"}, {GGML_OP_RMS_NORM, "RMS_NORM"}, {GGML_OP_MUL_MAT, "MUL_MAT"}, {GGML_OP_MUL_MAT_ID, "MUL_MAT_ID"}, {GGML_OP_SCALE, "SCALE"},
        {GGML_OP_CPY, "CPY"}, {GGML_OP_CONT, "CONT"}, {GGML_OP_RESHAPE, "RESHAPE"}, {GGML_OP_VIEW, "VIEW"}, {GGML_OP_PERMUTE, "PERMUTE"},
        {GGML_OP_TRANSPOSE, "TRANSPOSE"}, {GGML_OP_GET_ROWS, "GET_ROWS"}, {GGML_OP_SET_ROWS, "SET_ROWS"}, {GGML_OP_SOFT_MAX, "SOFT_MAX"},
        {GGML_OP_ROPE, "ROPE"}, {GGML_OP_FLASH_ATTN_EXT, "FLASH_ATTN_EXT"}, {GGML_OP_UNARY, "UNARY"}, {GGML_OP_GLU, "GLU"},
    };
    bool ok = true;
    for (const auto & e : k_expect) {
        const char * got = host_name(e.op);
        if (!got || strcmp(got, e.name) != 0) {
            MI_ERR("operator numbering mismatch: this library was built with %s = %d, the host calls op %d \"%s\" — rebuild against the host's ggml.h (tools/abi_dump.c prints both tables)",
                   e.name, e.op, e.op, got ? got : "(null)");
            ok = false;
        }
    }
    return ok;
}

}  // namespace mi355x

extern "C" {

__attribute__((visibility("default"))) ggml_backend_reg_t ggml_backend_mi355x_reg(void) {
    std::call_once(mi355x::g_reg_once, mi355x::init_reg);
    return &mi355x::g_reg;
}

__attribute__((visibility("default"))) ggml_backend_reg_t ggml_backend_init(void) {
    ggml_backend_reg_t reg = ggml_backend_mi355x_reg();
    if (mi355x::g_reg_ctx.devices.empty()) {
        MI_ERR("no gfx950 (MI355X) device visible: backend not registered");
        return nullptr;
    }
    static const bool ops_ok = mi355x::op_numbering_matches_host();
    if (!ops_ok) return nullptr;
    return reg;
}

__attribute__((visibility("default"))) int ggml_backend_score(void) { return mi355x::count_gfx950() > 0 ? 100 : 0; }
}
