// split.cpp — "ggml_backend_split_buffer_type": the row-split weight buffer of `-sm row --tensor-split a,b,c,...`, one process
// driving N MI355X.
//
// How it is reached: llama-box parses -sm row (/root/reference/llama-box/engine_param.hpp:902-916) and -ts
// (:821-842) into llama.cpp's model params; llama.cpp then asks the backend registry of the main GPU for the proc address
// "ggml_backend_split_buffer_type" and calls it with (main_device, tensor_split[]).  Mat-mul weights are allocated in the buffer
// type it returns; every other tensor (activations, KV cache) stays in the main device's ordinary buffer, and the graph is
// computed by the MAIN device's backend alone.  So the contract is upstream's row split, not a Megatron layout: a weight
// [K, N] is cut by ROWS (output features) in the given proportions, device d owns rows [row0[d], row0[d+1]), and a MUL_MAT on
// it is: broadcast the activations, every device computes its rows, the row ranges are gathered into the main device's dst.  No
// reduction is involved (rows are independent), so results are bit-identical to the single-device kernels.
//
// Round 3: the TENSOR-PARALLEL layout north_star asks for is reachable through this interface.  A weight whose name says it consumes a
// sharded activation — attn_output, ffn_down — is cut along K instead (whole 256-value super-blocks: "row-parallel"), in the same
// proportions (256-row granules) as the rows of the weights that produce that activation.  Then
//   * the FFN runs sharded end to end (run_split_ffn): x is broadcast once, every device computes gate / up / SwiGLU for ITS rows of the
//     hidden layer and multiplies them with ITS K range of ffn_down; the only exchange is the sum of the n_dev partial rows — one
//     in-stream reduction by the main device (a kernel that reads the partials out of the peers' memory over xGMI, device order, then the
//     residual), no gather of the 2 x n_ff hidden values;
//   * attn_output takes the K ranges of the attention result (scatter instead of broadcast) and ends in the same reduction:
//     two reductions per layer, `allreduces` counts them.  Attention itself and the KV cache stay on the main device — the host
//     allocates them there (llama.cpp keeps KV and non-matmul tensors in the main GPU's buffer under -sm row).
//
// MI355X-first choices: slices live in per-device hipMalloc allocations made at init_tensor time (GGUF bytes verbatim: the
// kernels address a slice exactly like a whole tensor); the exchange is P2P over xGMI — hipMemcpyPeerAsync on the owning device's
// stream, ordered by events against the main stream — 16 KB of activations out and N_d floats back per device for a decode token;
// each device runs the same hand-written mat-vec kernels (mmvq.hip) on its slice.  Honest limit, stated in DESIGN.md §6: with the
// KV cache and attention pinned to the main device by the host, this interface cannot reach the >= 3.5x decode target — the
// one-process-per-GPU tensor-parallel form (tp.cpp: sharded heads, two all-reduces per layer) is the scaling path; this one
// exists so that `-sm row` WORKS through the reference's own interface.
// GGML_MI355X_FAKE_DEVICES=N registers N logical devices on every physical one: the tests exercise the whole split path (placement,
// scatter/gather, multi-stream execution) on a single-GPU box.
#include <algorithm>
#include <mutex>
#include <unordered_map>

#include "common.h"
#include "kernels.h"

namespace mi355x {

// rows [row0[d], row0[d+1]) of an nrows-row tensor go to device d: cumulative proportions (all-zero or missing = even), boundaries
// rounded down to `granule` rows (upstream rounds to its mat-mul tile height; here 64 keeps rotary pairs, heads' halves and the
// 16-row work units of the mat-vec kernels together), the last device takes the remainder
void split_rows(int64_t nrows, const float * tensor_split, int n_dev, int64_t granule, int64_t * row0) {
    double total = 0.0;
    for (int d = 0; d < n_dev; ++d) total += tensor_split ? std::max(0.0f, tensor_split[d]) : 0.0f;
    double cum = 0.0;
    for (int d = 0; d < n_dev; ++d) {
        const double frac = total > 0.0 ? cum / total : (double) d / n_dev;
        int64_t r = (int64_t) (frac * (double) nrows);
        r -= r % granule;
        row0[d] = std::min(std::max<int64_t>(r, d ? row0[d - 1] : 0), nrows);
        cum += tensor_split && total > 0.0 ? std::max(0.0f, tensor_split[d]) : 0.0;
    }
    row0[0] = 0;
    row0[n_dev] = nrows;
}

struct split_buft_ctx {
    int main_device = 0;  // logical device index
    int n_dev = 0;
    float split[GGML_MI355X_MAX_DEVICES] = {0};
    std::string name;
};
struct split_buffer_ctx {
    split_buft_ctx * bt = nullptr;
    std::unordered_map<const ggml_tensor *, split_tensor_info> tensors;
};

static void sbuf_free(ggml_backend_buffer_t b) {
    split_buffer_ctx * c = (split_buffer_ctx *) b->context;
    for (auto & kv : c->tensors)
        for (int d = 0; d < kv.second.n_dev; ++d)
            if (kv.second.slice[d]) {
                HIP_SOFT(hipSetDevice(logical_device_ordinal(d)));
                HIP_NOTE(hipFree(kv.second.slice[d]));
            }
    delete c;
}
static void * sbuf_get_base(ggml_backend_buffer_t) { return (void *) 0x1000; }  // never dereferenced (as upstream's split buffer)
static enum ggml_status sbuf_init_tensor(ggml_backend_buffer_t b, ggml_tensor * t) {
    split_buffer_ctx * c = (split_buffer_ctx *) b->context;
    if (t->view_src != nullptr || !ggml_abi_is_contiguous(t) || t->ne[2] != 1 || t->ne[3] != 1) {
        MI_ERR("split buffer: tensor '%s' is a view / not a contiguous matrix", t->name);
        return GGML_STATUS_FAILED;
    }
    auto old = c->tensors.find(t);
    if (old != c->tensors.end()) {  // re-initialised (a host may call init_tensor again after a reset): the previous slices go first
        for (int d = 0; d < old->second.n_dev; ++d)
            if (old->second.slice[d]) {
                HIP_SOFT(hipSetDevice(logical_device_ordinal(d)));
                HIP_NOTE(hipFree(old->second.slice[d]));
            }
        c->tensors.erase(old);
    }
    split_tensor_info info{};
    info.n_dev = c->bt->n_dev;
    info.row_bytes = ggml_abi_row_size(t->type, t->ne[0]);
    const int64_t blck = ggml_abi_blck_size(t->type);
    // row-parallel weights by their GGUF names (llama.cpp's tensor names: model.patch:20-30 corroborates the style); needs whole
    // 256-value super-blocks per device and a quantised type; anything else is cut by rows
    const bool by_k = (strstr(t->name, "attn_output") || strstr(t->name, "ffn_down")) && blck > 1 && (t->ne[0] % 256) == 0 && !getenv("GGML_MI355X_SPLIT_ROWS_ONLY");
    info.kind = by_k ? 1 : 0;
    // The attention weights are cut at EIGHTHS of their rows (attn_output: of its K range): wq, wk, wv and attn_output then agree on which heads a
    // device owns whatever their sizes — a device's rows of wk / wv are whole KV heads (8 of them in the Llama-3 family, 4 in Qwen2-7B: halves of
    // the devices for -ts over 2 or 4), its rows of wq the query heads of exactly those groups, its K range of attn_output their outputs — which
    // is what lets the attention and the KV cache be sharded with them (tp_inproc.cpp; it checks, and declines where a head would be cut).
    const bool attn = strstr(t->name, "attn_q") || strstr(t->name, "attn_k") || strstr(t->name, "attn_v") || strstr(t->name, "attn_output");
    const int64_t cut = by_k ? t->ne[0] : t->ne[1];
    const int64_t eighth = attn && cut % 8 == 0 && (by_k ? (cut / 8) % 256 == 0 : (cut / 8) % 16 == 0) && !getenv("GGML_MI355X_SPLIT_LEGACY_ROWS") ? cut / 8 : 0;
    if (by_k) split_rows(t->ne[0], c->bt->split, info.n_dev, eighth ? eighth : 256, info.row0);
    else split_rows(t->ne[1], c->bt->split, info.n_dev, eighth ? eighth : (t->ne[1] % 256 == 0 ? 256 : 64), info.row0);  // (256-row granules where the rows can become another weight's K range)
    for (int d = 0; d < info.n_dev; ++d) {
        const int64_t part = info.row0[d + 1] - info.row0[d];
        info.slice_row_bytes[d] = by_k ? (size_t) (part / blck) * ggml_abi_type_size(t->type) : info.row_bytes;
        if (part <= 0) continue;
        const size_t bytes = by_k ? (size_t) t->ne[1] * info.slice_row_bytes[d] : (size_t) part * info.row_bytes;
        HIP_SOFT(hipSetDevice(logical_device_ordinal(d)));
        if (hipMalloc(&info.slice[d], bytes + 256) != hipSuccess) {
            (void) hipGetLastError();
            MI_ERR("split buffer: allocating %.1f MiB of '%s' on device %d failed", bytes / 1048576.0, t->name, d);
            return GGML_STATUS_ALLOC_FAILED;
        }
    }
    c->tensors[t] = info;
    return GGML_STATUS_SUCCESS;
}
static const split_tensor_info * find(ggml_backend_buffer_t b, const ggml_tensor * t) {
    split_buffer_ctx * c = (split_buffer_ctx *) b->context;
    auto it = c->tensors.find(t);
    return it == c->tensors.end() ? nullptr : &it->second;
}
// whole tensors only, like upstream's split buffer (the loader writes a tensor in one call)
static void sbuf_set_tensor(ggml_backend_buffer_t b, ggml_tensor * t, const void * data, size_t offset, size_t size) {
    const split_tensor_info * info = find(b, t);
    if (!info || offset != 0 || size != ggml_abi_nbytes(t)) {
        MI_ERR("split buffer: set_tensor of '%s' must cover the whole tensor", t->name);
        abort();
    }
    const size_t ts = ggml_abi_type_size(t->type), blck = (size_t) ggml_abi_blck_size(t->type);
    for (int d = 0; d < info->n_dev; ++d) {
        const int64_t rows = info->row0[d + 1] - info->row0[d];
        if (rows <= 0) continue;
        HIP_SOFT(hipSetDevice(logical_device_ordinal(d)));
        if (info->kind == 1)  // device d's K range of every row
            HIP_SOFT(hipMemcpy2D(info->slice[d], info->slice_row_bytes[d], (const char *) data + (size_t) info->row0[d] / blck * ts, info->row_bytes, info->slice_row_bytes[d], (size_t) t->ne[1], hipMemcpyHostToDevice));
        else
            HIP_SOFT(hipMemcpy(info->slice[d], (const char *) data + (size_t) info->row0[d] * info->row_bytes, (size_t) rows * info->row_bytes, hipMemcpyHostToDevice));
    }
}
static void sbuf_get_tensor(ggml_backend_buffer_t b, const ggml_tensor * t, void * data, size_t offset, size_t size) {
    const split_tensor_info * info = find(b, t);
    if (!info || offset != 0 || size != ggml_abi_nbytes(t)) {
        MI_ERR("split buffer: get_tensor of '%s' must cover the whole tensor", t->name);
        abort();
    }
    const size_t ts = ggml_abi_type_size(t->type), blck = (size_t) ggml_abi_blck_size(t->type);
    for (int d = 0; d < info->n_dev; ++d) {
        const int64_t rows = info->row0[d + 1] - info->row0[d];
        if (rows <= 0) continue;
        HIP_SOFT(hipSetDevice(logical_device_ordinal(d)));
        if (info->kind == 1)
            HIP_SOFT(hipMemcpy2D((char *) data + (size_t) info->row0[d] / blck * ts, info->row_bytes, info->slice[d], info->slice_row_bytes[d], info->slice_row_bytes[d], (size_t) t->ne[1], hipMemcpyDeviceToHost));
        else
            HIP_SOFT(hipMemcpy((char *) data + (size_t) info->row0[d] * info->row_bytes, info->slice[d], (size_t) rows * info->row_bytes, hipMemcpyDeviceToHost));
    }
}
static void sbuf_memset_tensor(ggml_backend_buffer_t b, ggml_tensor * t, uint8_t value, size_t offset, size_t size) {
    const split_tensor_info * info = find(b, t);
    if (!info || offset != 0 || size != ggml_abi_nbytes(t)) { MI_ERR("split buffer: memset_tensor must cover the whole tensor"); abort(); }
    for (int d = 0; d < info->n_dev; ++d) {
        const int64_t rows = info->row0[d + 1] - info->row0[d];
        if (rows <= 0) continue;
        HIP_SOFT(hipSetDevice(logical_device_ordinal(d)));
        HIP_SOFT(hipMemset(info->slice[d], value, info->kind == 1 ? (size_t) t->ne[1] * info->slice_row_bytes[d] : (size_t) rows * info->row_bytes));
    }
}
static void sbuf_clear(ggml_backend_buffer_t, uint8_t) {}  // (slices are written whole by set_tensor; nothing to clear before that)
static const ggml_backend_buffer_i k_split_buffer_iface = {sbuf_free, sbuf_get_base, sbuf_init_tensor, sbuf_memset_tensor, sbuf_set_tensor,
                                                           sbuf_get_tensor, /* cpy_tensor */ nullptr, sbuf_clear, nullptr};

bool buffer_is_split(ggml_backend_buffer_t b) { return b != nullptr && b->iface.free_buffer == sbuf_free; }
const split_tensor_info * split_info(const ggml_tensor * t) { return buffer_is_split(t->buffer) ? find(t->buffer, t) : nullptr; }

static const char * sbuft_name(ggml_backend_buffer_type_t buft) { return ((split_buft_ctx *) buft->context)->name.c_str(); }
static ggml_backend_buffer_t sbuft_alloc(ggml_backend_buffer_type_t buft, size_t size) {
    split_buffer_ctx * c = new split_buffer_ctx();
    c->bt = (split_buft_ctx *) buft->context;
    return make_backend_buffer(buft, k_split_buffer_iface, c, size);
}
static size_t sbuft_alignment(ggml_backend_buffer_type_t) { return 128; }
static size_t sbuft_alloc_size(ggml_backend_buffer_type_t, const ggml_tensor * t) { return (ggml_abi_nbytes(t) + 255) / 256 * 256; }
static bool sbuft_is_host(ggml_backend_buffer_type_t) { return false; }
static const ggml_backend_buffer_type_i k_split_buft_iface = {sbuft_name, sbuft_alloc, sbuft_alignment, /* max_size */ nullptr, sbuft_alloc_size, sbuft_is_host};

bool buft_is_split(ggml_backend_buffer_type_t buft) { return buft != nullptr && buft->iface.get_name == sbuft_name; }
int split_buft_main_device(ggml_backend_buffer_type_t buft) { return ((split_buft_ctx *) buft->context)->main_device; }

// the proc address: ggml_backend_split_buffer_type_t(int main_device, const float * tensor_split).  One buffer type object per distinct
// (main device, proportions); they live for the life of the process, like upstream's.
ggml_backend_buffer_type_t split_buffer_type(int main_device, const float * tensor_split) {
    static std::mutex mtx;
    static std::vector<ggml_backend_buffer_type *> all;
    const int n = logical_device_count();
    if (main_device < 0 || main_device >= n) return nullptr;
    std::lock_guard<std::mutex> lock(mtx);
    for (auto * bt : all) {
        split_buft_ctx * c = (split_buft_ctx *) bt->context;
        bool same = c->main_device == main_device;
        for (int d = 0; d < n && same; ++d) same = c->split[d] == (tensor_split ? tensor_split[d] : 0.0f);
        if (same) return bt;
    }
    split_buft_ctx * c = new split_buft_ctx();
    c->main_device = main_device;
    c->n_dev = n;
    for (int d = 0; d < n; ++d) c->split[d] = tensor_split ? tensor_split[d] : 0.0f;
    c->name = std::string(GGML_MI355X_NAME) + "_Split";
    ggml_backend_buffer_type * bt = new ggml_backend_buffer_type{k_split_buft_iface, logical_device(main_device), c};
    all.push_back(bt);
    return bt;
}

// ------------------------------------------------------------------------------------------------ execution
// per owning device: a stream, scratch for the activation copy / the result rows, an event
struct split_helper {
    int ordinal = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev = nullptr;
    char * ws = nullptr;
    size_t ws_size = 0;
};
static split_helper * helper_for(backend_ctx * c, int d) {
    if ((int) c->split_helpers.size() <= d) c->split_helpers.resize((size_t) d + 1, nullptr);
    if (!c->split_helpers[d]) {
        split_helper * h = new split_helper();
        h->ordinal = logical_device_ordinal(d);
        if (hipSetDevice(h->ordinal) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev, hipEventDisableTiming) != hipSuccess) {
            (void) hipGetLastError();
            MI_ERR("split buffer: no stream / event on device %d", h->ordinal);
            if (h->stream) (void) hipStreamDestroy(h->stream);
            delete h;
            (void) hipSetDevice(c->device);
            return nullptr;
        }
        if (h->ordinal != c->device) {
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, h->ordinal, c->device) == hipSuccess && can) {
                if (hipDeviceEnablePeerAccess(c->device, 0) != hipSuccess) (void) hipGetLastError();  // (already enabled is fine)
            }
            // ... and the main device reads the helpers' partial products in its reduction kernel (launch_reduce_parts)
            (void) hipSetDevice(c->device);
            if (hipDeviceCanAccessPeer(&can, c->device, h->ordinal) == hipSuccess && can) {
                if (hipDeviceEnablePeerAccess(h->ordinal, 0) != hipSuccess) (void) hipGetLastError();
            }
        }
        (void) hipSetDevice(c->device);
        c->split_helpers[d] = h;
    }
    return c->split_helpers[d];
}
// Called with the helper's device current, before its workspace is freed and regrown.  hipFree() waits for the allocation's OWN
// device only, but since round 3 the main device's k_reduce_parts reads the partial products out of this workspace over peer access
// on c->stream: wait for both streams, so that no reader — local or remote — can still be queued.
static bool quiesce_readers(backend_ctx * c, split_helper * h) {
    HIP_TRY(hipStreamSynchronize(h->stream), false);
    HIP_TRY(hipSetDevice(c->device), false);
    const hipError_t e = hipStreamSynchronize(c->stream);
    if (hipSetDevice(h->ordinal) != hipSuccess || e != hipSuccess) {
        (void) hipGetLastError();
        MI_ERR("split buffer: waiting for the main stream before a workspace regrow failed");
        return false;
    }
    return true;
}
// run_split_* are called with the main device current and return with it current on EVERY path: the early returns of HIP_TRY
// inside their per-device loops used to leave a helper's device current for graph_compute's failure path.
struct main_device_guard {
    int device;
    explicit main_device_guard(const backend_ctx * c) : device(c->device) {}
    ~main_device_guard() { if (hipSetDevice(device) != hipSuccess) (void) hipGetLastError(); }
};
void free_split_helpers(backend_ctx * c) {
    for (split_helper * h : c->split_helpers) {
        if (!h) continue;
        HIP_SOFT(hipSetDevice(h->ordinal));
        HIP_SOFT(hipStreamSynchronize(h->stream));
        if (h->ws) HIP_NOTE(hipFree(h->ws));
        HIP_NOTE(hipEventDestroy(h->ev));
        HIP_SOFT(hipStreamDestroy(h->stream));
        delete h;
    }
    c->split_helpers.clear();
    HIP_SOFT(hipSetDevice(c->device));
    if (c->split_ready) {
        HIP_NOTE(hipEventDestroy(c->split_ready));
        c->split_ready = nullptr;
    }
}

bool split_mul_mat_supported(const ggml_tensor * op) {
    const ggml_tensor * w = op->src[0];
    const ggml_tensor * b = op->src[1];
    const bool quant = w->type == GGML_TYPE_Q4_K || w->type == GGML_TYPE_Q5_K || w->type == GGML_TYPE_Q6_K || w->type == GGML_TYPE_Q8_0;
    return quant && (w->ne[0] % 256) == 0 && w->ne[2] == 1 && w->ne[3] == 1 && b->type == GGML_TYPE_F32 && ggml_abi_is_contiguous(b) && op->type == GGML_TYPE_F32 &&
           ggml_abi_is_contiguous(op) && (((uintptr_t) b->data) & 15) == 0;
}

// dst[N, M] = W[K, N] (rows split over the devices) x b[K, M]; called with the main device current and returns with it current
bool run_split_mul_mat(backend_ctx * c, const ggml_tensor * w, const ggml_tensor * b, ggml_tensor * dst) {
    const main_device_guard restore_device(c);
    const split_tensor_info * info = split_info(w);
    if (!info) return false;
    if (info->kind == 1) return run_split_rowpar(c, w, b, dst, nullptr);
    const int64_t K = w->ne[0], N = w->ne[1], M = b->ne[1] * b->ne[2] * b->ne[3];
    const size_t x_bytes = ggml_abi_nbytes(b);
    const int act_kind = w->type == GGML_TYPE_Q8_0 ? GGML_TYPE_Q8_0 : GGML_TYPE_Q8_K;
    if (!c->split_ready) HIP_TRY(hipEventCreateWithFlags(&c->split_ready, hipEventDisableTiming), false);  // one per backend, re-recorded per mat-mul
    hipEvent_t ready = c->split_ready;  // the activations exist on the main stream from here on
    HIP_TRY(hipEventRecord(ready, c->stream), false);
    for (int d = 0; d < info->n_dev; ++d) {
        const int64_t rows = info->row0[d + 1] - info->row0[d];
        if (rows <= 0) continue;
        split_helper * h = helper_for(c, d);
        if (!h) return false;
        HIP_TRY(hipSetDevice(h->ordinal), false);
        const size_t q_bytes = quantized_act_bytes(act_kind, K, M);
        const size_t need = ((x_bytes + 255) & ~(size_t) 255) + ((q_bytes + 255) & ~(size_t) 255) + (size_t) rows * M * sizeof(float) + 256;
        if (need > h->ws_size) {
            if (c->capturing) { (void) hipSetDevice(c->device); return false; }  // (sized on the eager first sighting of a topology; never inside a capture)
            if (!quiesce_readers(c, h)) { (void) hipSetDevice(c->device); return false; }
            if (h->ws) (void) hipFree(h->ws);
            h->ws = nullptr;
            h->ws_size = 0;
            if (hipMalloc((void **) &h->ws, need + (1u << 20)) != hipSuccess) {
                (void) hipGetLastError();
                MI_ERR("split buffer: %.1f MiB of scratch on device %d failed", (need + (1u << 20)) / 1048576.0, h->ordinal);
                (void) hipSetDevice(c->device);
                return false;  // -> GGML_STATUS_FAILED for this llama_decode, not abort()
            }
            h->ws_size = need + (1u << 20);
        }
        float * xd = (float *) h->ws;
        char * qd = h->ws + ((x_bytes + 255) & ~(size_t) 255);
        float * yd = (float *) (qd + ((q_bytes + 255) & ~(size_t) 255));
        HIP_TRY(hipStreamWaitEvent(h->stream, ready, 0), false);
        if (h->ordinal == c->device) HIP_TRY(hipMemcpyAsync(xd, b->data, x_bytes, hipMemcpyDeviceToDevice, h->stream), false);
        else HIP_TRY(hipMemcpyPeerAsync(xd, h->ordinal, b->data, c->device, x_bytes, h->stream), false);
        mmvq_args a{};
        a.W = (const uint8_t *) info->slice[d];
        a.w_nb1 = (int64_t) info->row_bytes;
        a.type = w->type;
        a.K = (int) K;
        a.N = (int) rows;
        a.dst = yd;
        a.dst_stride = rows;
        if (M == 1) {
            a.ncols = 1;
            a.x = xd;  // f32 prologue: the mat-vec quantises the row itself (mmvq.hip PRO 1)
            launch_mmvq(h->stream, a, 1);
        } else {
            tdesc xdsc{(char *) xd, {K, M, 1, 1}, {4, K * 4, K * M * 4, K * M * 4}, GGML_TYPE_F32};
            launch_quantize_act(h->stream, act_kind, xdsc, qd);
            a.ncols = (int) M;
            a.act = qd;
            launch_mmvq(h->stream, a, 1);  // (columns in chunks of 8: functional for prompt batches, not tuned — DESIGN.md §6)
        }
        // gather: rows [row0, row0 + rows) of every column of dst
        if (M == 1) HIP_TRY(hipMemcpyAsync((float *) dst->data + info->row0[d], yd, (size_t) rows * sizeof(float), hipMemcpyDeviceToDevice, h->stream), false);
        else HIP_TRY(hipMemcpy2DAsync((float *) dst->data + info->row0[d], (size_t) N * sizeof(float), yd, (size_t) rows * sizeof(float), (size_t) rows * sizeof(float), (size_t) M,
                                      hipMemcpyDeviceToDevice, h->stream), false);
        HIP_TRY(hipEventRecord(h->ev, h->stream), false);
        c->st.kernel_launches += 1;
    }
    HIP_TRY(hipSetDevice(c->device), false);
    for (int d = 0; d < info->n_dev; ++d)
        if (info->row0[d + 1] > info->row0[d]) HIP_TRY(hipStreamWaitEvent(c->stream, c->split_helpers[d]->ev, 0), false);
    return hipGetLastError() == hipSuccess;
}


// ------------------------------------------------------------------------------------------------ tensor-parallel forms
static bool helper_ws(backend_ctx * c, split_helper * h, size_t need) {
    if (need <= h->ws_size) return true;
    if (c->capturing) return false;  // (sized on the eager first sighting of a topology; never inside a capture)
    if (!quiesce_readers(c, h)) return false;
    if (h->ws) (void) hipFree(h->ws);
    h->ws = nullptr;
    h->ws_size = 0;
    if (hipMalloc((void **) &h->ws, need + (1u << 20)) != hipSuccess) {
        (void) hipGetLastError();
        MI_ERR("split buffer: %.1f MiB of scratch on device %d failed", (need + (1u << 20)) / 1048576.0, h->ordinal);
        return false;
    }
    h->ws_size = need + (1u << 20);
    return true;
}
static size_t al256(size_t n) { return (n + 255) & ~(size_t) 255; }
// columns of `M` f32 values each at `x` (row stride K): one mat-vec (M = 1: f32 prologue inside the kernel) or quantise + multi-column kernel
static void slice_matmul(hipStream_t s, int type, const void * W, const void * W2, int64_t w_nb1, int64_t K, int64_t N, int64_t M, const float * x, char * q8, float * out) {
    mmvq_args a{};
    a.W = (const uint8_t *) W;
    a.W2 = (const uint8_t *) W2;
    a.w_nb1 = w_nb1;
    a.type = type;
    a.K = (int) K;
    a.N = (int) N;
    a.dst = out;
    a.dst_stride = N;
    if (M == 1) {
        a.ncols = 1;
        a.x = x;
    } else {
        const int kind = type == GGML_TYPE_Q8_0 ? GGML_TYPE_Q8_0 : GGML_TYPE_Q8_K;
        tdesc xd{(char *) x, {K, M, 1, 1}, {4, K * 4, K * M * 4, K * M * 4}, GGML_TYPE_F32};
        launch_quantize_act(s, kind, xd, q8);
        a.ncols = (int) M;
        a.act = q8;
    }
    launch_mmvq(s, a, 1);
}

static bool join_and_reduce(backend_ctx * c, const split_tensor_info * info, float * const * parts, const ggml_tensor * add, ggml_tensor * dst) {
    HIP_TRY(hipSetDevice(c->device), false);
    reduce_parts rp{};
    for (int d = 0; d < info->n_dev; ++d) {
        if (info->row0[d + 1] <= info->row0[d]) continue;
        HIP_TRY(hipStreamWaitEvent(c->stream, c->split_helpers[d]->ev, 0), false);
        rp.part[rp.n++] = parts[d];
    }
    launch_reduce_parts(c->stream, rp, add ? (const float *) add->data : nullptr, (float *) dst->data, ggml_abi_nelements(dst));
    c->st.kernel_launches++;
    c->st.allreduces++;
    return hipGetLastError() == hipSuccess;
}

bool run_split_rowpar(backend_ctx * c, const ggml_tensor * w, const ggml_tensor * b, ggml_tensor * dst, const ggml_tensor * add) {
    const main_device_guard restore_device(c);
    const split_tensor_info * info = split_info(w);
    if (!info || info->kind != 1 || (ggml_abi_nelements(dst) % 4) != 0) return false;
    const int64_t K = w->ne[0], N = w->ne[1], M = b->ne[1] * b->ne[2] * b->ne[3];
    if (!c->split_ready) HIP_TRY(hipEventCreateWithFlags(&c->split_ready, hipEventDisableTiming), false);
    HIP_TRY(hipEventRecord(c->split_ready, c->stream), false);
    float * parts[GGML_MI355X_MAX_DEVICES] = {nullptr};
    for (int d = 0; d < info->n_dev; ++d) {
        const int64_t kd = info->row0[d + 1] - info->row0[d];
        if (kd <= 0) continue;
        split_helper * h = helper_for(c, d);
        if (!h) return false;
        HIP_TRY(hipSetDevice(h->ordinal), false);
        const size_t x_b = al256((size_t) kd * M * 4), q_b = al256(quantized_act_bytes(w->type == GGML_TYPE_Q8_0 ? GGML_TYPE_Q8_0 : GGML_TYPE_Q8_K, kd, M)), p_b = al256((size_t) N * M * 4);
        if (!helper_ws(c, h, x_b + q_b + p_b + 256)) { (void) hipSetDevice(c->device); return false; }
        float * xd = (float *) h->ws;
        char * qd = h->ws + x_b;
        float * pd = (float *) (h->ws + x_b + q_b);
        HIP_TRY(hipStreamWaitEvent(h->stream, c->split_ready, 0), false);
        // this device's K range of every column of b
        if (M == 1 || kd == K) HIP_TRY(hipMemcpyAsync(xd, (const float *) b->data + info->row0[d], (size_t) kd * M * 4, hipMemcpyDeviceToDevice, h->stream), false);
        else HIP_TRY(hipMemcpy2DAsync(xd, (size_t) kd * 4, (const float *) b->data + info->row0[d], (size_t) K * 4, (size_t) kd * 4, (size_t) M, hipMemcpyDeviceToDevice, h->stream), false);
        slice_matmul(h->stream, w->type, info->slice[d], nullptr, (int64_t) info->slice_row_bytes[d], kd, N, M, xd, qd, pd);
        HIP_TRY(hipEventRecord(h->ev, h->stream), false);
        c->st.kernel_launches += 1;
        parts[d] = pd;
    }
    return join_and_reduce(c, info, parts, add, dst);
}

bool split_ffn_applies(const ggml_tensor * wg, const ggml_tensor * wu, const ggml_tensor * wd) {
    const split_tensor_info * ig = split_info(wg), * iu = split_info(wu), * id = split_info(wd);
    if (!ig || !iu || !id || ig->kind != 0 || iu->kind != 0 || id->kind != 1) return false;
    if (wg->type != wu->type || wg->ne[0] != wu->ne[0] || wg->ne[1] != wu->ne[1] || wd->ne[0] != wg->ne[1] || wg->nb[1] != wu->nb[1]) return false;
    if ((wg->ne[0] % 256) != 0 || (wd->ne[1] % 4) != 0) return false;
    for (int d = 0; d <= ig->n_dev; ++d)
        if (ig->row0[d] != iu->row0[d] || ig->row0[d] != id->row0[d]) return false;  // a device's rows of the hidden layer ARE its K range of ffn_down
    return true;
}

bool run_split_ffn(backend_ctx * c, const ggml_tensor * wg, const ggml_tensor * wu, const ggml_tensor * wd, const ggml_tensor * x, ggml_tensor * dst, const ggml_tensor * add) {
    const main_device_guard restore_device(c);
    const split_tensor_info * ig = split_info(wg), * iu = split_info(wu), * id = split_info(wd);
    const int64_t E = wg->ne[0], N = wd->ne[1], M = x->ne[1] * x->ne[2] * x->ne[3];
    if (!c->split_ready) HIP_TRY(hipEventCreateWithFlags(&c->split_ready, hipEventDisableTiming), false);
    HIP_TRY(hipEventRecord(c->split_ready, c->stream), false);
    float * parts[GGML_MI355X_MAX_DEVICES] = {nullptr};
    for (int d = 0; d < ig->n_dev; ++d) {
        const int64_t fd = ig->row0[d + 1] - ig->row0[d];
        if (fd <= 0) continue;
        split_helper * h = helper_for(c, d);
        if (!h) return false;
        HIP_TRY(hipSetDevice(h->ordinal), false);
        const int kg = wg->type == GGML_TYPE_Q8_0 ? GGML_TYPE_Q8_0 : GGML_TYPE_Q8_K, kdn = wd->type == GGML_TYPE_Q8_0 ? GGML_TYPE_Q8_0 : GGML_TYPE_Q8_K;
        const size_t x_b = al256((size_t) E * M * 4), q1_b = al256(quantized_act_bytes(kg, E, M)), a_b = al256((size_t) fd * M * 4), q2_b = al256(quantized_act_bytes(kdn, fd, M)), p_b = al256((size_t) N * M * 4);
        if (!helper_ws(c, h, x_b + q1_b + a_b + q2_b + p_b + 256)) { (void) hipSetDevice(c->device); return false; }
        float * xd = (float *) h->ws;
        char * q1 = h->ws + x_b;
        float * ad = (float *) (h->ws + x_b + q1_b);
        char * q2 = h->ws + x_b + q1_b + a_b;
        float * pd = (float *) (h->ws + x_b + q1_b + a_b + q2_b);
        HIP_TRY(hipStreamWaitEvent(h->stream, c->split_ready, 0), false);
        if (h->ordinal == c->device) HIP_TRY(hipMemcpyAsync(xd, x->data, (size_t) E * M * 4, hipMemcpyDeviceToDevice, h->stream), false);
        else HIP_TRY(hipMemcpyPeerAsync(xd, h->ordinal, x->data, c->device, (size_t) E * M * 4, h->stream), false);
        // silu(Wg x) * (Wu x) for this device's rows of the hidden layer, then its K range of ffn_down: nothing leaves the device in between
        slice_matmul(h->stream, wg->type, ig->slice[d], iu->slice[d], (int64_t) ig->row_bytes, E, fd, M, xd, q1, ad);
        slice_matmul(h->stream, wd->type, id->slice[d], nullptr, (int64_t) id->slice_row_bytes[d], fd, N, M, ad, q2, pd);
        HIP_TRY(hipEventRecord(h->ev, h->stream), false);
        c->st.kernel_launches += 2;
        parts[d] = pd;
    }
    return join_and_reduce(c, id, parts, add, dst);
}

}  // namespace mi355x
