// common.h — internal declarations shared by the backend's translation units (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/ggml_mi355x.h"

#define MI_LOG(level, ...)                                    \
    do {                                                      \
        if ((level) <= mi355x::log_level()) {                 \
            fprintf(stderr, "ggml-mi355x: " __VA_ARGS__);     \
            fprintf(stderr, "\n");                            \
        }                                                     \
    } while (0)
#define MI_ERR(...) MI_LOG(0, __VA_ARGS__)
#define MI_INFO(...) MI_LOG(1, __VA_ARGS__)
#define MI_DBG(...) MI_LOG(2, __VA_ARGS__)

// abort-on-error, the convention of every ggml backend for non-graph entry points (SURVEY.md §8b "Error conventions")
#define HIP_CHECK(expr)                                                                                         \
    do {                                                                                                        \
        hipError_t err_ = (expr);                                                                               \
        if (err_ != hipSuccess) {                                                                               \
            fprintf(stderr, "ggml-mi355x: HIP error %d (%s) at %s:%d: %s\n", (int) err_, hipGetErrorString(err_), __FILE__, __LINE__, #expr); \
            abort();                                                                                            \
        }                                                                                                       \
    } while (0)

// Entry points of the ggml interface that return no status (set_tensor, get_tensor, memset, cpy, clear, synchronize, events, free): upstream's
// backends abort() there.  llama-box is a server: a failed copy must not take every other request down with it.  The failure is logged,
// cleared and REMEMBERED (mi355x::hip_failed()): from then on every graph_compute of this process's backends returns GGML_STATUS_FAILED ->
// llama_decode rc < 0 -> the engine fails the affected requests (llama-box/httpserver.hpp:3541-3545) and the operator sees the log.
#define HIP_SOFT(expr)                                                                                          \
    do {                                                                                                        \
        hipError_t err_ = (expr);                                                                               \
        if (err_ != hipSuccess) {                                                                               \
            (void) hipGetLastError();                                                                           \
            fprintf(stderr, "ggml-mi355x: HIP error %d (%s) at %s:%d: %s — graph_compute will report failure from here on\n", (int) err_, \
                    hipGetErrorString(err_), __FILE__, __LINE__, #expr);                                        \
            mi355x::note_hip_failure();                                                                         \
        }                                                                                                       \
    } while (0)

// Bookkeeping calls that move no tensor data (event create / destroy / elapsed time of the bench's timing, frees in teardown, memory and property
// queries): a failure there is logged and cleared and NOTHING else — it must not brick inference for the rest of the process.
#define HIP_NOTE(expr)                                                                                          \
    do {                                                                                                        \
        hipError_t err_ = (expr);                                                                               \
        if (err_ != hipSuccess) {                                                                               \
            (void) hipGetLastError();                                                                           \
            fprintf(stderr, "ggml-mi355x: HIP error %d (%s) at %s:%d: %s (ignored: no tensor data involved)\n", (int) err_, hipGetErrorString(err_), __FILE__, __LINE__, #expr); \
        }                                                                                                       \
    } while (0)

// inside graph execution a HIP failure must fail ONE llama_decode (GGML_STATUS_FAILED -> rc -2, llama-box/httpserver.hpp:3541-3545),
// not the whole llama-box process: log, clear the sticky error, hand `ret` to the caller
#define HIP_TRY(expr, ret)                                                                                      \
    do {                                                                                                        \
        hipError_t err_ = (expr);                                                                               \
        if (err_ != hipSuccess) {                                                                               \
            (void) hipGetLastError();                                                                           \
            fprintf(stderr, "ggml-mi355x: HIP error %d (%s) at %s:%d: %s\n", (int) err_, hipGetErrorString(err_), __FILE__, __LINE__, #expr); \
            return ret;                                                                                         \
        }                                                                                                       \
    } while (0)

namespace mi355x {

int log_level();
void note_hip_failure();  // backend.cpp: a data-path HIP call failed outside graph execution (HIP_SOFT)
bool hip_failed();
void clear_hip_failure();  // set_option("clear_failure", 1): the operator's way back without a restart (the log says what failed)

// device-side layout of a Q8_K-quantised activation block (the ggml block_q8_K fields, 16-byte aligned:
// qs | bsums | d) — produced by quantize kernels, consumed by the K-quant matvec / GEMM kernels
// small host -> device uploads batched into one launch (ops.hip: k_upload_multi)
struct upload_seg { char * dst; const char * src; size_t n; };
struct upload_batch { upload_seg seg[8]; int n; };

struct q8k_dev {
    int8_t qs[256];
    uint16_t bsums[16]; // block_q8_K.bsums (sums over 16 values, |.| <= 2032) stored as IEEE f16 — exact — because their only
                        // readers are the matrix-core GEMMs, which feed them to an f16 MFMA as they are
    int16_t bs32[8];    // sums over 32 values (what the Q4_K/Q5_K mins term needs per sub-block)
    float d;
    float pad[3];
};
static_assert(sizeof(q8k_dev) == 320, "q8k_dev");
// Q8_0-quantised activation block for Q8_0 weights: d already rounded through fp16 (as block_q8_0.d is)
struct q80_dev {
    int8_t qs[32];
    float d;
};
static_assert(sizeof(q80_dev) == 36, "q80_dev");

struct backend_ctx;

struct options {
    bool graphs = true;        // hipGraph capture + replay of repeated graphs
    bool fusion = true;        // node fusion (norm+mul, mul_mat+add, ...)
    bool prologue = true;      // fold RMS_NORM / activation quantisation into the mat-vec prologue
    bool qkv = true;           // fused Q/K/V + rope + cache store launch
    int mmvq_max_cols = 2;     // widest batch that takes the mat-vec fusions (gate/up/SwiGLU, +bias +residual in one launch); from three columns on the
                               // skinny matrix-core kernel + its separate SwiGLU launch is faster (-np 8 step 4.06 -> 3.45 ms, -np 3 3.77 -> 3.35)
    int mmq_min_cols = 3;      // batches at least this wide run on the matrix cores (measured, ms/step matrix cores vs multi-column mat-vec: 3 columns 4.1 / 4.4, 4: 3.9 / 4.2, 8: 4.4 / 5.7; 2 columns: 3.6 / 3.1)
    bool mmq_i8 = true;        // Q4_K/Q5_K batches on the int8 matrix cores (mmq_i8.hip) instead of the f16 variant (mmq.hip)
    bool mm_merge = true;      // batches: sibling mat-muls over the same activations (wq/wk/wv, gate/up) as one launch
    bool ss_partials = true;    // residual-stream mat-vecs leave the sum of squares of their result for the next RMS_NORM prologue (GGML_MI355X_SS_PARTIALS=0: off)
    bool fa_self_merge = false; // split attention (decode): the last split workgroup merges the partial records, no combine launch.  Off: measured
                               // one token, n_kv 2100: 16.5 us against 7.6 + 5.0 for split + combine (record write-through, counter and re-read are a longer
                               // dependent chain than a launch); -np 32: 18.3 against 19.6 us per layer (4.45 vs 4.49 ms per step)
    bool attn_nf = true;       // -np decode steps on the non-flash path: K.q -> SOFT_MAX -> V^T.p as one launch over the tokens' visible-cell lists (attn_nf.hip)
    bool softmax_mm = true;    // decode on the non-flash path: SOFT_MAX folded into the V^T.p product that reads it (mmf.hip)
    bool skinny_rope = true;   // -np decode steps: ROPE(q), ROPE(k) and both KV-cache stores in the epilogue of the skinny QKV launch(es)
    bool skinny_mix = true;    // -np decode steps: sibling mat-muls stored in two K-quant formats (Q4_K wq / wk + Q6_K wv) share one skinny launch
    bool mmq_skinny = true;    // 2..32 columns: weight-streaming matrix-core kernel (mmq_skinny.hip) instead of the tiled GEMM
    int mmq_bn = 0;            // force the weight-panel height of mmq_i8 (64 / 128); 0 = pick by grid size
    int fa_splits = 0;         // 0 = auto
    bool fa_wo = false;        // decode: attention as few fat splits whose merge is the wo mat-vec's prologue (no combine launch).  Correct and
                               // tested, OFF by default: a KV trip of the lane-parallel kernel is a ~4 us dependent chain, so 9 splits of 2
                               // trips (12.0 us) + the merging prologue (wo 6.8 -> 10.4 us) lose to 36 one-trip splits + combine (13.6 + 6.8)
    bool small_uploads = true; // set_tensor_async of <= 64 KiB: pinned ring + copy kernel instead of a blit
    bool small_downloads = true; // get_tensor_async of <= 8 MiB into this backend's pinned host buffer type: a copy kernel instead of a blit
    bool timing = false;       // hipEvent-bracket kernel classes (bench only; disables graphs)
    int decode_copy_headroom_gib = 2;  // a buffer's decode copy is made only while this much device memory stays free beside it (tests raise it to force the fallback)
    bool decode_copy = [] { const char * e = getenv("GGML_MI355X_DECODE_COPY"); return e ? atoi(e) != 0 : true; }();  // batch-1 mat-vecs read the plane-layout copy of their weights
    int exec_update = [] { const char * e = getenv("GGML_MI355X_EXEC_UPDATE"); return e ? atoi(e) : 1; }();  // patch the predecessor's executable graph at a capture at
    int shadow_capture = [] { const char * e = getenv("GGML_MI355X_SHADOW_CAPTURE"); return e ? atoi(e) : 1; }();  // a capture at first sighting runs BEHIND the step's own eager launches
    int q80_min_cols = [] { const char * e = getenv("GGML_MI355X_Q80_MIN_COLS"); return e ? atoi(e) : 33; }();  // Q8_0 weights: batches of at least this many columns take the matrix-core GEMM (mmq_q80.hip), smaller ones 8-column mat-vec passes
                               // first sighting (graph.cpp); 2 = run the update and treat it as failed (tests)
};

struct stats {
    int64_t graph_launches = 0, graph_captures = 0, eager_graphs = 0, kernel_launches = 0, fused_nodes = 0, allreduces = 0;
    int64_t p2p_allreduces = 0;        // row-parallel sums served by the one-shot peer-to-peer kernel (tp_p2p.hip) instead of RCCL
    int64_t ss_handoffs = 0;           // RMS_NORM prologues that took the sum of squares from the producing mat-vec's partial sums
    int64_t skinny_launches = 0;       // mat-muls of 2..32 columns served by the weight-streaming matrix-core kernel
    int64_t wide_launches = 0;         // prompt-batch mat-muls served by its wide form
    int64_t tiled_launches = 0;        // batch mat-muls served by the LDS-tiled int8 GEMM (mmq_i8.hip)
    int64_t nf_mma_chains = 0;         // non-flash K.q -> SOFT_MAX -> V^T.p chains of a prompt micro-batch served by the two-pass matrix-core kernel (round 4)
    int64_t fa_list_launches = 0;      // FLASH_ATTN_EXT nodes served over per-token position lists (2..32 tokens; 33..256 when the mask is known to be sparse)
    int64_t rope_epilogues = 0;        // batches whose rope + KV-cache stores rode in the skinny QKV launches
    int64_t graph_launch_host_ns = 0;  // host time spent inside hipGraphLaunch (replays only)
    int64_t graph_key_host_ns = 0;     // replays only: host time from entering graph_compute to calling hipGraphLaunch (recognising the graph)
    int64_t graph_compute_host_ns = 0; // host time inside graph_compute, all paths
    int64_t graph_key_fast_hits = 0;   // replays recognised by comparing against the graph replayed last (no key built, no hash)
    int64_t kv_native_nodes = 0;       // ... and those whose K / V in such a type were read in place by the lane-parallel kernel's DQ form
    int64_t kv_image_nodes = 0;        // FLASH_ATTN_EXT nodes whose K / V (kept in q4_0, q4_1, q5_0, q5_1, iq4_nl, bf16, f32 ...) were read through an f16 image
    int64_t kernel_downloads = 0;      // get_tensor_async calls served by a copy kernel writing mapped pinned memory
    int64_t graph_early_captures = 0;  // graphs captured at their FIRST sighting (same step as the one replayed last, over a grown cache)
    int64_t graph_shadow_captures = 0; // ... of which the capture ran behind the step's own eager launches (the GPU busy meanwhile) and serves the NEXT step
    int64_t graph_capture_walk_ns = 0, graph_exec_update_ns = 0, graph_shadow_eager_ns = 0;  // host time of: the walk into a capture (begin..end), hipGraphExecUpdate / instantiate, a shadow capture's eager walk
    int64_t graph_exec_updates = 0;    // ... of them, served by patching the predecessor's executable graph (hipGraphExecUpdate) instead of instantiating
    int64_t decode_copy_tensors = 0;   // weight matrices repacked into the decode copy by this backend instance ...
    int64_t decode_copy_bytes = 0;     // ... and their bytes
    int64_t decode_copy_launches = 0;  // mat-vec / fused Q/K/V launches that streamed a decode copy
    int64_t elided_conts = 0;          // cont(permute(kqv)) copies of a one-token non-flash attention whose bytes the producing launch wrote in place (round 6)
    int64_t step_heads = 0;            // decode-step heads served by one launch: GET_ROWS + mask cast + rotary table (ops.hip: k_step_head)
    int64_t graph_exec_update_failures = 0; // ... and updates that failed: the predecessor's (possibly half-patched) executable graph is destroyed, both entries start over
    int64_t graph_evictions = 0;       // cache entries dropped because they had not been used for 256 graphs
    int64_t graph_key_collisions = 0;  // two different graph keys with one hash (each keeps its own entry)
};

struct tp_state;      // tp.cpp
struct split_helper;  // split.cpp
struct ip_engine;     // tp_inproc.cpp

struct cached_graph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    int seen = 0;
    uint64_t last_use = 0;
    int64_t allreduces = 0;  // reductions / collectives recorded in the graph (counted again at every replay: stats::allreduces)
    int n_nodes = 0;
    bool early_failed = false;  // a capture at first sighting was tried and failed (graph.cpp)
    std::vector<uint64_t> key;  // the graph this entry stands for, word by word (graph.cpp: walk_key) — an entry is used only when these are equal
};

struct timing_slot {
    double total_ms = 0;
    int64_t count = 0;
};

struct backend_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string name;
    options opt;
    stats st;
    // scratch (quantised activations, attention partials); grown outside capture only
    void * ws = nullptr;
    size_t ws_size = 0;
    // activation-quantisation cache: which src1 tensor the q8 scratch currently holds
    const void * q8_src = nullptr;
    int q8_kind = 0;
    size_t q8_bytes = 0;
    uint64_t q8_epoch = 0;
    // hipGraph cache
    std::unordered_map<uint64_t, cached_graph> graphs;
    cached_graph * last_graph = nullptr;  // the entry looked up last: compared first, in place (graph.cpp: key_equals)
    std::vector<uint64_t> key_scratch;
    uint64_t tick = 0;
    uint64_t decode_epoch = 0;  // decode_copy_epoch() the cached graphs were captured under
    bool capturing = false;
    bool mm_q5_major = false;  // the graph being planned / run holds more Q5_K than Q4_K weight bytes (graph.cpp: mmq_min_cols_for)
    // tensor parallel
    tp_state * tp = nullptr;
    ip_engine * ip = nullptr;  // -sm row served as in-process tensor parallelism (tp_inproc.cpp); owned by the main device's backend
    // -sm row: per owning device a stream / scratch / event for the slices of row-split weights (split.cpp), created on first use
    std::vector<split_helper *> split_helpers;
    hipEvent_t split_ready = nullptr;  // "the activations of this split mat-mul exist on the main stream"
    // small host->device uploads (token ids, positions, cache indices, one mask row): staged in a pinned ring and moved by a
    // tiny kernel — a blit through hipMemcpyAsync costs ~25 us of stream time per copy, five of them per decode step
    char * up_ring = nullptr;
    size_t up_cap = 0, up_head = 0;
    upload_batch up_pending{};  // staged in the ring, not yet launched: flushed as ONE kernel before anything else enters the stream
    // attention over a unified cache with a few query tokens: per-token lists of visible tiles (fattn.hip, k_fattn_tile_scan),
    // built once per graph execution and shared by the attention nodes of all layers
    double * ss_buf = nullptr;       // 256 partial sums of squares: from the mat-vec that writes a residual stream to the norm prologue that reads it (mmvq_args::ss_out)
    unsigned * fa_arrive = nullptr;  // arrival counters of the self-merging attention splits (zero between launches)
    static constexpr int fa_arrive_slots = 16384;
    float * rope_tab = nullptr;   // (cos, sin) per (token, rotation pair) of a small batch: written once per graph run, read by every layer's QKV epilogue
    static constexpr int rope_tab_floats = 4096 * 256;  // (cos, sin) of up to 4096 tokens x 128 pairs (round 4: prompt micro-batches use the table too)
    int * fa_lists = nullptr;
    size_t fa_lists_bytes = 0;
    // per-class kernel timing (bench)
    std::map<std::string, timing_slot> timing;
    std::vector<std::pair<std::string, std::pair<hipEvent_t, hipEvent_t>>> pending_events;
};

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the (kernel, DEVICE) pair: llama-box's usual multi-GPU mode is one process
// driving several MI355X, so "raised once per process" would leave every device but the first at the 64 KB default and its launches
// failing.  `done` is one bit per device ordinal, owned by the launcher of that kernel instantiation.
inline bool ensure_dyn_lds(const void * kernel, size_t bytes, std::atomic<uint32_t> & done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) dev = 0;
    const uint32_t bit = 1u << dev;
    if (done.load(std::memory_order_acquire) & bit) return true;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes) != hipSuccess) {
        (void) hipGetLastError();
        MI_ERR("hipFuncSetAttribute(MaxDynamicSharedMemorySize = %zu) failed on device %d", bytes, dev);
        return false;
    }
    done.fetch_or(bit, std::memory_order_release);
    return true;
}

void flush_uploads(backend_ctx * c);  // launches the small uploads staged by set_tensor_async (backend.cpp)

// ---- buffers (backend.cpp) ----
struct buffer_ctx {
    int device = 0;
    void * base = nullptr;
    size_t size = 0;
    bool rowpar = false;  // tensor-parallel reducing buffer type
    // THE DECODE COPY (round 6; repack.hip, mmvq_types.h): a weights buffer's K-quant matrices a second time, in the plane layout the batch-1 mat-vec kernels stream
    // with non-temporal loads — same size, a tensor's copy sits at the tensor's own offset.  Allocated at the first use by a graph (never for buffers that are not
    // usage WEIGHTS), a tensor's copy written then too; anything that writes into the buffer afterwards drops the copies it overlaps (and every captured graph).
    void * shadow = nullptr;
    bool shadow_failed = false;
    std::mutex sh_mtx;
    std::unordered_map<size_t, size_t> sh_valid;  // offset of a repacked tensor -> its bytes
};
// the plane-layout copy of weight matrix w for the batch-1 mat-vec kernels, or nullptr (no copy: views, split / row-parallel buffers, option off,
// allocation failed, or asked during a capture for a tensor not yet repacked)
const uint8_t * decode_copy(backend_ctx * c, const ggml_tensor * w);
const uint8_t * q80_panel_copy(backend_ctx * c, const ggml_tensor * w);  // a Q8_0 matrix's panel copy (mmq_q80.hip: the 9 .. 32-column kernel), or nullptr
uint64_t decode_copy_epoch();  // bumped whenever a copy was dropped: graphs captured before hold pointers to it
bool buffer_is_ours(ggml_backend_buffer_t b);
bool buffer_is_rowpar(ggml_backend_buffer_t b);
ggml_backend_buffer_t make_backend_buffer(ggml_backend_buffer_type_t buft, const ggml_backend_buffer_i & iface, void * context, size_t size);
int logical_device_count();                 // devices of the registration (GGML_MI355X_FAKE_DEVICES multiplies them for tests)
int logical_device_ordinal(int i);          // HIP ordinal behind logical device i
ggml_backend_dev_t logical_device(int i);

// ---- row-split weight buffers, -sm row (split.cpp) ----
struct split_tensor_info {
    int n_dev;
    int kind;                                   // 0: cut by ROWS (output features: wq / wk / wv / ffn_gate / ffn_up / output — "column-parallel"),
                                                // 1: cut along K (input features, whole 256-value super-blocks: attn_output / ffn_down — "row-parallel":
                                                //    every device holds all rows of its K range and produces a partial sum)
    int64_t row0[GGML_MI355X_MAX_DEVICES + 1];  // kind 0: device d owns rows [row0[d], row0[d + 1]); kind 1: VALUES [row0[d], row0[d + 1]) of every row
    void * slice[GGML_MI355X_MAX_DEVICES];      // its part, GGUF bytes verbatim (kind 1: rows of (row0[d+1] - row0[d]) / blck blocks), in device d's memory
    size_t row_bytes;                           // bytes of a FULL row
    size_t slice_row_bytes[GGML_MI355X_MAX_DEVICES];  // bytes of a row of device d's slice (kind 0: = row_bytes)
};
void split_rows(int64_t nrows, const float * tensor_split, int n_dev, int64_t granule, int64_t * row0);
ggml_backend_buffer_type_t split_buffer_type(int main_device, const float * tensor_split);
bool buft_is_split(ggml_backend_buffer_type_t buft);
int split_buft_main_device(ggml_backend_buffer_type_t buft);
bool buffer_is_split(ggml_backend_buffer_t b);
const split_tensor_info * split_info(const ggml_tensor * t);
bool split_mul_mat_supported(const ggml_tensor * op);
bool run_split_mul_mat(backend_ctx * c, const ggml_tensor * w, const ggml_tensor * b, ggml_tensor * dst);
// row-parallel weight (kind 1): scatter b's K ranges, per-device partial products, in-stream sum on the main device (+ optional addend)
bool run_split_rowpar(backend_ctx * c, const ggml_tensor * w, const ggml_tensor * b, ggml_tensor * dst, const ggml_tensor * add);
// the whole FFN on sharded weights: broadcast x once, gate / up / SwiGLU / down on every device's own rows of the hidden layer, ONE sum
bool split_ffn_applies(const ggml_tensor * wg, const ggml_tensor * wu, const ggml_tensor * wd);
bool run_split_ffn(backend_ctx * c, const ggml_tensor * wg, const ggml_tensor * wu, const ggml_tensor * wd, const ggml_tensor * x, ggml_tensor * dst, const ggml_tensor * add);
void free_split_helpers(backend_ctx * c);

// ---- graph execution (graph.cpp) ----
bool supports_op(const ggml_tensor * op);
enum ggml_status graph_compute(backend_ctx * ctx, ggml_cgraph * g);
void free_graph_cache(backend_ctx * ctx);
bool graph_key_equals(const ggml_cgraph * g, const std::vector<uint64_t> & key);  // the graph-key words (graph.cpp: walk_key), compared in place ...
void graph_key_build(const ggml_cgraph * g, std::vector<uint64_t> & key);         // ... or materialised

// ---- in-process tensor parallel over the devices of a row-split model, -sm row (tp_inproc.cpp) ----
struct ip_engine;
// handled = false: the graph is not one the engine takes (no split weights, or a shape it declines): the caller goes on as before
enum ggml_status ip_graph_compute(backend_ctx * ctx, ggml_cgraph * g, bool * handled);
void ip_free(backend_ctx * ctx);
int64_t ip_stat(const backend_ctx * ctx, const char * key);  // -1: unknown key
// the host is about to read / has written memory of `b` behind the engine's back (get_tensor, set_tensor, copies, a graph the engine does not take):
// the per-device KV shards are gathered into / re-scattered from the host's cache tensors lazily
void ip_host_access(ggml_backend_buffer_t b, bool write);
void ip_host_buffer_freed(ggml_backend_buffer_t b);  // the host frees a buffer the engine mirrors
bool ip_any();

// ---- internal buffer objects for tensors the backend creates itself (backend.cpp) ----
struct buffer_ctx;
void make_internal_buffer(ggml_backend_buffer * out, buffer_ctx * bc, int device_ordinal, bool rowpar);
ggml_backend_t internal_backend(int logical_device);  // a second backend instance on a device of the registration (its own stream and scratch)

// ---- host-side shadow of what the engine uploads as attention masks (backend.cpp) ----
// llama.cpp fills the KQ mask on the host and hands it over with set_tensor(_async) before every graph: while the bytes pass through, the
// backend notes how many cells the batch's tokens can see.  graph_compute reads that to choose kernels whose cost depends on the mask's
// CONTENT (position lists against the dense matrix-core kernel for 33+ tokens; the list form of the non-flash chain only when no token's
// list exceeds its capacity) — decisions a captured graph cannot take from device memory.  Unknown mask (never uploaded through this
// backend, or too large to be worth scanning): callers keep their shape-only rules.
struct mask_stats {
    int rows = 0;          // rows with at least one visible cell
    int max_visible = 0;   // the longest row
    float density = 1.0f;  // visible cells / (rows x row length)
};
void note_mask_upload(const ggml_tensor * t, const void * host, size_t offset, size_t size);
bool lookup_mask_stats(const void * dev_ptr, mask_stats * out);
void forget_mask_stats(const void * dev_ptr, size_t nbytes);  // a device-side write landed there
void copy_mask_stats(const ggml_tensor * src, const ggml_tensor * dst, size_t nbytes);  // ... unless it was a whole-tensor copy of a tensor on record: the record follows the bytes
// 1: the mask at dev_ptr is known and sparse enough (<= a quarter visible) that per-token position lists beat a dense tile kernel
int mask_sparse_hint(const void * dev_ptr);

// ---- tensor parallel (tp.cpp) ----
int tp_init(backend_ctx * ctx, int rank, int world, const void * uid, size_t uid_size);
int tp_get_unique_id(void * out, size_t size);
bool tp_active(const backend_ctx * ctx);
int tp_p2p_export(backend_ctx * ctx, int rank, int world, void * handle_out, size_t size);  // -> this rank's mailbox as a hipIpcMemHandle_t (64 bytes)
int tp_p2p_attach(backend_ctx * ctx, const void * handles, size_t size);                    // world handles in rank order, own slot ignored
int64_t tp_p2p_timeouts(backend_ctx * ctx);
int tp_attach_local(backend_ctx * const * ctxs, int n);  // the devices of this process as one group (ranks = array order); 0 or < 0
bool tp_p2p_enable(backend_ctx * ctx, bool on);  // false: refused (switching the only transport of a group off)
bool tp_p2p_reset(backend_ctx * ctx);           // forget a time-out (every rank, all idle)
bool tp_check(backend_ctx * ctx);               // false: an all-reduce of an earlier graph timed out — the caller fails its graph_compute
// in-stream sum all-reduce of n floats at ptr (capturable)
bool tp_all_reduce(backend_ctx * ctx, float * ptr, size_t n);
// the same with the residual ADD that follows folded in: out[i] = sum_ranks(ptr[i]) + add[i] (add: n values), and the sum of squares of `out`
// left as partial sums (ss_out, *ss_n of them) — served by ONE peer-to-peer launch; returns false (nothing done) when that form does not apply
bool tp_all_reduce_fused(backend_ctx * ctx, float * ptr, size_t n, const float * add, float * out, double * ss_out, int * ss_n);
void tp_free(backend_ctx * ctx);

}  // namespace mi355x
