// graph.cpp — ggml_backend_i::graph_compute for the MI355X backend: node dispatch, chain fusion, activation-
// quantisation reuse, scratch sizing and hipGraph capture/replay.
//
// Entry: ggml_backend_sched_graph_compute -> backend.graph_compute (direct use at
// /root/reference/llama-box/rpcserver.hpp:1390).  llama-box's decode loop submits the SAME graph topology every
// step (one llama_decode per engine iteration, llama-box/httpserver.hpp:3591), ~290 nodes of 3-90 us each, which is
// host-launch-bound if issued eagerly (MI355X_MICROARCH.md "graph-replay-floor").  So: a graph whose fingerprint
// (ops, shapes, strides, addresses, params) repeats is captured once into a hipGraph and replayed; the stock
// backend's CUDA-graph path (GGML_CUDA_GRAPHS, /root/reference/CMakeLists.txt:56-58) plays the same role there.
// Errors inside compute return GGML_STATUS_FAILED (-> llama_decode rc -2, llama-box/httpserver.hpp:3541-3545).
#include <algorithm>
#include <chrono>
#include <cmath>

#include "kernels.h"

namespace mi355x {

static bool is_quant(int t) { return t == GGML_TYPE_Q4_K || t == GGML_TYPE_Q5_K || t == GGML_TYPE_Q6_K || t == GGML_TYPE_Q8_0; }
// batches at least this wide run on the matrix cores.  The option's default (3) is per weight type since round 6: Q4_K / Q6_K matrices take the weight-streaming
// kernel from TWO columns on (a 2-sequence step of Llama-3-8B Q4_K_M: 2.89-3.03 -> 2.78 ms — the multi-column mat-vec path has none of the batch path's fusions,
// 17 launches a layer), Q5_K stays at 3 (its skinny unit is slower: Qwen2-7B Q5_K_M 3.13 -> 3.79 ms with 2; profiles/r06_np2_min_cols.txt)
// (Q6_K follows the model: with 2 where Q4_K carries the graph's weight bytes — Q4_K_M; 3 in a Q5_K_M model, whose Q6_K matrices measured slower on the unit at 2
// columns: Qwen2 3.13 -> 3.30 ms when only they moved.  plan_ws counts the bytes.)
static inline int mmq_min_cols_for(const backend_ctx * c, int wtype) {
    if (c->opt.mmq_min_cols != 3) return c->opt.mmq_min_cols;
    return (wtype == GGML_TYPE_Q4_K || (wtype == GGML_TYPE_Q6_K && !c->mm_q5_major)) ? 2 : 3;
}
static int act_kind(int wtype) { return wtype == GGML_TYPE_Q8_0 ? GGML_TYPE_Q8_0 : GGML_TYPE_Q8_K; }
static bool is_f32_contig(const ggml_tensor * t) { return t->type == GGML_TYPE_F32 && ggml_abi_is_contiguous(t); }
static bool rows_contig(const ggml_tensor * t) { return t->nb[0] == ggml_abi_type_size(t->type); }
static bool same_shape(const ggml_tensor * a, const ggml_tensor * b) {
    for (int i = 0; i < 4; ++i) if (a->ne[i] != b->ne[i]) return false;
    return true;
}

// ------------------------------------------------------------------------------------------------ supports_op
// MUL_MAT whose src0 is a KV-cache view kept in a block format or bf16 (K.q of the non-flash path with -ctk q8_0 / q4_0 / ...: llama.cpp asks flash
// attention only of a quantised V): the view is expanded to f16 (kv_types.hip) and the f16 product runs on the image.  Weight matrices of the
// mat-vec formats (2-D Q8_0 ...) keep their own kernels.
static bool mm_cache_image_ok(const ggml_tensor * op) {
    const ggml_tensor * a = op->src[0];
    const ggml_tensor * b = op->src[1];
    if (!a || !b || b->type != GGML_TYPE_F32 || op->type != GGML_TYPE_F32 || !ggml_abi_is_contiguous(op)) return false;
    if (!(kv_type_is_block(a->type) || a->type == GGML_TYPE_BF16 || a->type == GGML_TYPE_Q8_0)) return false;
    // a VIEW into a cache (or a batch of matrices), never a plain 2-D weight: llama.cpp's loader probes every weight type with a plain tensor, and a model stored
    // in q4_0 / q5_1 / bf16 ... must get the same answer there as at graph time — its mat-muls stay where they were (the CPU backend), they do not take an
    // image of a whole weight matrix per step
    if (a->view_src == nullptr && a->ne[2] == 1) return false;
    // ... nor a view of one: a 2-D view of a Q8_0 weight keeps the quantised mat-vec / mat-mul kernels (the integer-dot arithmetic of the reference), and
    // weights in the other block formats stay where they were
    const ggml_backend_buffer_t root = a->view_src && a->view_src->buffer ? a->view_src->buffer : a->buffer;
    if (root && root->usage == GGML_BACKEND_BUFFER_USAGE_WEIGHTS) return false;
    if (a->nb[0] != ggml_abi_type_size(a->type) || (a->ne[0] % 32) != 0 || (a->nb[1] % 2) || (a->nb[2] % 2) || a->ne[3] != 1 || b->ne[3] != 1 || b->nb[0] != 4) return false;
    return a->ne[2] > 0 && b->ne[2] % a->ne[2] == 0;
}

// which attention path serves this FLASH_ATTN_EXT node: 0 none (the host keeps it on the CPU backend), 1 the f16 kernels on the cache views in place,
// 2 the lane-parallel kernel on block_q8_0 K / V, 3 the f16 kernels on an f16 IMAGE of whichever of K / V is kept in another type (kv_types.hip:
// -ctk / -ctv q4_0, q4_1, q5_0, q5_1, iq4_nl, bf16, f32, mixed pairs, and q8_0 shapes route 2 does not take)
static int fa_route(const ggml_tensor * op) {
    const ggml_tensor * a = op->src[0];
    const ggml_tensor * k = op->src[1];
    const ggml_tensor * v = op->src[2];
    const ggml_tensor * m = op->src[3];
    if (!a || !k || !v || a->type != GGML_TYPE_F32) return 0;
    if (k->type == GGML_TYPE_Q8_0 && v->type == GGML_TYPE_Q8_0) {
        // quantised KV cache: served by the lane-parallel kernel (head_dim 128, 1 .. 8 query heads per KV head, no soft-capping / ALiBi)
        const int64_t g = k->ne[2] ? a->ne[2] / k->ne[2] : 0;
        const bool shape = k->ne[0] == 128 && v->ne[0] == 128 && a->nb[0] == 4 && !(a->nb[1] % 16) && !(a->nb[2] % 16) && a->ne[2] % k->ne[2] == 0 && k->ne[2] == v->ne[2] &&
                           g >= 1 && g <= 8 && ggml_abi_op_param_f32(op, 1) == 0.0f && ggml_abi_op_param_f32(op, 2) == 0.0f &&
                           !(k->nb[1] % 2) && !(v->nb[1] % 2) && !(k->nb[2] % 2) && !(v->nb[2] % 2);
        if (shape) return (!m || (m->type == GGML_TYPE_F16 && m->ne[2] == 1 && rows_contig(m))) ? 2 : 0;
    }
    const bool k16 = k->type == GGML_TYPE_F16, v16 = v->type == GGML_TYPE_F16;
    if (!(k16 || kv_image_type(k->type)) || !(v16 || kv_image_type(v->type))) return 0;
    if (k->ne[0] != v->ne[0] || (k->ne[0] != 64 && k->ne[0] != 128)) return 0;
    if (a->nb[0] != 4) return 0;
    for (const ggml_tensor * t : {k, v}) {
        if (t->type == GGML_TYPE_F16) {
            if (t->nb[0] != 2 || (t->nb[1] % 16) || (t->nb[2] % 16)) return 0;
        } else {  // read block by block into the image: whole blocks per head row, element-aligned strides, one cache stream
            const size_t al = t->type == GGML_TYPE_F32 ? 4 : 2;
            if (t->nb[0] != ggml_abi_type_size(t->type) || (t->nb[1] % al) || (t->nb[2] % al) || t->ne[3] != 1) return 0;
        }
    }
    if (m && (m->type != GGML_TYPE_F16 || m->ne[2] != 1 || !rows_contig(m))) return 0;
    if (k->ne[2] <= 0 || a->ne[2] % k->ne[2] != 0 || k->ne[2] != v->ne[2]) return 0;
    const int64_t g = a->ne[2] / k->ne[2];
    if (g < 1 || g > 8) return 0;  // (1 .. 8 query heads per KV head: MHA, Llama-3.2-3B's 3, Qwen2.5's 6 / 7, Llama-3's 4 / 8; round 6 — until then 3, 5, 6 stayed on the CPU backend)
    return (k16 && v16) ? 1 : 3;
}

bool supports_op(const ggml_tensor * op) {
    const ggml_tensor * a = op->src[0];
    const ggml_tensor * b = op->src[1];
    // A tensor in a row-split buffer (-sm row) has no dereferenceable ->data (split.cpp: its bytes are per-device slices behind a map), so
    // the one op that may read it is a MUL_MAT taking it as src0.  llama.cpp's loader probes EVERY weight against the split buffer type
    // first (weight_buft_supported builds MUL / ADD / ROPE / GET_ROWS probes with w->buffer set to a dummy split buffer): answering true
    // there would put norm weights, biases, rope_freqs and token_embd behind the fake base address (as ggml-cuda refuses them).
    for (int s = 0; s < GGML_MAX_SRC; ++s)
        if (op->src[s] && buffer_is_split(op->src[s]->buffer) && !(op->op == GGML_OP_MUL_MAT && s == 0)) return false;
    if (buffer_is_split(op->buffer)) return false;
    switch (op->op) {
        case GGML_OP_NONE: case GGML_OP_VIEW: case GGML_OP_RESHAPE: case GGML_OP_PERMUTE: case GGML_OP_TRANSPOSE:
            return true;
        case GGML_OP_MUL_MAT: {
            if (!a || !b || b->type != GGML_TYPE_F32 || op->type != GGML_TYPE_F32 || !ggml_abi_is_contiguous(op)) return false;
            if (buffer_is_split(a->buffer)) return split_mul_mat_supported(op);  // -sm row weights: every device computes its rows (split.cpp)
            if (mm_cache_image_ok(op)) return true;
            if (is_quant(a->type)) {
                return a->ne[2] == 1 && a->ne[3] == 1 && rows_contig(a) && b->nb[0] == 4 && a->ne[0] % ggml_abi_blck_size(a->type) == 0;
            }
            if (a->type == GGML_TYPE_F16 || a->type == GGML_TYPE_F32) return true;
            return false;
        }
        case GGML_OP_ADD: case GGML_OP_SUB: case GGML_OP_MUL: case GGML_OP_DIV:
            return a->type == GGML_TYPE_F32 && b->type == GGML_TYPE_F32 && op->type == GGML_TYPE_F32;
        case GGML_OP_SCALE:
            return is_f32_contig(a) && is_f32_contig(op);
        case GGML_OP_RMS_NORM:
            return a->type == GGML_TYPE_F32 && a->nb[0] == 4 && op->nb[0] == 4;
        case GGML_OP_UNARY: {
            const int u = op->op_params[0];
            const bool ok = u == GGML_UNARY_OP_SILU || u == GGML_UNARY_OP_RELU || u == GGML_UNARY_OP_NEG || u == GGML_UNARY_OP_EXP ||
                            u == GGML_UNARY_OP_TANH || u == GGML_UNARY_OP_SIGMOID;
            return ok && is_f32_contig(a) && is_f32_contig(op);
        }
        case GGML_OP_GLU:
            return op->op_params[0] == GGML_GLU_OP_SWIGLU && a->type == GGML_TYPE_F32 && a->nb[0] == 4 && (!b || (b->type == GGML_TYPE_F32 && b->nb[0] == 4)) &&
                   a->ne[2] == 1 && a->ne[3] == 1 && op->nb[0] == 4;
        case GGML_OP_CPY: case GGML_OP_DUP: case GGML_OP_CONT: {
            const int st = a->type, dt = op->type;
            if (st == GGML_TYPE_I32 && dt == GGML_TYPE_I32) return true;
            if ((st == GGML_TYPE_Q8_0 && dt == GGML_TYPE_F32) || (st == GGML_TYPE_F32 && dt == GGML_TYPE_Q8_0))  // K-shift of a quantised cache
                return ggml_abi_is_contiguous(a) && ggml_abi_is_contiguous(op) && (a->ne[0] % 32) == 0;
            if ((kv_store_type(st) && dt == GGML_TYPE_F32) || (st == GGML_TYPE_F32 && kv_store_type(dt)))  // ... of a cache in one of the other types (kv_types.hip)
                return ggml_abi_is_contiguous(a) && ggml_abi_is_contiguous(op) && (a->ne[0] % 32) == 0 && (((uintptr_t) a->data | (uintptr_t) op->data) & 15) == 0;
            return (st == GGML_TYPE_F32 || st == GGML_TYPE_F16) && (dt == GGML_TYPE_F32 || dt == GGML_TYPE_F16);
        }
        case GGML_OP_GET_ROWS:
            return b->type == GGML_TYPE_I32 && op->type == GGML_TYPE_F32 && rows_contig(a) && op->nb[0] == 4 &&
                   (is_quant(a->type) || a->type == GGML_TYPE_F16 || a->type == GGML_TYPE_F32);
        case GGML_OP_SET_ROWS:
            if (kv_store_type(op->type))  // a cache row in q4_0 / q4_1 / q5_0 / q5_1 / iq4_nl / bf16: the type's from_float per block of 32 (kv_types.hip)
                return a->type == GGML_TYPE_F32 && b->type == GGML_TYPE_I64 && a->nb[0] == 4 && (a->ne[0] % 32) == 0 && (a->nb[1] % 16) == 0 && (a->nb[2] % 16) == 0 && (a->nb[3] % 16) == 0 &&
                       op->nb[0] == ggml_abi_type_size(op->type);
            if (op->type == GGML_TYPE_Q8_0) return a->type == GGML_TYPE_F32 && b->type == GGML_TYPE_I64 && a->nb[0] == 4 && (a->ne[0] % 32) == 0 && (a->nb[1] % 16) == 0 && (a->nb[2] % 16) == 0 && (a->nb[3] % 16) == 0;
            return a->type == GGML_TYPE_F32 && b->type == GGML_TYPE_I64 && a->nb[0] == 4 && (op->type == GGML_TYPE_F16 || op->type == GGML_TYPE_F32) &&
                   op->nb[0] == ggml_abi_type_size(op->type);
        case GGML_OP_SOFT_MAX:
            return is_f32_contig(a) && is_f32_contig(op) && (!b || ((b->type == GGML_TYPE_F16 || b->type == GGML_TYPE_F32) && rows_contig(b)));
        case GGML_OP_ROPE: {
            const int mode = op->op_params[2];
            if (mode & GGML_ROPE_TYPE_MROPE) {
                // ggml_rope_multi (Qwen2-VL sections, vision towers).  llama-box removes upstream's "some section > 0" assertion
                // (mrope.patch:5-26); with all sections zero the reference loop divides by zero, so that case is refused
                const int32_t * sec = op->op_params + 11;
                const int64_t sect_dims = (int64_t) sec[0] + sec[1] + sec[2] + sec[3];
                if (sec[0] < 0 || sec[1] < 0 || sec[2] < 0 || sec[3] < 0 || sect_dims <= 0 || sect_dims > a->ne[0]) return false;
                if (!ggml_abi_is_contiguous(b) || ggml_abi_nelements(b) < 4 * a->ne[2]) return false;
                if (mode == GGML_ROPE_TYPE_VISION && op->op_params[1] != a->ne[0] / 2) return false;
                if (mode != GGML_ROPE_TYPE_VISION && op->op_params[1] > a->ne[0]) return false;
            }
            return (a->type == GGML_TYPE_F32 || a->type == GGML_TYPE_F16) && a->type == op->type && rows_contig(a) && rows_contig(op) && b->type == GGML_TYPE_I32;
        }
        case GGML_OP_FLASH_ATTN_EXT:
            return fa_route(op) != 0;
        case GGML_OP_ARGMAX:
            return a->type == GGML_TYPE_F32 && a->nb[0] == 4;
        default:
            return false;
    }
}

// ------------------------------------------------------------------------------------------------ scratch
static bool ensure_ws(backend_ctx * c, size_t need) {
    if (need <= c->ws_size) return true;
    if (c->capturing) return false;
    HIP_SOFT(hipStreamSynchronize(c->stream));
    if (c->ws) HIP_NOTE(hipFree(c->ws));
    c->ws = nullptr;
    c->ws_size = 0;
    const size_t sz = (need + (size_t) (8u << 20)) & ~(size_t) 255;
    if (hipMalloc(&c->ws, sz) != hipSuccess) {
        (void) hipGetLastError();
        MI_ERR("failed to allocate %.1f MiB of scratch", sz / 1048576.0);
        return false;
    }
    c->ws_size = sz;
    c->q8_src = nullptr;
    free_graph_cache(c);  // captured graphs hold the old scratch address
    return true;
}

struct ws_plan {
    size_t act_bytes = 0;  // region A: quantised activations
    size_t aux_bytes = 0;  // region B: attention partials
};

static size_t fa_image_offset(const tdesc & q, const tdesc & k, const tdesc & v) { return (fattn_workspace_bytes(q, k, v, 64, GGML_TYPE_F16) + 255) & ~(size_t) 255; }
static ws_plan plan_ws(backend_ctx * c, const ggml_cgraph * g) {
    ws_plan p;
    {   // which K-quant carries this graph's weight bytes (mmq_min_cols_for)
        size_t q4 = 0, q5 = 0;
        for (int i = 0; i < g->n_nodes; ++i) {
            const ggml_tensor * n = g->nodes[i];
            if (n->op != GGML_OP_MUL_MAT || !n->src[0]) continue;
            if (n->src[0]->type == GGML_TYPE_Q4_K) q4 += ggml_abi_nbytes(n->src[0]);
            else if (n->src[0]->type == GGML_TYPE_Q5_K) q5 += ggml_abi_nbytes(n->src[0]);
        }
        c->mm_q5_major = q5 > q4;
    }
    for (int i = 0; i < g->n_nodes; ++i) {
        const ggml_tensor * n = g->nodes[i];
        if (n->op == GGML_OP_MUL_MAT && is_quant(n->src[0]->type)) {
            const ggml_tensor * b = n->src[1];
            const int64_t Mc = b->ne[1] * b->ne[2] * b->ne[3];
            // (2..32 columns: the skinny matrix-core kernel fetches 32 columns' worth of activation bytes whatever M is)
            // (and the wide form of the same unit fetches whole groups of 128 columns of a prompt batch)
            p.act_bytes = std::max(p.act_bytes, quantized_act_bytes(act_kind(n->src[0]->type), b->ne[0], (Mc >= 2 && Mc < 32) ? 32 : (Mc >= 33 ? (Mc + 127) / 128 * 128 : Mc)));
            if (Mc >= mmq_min_cols_for(c, n->src[0]->type) && (n->ne[0] % 4) == 0) p.aux_bytes = std::max(p.aux_bytes, 3 * mmq_workspace_bytes(n->src[0]->type, b->ne[0], n->src[0]->ne[1], Mc, c->opt.mmq_skinny));  // (x3: up to three sibling matrices share a launch)
        } else if (n->op == GGML_OP_MUL_MAT && mm_cache_image_ok(n) && !buffer_is_split(n->src[0]->buffer)) {
            const tdesc a16 = kv_image_desc(TD(n->src[0]), nullptr);
            p.aux_bytes = std::max(p.aux_bytes, ((mul_mat_f_workspace_bytes(a16, TD(n->src[1])) + 255) & ~(size_t) 255) + kv_image_bytes(a16));
        } else if (n->op == GGML_OP_MUL_MAT && n->src[0]->type == GGML_TYPE_F16) {
            p.aux_bytes = std::max(p.aux_bytes, mul_mat_f_workspace_bytes(TD(n->src[0]), TD(n->src[1])));
            if (c->opt.attn_nf) p.aux_bytes = std::max(p.aux_bytes, attn_nf_list_scratch_bytes(TD(n->src[1]), TD(n->src[0]), nullptr));
            // (K.q of a prompt micro-batch on the non-flash path: statistics + partial records of the two-pass matrix-core form)
            if (c->opt.attn_nf && n->src[1]->type == GGML_TYPE_F32 && n->src[1]->ne[1] >= fattn_mma_min_q() && n->src[0]->ne[0] == 128 && n->src[1]->ne[3] == 1)
                p.aux_bytes = std::max(p.aux_bytes, attn_nf_mma_ws_bytes(TD(n->src[1]), fattn_mma_pick_splits(TD(n->src[1]), TD(n->src[0]))));
        } else if (n->op == GGML_OP_FLASH_ATTN_EXT) {
            const tdesc q = TD(n->src[0]), k = TD(n->src[1]), v = TD(n->src[2]);
            // (both forms a 33+-token batch may take — matrix-core tiles or, for a mask known to be sparse, position lists — fit this)
            const int ns = std::min(64, c->opt.fa_splits > 0 ? c->opt.fa_splits : std::max(fattn_pick_splits(q, k), 16));
            if (fa_route(n) == 3) {  // the f16 images of K / V behind the largest partial-record area the f16 kernels may ask for
                p.aux_bytes = std::max(p.aux_bytes, fa_image_offset(q, k, v) + (k.type != GGML_TYPE_F16 ? kv_image_bytes(k) : 0) + (v.type != GGML_TYPE_F16 ? kv_image_bytes(v) : 0));
                continue;
            }
            p.aux_bytes = std::max(p.aux_bytes, fattn_workspace_bytes(q, k, v, ns, n->src[1]->type));
        }
    }
    p.act_bytes = (p.act_bytes + 255) & ~(size_t) 255;
    return p;
}

// ------------------------------------------------------------------------------------------------ attention-mask content hints
// the uploaded tensor behind an attention mask: llama.cpp hands FLASH_ATTN_EXT a ggml_cast(F16) of the F32 KQ_mask it fills on the host
// (a CPY node whose src[0] is the input), possibly through views — the statistics of backend.cpp are keyed by the address that went
// through set_tensor
static const void * mask_upload_ptr(const ggml_tensor * m) {
    for (int hop = 0; m && hop < 4; ++hop) {
        const bool whole_view = (m->op == GGML_OP_VIEW || m->op == GGML_OP_RESHAPE) && m->src[0] && m->view_offs == 0 && ggml_abi_nelements(m) == ggml_abi_nelements(m->src[0]);
        if (!((m->op == GGML_OP_CPY || m->op == GGML_OP_DUP || m->op == GGML_OP_CONT || whole_view) && m->src[0])) break;
        m = m->src[0];
    }
    return m ? m->data : nullptr;
}
static int fa_mask_hint(const ggml_tensor * n) {  // kernel choices that follow the CONTENT of the attention mask (common.h: mask_stats)
    if (n->op == GGML_OP_FLASH_ATTN_EXT && n->src[3]) return n->src[0]->ne[1] >= 33 ? mask_sparse_hint(mask_upload_ptr(n->src[3])) : 0;
    if (n->op == GGML_OP_SOFT_MAX && n->src[1] && n->src[0]->ne[1] >= 2 && n->src[0]->ne[1] <= 32) {
        mask_stats ms;
        return lookup_mask_stats(mask_upload_ptr(n->src[1]), &ms) ? (ms.max_visible > 1024 ? 2 : 1) : 0;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------ timing (bench)
struct timed_scope {
    backend_ctx * c;
    std::string cls;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool probe;
    // probe = true: the class is ONE streaming mat-vec launch whose launcher can time the kernel itself (launch_probe)
    timed_scope(backend_ctx * c_, const char * cls_, double bytes, bool probe_ = false) : c(c_), cls(cls_), probe(probe_) {
        if (!c->opt.timing || c->capturing) return;
        HIP_NOTE(hipEventCreate(&e0));
        HIP_NOTE(hipEventCreate(&e1));
        if (probe) {
            g_launch_probe.e0 = e0;
            g_launch_probe.e1 = e1;
            g_launch_probe.armed = true;
            g_launch_probe.used = false;
        } else {
            HIP_SOFT(hipEventRecord(e0, c->stream));
        }
        c->timing[cls].total_ms += 0;  // create slot
        c->timing["bytes:" + cls].total_ms += bytes;
    }
    ~timed_scope() {
        if (!e0) return;
        if (probe) {
            const bool used = g_launch_probe.used;
            g_launch_probe = launch_probe();
            if (!used) {  // the launcher took a path without the probe: nothing was recorded
                HIP_NOTE(hipEventDestroy(e0));
                HIP_NOTE(hipEventDestroy(e1));
                return;
            }
        } else {
            HIP_SOFT(hipEventRecord(e1, c->stream));
        }
        c->pending_events.push_back({cls, {e0, e1}});
    }
};

// ------------------------------------------------------------------------------------------------ node execution
struct deferred_norm {
    const ggml_tensor * x;  // RMS_NORM input (residual stream)
    const ggml_tensor * w;  // norm weight
    float eps;
    const ggml_tensor * out;  // the MUL node: its memory is still written (by workgroup 0 of every consumer launch), see use_count()
};
struct exec_state {
    backend_ctx * c;
    ggml_cgraph * g;
    std::unordered_map<const ggml_tensor *, int> uses;
    mutable std::unordered_map<const ggml_tensor *, int> uses_global;  // memo of graph_use_count()
    size_t act_off = 0, aux_off = 0;
    // [RMS_NORM -> MUL] pairs whose result is consumed only by single-column K-quant mat-vecs: nothing is launched
    // for them; each consumer recomputes norm*w in its prologue (mmvq.hip PRO=2)
    std::unordered_map<const ggml_tensor *, deferred_norm> deferred;
    std::vector<char> done;  // nodes already executed out of order by a multi-chain fusion
    std::vector<int> ooo;    // ... those flagged during the current run_node call (quantised-activation cache invalidation)
    bool q8_fresh = false;   // the node just executed produced the quantised-activation cache for its own output
    // tile lists (fattn.hip) valid for this mask tensor / tile size during this execution of the graph
    const void * fa_list_mask = nullptr;
    int fa_list_tile = 0;
    // decode attention left as fat-split partial records for the prologue of the mat-vec that reads it (wo): fattn.hip / mmvq.hip PRO 3
    struct { const ggml_tensor * b = nullptr; const ggml_tensor * fa = nullptr; const float * part = nullptr; int splits = 0; } fa_wo;
    // a split-K mat-mul whose partial products are summed by the norm + quantise kernel that reads it next (no reduce pass)
    const ggml_tensor * sk_dst = nullptr;
    splitk_src sk{};
    const ggml_tensor * epi_dst = nullptr;  // run_mul_mat_q: the skinny launch that produces this tensor carries `epi` (a sibling of another weight format)
    mmq_epi epi{};
    // the (cos, sin) table of the QKV epilogues (ops.hip: k_rope_table) is valid for these positions / parameters during this execution of the graph
    const void * rope_tab_pos = nullptr;
    const void * rope_tab_ff = nullptr;
    rope_params rope_tab_p{};
    int rope_tab_m = 0;
    int sk_next = -1;  // node index at which run_mul_mat_q may look ahead (set by the caller: first node after the consumed ones)
    // the tensor whose sum of squares sits in c->ss_buf as ss_n partial sums (written by the mat-vec launch that produced the tensor: mmvq_args::ss_out),
    // for the RMS_NORM prologue that reads it next; reset when any other node writes into its memory
    const ggml_tensor * ss_tensor = nullptr;
    int ss_n = 0;
    // merged Q/K/V projections whose split-K partial products are summed by the rope + cache-store kernel at node `node`
    struct { int node = -1, n = 0, ks = 0, M = 0; mmq_mat_desc mats[3]; const ggml_tensor * dst[3]; const float * part = nullptr; } rs_sk;
};

// Number of readers of t.  A backend is handed ONE split of the host's graph (ggml_backend_sched: ggml_graph_view of the full
// graph), so counting the nodes it was given is not enough to call a tensor dead: the reader may sit in another backend's
// split.  Since the fusion helpers (ggml_can_fuse) the cgraph carries use_counts for the WHOLE graph, indexed by the slot of
// visited_hash_set, and views share the parent's table — that count wins when it is larger.  A host that provides no table
// (use_counts == NULL) gets local counts only; the fusions that skip a write then rely on GGML_TENSOR_FLAG_OUTPUT alone, which is
// why the one elision llama.cpp is known to read behind (result_norm = t_embd) is never an elision here: a deferred
// RMS_NORM * w is still written, by workgroup 0 of its consumer (deferred_norm::out).
static int graph_use_count(const ggml_cgraph * g, const ggml_tensor * t) {
    const ggml_hash_set & hs = g->visited_hash_set;
    if (!g->use_counts || !hs.size || !hs.used || !hs.keys) return -1;
    const size_t h = ((size_t) (uintptr_t) t >> 4) % hs.size;  // ggml_hash / ggml_hash_find
    size_t i = h;
    while (hs.used[i >> 5] & (1u << (i & 31))) {
        if (hs.keys[i] == t) return g->use_counts[i];
        i = (i + 1) % hs.size;
        if (i == h) break;
    }
    return -1;
}
// (cos, sin) per (token, rotation pair) of the batch at `pos` (ops.hip: k_rope_table), valid for this execution of the graph: every
// layer rotates with the same positions and parameters, so the chain of multiplies and the accurate cosf / sinf run once per graph run
static const float * ensure_rope_table(exec_state & st, const int32_t * pos, const float * ff, const rope_params & p, int M) {
    backend_ctx * c = st.c;
    static const bool on = !getenv("GGML_MI355X_ROPE_TABLE") || atoi(getenv("GGML_MI355X_ROPE_TABLE")) != 0;
    if (!on || !c->rope_tab || M < 1 || (int64_t) M * p.n_dims > (int64_t) backend_ctx::rope_tab_floats) return nullptr;
    if (st.rope_tab_pos != (const void *) pos || st.rope_tab_ff != (const void *) ff || st.rope_tab_m != M || memcmp(&st.rope_tab_p, &p, sizeof(rope_params)) != 0) {
        launch_rope_table(c->stream, pos, ff, p, M, c->rope_tab);
        c->st.kernel_launches++;
        st.rope_tab_pos = pos;
        st.rope_tab_ff = ff;
        st.rope_tab_p = p;
        st.rope_tab_m = M;
    }
    return c->rope_tab;
}
static void mark_done(exec_state & st, int k) {
    st.done[k] = 1;
    st.ooo.push_back(k);
}
static int use_count(const exec_state & st, const ggml_tensor * t) {
    auto it = st.uses.find(t);
    const int local = it == st.uses.end() ? 0 : it->second;
    auto ig = st.uses_global.find(t);
    int glob;
    if (ig != st.uses_global.end()) glob = ig->second;
    else {
        glob = graph_use_count(st.g, t);
        st.uses_global.emplace(t, glob);
    }
    return std::max(local, glob);
}
static bool single_use(const exec_state & st, const ggml_tensor * t) { return use_count(st, t) == 1 && !(t->flags & GGML_TENSOR_FLAG_OUTPUT); }

// returns the device pointer of src1 quantised for weight type `wtype`, quantising only if the scratch does not
// already hold exactly this tensor (Q/K/V and gate/up share their input)
static const void * quantized_src1(exec_state & st, const ggml_tensor * b, int wtype) {  // (wtype MI_ACT_Q80_PANEL: Q8_0 blocks in the panel order of the 9 .. 32-column kernel)
    backend_ctx * c = st.c;
    const int kind = wtype == MI_ACT_Q80_PANEL ? MI_ACT_Q80_PANEL : act_kind(wtype);
    void * dst = (char *) c->ws + st.act_off;
    if (c->q8_src == b->data && c->q8_kind == kind && c->q8_bytes == ggml_abi_nbytes(b)) return dst;
    {
        timed_scope ts(c, "quantize_act", (double) ggml_abi_nbytes(b));
        if (kind == MI_ACT_Q80_PANEL) launch_quantize_q80_panel(c->stream, TD(b), dst);
        else launch_quantize_act(c->stream, kind, TD(b), dst);
        c->st.kernel_launches++;
    }
    c->q8_src = b->data;
    c->q8_kind = kind;
    c->q8_bytes = ggml_abi_nbytes(b);
    return dst;
}

static const char * type_tag(int t) {
    switch (t) {
        case GGML_TYPE_Q4_K: return "q4_K";
        case GGML_TYPE_Q5_K: return "q5_K";
        case GGML_TYPE_Q6_K: return "q6_K";
        case GGML_TYPE_Q8_0: return "q8_0";
        default: return "f";
    }
}

static bool is_view_op(const ggml_tensor * t);
static bool ranges_overlap(const ggml_tensor * x, const ggml_tensor * y);
static bool quant_consumers_only(const exec_state & st, int at, const ggml_tensor * t);
static bool same_shape(const ggml_tensor * a, const ggml_tensor * b);
// quantised mat-mul, optionally with fused epilogue; w2 != null -> SwiGLU over (w, w2)
static bool run_mul_mat_q(exec_state & st, const ggml_tensor * w, const ggml_tensor * w2, const ggml_tensor * b, ggml_tensor * dst,
                          const ggml_tensor * add, const ggml_tensor * add2) {
    backend_ctx * c = st.c;
    const int64_t K = w->ne[0], N = w->ne[1];
    const int64_t M = b->ne[1] * b->ne[2] * b->ne[3];
    const double wbytes = (double) ggml_abi_row_size(w->type, K) * (double) N * (w2 ? 2.0 : 1.0);
    const bool kquant = w->type == GGML_TYPE_Q4_K || w->type == GGML_TYPE_Q5_K || w->type == GGML_TYPE_Q6_K || (w->type == GGML_TYPE_Q8_0 && (K % 256) == 0);  // formats with the mat-vec prologue
    auto dn = st.deferred.find(b);
    const bool pro_norm = dn != st.deferred.end();
    const bool pro_fa = st.fa_wo.b == b && st.fa_wo.part != nullptr;  // (set by the FLASH_ATTN_EXT node after checking this very mat-vec)
    static const bool dbg_no_pro_f32 = getenv("GGML_MI355X_DBG_NO_PRO_F32") != nullptr;
    const bool pro_f32 = !dbg_no_pro_f32 && !pro_norm && !pro_fa && c->opt.fusion && c->opt.prologue && kquant && M == 1 && b->type == GGML_TYPE_F32 && b->nb[0] == 4 && (((uintptr_t) b->data) & 15) == 0;
    if (pro_fa && (w2 || pro_norm || !kquant || M != 1)) {
        MI_ERR("graph_compute: attention partials were left for a mat-vec that cannot merge them");
        return false;
    }
    if (pro_norm || pro_f32 || pro_fa) {
        mmvq_args a{};
        a.W = (const uint8_t *) w->data;
        a.W2 = w2 ? (const uint8_t *) w2->data : nullptr;
        a.Wp = decode_copy(c, w);  // (the plane-layout copies the streaming kernels read with non-temporal loads: backend.cpp, repack.hip)
        a.W2p = w2 && a.Wp ? decode_copy(c, w2) : nullptr;
        if (w2 && !a.W2p) a.Wp = nullptr;
        c->st.decode_copy_launches += a.Wp != nullptr;
        a.w_nb1 = (int64_t) w->nb[1];
        a.type = w->type;
        a.K = (int) K;
        a.N = (int) N;
        a.ncols = 1;
        a.dst = (float *) dst->data;
        a.dst_stride = (int64_t) (dst->nb[1] / 4);
        a.add = add ? (const float *) add->data : nullptr;
        a.add2 = add2 ? (const float *) add2->data : nullptr;
        a.x = pro_norm ? (const float *) dn->second.x->data : (pro_fa ? nullptr : (const float *) b->data);
        if (pro_fa) {
            a.fa_part = st.fa_wo.part;
            a.fa_splits = st.fa_wo.splits;
            // the attention result itself stays a tensor of the graph: written on the way, unless this launch's result recycled its block
            a.x_out = ranges_overlap(dst, st.fa_wo.fa) ? nullptr : (float *) st.fa_wo.fa->data;
            st.fa_wo.b = nullptr;
            st.fa_wo.part = nullptr;
        }
        a.norm_w = pro_norm ? (const float *) dn->second.w->data : nullptr;
        a.eps = pro_norm ? dn->second.eps : 0.0f;
        // (the allocator may have given THIS result the block of the norm's MUL node, free after its last reader — then that node is
        // provably dead and is not written: the two stores would race inside one launch)
        // ... and so is a MUL node that the allocator placed over the norm's INPUT x (RMS_NORM and MUL run in place when the norm is x's last reader — the
        // final norm in front of the output matrix): workgroup 0 would replace x by norm(x) * w while the other workgroups' prologues still read x.  On
        // one otherwise idle GPU all 256 workgroups start together and read x before workgroup 0 gets that far; with anything else on the GPU (a second
        // rank or model, the logical devices of the tests) late workgroups read the overwritten row: logits 4e-2 off at random steps (round 5).  The
        // node is dead by construction: can_defer_norm admitted the pair only because every reader of it takes the prologue.
        a.norm_out = (pro_norm && !ranges_overlap(dst, dn->second.out) && !ranges_overlap(dn->second.x, dn->second.out)) ? (float *) dn->second.out->data : nullptr;
        if (pro_norm && st.ss_tensor != nullptr && st.ss_tensor == dn->second.x) {  // x came out of a launch that left its sum of squares behind
            a.ss_in = c->ss_buf;
            a.ss_n = st.ss_n;
            c->st.ss_handoffs++;
        }
        if (st.ss_tensor != nullptr && ranges_overlap(dst, st.ss_tensor)) st.ss_tensor = nullptr;  // (this result recycles that tensor's block)
        // (never from a launch that CONSUMES the record — a deferred-norm prologue with a bias or residual add, e.g. Qwen2's Q/K/V + bias when the fused
        // QKV launch declines: a workgroup that finishes early would overwrite ss_buf[blockIdx.x] before a late one has read all ss_n partials)
        if (c->opt.ss_partials && c->ss_buf != nullptr && (add || add2) && !w2 && !a.ss_in && launch_mmvq_ss_count(a) > 0) {
            // a residual stream leaves this launch (wo + residual, ffn_down + residual): the RMS_NORM that reads it next takes the sum of squares from here
            a.ss_out = c->ss_buf;
            st.ss_tensor = dst;
            st.ss_n = launch_mmvq_ss_count(a);
        }
        char cls[64];
        snprintf(cls, sizeof(cls), "mmvq_%s%s_%s", type_tag(w->type), w2 ? "_glu" : "", pro_norm ? "normpro" : (pro_fa ? "attnpro" : "f32pro"));
        timed_scope ts(c, cls, wbytes, true);
        launch_mmvq(c->stream, a, 1);
        c->st.kernel_launches++;
        return true;
    }
    const bool q80_skinny = w->type == GGML_TYPE_Q8_0 && !w2 && !add2 && mmq_q80_skinny_supported(w->type, K, N, M);
    // the prompt GEMM over Q8_0 weights reads both operands in panel order when the weights have their panel copy (GGML_MI355X_Q80_GEMM_PANELS=0: as before round 6)
    static const bool gemm_panels = !getenv("GGML_MI355X_Q80_GEMM_PANELS") || atoi(getenv("GGML_MI355X_Q80_GEMM_PANELS")) != 0;
    const bool q80_gemm = !q80_skinny && w->type == GGML_TYPE_Q8_0 && M >= c->opt.q80_min_cols && !w2 && !add2 && mmq_q80_supported(w->type, K, N, M);
    const uint8_t * q80_gemm_wp = (q80_gemm && gemm_panels) ? q80_panel_copy(c, w) : nullptr;
    const void * act = quantized_src1(st, b, (q80_skinny || q80_gemm_wp) ? MI_ACT_Q80_PANEL : w->type);
    if (q80_skinny) {  // 9 .. 32 columns of a -np decode step (round 6)
        const uint8_t * wp = q80_panel_copy(c, w);
        c->st.decode_copy_launches += wp != nullptr;
        timed_scope ts(c, "mmq_q8_0_skinny", wbytes);
        const int64_t arows = add ? add->ne[1] * add->ne[2] * add->ne[3] : 0;
        launch_mmq_q80_skinny(c->stream, (const uint8_t *) w->data, wp, (int64_t) w->nb[1], (int) K, (int) N, (int) M, act, (float *) dst->data, (int64_t) (dst->nb[1] / 4),
                              add ? (const float *) add->data : nullptr, (!add || arows == 1) ? 0 : (int64_t) (add->nb[1] / 4));
        c->st.kernel_launches++;
        c->st.skinny_launches++;
        return true;
    }
    if (q80_gemm) {
        timed_scope ts(c, "mmq_q8_0", wbytes);
        const int64_t arows = add ? add->ne[1] * add->ne[2] * add->ne[3] : 0;
        c->st.decode_copy_launches += q80_gemm_wp != nullptr;
        launch_mmq_q80(c->stream, (const uint8_t *) w->data, q80_gemm_wp, (int64_t) w->nb[1], (int) K, (int) N, (int) M, act, (float *) dst->data, (int64_t) (dst->nb[1] / 4),
                       add ? (const float *) add->data : nullptr, (!add || arows == 1) ? 0 : (int64_t) (add->nb[1] / 4));
        c->st.kernel_launches++;
        return true;
    }
    const bool i8 = c->opt.mmq_i8 && mmq_i8_supported(w->type, K, N, M);
    if (M >= mmq_min_cols_for(c, w->type) && !w2 && !add2 && (!add || i8) && (i8 || mmq_supported(w->type, K, N, M))) {
        timed_scope ts(c, (std::string("mmq_") + type_tag(w->type) + "_n" + std::to_string(N) + "_k" + std::to_string(K)).c_str(), wbytes, i8);
        const int ks = st.epi_dst == dst ? 1 : ((N % 4) == 0 && (dst->nb[1] % 16) == 0 ? mmq_pick_ksplit(K, N, M, c->opt.mmq_skinny, w->type) : 1);
        float * part = (float *) ((char *) c->ws + st.aux_off);
        if (i8) {
            const int64_t arows = add ? add->ne[1] * add->ne[2] * add->ne[3] : 0;
            const int64_t add_stride = (!add || arows == 1) ? 0 : (int64_t) (add->nb[1] / 4);
            // K split, and the next kernel is the RMS_NORM -> MUL -> Q8_K of exactly this result: it sums the partial products itself
            // (writing the f32 result on the way) — one launch instead of reduce + norm
            bool defer = false;
            if (ks > 1 && c->opt.fusion && c->opt.prologue && st.sk_next >= 0 && ggml_abi_is_contiguous(dst) && (N % 256) == 0 && N <= 16384 && (!add || (add_stride % 4) == 0)) {
                const ggml_cgraph * g = st.g;
                int j = st.sk_next;
                while (j < g->n_nodes && (st.done[j] || is_view_op(g->nodes[j]))) ++j;
                if (j + 1 < g->n_nodes) {
                    const ggml_tensor * rn = g->nodes[j];
                    const ggml_tensor * m = g->nodes[j + 1];
                    if (rn->op == GGML_OP_RMS_NORM && rn->src[0] == dst && m->op == GGML_OP_MUL && single_use(st, rn) && m->type == GGML_TYPE_F32 && m->nb[0] == 4) {
                        const ggml_tensor * wn = m->src[0] == rn ? m->src[1] : (m->src[1] == rn ? m->src[0] : nullptr);
                        defer = wn && wn->type == GGML_TYPE_F32 && ggml_abi_is_contiguous(wn) && wn->ne[0] == rn->ne[0] && ggml_abi_nelements(wn) == wn->ne[0] && same_shape(m, rn) &&
                                !(((uintptr_t) wn->data) & 15) && !(((uintptr_t) dst->data) & 15) && (!add || !(((uintptr_t) add->data) & 15)) && quant_consumers_only(st, j + 1, m);
                    }
                }
            }
            const mmq_epi * epi = st.epi_dst == dst ? &st.epi : nullptr;
            const int served = launch_mmq_i8(c->stream, w->type, (const uint8_t *) w->data, (int64_t) w->nb[1], (int) K, (int) N, (int) M, act, (float *) dst->data, (int64_t) (dst->nb[1] / 4), c->opt.mmq_bn, epi ? 1 : ks, part,
                                             add ? (const float *) add->data : nullptr, add_stride, !defer, c->opt.mmq_skinny, epi);
            c->st.skinny_launches += served == 1;
            c->st.wide_launches += served == 2;
            c->st.tiled_launches += served == 0;
            if (defer) {
                st.sk_dst = dst;
                st.sk = splitk_src{part, ks, (int64_t) M * N, add ? (const float *) add->data : nullptr, add_stride, (float *) dst->data, (int64_t) (dst->nb[1] / 4)};
                c->st.kernel_launches++;
                st.sk_next = -1;
                return true;
            }
        }
        else
            launch_mmq(c->stream, w->type, (const uint8_t *) w->data, (int64_t) w->nb[1], (int) K, (int) N, (int) M, act, (float *) dst->data, (int64_t) (dst->nb[1] / 4), ks, part);
        if (ks > 1) c->st.kernel_launches++;
        c->st.kernel_launches++;
        return true;
    }
    mmvq_args a{};
    a.W = (const uint8_t *) w->data;
    a.W2 = w2 ? (const uint8_t *) w2->data : nullptr;
    a.w_nb1 = (int64_t) w->nb[1];
    a.type = w->type;
    a.K = (int) K;
    a.N = (int) N;
    a.ncols = (int) M;
    a.act = act;
    if (M == 1) {
        a.Wp = decode_copy(c, w);
        a.W2p = w2 && a.Wp ? decode_copy(c, w2) : nullptr;
        if (w2 && !a.W2p) a.Wp = nullptr;
        c->st.decode_copy_launches += a.Wp != nullptr;
    }
    a.dst = (float *) dst->data;
    a.dst_stride = (int64_t) (dst->nb[1] / 4);
    auto addend = [&](const ggml_tensor * t, const float *& p, int64_t & stride) {
        p = nullptr;
        stride = 0;
        if (!t) return;
        p = (const float *) t->data;
        stride = (t->ne[1] * t->ne[2] * t->ne[3] == 1) ? 0 : (int64_t) (t->nb[1] / 4);
    };
    addend(add, a.add, a.add_stride);
    addend(add2, a.add2, a.add2_stride);
    const int rpw = (M == 1 && N >= 2048) ? 2 : 1;
    char cls[64];
    snprintf(cls, sizeof(cls), "mmvq_%s%s_nc%d", type_tag(w->type), w2 ? "_glu" : "", (int) std::min<int64_t>(M, 8));
    // the column loop lives in launch_mmvq (chunks of <= 8); weights are re-streamed once per chunk
    timed_scope ts(c, cls, wbytes * (double) ((M + 7) / 8));
    launch_mmvq(c->stream, a, rpw);
    c->st.kernel_launches += (M + 7) / 8;
    return true;
}

static bool quant_mm_ok(const ggml_tensor * n) {
    return n->op == GGML_OP_MUL_MAT && is_quant(n->src[0]->type) && ggml_abi_is_contiguous(n);
}
// an ADD whose one operand is `x` and whose other operand is a bias row or a same-shape residual; returns the other
static const ggml_tensor * add_partner(const ggml_tensor * add, const ggml_tensor * x) {
    if (add->op != GGML_OP_ADD || add->type != GGML_TYPE_F32 || !ggml_abi_is_contiguous(add)) return nullptr;
    const ggml_tensor * o = add->src[0] == x ? add->src[1] : (add->src[1] == x ? add->src[0] : nullptr);
    if (!o || o == x || o->type != GGML_TYPE_F32 || !ggml_abi_is_contiguous(o)) return nullptr;
    if (add->src[0] != x && !same_shape(add->src[0], add)) return nullptr;  // x must be the broadcast target when it is src1
    if (o->ne[0] != x->ne[0]) return nullptr;
    const int64_t orows = o->ne[1] * o->ne[2] * o->ne[3];
    if (orows != 1 && !same_shape(o, x)) return nullptr;
    if (!same_shape(add, x)) return nullptr;
    return o;
}

// Batches (M > 1): may the producer of `t` (node index `at`) write ONLY the Q8_K blocks of its result into the activation
// scratch?  Yes when every use of t is src1 of a K-quant MUL_MAT and no other quantised mat-mul (which would overwrite the
// scratch) runs before the last of them.
static bool quant_consumers_only(const exec_state & st, int at, const ggml_tensor * t) {
    if ((t->flags & GGML_TENSOR_FLAG_OUTPUT) || t->type != GGML_TYPE_F32 || (t->ne[0] % 256) != 0 || t->ne[0] > 16384 * 4) return false;
    const ggml_cgraph * g = st.g;
    int last = -1, n_cons = 0;
    for (int j = at + 1; j < g->n_nodes; ++j) {
        const ggml_tensor * u = g->nodes[j];
        for (int s = 0; s < GGML_MAX_SRC; ++s) {
            if (u->src[s] != t) continue;
            const ggml_tensor * wt = u->src[0];
            if (!(u->op == GGML_OP_MUL_MAT && s == 1 && (wt->type == GGML_TYPE_Q4_K || wt->type == GGML_TYPE_Q5_K || wt->type == GGML_TYPE_Q6_K))) return false;
            if (buffer_is_split(wt->buffer)) return false;  // (split weights read the f32 activations on other devices: the tensor itself must exist)
            last = j;
            n_cons++;
        }
    }
    if (n_cons == 0 || n_cons != use_count(st, t)) return false;
    for (int j = at + 1; j <= last; ++j) {
        const ggml_tensor * u = g->nodes[j];
        if (u->op == GGML_OP_MUL_MAT && is_quant(u->src[0]->type) && u->src[1] != t) return false;
        if (u->op == GGML_OP_MUL_MAT_ID) return false;
    }
    return true;
}
// ... and the same question for Q8_0 weights (round 6): every reader of t is a Q8_0 mat-mul that takes the 9 .. 128-column matrix-core kernel (its activations in panel
// order), so a producer may leave Q8_0 panel blocks instead of f32
static bool q80_panel_consumers_only(const exec_state & st, int at, const ggml_tensor * t) {
    if ((t->flags & GGML_TENSOR_FLAG_OUTPUT) || t->type != GGML_TYPE_F32 || (t->ne[0] % 128) != 0) return false;
    const int64_t M = t->ne[1] * t->ne[2] * t->ne[3];
    const ggml_cgraph * g = st.g;
    int last = -1, n_cons = 0;
    for (int j = at + 1; j < g->n_nodes; ++j) {
        const ggml_tensor * u = g->nodes[j];
        for (int s = 0; s < GGML_MAX_SRC; ++s) {
            if (u->src[s] != t) continue;
            const ggml_tensor * wt = u->src[0];
            if (!(u->op == GGML_OP_MUL_MAT && s == 1 && wt->type == GGML_TYPE_Q8_0 && wt->ne[2] == 1 && wt->ne[3] == 1 && rows_contig(wt) && !buffer_is_split(wt->buffer) &&
                  mmq_q80_skinny_supported(wt->type, wt->ne[0], wt->ne[1], M) && u->type == GGML_TYPE_F32 && ggml_abi_is_contiguous(u)))
                return false;
            if (tp_active(st.c) && buffer_is_rowpar(wt->view_src ? wt->view_src->buffer : wt->buffer)) return false;
            last = j;
            n_cons++;
        }
    }
    if (n_cons == 0 || n_cons != use_count(st, t)) return false;
    for (int j = at + 1; j <= last; ++j) {  // (the activation area is shared: no other quantised mat-mul may run before the last reader)
        const ggml_tensor * u = g->nodes[j];
        if (u->op == GGML_OP_MUL_MAT && is_quant(u->src[0]->type) && u->src[1] != t) return false;
        if (u->op == GGML_OP_MUL_MAT_ID) return false;
    }
    return true;
}
static void mark_q8_cache(exec_state & st, const ggml_tensor * t, int kind = GGML_TYPE_Q8_K);
static void mark_q8_cache(exec_state & st, const ggml_tensor * t, int kind) {
    st.c->q8_src = t->data;
    st.c->q8_kind = kind;
    st.c->q8_bytes = ggml_abi_nbytes(t);
    st.q8_fresh = true;
}

// May the pair [RMS_NORM n (node i), MUL m (node i+1)] be left un-launched?  Yes when m is a single row consumed ONLY
// as src1 of K-quant MUL_MATs (each recomputes norm*w in its prologue), and no node up to the last consumer writes
// memory overlapping x — the host allocator may already have recycled x's block if the norm was its last reader.
static bool can_defer_norm(const exec_state & st, int i, const ggml_tensor * n, const ggml_tensor * m, const ggml_tensor * x, const ggml_tensor * w) {
    if (ggml_abi_nrows(m) != 1 || (m->flags & GGML_TENSOR_FLAG_OUTPUT) || (n->flags & GGML_TENSOR_FLAG_OUTPUT)) return false;
    if (!ggml_abi_is_contiguous(x) || !ggml_abi_is_contiguous(m) || ((((uintptr_t) x->data) | ((uintptr_t) w->data) | ((uintptr_t) m->data)) & 15) || (x->ne[0] % 256) != 0 || x->ne[0] > 16384) return false;
    const ggml_cgraph * g = st.g;
    int last = -1, n_cons = 0;
    for (int j = i + 2; j < g->n_nodes; ++j) {
        const ggml_tensor * t = g->nodes[j];
        for (int s = 0; s < GGML_MAX_SRC; ++s) {
            if (t->src[s] != m) continue;
            const ggml_tensor * wt = t->src[0];
            const bool ok = t->op == GGML_OP_MUL_MAT && s == 1 && (wt->type == GGML_TYPE_Q4_K || wt->type == GGML_TYPE_Q5_K || wt->type == GGML_TYPE_Q6_K || wt->type == GGML_TYPE_Q8_0) &&
                            wt->ne[2] == 1 && wt->ne[3] == 1 && ggml_abi_is_contiguous(t) && !(st.c->tp && buffer_is_rowpar(wt->view_src ? wt->view_src->buffer : wt->buffer)) &&
                            !buffer_is_split(wt->buffer);  // (a split weight's devices are sent the f32 row: it has to be written)
            if (!ok) return false;
            last = j;
            n_cons++;
        }
    }
    if (n_cons == 0 || n_cons != use_count(st, m)) return false;
    const char * x0 = (const char *) x->data;
    const char * x1 = x0 + ggml_abi_nbytes(x);
    for (int j = i + 2; j <= last; ++j) {
        const ggml_tensor * t = g->nodes[j];
        if (t->op == GGML_OP_NONE || t->op == GGML_OP_VIEW || t->op == GGML_OP_RESHAPE || t->op == GGML_OP_PERMUTE || t->op == GGML_OP_TRANSPOSE) continue;
        const char * t0 = (const char *) t->data;
        const char * t1 = t0 + ggml_abi_nbytes(t);
        if (t0 < x1 && x0 < t1) return false;
    }
    return true;
}

// ------------------------------------------------------------------------------------------------ fused Q/K/V
static bool ranges_overlap(const ggml_tensor * x, const ggml_tensor * y);
static bool is_view_op(const ggml_tensor * t) {
    return t->op == GGML_OP_NONE || t->op == GGML_OP_VIEW || t->op == GGML_OP_RESHAPE || t->op == GGML_OP_PERMUTE || t->op == GGML_OP_TRANSPOSE;
}
struct qkv_chain {
    const ggml_tensor * mm = nullptr;
    const ggml_tensor * bias = nullptr;
    const ggml_tensor * rope = nullptr;
    const ggml_tensor * store = nullptr;    // SET_ROWS node (f16 cache rows)
    bool scatter = false;                   // ... whose rows are single elements (transposed V cache): one index per value
    const ggml_tensor * out_f32 = nullptr;  // otherwise: last materialised f32 tensor of the chain
    std::vector<int> nodes;
};
// the block-format / bf16 cache-row stores ride in the fused Q/K/V launch (round 6); GGML_MI355X_QKV_KV_STORE=0: the separate SET_ROWS launch of round 5 (A/B)
static bool qkv_kv_store_on() {
    static const bool on = !getenv("GGML_MI355X_QKV_KV_STORE") || atoi(getenv("GGML_MI355X_QKV_KV_STORE")) != 0;
    return on;
}
// follows mm -> [ADD bias] -> (RESHAPE)* -> [ROPE] -> (RESHAPE)* -> [SET_ROWS]; stops at the first consumer it cannot absorb
static bool follow_qkv_chain(const exec_state & st, int start, int limit, qkv_chain & ch) {
    const ggml_cgraph * g = st.g;
    const ggml_tensor * cur = g->nodes[start];
    ch.mm = cur;
    ch.nodes.push_back(start);
    const int64_t N = cur->ne[0];
    for (;;) {
        if (!single_use(st, cur)) break;
        int j = -1;
        for (int k = start + 1; k < limit && j < 0; ++k)
            for (int sidx = 0; sidx < GGML_MAX_SRC; ++sidx)
                if (g->nodes[k]->src[sidx] == cur) { j = k; break; }
        if (j < 0) break;
        const ggml_tensor * c = g->nodes[j];
        if (c->op == GGML_OP_RESHAPE && c->src[0] == cur) {
            cur = c;
            ch.nodes.push_back(j);
            continue;
        }
        if (c->op == GGML_OP_ADD && !ch.bias && !ch.rope) {
            const ggml_tensor * o = add_partner(c, cur);
            if (o && ggml_abi_nelements(o) == N && ggml_abi_nrows(c) == 1) {
                ch.bias = o;
                cur = c;
                ch.nodes.push_back(j);
                continue;
            }
            break;
        }
        if (c->op == GGML_OP_ROPE && c->src[0] == cur && !ch.rope && c->type == GGML_TYPE_F32 && ggml_abi_is_contiguous(c)) {
            ch.rope = c;
            cur = c;
            ch.nodes.push_back(j);
            continue;
        }
        if (c->op == GGML_OP_SET_ROWS && c->src[0] == cur && c->ne[0] == N && cur->ne[0] == N && ggml_abi_nelements(cur) == N &&
            c->src[1]->type == GGML_TYPE_I64 && ggml_abi_nelements(c->src[1]) == 1 &&
            ((c->type == GGML_TYPE_F16 && c->nb[0] == 2) || (c->type == GGML_TYPE_BF16 && c->nb[0] == 2 && qkv_kv_store_on()) ||
             ((c->type == GGML_TYPE_Q8_0 || (kv_type_is_block(c->type) && qkv_kv_store_on())) && (N % 32) == 0))) {
            ch.store = c;
            ch.nodes.push_back(j);
            return true;
        }
        // the non-flash path keeps V transposed: the projection is viewed as N rows of one element and each goes to its own cache row
        // (llama.cpp's v_idxs: element j of the token -> j * n_ctx + cell)
        if (c->op == GGML_OP_SET_ROWS && c->src[0] == cur && cur->ne[0] == 1 && c->ne[0] == 1 && ggml_abi_nelements(cur) == N &&
            c->src[1]->type == GGML_TYPE_I64 && ggml_abi_nelements(c->src[1]) == N && ggml_abi_is_contiguous(c->src[1]) &&
            c->type == GGML_TYPE_F16 && c->nb[0] == 2 && c->nb[1] == 2) {
            ch.store = c;
            ch.scatter = true;
            ch.nodes.push_back(j);
            return true;
        }
        break;
    }
    // the chain ends in front of a consumer it cannot absorb (e.g. the SET_ROWS of a cache kept in another type, kv_types.hip): trailing views are not
    // part of it — they do nothing, and counting them would stretch the fused window over nodes that sit between them (SET_ROWS(k) in front of
    // V's reshape) and make try_fuse_qkv give up
    while (ch.nodes.size() > 1 && is_view_op(g->nodes[ch.nodes.back()]) && g->nodes[ch.nodes.back()] != ch.rope) ch.nodes.pop_back();
    cur = g->nodes[ch.nodes.back()];
    ch.out_f32 = cur;
    return ggml_abi_is_contiguous(cur) && cur->type == GGML_TYPE_F32;
}

// at a single-column K-quant MUL_MAT: try to run it together with its siblings (same src1) and their bias / rope / cache
// store chains as ONE launch (qkv.hip).  Returns true if launched; the absorbed nodes are flagged in st.done.
static bool try_fuse_qkv(exec_state & st, int i) {
    backend_ctx * c = st.c;
    ggml_cgraph * g = st.g;
    const ggml_tensor * n = g->nodes[i];
    const ggml_tensor * X = n->src[1];
    auto kq = [](const ggml_tensor * w) { return (w->type == GGML_TYPE_Q4_K || w->type == GGML_TYPE_Q5_K || w->type == GGML_TYPE_Q6_K || w->type == GGML_TYPE_Q8_0) && w->ne[2] == 1 && w->ne[3] == 1 && rows_contig(w) && (w->ne[1] % 2) == 0; };
    if (ggml_abi_nrows(X) != 1 || X->type != GGML_TYPE_F32 || (X->ne[0] % 256) != 0) return false;
    auto dn = st.deferred.find(X);
    const bool norm = dn != st.deferred.end();
    if (!norm && (X->nb[0] != 4 || (((uintptr_t) X->data) & 15))) return false;
    if (norm && X->ne[0] > 8192) return false;  // qkv.hip holds the whole row in one prologue batch
    const int limit = std::min(g->n_nodes, i + 40);
    std::vector<qkv_chain> chains;
    for (int k = i; k < limit && chains.size() < 3; ++k) {
        const ggml_tensor * t = g->nodes[k];
        if (t->op != GGML_OP_MUL_MAT || t->src[1] != X) continue;
        if (!kq(t->src[0]) || !ggml_abi_is_contiguous(t) || st.done[k]) return false;
        if (c->tp && buffer_is_rowpar(t->src[0]->view_src ? t->src[0]->view_src->buffer : t->src[0]->buffer)) return false;
        qkv_chain ch;
        if (!follow_qkv_chain(st, k, limit, ch)) return false;
        chains.push_back(ch);
    }
    if (chains.size() < 2) return false;
    {
        // a block-format cache row is assembled from 16 consecutive row PAIRS of a workgroup trip: the NeoX layout pairs rows (i, i + d/2) of two different
        // blocks.  Such a model keeps round 5's form for those caches — the chains end in f32 in front of their SET_ROWS, which run as their own launches —
        // instead of losing the whole fused launch (as a q8_0 cache did until round 6: Qwen2-7B at 8 k context 356 tok/s with -ctk q8_0 against 481 with f16)
        bool neox = false;
        for (auto & ch : chains) neox = neox || (ch.rope && ((ch.rope->op_params[2] & GGML_ROPE_TYPE_NEOX) || (ch.rope->ne[0] % 32) != 0)) || (ch.mm->ne[0] % 32) != 0;
        if (neox)
            for (auto & ch : chains)
                if (ch.store && !ch.scatter && (kv_type_is_block(ch.store->type) || ch.store->type == GGML_TYPE_Q8_0)) {
                    ch.store = nullptr;
                    ch.nodes.pop_back();
                    while (ch.nodes.size() > 1 && is_view_op(g->nodes[ch.nodes.back()]) && g->nodes[ch.nodes.back()] != ch.rope) ch.nodes.pop_back();
                    ch.out_f32 = g->nodes[ch.nodes.back()];
                    if (!ggml_abi_is_contiguous(ch.out_f32) || ch.out_f32->type != GGML_TYPE_F32) return false;
                }
    }
    bool attn = false;  // only attention projections (something rotates or lands in the KV cache); gate/up siblings have their own fusion
    for (auto & ch : chains) attn = attn || ch.rope || ch.store;
    if (!attn) return false;
    // every consumer of X must be one of the collected mat-muls (otherwise a fourth sibling would be left behind; fine, but keep it simple)
    int last = i;
    std::vector<char> in_set(g->n_nodes, 0);
    for (auto & ch : chains) for (int k : ch.nodes) { in_set[k] = 1; last = std::max(last, k); }
    for (int k = i; k <= last; ++k)
        if (!in_set[k] && !is_view_op(g->nodes[k])) return false;  // an unrelated node sits inside the window: do not reorder around it
    // rope / store consistency
    const ggml_tensor * rope0 = nullptr;
    const ggml_tensor * idx0 = nullptr;
    bool q8_store = false;  // a cache row in a block format (q8_0, or q4_0 / q4_1 / q5_0 / q5_1 / iq4_nl: kv_quant.h): assembled per block by the launch's workgroups
    for (auto & ch : chains) q8_store = q8_store || (ch.store && !ch.scatter && (ch.store->type == GGML_TYPE_Q8_0 || kv_type_is_block(ch.store->type)));
    for (auto & ch : chains) {
        if (q8_store) {
            // quantised KV cache: a workgroup trip (16 row pairs) must be exactly one block_q8_0 of the cache row, which the
            // interleaved pair layout gives when every segment is a multiple of 32 rows; the NeoX layout pairs rows
            // (i, i + d/2) of two different blocks and stays on the unfused path
            if ((ch.mm->ne[0] % 32) != 0) return false;
            if (ch.rope && ((ch.rope->op_params[2] & GGML_ROPE_TYPE_NEOX) || (ch.rope->ne[0] % 32) != 0)) return false;
        }
        if (ch.rope) {
            const ggml_tensor * r = ch.rope;
            const int mode = r->op_params[2];
            if ((mode & GGML_ROPE_TYPE_MROPE) || r->op_params[1] != r->ne[0] || r->ne[2] != 1 || r->ne[3] != 1 || (r->ne[0] % 2) || r->ne[0] > 256) return false;
            if (r->ne[0] * r->ne[1] != ch.mm->ne[0]) return false;
            if (rope0 && (memcmp(rope0->op_params, r->op_params, sizeof(r->op_params)) != 0 || rope0->src[1] != r->src[1] || rope0->src[2] != r->src[2] || rope0->ne[0] != r->ne[0])) return false;
            rope0 = r;
        }
        if (ch.store && !ch.scatter) {
            if (idx0 && idx0->data != ch.store->src[1]->data) return false;
            idx0 = ch.store->src[1];
        }
    }
    // every workgroup reads X in its prologue while others may already be storing results: when X is a materialised tensor
    // (no deferred norm), none of the launch's outputs may live in memory the allocator recycled from X
    if (!norm)
        for (auto & ch : chains) {
            const ggml_tensor * o = ch.store ? nullptr : ch.out_f32;
            if (o && ranges_overlap(o, X)) return false;
        }
    // rope parameters shared by every launch of this group
    qkv_args base{};
    base.K = (int) X->ne[0];
    base.x = norm ? (const float *) dn->second.x->data : (const float *) X->data;
    base.norm_w = norm ? (const float *) dn->second.w->data : nullptr;
    base.eps = norm ? dn->second.eps : 0.0f;
    base.norm_out = (norm && !ranges_overlap(dn->second.x, dn->second.out)) ? (float *) dn->second.out->data : nullptr;  // (never over the norm's own input: see run_mul_mat_q)
    if (norm && st.ss_tensor != nullptr && st.ss_tensor == dn->second.x) {
        base.ss_in = c->ss_buf;
        base.ss_n = st.ss_n;
        c->st.ss_handoffs++;
    }
    if (norm)  // a chain tensor that recycled the MUL node's block proves that node dead after this launch: do not write it (race)
        for (auto & ch : chains)
            for (int k : ch.nodes)
                if (ranges_overlap(g->nodes[k], dn->second.out)) base.norm_out = nullptr;
    if (rope0) {
        rope_params p;
        p.n_dims = rope0->op_params[1];
        p.mode = rope0->op_params[2];
        p.n_ctx_orig = rope0->op_params[4];
        p.freq_base = ggml_abi_op_param_f32(rope0, 5);
        p.freq_scale = ggml_abi_op_param_f32(rope0, 6);
        p.ext_factor = ggml_abi_op_param_f32(rope0, 7);
        p.attn_factor = ggml_abi_op_param_f32(rope0, 8);
        p.beta_fast = ggml_abi_op_param_f32(rope0, 9);
        p.beta_slow = ggml_abi_op_param_f32(rope0, 10);
        rope_host_consts(p, base.theta_scale, base.corr0, base.corr1);
        base.head_dim = (int) rope0->ne[0];
        base.neox = (p.mode & GGML_ROPE_TYPE_NEOX) ? 1 : 0;
        base.pos = (const int32_t *) rope0->src[1]->data;
        base.freq_factors = rope0->src[2] ? (const float *) rope0->src[2]->data : nullptr;
        base.freq_scale = p.freq_scale;
        base.ext_factor = p.ext_factor;
        base.attn_factor = p.attn_factor;
        memset(p.sections, 0, sizeof(p.sections));
        base.rope_tab = ensure_rope_table(st, base.pos, base.freq_factors, p, 1);
    } else {
        base.head_dim = 2;
        base.pos = nullptr;
    }
    base.slot = idx0 ? (const int64_t *) idx0->data : nullptr;
    // one launch for up to two weight formats (Q4_K_M: {wq, wk} Q4_K and, in the "more bits" layers, {wv} Q6_K — each format gets
    // its own range of workgroups inside the launch); a third format, or a pair qkv.hip has no instantiation for, launches separately
    std::vector<char> launched(chains.size(), 0);
    for (size_t first = 0; first < chains.size(); ++first) {
        if (launched[first]) continue;
        const int type = chains[first].mm->src[0]->type;
        int type_b = type;
        for (size_t s = first + 1; s < chains.size(); ++s) {
            const int t2 = chains[s].mm->src[0]->type;
            if (!launched[s] && t2 != type && qkv_types_supported(type, t2)) { type_b = t2; break; }
        }
        qkv_args a = base;
        a.nseg = 0;
        a.planes = 1;
        double bytes = 0;
        for (size_t s = first; s < chains.size(); ++s) {
            const qkv_chain & ch = chains[s];
            const ggml_tensor * w = ch.mm->src[0];
            if (launched[s] || (w->type != type && w->type != type_b)) continue;
            launched[s] = 1;
            qkv_seg & sg = a.seg[a.nseg++];
            sg.W = (const uint8_t *) w->data;
            sg.w_nb1 = (int64_t) w->nb[1];
            sg.alt = w->type == type ? 0 : 1;
            sg.N = (int) w->ne[1];
            sg.bias = ch.bias ? (const float *) ch.bias->data : nullptr;
            sg.rope = ch.rope ? 1 : 0;
            sg.store = !ch.store ? 0 : ch.scatter ? 3 : (ch.store->type == GGML_TYPE_F16 ? 1 : ch.store->type == GGML_TYPE_BF16 ? 4 : 2);
            sg.kvt = (uint8_t) (ch.store && sg.store == 2 ? ch.store->type : 0);
            sg.out = ch.store ? (char *) ch.store->data : (char *) ch.out_f32->data;
            sg.row_stride = !ch.store ? 0 : ch.scatter ? (int64_t) (uintptr_t) ch.store->src[1]->data : (int64_t) ch.store->nb[1];
            bytes += (double) ggml_abi_row_size(w->type, w->ne[0]) * (double) w->ne[1];
        }
        {   // the decode copies (plane layout) of every matrix of the launch, or the block layout for all of them
            const uint8_t * wp[3] = {nullptr, nullptr, nullptr};
            int k = 0;
            for (size_t s2 = first; s2 < chains.size() && a.planes; ++s2) {
                const ggml_tensor * w = chains[s2].mm->src[0];
                if (k >= a.nseg || (const uint8_t *) w->data != a.seg[k].W) continue;
                wp[k] = decode_copy(c, w);
                if (!wp[k]) a.planes = 0;
                ++k;
            }
            if (k != a.nseg) a.planes = 0;
            if (a.planes) {
                for (int q = 0; q < a.nseg; ++q) a.seg[q].W = wp[q];
                c->st.decode_copy_launches++;
            }
        }
        char cls[64];
        if (type_b == type) snprintf(cls, sizeof(cls), "qkv_fused_%s_%s", type_tag(type), norm ? "normpro" : "f32pro");
        else snprintf(cls, sizeof(cls), "qkv_fused_%s+%s_%s", type_tag(type), type_tag(type_b), norm ? "normpro" : "f32pro");
        timed_scope ts(c, cls, bytes);
        launch_qkv(c->stream, a, type, type_b);
        c->st.kernel_launches++;
    }
    for (auto & ch : chains) for (int k : ch.nodes) { mark_done(st, k); c->st.fused_nodes++; }
    return true;
}

struct rope_store_plan {
    rope_store_args a{};
    int T = 0;
    int nodes[3] = {-1, -1, -1};                            // ROPE(k), SET_ROWS(k), SET_ROWS(v)
    const ggml_tensor * src[3] = {nullptr, nullptr, nullptr};  // what ROPE(q), ROPE(k) and SET_ROWS(v) read
};
static bool plan_rope_store(const exec_state & st, int i, rope_store_plan & pl);
static const ggml_tensor * through_views(const ggml_tensor * t);

// ------------------------------------------------------------------------------------------------ batches: sibling mat-muls
// A batch's wq / wk / wv (and ffn_gate / ffn_up) multiply the same activations.  Launched one by one, the small ones cannot
// fill the chip without a K split and its second kernel; as ONE launch over the concatenated row panels they can (mmq_i8.hip).
// The siblings that follow node i in the graph are executed EARLY, at node i: legal when nothing between node i and a
// sibling's own position reads or occupies the memory the sibling's result is written to (the graph allocator may have given
// it a block that an earlier tensor still owns at node i's time).  Bias / residual ADDs directly after a member ride in its store.
static bool ranges_overlap(const ggml_tensor * x, const ggml_tensor * y) {
    if (!x->data || !y->data) return false;
    const char * x0 = (const char *) x->data, * y0 = (const char *) y->data;
    return x0 < y0 + ggml_abi_nbytes(y) && y0 < x0 + ggml_abi_nbytes(x);
}
static int try_merge_mm_batch(exec_state & st, int i) {  // returns the number of nodes consumed at position i (0: not merged)
    backend_ctx * c = st.c;
    ggml_cgraph * g = st.g;
    ggml_tensor * n0 = g->nodes[i];
    const ggml_tensor * X = n0->src[1];
    const int type = n0->src[0]->type;
    const int64_t K = n0->src[0]->ne[0], M = X->ne[1] * X->ne[2] * X->ne[3];
    struct member { int k; ggml_tensor * dst; const ggml_tensor * add; int n_nodes; };
    auto eligible = [&](const ggml_tensor * t, bool same_type = true) {
        const ggml_tensor * w = t->src[0];
        return t->op == GGML_OP_MUL_MAT && t->src[1] == X && (w->type == type) == same_type && w->ne[0] == K && w->ne[2] == 1 && w->ne[3] == 1 && rows_contig(w) &&
               mmq_i8_supported(w->type, K, w->ne[1], M) && (w->ne[1] % 4) == 0 && t->type == GGML_TYPE_F32 && ggml_abi_is_contiguous(t) &&
               !(tp_active(c) && buffer_is_rowpar(w->view_src ? w->view_src->buffer : w->buffer));
    };
    auto with_add = [&](int k) {  // member at node k, with the ADD that directly follows it folded in
        ggml_tensor * t = g->nodes[k];
        ggml_tensor * a1 = k + 1 < g->n_nodes ? g->nodes[k + 1] : nullptr;
        const ggml_tensor * o1 = (a1 && !st.done[k + 1] && single_use(st, t)) ? add_partner(a1, t) : nullptr;
        if (o1 && ggml_abi_is_contiguous(a1) && a1->type == GGML_TYPE_F32) {
            // the sibling runs early, at node i: what it adds must exist by then (a bias is a weight; a residual produced
            // between node i and the sibling's own position is not there yet)
            bool ready = true;
            for (int j = i; j <= k && ready; ++j) ready = !ranges_overlap(g->nodes[j], o1) || is_view_op(g->nodes[j]);
            if (ready) return member{k, a1, o1, 2};
        }
        return member{k, t, nullptr, 1};
    };
    if (!eligible(n0)) return 0;
    std::vector<member> ms{with_add(i)};
    const int limit = std::min(g->n_nodes, i + 24);
    std::vector<member> others;  // siblings stored in another format: not part of the launch, but pulled forward behind it (their
                                 // projections then no longer sit between the ropes and the cache stores: try_fuse_rope_store)
    for (int k = i + ms[0].n_nodes; k < limit && ms.size() + others.size() < 3; ++k) {
        ggml_tensor * t = g->nodes[k];
        if (st.done[k]) continue;
        const bool same = eligible(t), other = !same && eligible(t, false);
        if (!same && !other) continue;
        const member m = with_add(k);
        bool ok = true;
        for (auto & o : ms) ok = ok && !ranges_overlap(m.dst, o.dst) && !ranges_overlap(m.dst, g->nodes[o.k]);
        for (auto & o : others) ok = ok && !ranges_overlap(m.dst, o.dst) && !ranges_overlap(m.dst, g->nodes[o.k]);
        for (int j = i + 1; j < k && ok; ++j) {  // everything that runs between node i and the sibling's own position
            const ggml_tensor * u = g->nodes[j];
            bool is_member = false;
            for (auto & o : ms) is_member = is_member || j == o.k || (o.n_nodes == 2 && j == o.k + 1);
            for (auto & o : others) is_member = is_member || j == o.k || (o.n_nodes == 2 && j == o.k + 1);
            if (!is_view_op(u) && !is_member && ranges_overlap(m.dst, u)) ok = false;
            for (int sidx = 0; sidx < GGML_MAX_SRC && ok; ++sidx)
                if (u->src[sidx] && ranges_overlap(m.dst, u->src[sidx])) ok = false;
        }
        if (ok && m.add && ranges_overlap(m.dst, m.add) && m.add->data != m.dst->data) ok = false;
        if (ok) (same ? ms : others).push_back(m);
    }
    // a decode step of a continuous batch: the sibling stored in another K-quant format (Q4_K_M keeps wv as Q6_K in half the layers) joins
    // the launch — the skinny kernel serves two formats in two passes — instead of running alone on 32 workgroups behind it
    bool mixed = false;
    if (c->opt.mmq_skinny && c->opt.skinny_mix && M >= 2 && M <= 32 && ms.size() == 2 && others.size() == 1) {
        int ty[3];
        int64_t Ns[3], nb[3];
        const member * all[3] = {&ms[0], &ms[1], &others[0]};
        for (int q = 0; q < 3; ++q) {
            const ggml_tensor * w = g->nodes[all[q]->k]->src[0];
            ty[q] = w->type;
            Ns[q] = w->ne[1];
            nb[q] = (int64_t) w->nb[1];
        }
        if (mmq_skinny_mix_ok(ty, Ns, nb, 3, K, M)) {
            ms.push_back(others[0]);
            others.clear();
            mixed = true;
        }
    }
    if (ms.size() < 2) return 0;
    int64_t n_total = 0;
    mmq_mat_desc mats[3];
    double wbytes = 0;
    for (size_t q = 0; q < ms.size(); ++q) {
        const ggml_tensor * w = g->nodes[ms[q].k]->src[0];
        const ggml_tensor * add = ms[q].add;
        const int64_t arows = add ? add->ne[1] * add->ne[2] * add->ne[3] : 0;
        if ((ms[q].dst->nb[1] % 16) != 0) return 0;
        mats[q] = {(const uint8_t *) w->data, (int64_t) w->nb[1], (int) w->ne[1], (float *) ms[q].dst->data, (int64_t) (ms[q].dst->nb[1] / 4),
                   add ? (const float *) add->data : nullptr, (!add || arows == 1) ? 0 : (int64_t) (add->nb[1] / 4), (int) w->type == type ? 0 : (int) w->type};
        n_total += w->ne[1];
        wbytes += (double) ggml_abi_row_size(w->type, K) * (double) w->ne[1];
    }
    int ks = mmq_pick_ksplit(K, n_total, M, c->opt.mmq_skinny, type);
    // -np decode steps: when these are the attention projections and ROPE(q), ROPE(k), SET_ROWS(k), SET_ROWS(v) follow (the window
    // plan_rope_store recognises), the skinny launches rotate and store in their epilogue — no K split, no f32 projections, no rope launch
    rope_store_plan epl;
    int epi_node = -1;
    mmq_epi epi{};
    if (c->opt.fusion && c->opt.mmq_skinny && c->opt.skinny_rope && M <= 32 && st.rs_sk.node < 0) {
        for (int k = i + ms[0].n_nodes; k < std::min(g->n_nodes, i + 24) && epi_node < 0; ++k)
            if (!st.done[k] && !is_view_op(g->nodes[k])) {
                bool member = false;
                for (auto & o : ms) member = member || k == o.k || (o.n_nodes == 2 && k == o.k + 1);
                for (auto & o : others) member = member || k == o.k || (o.n_nodes == 2 && k == o.k + 1);
                if (!member) epi_node = k;
            }
        bool ok = epi_node >= 0 && g->nodes[epi_node]->op == GGML_OP_ROPE;
        if (ok) {
            // (the members behind node i are flagged done below; the plan's window walk must already see them that way)
            std::vector<int> flagged;
            for (size_t q = 1; q < ms.size(); ++q) for (int d = 0; d < ms[q].n_nodes; ++d) if (!st.done[ms[q].k + d]) { st.done[ms[q].k + d] = true; flagged.push_back(ms[q].k + d); }
            for (auto & o : others) for (int d = 0; d < o.n_nodes; ++d) if (!st.done[o.k + d]) { st.done[o.k + d] = true; flagged.push_back(o.k + d); }
            ok = plan_rope_store(st, epi_node, epl);
            for (int f : flagged) st.done[f] = false;
        }
        // (all three in ONE launch only — a wv stored in another K-quant format has joined it above (skinny_mix).  A sibling left outside would run
        // alone without its K split, 32 workgroups for 1024 rows, and cost more than the rope launch saves: measured 4.27 -> 4.40 ms per
        // -np 32 step in round 2)
        ok = ok && epl.a.p.mode == 0 && (epl.a.head_dim % 2) == 0 && ms.size() == 3 && others.empty();  // (V: cache rows, or — non-flash path — the elements of the transposed cache)
        auto role_of = [&](const member & m) {
            for (int sidx = 0; sidx < 3; ++sidx)
                if (through_views(epl.src[sidx]) == m.dst) return sidx;
            return -1;
        };
        auto member_ok = [&](const member & m, int64_t N) {
            const int r = role_of(m);
            const int64_t arows = m.add ? m.add->ne[1] * m.add->ne[2] * m.add->ne[3] : 0;
            const ggml_tensor * w = g->nodes[m.k]->src[0];
            return r >= 0 && use_count(st, m.dst) == 1 && !(m.dst->flags & GGML_TENSOR_FLAG_OUTPUT) && (!m.add || arows == 1) && (N % epl.a.head_dim) == 0 &&
                   mmq_skinny_supported(w->type, K, N, M, (int64_t) w->nb[1]) &&
                   N == (r == 0 ? (int64_t) epl.a.nh : (int64_t) epl.a.nkv) * epl.a.head_dim;
        };
        for (size_t q = 0; q < ms.size() && ok; ++q) ok = member_ok(ms[q], mats[q].N);
        for (auto & o : others) ok = ok && member_ok(o, g->nodes[o.k]->src[0]->ne[1]);
        // the rotation's (cos, sin) per (token, pair): the same for every layer of this graph run — computed once, by the first epilogue
        if (ok) {
            epi.tab = ensure_rope_table(st, epl.a.pos, epl.a.ff, epl.a.p, (int) M);
            ok = epi.tab != nullptr;
        }
        if (ok) {
            rope_host_consts(epl.a.p, epi.theta_scale, epi.corr0, epi.corr1);
            epi.freq_scale = epl.a.p.freq_scale;
            epi.ext_factor = epl.a.p.ext_factor;
            epi.attn_factor = epl.a.p.attn_factor;
            epi.pos = epl.a.pos;
            epi.ff = epl.a.ff;
            epi.idx = epl.a.idx;
            epi.v_idx = epl.a.v_idx;
            epi.head_dim = epl.a.head_dim;
            epi.n_dims = epl.a.p.n_dims;
            ks = 1;
        } else
            epi_node = -1;
    }
    auto epi_fill = [&](mmq_epi & e, int slot, int role) {  // role: 0 q, 1 k, 2 v
        e.kind[slot] = role == 0 ? 1 : (role == 1 ? 2 : (epl.a.v_idx ? 4 : 3));
        e.out[slot] = role == 0 ? epl.a.q_dst : (role == 1 ? epl.a.k_cache : epl.a.v_cache);
        e.nb1[slot] = role == 0 ? epl.a.qd_nb1 : (role == 1 ? epl.a.kc_nb1 : epl.a.vc_nb1);
        e.nb2[slot] = role == 0 ? epl.a.qd_nb2 : 0;
    };
    if ((size_t) ks * (size_t) M * (size_t) n_total * sizeof(float) > c->ws_size - st.aux_off && ks > 1) return 0;
    const void * act = quantized_src1(st, X, type);
    for (size_t q = 1; q < ms.size(); ++q)
        for (int d = 0; d < ms[q].n_nodes; ++d) { mark_done(st, ms[q].k + d); c->st.fused_nodes++; }
    for (auto & o : others)
        for (int d = 0; d < o.n_nodes; ++d) mark_done(st, o.k + d);
    // K split and the results are read only by the rope + cache-store kernel that follows (attention projections of a small batch):
    // that kernel sums the partial products itself — no reduce pass, and the f32 projections are never written
    bool defer = false;
    int jr = -1;
    if (ks > 1 && c->opt.fusion && st.rs_sk.node < 0 && epi_node < 0) {
        for (int k = i + ms[0].n_nodes; k < std::min(g->n_nodes, i + 24) && jr < 0; ++k)
            if (!st.done[k] && !is_view_op(g->nodes[k])) jr = k;
        rope_store_plan pl;
        if (jr >= 0 && plan_rope_store(st, jr, pl)) {
            defer = true;
            for (size_t q = 0; q < ms.size() && defer; ++q) {
                bool mapped = false;
                for (int sidx = 0; sidx < 3; ++sidx) mapped = mapped || through_views(pl.src[sidx]) == ms[q].dst;
                const int64_t arows = ms[q].add ? ms[q].add->ne[1] * ms[q].add->ne[2] * ms[q].add->ne[3] : 0;
                // every reader of the projection must be that kernel (one reshape in between), and a folded ADD must be a bias row
                defer = mapped && use_count(st, ms[q].dst) == 1 && !(ms[q].dst->flags & GGML_TENSOR_FLAG_OUTPUT) && (!ms[q].add || arows == 1) &&
                        (mats[q].N % pl.a.head_dim) == 0;
            }
        }
    }
    float * part = (float *) ((char *) c->ws + st.aux_off);
    {
        timed_scope ts(c, (std::string("mmq_") + type_tag(type) + (mixed ? std::string("+") + type_tag(g->nodes[ms.back().k]->src[0]->type) : std::string()) + (ms.size() == 3 ? "_x3" : "_x2") + "_n" +
                           std::to_string(n_total) + "_k" + std::to_string(K)).c_str(), wbytes, true);
        if (epi_node >= 0) {
            auto role_of2 = [&](const member & m) {
                for (int sidx = 0; sidx < 3; ++sidx)
                    if (through_views(epl.src[sidx]) == m.dst) return sidx;
                return -1;
            };
            for (size_t q = 0; q < ms.size(); ++q) epi_fill(epi, (int) q, role_of2(ms[q]));
        }
        const int served = launch_mmq_i8_multi(c->stream, type, (int) ms.size(), mats, (int) K, (int) M, act, c->opt.mmq_bn, ks, part, !defer, c->opt.mmq_skinny, epi_node >= 0 ? &epi : nullptr);
        c->st.skinny_launches += served == 1;
        c->st.wide_launches += served == 2;
            c->st.tiled_launches += served == 0;
    }
    c->st.kernel_launches += (ks > 1 && !defer) ? 2 : 1;
    c->st.fused_nodes += ms[0].n_nodes - 1;
    size_t aux_bump = 0;
    if (defer) {
        st.rs_sk.node = jr;
        st.rs_sk.n = (int) ms.size();
        st.rs_sk.ks = ks;
        st.rs_sk.M = (int) M;
        st.rs_sk.part = part;
        for (size_t q = 0; q < ms.size(); ++q) { st.rs_sk.mats[q] = mats[q]; st.rs_sk.dst[q] = ms[q].dst; }
        aux_bump = ((size_t) ks * (size_t) M * (size_t) n_total * sizeof(float) + 255) & ~(size_t) 255;  // the partials stay live: later launches use the space behind them
    }
    st.aux_off += aux_bump;
    for (auto & o : others) {
        if (epi_node >= 0) {
            int role = -1;
            for (int sidx = 0; sidx < 3; ++sidx)
                if (through_views(epl.src[sidx]) == o.dst) role = sidx;
            st.epi = epi;
            for (int q = 0; q < 3; ++q) st.epi.kind[q] = 0;
            epi_fill(st.epi, 0, role);
            st.epi_dst = o.dst;
        }
        const bool ok_o = run_mul_mat_q(st, g->nodes[o.k]->src[0], nullptr, X, o.dst, o.add, nullptr);
        st.epi_dst = nullptr;
        if (!ok_o) { st.aux_off -= aux_bump; return -1; }
        c->st.fused_nodes += o.n_nodes - 1;
    }
    st.aux_off -= aux_bump;
    if (epi_node >= 0) {  // ROPE(q), ROPE(k), SET_ROWS(k), SET_ROWS(v) happened in the epilogues
        mark_done(st, epi_node);
        for (int k : epl.nodes) mark_done(st, k);
        c->st.fused_nodes += 4;
        c->st.rope_epilogues++;
    }
    return ms[0].n_nodes;
}

// Q8_0 weights, 9 .. 128 columns (round 6): wq / wk / wv (and gate / up) of a batch multiply the same activations — one launch of the weight-streaming matrix-core
// kernel over the concatenated 32-row panels (mmq_q80.hip), bias ADDs folded into the store.  The member search is try_merge_mm_batch's (same dependence rules);
// no K split, no epilogues: with the K / V projections out of the way the ropes and cache stores that follow fuse as they do for the K-quants (try_fuse_rope_store).
static int try_merge_q80_skinny(exec_state & st, int i) {  // returns the number of nodes consumed at position i (0: not merged)
    backend_ctx * c = st.c;
    ggml_cgraph * g = st.g;
    ggml_tensor * n0 = g->nodes[i];
    const ggml_tensor * X = n0->src[1];
    const int64_t K = n0->src[0]->ne[0], M = X->ne[1] * X->ne[2] * X->ne[3];
    struct member { int k; ggml_tensor * dst; const ggml_tensor * add; int n_nodes; };
    auto eligible = [&](const ggml_tensor * t) {
        const ggml_tensor * w = t->src[0];
        return t->op == GGML_OP_MUL_MAT && t->src[1] == X && w->type == GGML_TYPE_Q8_0 && w->ne[0] == K && w->ne[2] == 1 && w->ne[3] == 1 && rows_contig(w) && !buffer_is_split(w->buffer) &&
               mmq_q80_skinny_supported(w->type, K, w->ne[1], M) && t->type == GGML_TYPE_F32 && ggml_abi_is_contiguous(t) && !(tp_active(c) && buffer_is_rowpar(w->view_src ? w->view_src->buffer : w->buffer));
    };
    auto with_add = [&](int k) {  // member at node k, with the ADD that directly follows it folded in
        ggml_tensor * t = g->nodes[k];
        ggml_tensor * a1 = k + 1 < g->n_nodes ? g->nodes[k + 1] : nullptr;
        const ggml_tensor * o1 = (a1 && !st.done[k + 1] && single_use(st, t)) ? add_partner(a1, t) : nullptr;
        if (o1 && ggml_abi_is_contiguous(a1) && a1->type == GGML_TYPE_F32) {
            bool ready = true;  // (the sibling runs early, at node i: what it adds must exist by then)
            for (int j = i; j <= k && ready; ++j) ready = !ranges_overlap(g->nodes[j], o1) || is_view_op(g->nodes[j]);
            if (ready) return member{k, a1, o1, 2};
        }
        return member{k, t, nullptr, 1};
    };
    if (!eligible(n0)) return 0;
    std::vector<member> ms{with_add(i)};
    const int limit = std::min(g->n_nodes, i + 24);
    for (int k = i + ms[0].n_nodes; k < limit && ms.size() < 3; ++k) {
        ggml_tensor * t = g->nodes[k];
        if (st.done[k] || !eligible(t)) continue;
        const member m = with_add(k);
        bool ok = true;
        for (auto & o : ms) ok = ok && !ranges_overlap(m.dst, o.dst) && !ranges_overlap(m.dst, g->nodes[o.k]);
        for (int j = i + 1; j < k && ok; ++j) {  // everything that runs between node i and the sibling's own position
            const ggml_tensor * u = g->nodes[j];
            bool is_member = false;
            for (auto & o : ms) is_member = is_member || j == o.k || (o.n_nodes == 2 && j == o.k + 1);
            if (!is_view_op(u) && !is_member && ranges_overlap(m.dst, u)) ok = false;
            for (int sidx = 0; sidx < GGML_MAX_SRC && ok; ++sidx)
                if (u->src[sidx] && ranges_overlap(m.dst, u->src[sidx])) ok = false;
        }
        if (ok && m.add && ranges_overlap(m.dst, m.add) && m.add->data != m.dst->data) ok = false;
        if (ok) ms.push_back(m);
    }
    if (ms.size() < 2) return 0;
    mmq80s_desc mats[3];
    double wbytes = 0;
    int64_t n_total = 0;
    int with_copy = 0;
    for (size_t q = 0; q < ms.size(); ++q) {
        const ggml_tensor * w = g->nodes[ms[q].k]->src[0];
        const ggml_tensor * add = ms[q].add;
        const int64_t arows = add ? add->ne[1] * add->ne[2] * add->ne[3] : 0;
        mats[q] = {(const uint8_t *) w->data, q80_panel_copy(c, w), (int64_t) w->nb[1], (int) w->ne[1], (float *) ms[q].dst->data, (int64_t) (ms[q].dst->nb[1] / 4),
                   add ? (const float *) add->data : nullptr, (!add || arows == 1) ? 0 : (int64_t) (add->nb[1] / 4)};
        with_copy += mats[q].W_panels != nullptr;
        n_total += w->ne[1];
        wbytes += (double) ggml_abi_row_size(w->type, K) * (double) w->ne[1];
    }
    if (with_copy != 0 && with_copy != (int) ms.size()) return 0;  // (one template form per launch: all from their panel copies, or none)
    const void * act = quantized_src1(st, X, MI_ACT_Q80_PANEL);
    for (size_t q = 1; q < ms.size(); ++q)
        for (int d = 0; d < ms[q].n_nodes; ++d) { mark_done(st, ms[q].k + d); c->st.fused_nodes++; }
    {
        timed_scope ts(c, (std::string("mmq_q8_0_skinny") + (ms.size() == 3 ? "_x3" : "_x2") + "_n" + std::to_string(n_total) + "_k" + std::to_string(K)).c_str(), wbytes, true);
        launch_mmq_q80_skinny_multi(c->stream, (int) ms.size(), mats, (int) K, (int) M, act);
    }
    c->st.decode_copy_launches += with_copy != 0;
    c->st.skinny_launches++;
    c->st.kernel_launches++;
    c->st.fused_nodes += ms[0].n_nodes - 1;
    return ms[0].n_nodes;
}

// ------------------------------------------------------------------------------------------------ batches: rope + cache stores
// At ROPE(q) of a batch: ROPE(k) with the same parameters and positions, SET_ROWS(k cache <- rope(k)) and SET_ROWS(v cache <- v)
// with one index vector follow (llama.cpp's build_attn: q, k, v expanded together, then cpy_k / cpy_v) — four launches that
// each sit at the launch floor for a few dozen tokens.  When only views (and the K / V projections already executed with the Q
// projection: try_merge_mm_batch) sit between them, they run as one (ops.hip: k_rope_qk_store); absorbed nodes are flagged done.
static const ggml_tensor * through_views(const ggml_tensor * t) {
    while (t && (t->op == GGML_OP_RESHAPE || t->op == GGML_OP_VIEW) && t->src[0] && t->data == t->src[0]->data) t = t->src[0];
    return t;
}
static bool plan_rope_store(const exec_state & st, int i, rope_store_plan & pl) {
    ggml_cgraph * g = st.g;
    const ggml_tensor * rq = g->nodes[i];
    if (rq->op != GGML_OP_ROPE) return false;
    const ggml_tensor * q = rq->src[0];
    if (rq->type != GGML_TYPE_F32 || !ggml_abi_is_contiguous(rq) || q->type != GGML_TYPE_F32 || q->nb[0] != 4 || rq->ne[3] != 1 || rq->ne[2] < 2) return false;
    if ((rq->op_params[2] & GGML_ROPE_TYPE_MROPE) || rq->src[1]->type != GGML_TYPE_I32 || !ggml_abi_is_contiguous(rq->src[1])) return false;
    const int limit = std::min(g->n_nodes, i + 16);
    int jk = -1, jks = -1, jvs = -1;
    for (int k = i + 1; k < limit; ++k) {
        const ggml_tensor * t = g->nodes[k];
        if (st.done[k] || is_view_op(t)) continue;
        if (t->op == GGML_OP_ROPE && jk < 0) { jk = k; continue; }
        if (t->op == GGML_OP_SET_ROWS && jk >= 0 && jks < 0 && through_views(t->src[0]) == g->nodes[jk]) { jks = k; continue; }
        if (t->op == GGML_OP_SET_ROWS && jks >= 0 && jvs < 0) { jvs = k; break; }
        return false;  // anything else inside the window: leave the nodes alone
    }
    if (jk < 0 || jks < 0 || jvs < 0) return false;
    const ggml_tensor * rk = g->nodes[jk];
    const ggml_tensor * ks = g->nodes[jks];
    const ggml_tensor * vs = g->nodes[jvs];
    const ggml_tensor * ksrc = rk->src[0];
    const ggml_tensor * vsrc = vs->src[0];
    if (memcmp(rk->op_params, rq->op_params, sizeof(rq->op_params)) != 0 || rk->src[1] != rq->src[1] || rk->src[2] != rq->src[2]) return false;
    if (rk->type != GGML_TYPE_F32 || ksrc->type != GGML_TYPE_F32 || ksrc->nb[0] != 4 || rk->ne[0] != rq->ne[0] || rk->ne[2] != rq->ne[2] || rk->ne[3] != 1) return false;
    if (use_count(st, rk) != 1 || (rk->flags & GGML_TENSOR_FLAG_OUTPUT)) return false;  // rope(k) lives only in the cache
    const int64_t HD = rq->ne[0], NH = rq->ne[1], NKV = rk->ne[1], T = rq->ne[2];
    // cache stores: f16 rows of NKV * HD values, one I64 index per token, both stores through the same indices
    auto store_ok = [&](const ggml_tensor * sr, const ggml_tensor * src) {
        return sr->type == GGML_TYPE_F16 && sr->nb[0] == 2 && sr->ne[0] == NKV * HD && src->ne[0] == NKV * HD && ggml_abi_nelements(src) == NKV * HD * T &&
               sr->src[1]->type == GGML_TYPE_I64 && ggml_abi_nelements(sr->src[1]) == T && ggml_abi_is_contiguous(sr->src[1]);
    };
    // ... or, on the non-flash path, V transposed: the projection viewed as rows of ONE element, an index per element
    const bool v_scatter = vs->type == GGML_TYPE_F16 && vs->ne[0] == 1 && vs->nb[0] == 2 && vs->nb[1] == 2 && vsrc->ne[0] == 1 && ggml_abi_nelements(vsrc) == NKV * HD * T &&
                           vs->src[1]->type == GGML_TYPE_I64 && ggml_abi_nelements(vs->src[1]) == NKV * HD * T && ggml_abi_is_contiguous(vs->src[1]);
    if (!store_ok(ks, ks->src[0])) return false;
    if (!v_scatter && (!store_ok(vs, vsrc) || ks->src[1]->data != vs->src[1]->data)) return false;
    if (vsrc->type != GGML_TYPE_F32 || !ggml_abi_is_contiguous(vsrc) || !ggml_abi_is_contiguous(ksrc) || ksrc->ne[0] != HD || ksrc->ne[1] != NKV) return false;
    if ((rq->op_params[1] % 2) != 0 || rq->op_params[1] > HD) return false;
    rope_store_args a{};
    a.q_src = (const char *) q->data;
    a.q_dst = (char *) rq->data;
    a.k_src = (const char *) ksrc->data;
    a.v_src = (const char *) vsrc->data;
    a.q_nb1 = (int64_t) q->nb[1]; a.q_nb2 = (int64_t) q->nb[2];
    a.qd_nb1 = (int64_t) rq->nb[1]; a.qd_nb2 = (int64_t) rq->nb[2];
    a.k_nb1 = (int64_t) ksrc->nb[1]; a.k_nb2 = (int64_t) ksrc->nb[2];
    a.v_nb1 = HD * 4; a.v_nb2 = NKV * HD * 4;
    a.k_cache = (char *) ks->data; a.v_cache = (char *) vs->data;
    a.kc_nb1 = (int64_t) ks->nb[1]; a.vc_nb1 = (int64_t) vs->nb[1];
    a.idx = (const int64_t *) ks->src[1]->data;
    a.v_idx = v_scatter ? (const int64_t *) vs->src[1]->data : nullptr;
    a.pos = (const int32_t *) rq->src[1]->data;
    a.ff = rq->src[2] ? (const float *) rq->src[2]->data : nullptr;
    a.p.n_dims = rq->op_params[1];
    a.p.mode = rq->op_params[2];
    a.p.n_ctx_orig = rq->op_params[4];
    a.p.freq_base = ggml_abi_op_param_f32(rq, 5);
    a.p.freq_scale = ggml_abi_op_param_f32(rq, 6);
    a.p.ext_factor = ggml_abi_op_param_f32(rq, 7);
    a.p.attn_factor = ggml_abi_op_param_f32(rq, 8);
    a.p.beta_fast = ggml_abi_op_param_f32(rq, 9);
    a.p.beta_slow = ggml_abi_op_param_f32(rq, 10);
    memset(a.p.sections, 0, sizeof(a.p.sections));
    a.nh = (int) NH; a.nkv = (int) NKV; a.head_dim = (int) HD;
    pl.a = a;
    pl.T = (int) T;
    pl.nodes[0] = jk; pl.nodes[1] = jks; pl.nodes[2] = jvs;
    pl.src[0] = q; pl.src[1] = ksrc; pl.src[2] = vsrc;
    return true;
}
static void flush_deferred_qkv(exec_state & st);
static bool try_fuse_rope_store(exec_state & st, int i) {
    backend_ctx * c = st.c;
    rope_store_plan pl;
    if (!plan_rope_store(st, i, pl)) {
        flush_deferred_qkv(st);
        return false;
    }
    if (st.rs_sk.node == i) {
        // the projections' split-K partial products were left unsummed for this kernel (try_merge_mm_batch): wire them in
        const float * pp = st.rs_sk.part;
        for (int q = 0; q < st.rs_sk.n; ++q) {
            for (int sidx = 0; sidx < 3; ++sidx) {
                if (through_views(pl.src[sidx]) != st.rs_sk.dst[q]) continue;
                pl.a.sk[sidx].part = pp;
                pl.a.sk[sidx].mn = (int64_t) st.rs_sk.M * st.rs_sk.mats[q].N;
                pl.a.sk[sidx].bias = st.rs_sk.mats[q].add;
                pl.a.sk[sidx].n = st.rs_sk.mats[q].N;
            }
            pp += (size_t) st.rs_sk.ks * st.rs_sk.M * st.rs_sk.mats[q].N;
        }
        pl.a.ks = st.rs_sk.ks;
        st.rs_sk.node = -1;
    }
    timed_scope ts(c, "rope_qk_store", (double) ggml_abi_nbytes(st.g->nodes[i]) * 2);
    const float * tab = rope_qk_store_vec_ok(pl.a, pl.T) ? ensure_rope_table(st, pl.a.pos, pl.a.ff, pl.a.p, pl.T) : nullptr;
    if (tab) launch_rope_qk_store_vec(c->stream, pl.a, pl.T, tab);
    else launch_rope_qk_store(c->stream, pl.a, pl.T);
    c->st.kernel_launches++;
    for (int k : pl.nodes) { mark_done(st, k); c->st.fused_nodes++; }
    return true;
}
// the reader the unsummed projections were left for did not materialise: run their reduce pass now
static void flush_deferred_qkv(exec_state & st) {
    if (st.rs_sk.node < 0) return;
    launch_splitk_reduce_mats(st.c->stream, st.rs_sk.n, st.rs_sk.mats, st.rs_sk.part, st.rs_sk.ks, st.rs_sk.M);
    st.c->st.kernel_launches++;
    st.rs_sk.node = -1;
}

// a deferred split-K sum that its designated reader did not pick up after all: run the plain reduce pass now
static void flush_deferred_splitk(exec_state & st) {
    if (!st.sk_dst) return;
    launch_splitk_reduce(st.c->stream, st.sk.part, st.sk.ks, (int) (st.sk.mn / st.sk_dst->ne[0]), (int) st.sk_dst->ne[0], st.sk.out, st.sk.out_stride, st.sk.add, st.sk.add_stride);
    st.c->st.kernel_launches++;
    st.sk_dst = nullptr;
}

// executes node i (possibly fusing followers); returns number of nodes consumed, or -1 on failure
// ------------------------------------------------------------------------------------------------ non-flash attention of a small batch
// kq = MUL_MAT(K view, q) -> SOFT_MAX(kq, mask, scale) -> MUL_MAT(V^T view, p) for 2..32 tokens (llama-box's default path, -fa off, in a
// `-np` decode step): one launch over the tokens' lists of visible cells (attn_nf.hip) instead of three dense ones.  At node i = kq.
static bool try_fuse_attn_nf(exec_state & st, int i) {
    backend_ctx * c = st.c;
    ggml_cgraph * g = st.g;
    const ggml_tensor * kq = g->nodes[i];
    const ggml_tensor * K = kq->src[0], * Q = kq->src[1];
    if (!c->fa_lists || K->type != GGML_TYPE_F16 || Q->type != GGML_TYPE_F32 || Q->ne[1] < 2 || Q->ne[1] > 32 || !single_use(st, kq)) return false;
    if (buffer_is_split(K->buffer)) return false;
    auto next_real = [&](int from) {
        for (int k = from; k < std::min(g->n_nodes, from + 6); ++k) {
            if (st.done[k] || is_view_op(g->nodes[k])) continue;
            return k;
        }
        return -1;
    };
    const int js = next_real(i + 1);
    if (js < 0) return false;
    const ggml_tensor * sm = g->nodes[js];
    if (sm->op != GGML_OP_SOFT_MAX || sm->src[0] != kq || !sm->src[1] || sm->src[2] || ggml_abi_op_param_f32(sm, 1) != 0.0f || !single_use(st, sm)) return false;
    const int jv = next_real(js + 1);
    if (jv < 0) return false;
    const ggml_tensor * kqv = g->nodes[jv];
    if (kqv->op != GGML_OP_MUL_MAT || kqv->src[1] != sm || kqv->src[0]->type != GGML_TYPE_F16 || kqv->type != GGML_TYPE_F32 || !ggml_abi_is_contiguous(kqv)) return false;
    const ggml_tensor * V = kqv->src[0], * M = sm->src[1];
    if (buffer_is_split(V->buffer)) return false;
    // the position lists come from the mask (fattn.hip: k_fattn_pos_scan reads four cells per thread); llama.cpp's mask is f32 when flash
    // attention is off
    const int malign = M->type == GGML_TYPE_F32 ? 16 : 8;
    if ((M->type != GGML_TYPE_F16 && M->type != GGML_TYPE_F32) || (K->ne[1] % 4) != 0 || (M->nb[1] % malign) != 0 || ((uintptr_t) M->data & (malign - 1)) != 0 ||
        M->ne[0] < K->ne[1] || M->ne[1] < Q->ne[1] || M->ne[2] != 1 || M->ne[3] != 1)
        return false;
    if ((size_t) (Q->ne[1] * (K->ne[1] + 1)) * sizeof(int) > c->fa_lists_bytes) return false;
    {
        // The list kernel serves a token's first 1024 visible cells from registers / LDS and anything beyond through scratch memory (correct,
        // ~100 us per layer).  When the mask passed through set_tensor the LONGEST row is known exactly (common.h: mask_stats): one 20 k-cell
        // sequence among 31 short ones keeps the batch on the dense kernels (ADVICE r02 / VERDICT r03 #7).  Unknown mask: the average rule
        // inside attn_nf_list_scratch_bytes alone decides, as before.
        mask_stats ms;
        if (lookup_mask_stats(mask_upload_ptr(M), &ms) && ms.max_visible > 1024) return false;
    }
    const tdesc qd = TD(Q), kd = TD(K), vd = TD(V), md = TD(M);
    int dq_n = 1;
    (void) attn_nf_list_scratch_bytes(qd, kd, &dq_n);
    // llm_build_* continues with CONT(PERMUTE(kqv, 0, 2, 1, 3)) -> [D * NH, T]: when that copy is the result's only reader, the rows go
    // straight to its memory in that order (one launch less)
    tdesc od = TD(kqv);
    int jc = -1;
    if (single_use(st, kqv)) {
        const int jn = next_real(jv + 1);
        if (jn >= 0) {
            const ggml_tensor * ct = g->nodes[jn];
            const ggml_tensor * pv = ct->src[0];
            if (ct->op == GGML_OP_CONT && pv && pv->op == GGML_OP_PERMUTE && pv->src[0] == kqv && pv->data == kqv->data && ct->type == GGML_TYPE_F32 && ggml_abi_is_contiguous(ct) &&
                pv->ne[0] == kqv->ne[0] && pv->ne[1] == kqv->ne[2] && pv->ne[2] == kqv->ne[1] && pv->ne[3] == 1 && ggml_abi_nelements(ct) == ggml_abi_nelements(kqv)) {
                const int64_t tok_nb = (int64_t) kqv->ne[0] * kqv->ne[2] * 4, head_nb = (int64_t) kqv->ne[0] * 4;
                // The copy's block is usually q's recycled one (same size, q dies at K.q).  That is safe exactly when it is the SAME rows:
                // a workgroup (token, KV head) then overwrites only the q rows it alone reads, after it has read them — which needs one
                // workgroup per (token, KV head), i.e. no slices of the head dimensions.
                const bool same_rows = ct->data == Q->data && Q->nb[0] == 4 && (int64_t) Q->nb[1] == tok_nb && (int64_t) Q->nb[2] == head_nb && dq_n == 1;
                // ct's block is written at node i, before the allocator's lifetime for it begins (node jn): besides q, the launch reads the
                // mask and the K / V views — it must alias none of them (kq, the soft-max and kqv themselves are never materialised here)
                const bool clear_of_inputs = !ranges_overlap(ct, M) && !ranges_overlap(ct, K) && !ranges_overlap(ct, V);
                if (clear_of_inputs && (!ranges_overlap(ct, Q) || same_rows)) {
                    jc = jn;
                    od.data = (char *) ct->data;
                    od.nb[0] = 4;
                    od.nb[1] = tok_nb;
                    od.nb[2] = head_nb;
                }
            }
        }
    }
    // workgroups store rows of the result while others still read q: written to its own tensor, the result must not sit in q's recycled block
    if (jc < 0 && ranges_overlap(kqv, Q)) return false;
    float * scratch = c->ws ? (float *) ((char *) c->ws + st.aux_off) : nullptr;
    const size_t scratch_bytes = c->ws ? c->ws_size - st.aux_off : 0;
    const size_t need = attn_nf_list_scratch_bytes(qd, kd, nullptr);
    if (need == 0 || need > scratch_bytes) return false;
    if (st.fa_list_mask != M->data || st.fa_list_tile != 1) {  // first attention of this graph run: list every token's visible cells
        launch_fattn_tile_scan(c->stream, md, (int) Q->ne[1], (int) K->ne[1], 1, c->fa_lists);
        c->st.kernel_launches++;
        st.fa_list_mask = M->data;
        st.fa_list_tile = 1;
    }
    // the copy is read only by quantised mat-muls (wo): the rows leave the launch as Q8_K blocks in the activation scratch (no f32, no quantiser launch)
    void * q8_out = nullptr;
    if (jc >= 0 && dq_n == 1 && (Q->ne[2] / K->ne[2]) % 2 == 0 && c->opt.prologue && !(g->nodes[jc]->flags & GGML_TENSOR_FLAG_OUTPUT) && quant_consumers_only(st, jc, g->nodes[jc]))
        q8_out = (char *) c->ws + st.act_off;
    timed_scope ts(c, "attn_nf_list", (double) ggml_abi_nbytes(kqv));
    if (!launch_attn_nf_list(c->stream, qd, kd, vd, md, od, c->fa_lists, (int) K->ne[1] + 1, scratch, scratch_bytes, ggml_abi_op_param_f32(sm, 0), q8_out)) return false;
    if (q8_out) {
        mark_q8_cache(st, g->nodes[jc]);
        c->st.fused_nodes++;
    }
    c->st.kernel_launches++;
    mark_done(st, js);
    mark_done(st, jv);
    c->st.fused_nodes += 2;
    if (jc >= 0) {
        mark_done(st, jc);
        c->st.fused_nodes++;
    }
    return true;
}

// The same chain for a batch of 33 tokens and more (prompt micro-batches, speculative-decoding batches): two passes of the matrix-core kernel k_attn_nf_mma (fattn_mma.hip) — the row maxima
// and sums, then p = e / sum and V^T.p — instead of three dense launches over the [n_kv, tokens, heads] score matrix (round 4).  At node i = kq.
static bool try_fuse_attn_nf_mma(exec_state & st, int i) {
    backend_ctx * c = st.c;
    ggml_cgraph * g = st.g;
    const ggml_tensor * kq = g->nodes[i];
    const ggml_tensor * K = kq->src[0], * Q = kq->src[1];
    if (!c->fa_lists || !c->ws || K->type != GGML_TYPE_F16 || Q->type != GGML_TYPE_F32 || Q->ne[1] < fattn_mma_min_q() || !single_use(st, kq)) return false;
    if (buffer_is_split(K->buffer)) return false;
    auto next_real = [&](int from) {
        for (int k = from; k < std::min(g->n_nodes, from + 6); ++k) {
            if (st.done[k] || is_view_op(g->nodes[k])) continue;
            return k;
        }
        return -1;
    };
    const int js = next_real(i + 1);
    if (js < 0) return false;
    const ggml_tensor * sm = g->nodes[js];
    if (sm->op != GGML_OP_SOFT_MAX || sm->src[0] != kq || !sm->src[1] || sm->src[2] || ggml_abi_op_param_f32(sm, 1) != 0.0f || !single_use(st, sm)) return false;
    const int jv = next_real(js + 1);
    if (jv < 0) return false;
    const ggml_tensor * kqv = g->nodes[jv];
    if (kqv->op != GGML_OP_MUL_MAT || kqv->src[1] != sm || kqv->src[0]->type != GGML_TYPE_F16 || kqv->type != GGML_TYPE_F32 || !ggml_abi_is_contiguous(kqv)) return false;
    const ggml_tensor * V = kqv->src[0], * M = sm->src[1];
    if (buffer_is_split(V->buffer) || M->ne[2] != 1 || M->ne[3] != 1) return false;
    const tdesc qd = TD(Q), kd = TD(K), vd = TD(V), md = TD(M);
    if (!attn_nf_mma_applies(qd, kd, vd, md) || fattn_vis_bytes(qd, kd) > c->fa_lists_bytes) return false;
    // where the rows go: kqv [D, T, NH], or — when CONT(PERMUTE(kqv, 0, 2, 1, 3)) is its only reader — that copy [D * NH, T] (try_fuse_attn_nf has the
    // aliasing argument: the copy is usually q's recycled block and holds the SAME rows; a workgroup reads its queries before anything is written)
    tdesc od = TD(kqv);
    od.nb[1] = kqv->nb[2];  // (the kernel's dst: nb[1] = head stride, nb[2] = token stride)
    od.nb[2] = kqv->nb[1];
    int jc = -1;
    if (single_use(st, kqv)) {
        const int jn = next_real(jv + 1);
        if (jn >= 0) {
            const ggml_tensor * ct = g->nodes[jn];
            const ggml_tensor * pv = ct->src[0];
            if (ct->op == GGML_OP_CONT && pv && pv->op == GGML_OP_PERMUTE && pv->src[0] == kqv && pv->data == kqv->data && ct->type == GGML_TYPE_F32 && ggml_abi_is_contiguous(ct) &&
                pv->ne[0] == kqv->ne[0] && pv->ne[1] == kqv->ne[2] && pv->ne[2] == kqv->ne[1] && pv->ne[3] == 1 && ggml_abi_nelements(ct) == ggml_abi_nelements(kqv)) {
                const int64_t tok_nb = (int64_t) kqv->ne[0] * kqv->ne[2] * 4, head_nb = (int64_t) kqv->ne[0] * 4;
                const bool same_rows = ct->data == Q->data && Q->nb[0] == 4 && (int64_t) Q->nb[1] == tok_nb && (int64_t) Q->nb[2] == head_nb;
                const bool clear_of_inputs = !ranges_overlap(ct, M) && !ranges_overlap(ct, K) && !ranges_overlap(ct, V);
                if (clear_of_inputs && (!ranges_overlap(ct, Q) || same_rows) && (((uintptr_t) ct->data) & 15) == 0) {
                    jc = jn;
                    od.data = (char *) ct->data;
                    od.nb[0] = 4;
                    od.nb[1] = head_nb;
                    od.nb[2] = tok_nb;
                }
            }
        }
    }
    if (jc < 0 && (ranges_overlap(kqv, Q) || (((uintptr_t) kqv->data) & 15) != 0)) return false;
    const int n_splits = fattn_mma_pick_splits(qd, kd);
    const size_t need = attn_nf_mma_ws_bytes(qd, n_splits);
    if (need > c->ws_size - st.aux_off) return false;
    if (st.fa_list_mask != M->data || st.fa_list_tile != -1) {  // first attention of this graph run: the (query tile, kv tile) states
        launch_fattn_vis_scan(c->stream, md, (int) Q->ne[1], (int) K->ne[1], (uint8_t *) c->fa_lists);
        c->st.kernel_launches++;
        st.fa_list_mask = M->data;
        st.fa_list_tile = -1;
    }
    // the copy is read only by quantised mat-muls (wo) and the splits' sum is the row-parallel combine: it writes the Q8_K blocks itself
    void * q8_out = nullptr;
    if (jc >= 0 && n_splits > 1 && (Q->ne[2] % 2) == 0 && c->opt.prologue && !(g->nodes[jc]->flags & GGML_TENSOR_FLAG_OUTPUT) && quant_consumers_only(st, jc, g->nodes[jc]) &&
        fattn_combine_rows_applies(128, Q->ne[1], Q->ne[2], 1, n_splits, nullptr))
        q8_out = (char *) c->ws + st.act_off;
    timed_scope ts(c, "attn_nf_mma", (double) ggml_abi_nbytes(kqv));
    launch_attn_nf_mma(c->stream, qd, kd, vd, md, od, ggml_abi_op_param_f32(sm, 0), n_splits, (const uint8_t *) c->fa_lists, (char *) c->ws + st.aux_off, q8_out);
    if (q8_out) {
        mark_q8_cache(st, g->nodes[jc]);
        c->st.fused_nodes++;
    }
    c->st.kernel_launches += n_splits > 1 ? 3 : 2;
    c->st.nf_mma_chains++;
    mark_done(st, js);
    mark_done(st, jv);
    c->st.fused_nodes += 2;
    if (jc >= 0) {
        mark_done(st, jc);
        c->st.fused_nodes++;
    }
    return true;
}

static int run_node(exec_state & st, int i) {
    backend_ctx * c = st.c;
    ggml_cgraph * g = st.g;
    ggml_tensor * n = g->nodes[i];
    const ggml_tensor * a = n->src[0];
    const ggml_tensor * b = n->src[1];
    hipStream_t s = c->stream;
    const bool fuse = c->opt.fusion;
    auto next = [&](int k) -> ggml_tensor * { return i + k < g->n_nodes ? g->nodes[i + k] : nullptr; };

    for (int sidx = 0; sidx < GGML_MAX_SRC; ++sidx)
        if (n->src[sidx] && buffer_is_split(n->src[sidx]->buffer) && !(n->op == GGML_OP_MUL_MAT && sidx == 0)) {
            MI_ERR("graph_compute: node %d '%s' (op %d) reads '%s' from a row-split buffer: only MUL_MAT weights may live there", i, n->name, (int) n->op, n->src[sidx]->name);
            return -1;
        }
    // (a node that writes into the memory of the tensor whose sum of squares is on record — an in-place op — makes that record stale)
    if (st.ss_tensor && n != st.ss_tensor && !is_view_op(n) && ranges_overlap(n, st.ss_tensor)) st.ss_tensor = nullptr;
    if (st.sk_dst && !is_view_op(n) && !(n->op == GGML_OP_RMS_NORM && a == st.sk_dst)) flush_deferred_splitk(st);
    if (st.rs_sk.node >= 0 && st.rs_sk.node != i && !is_view_op(n)) flush_deferred_qkv(st);
    switch (n->op) {
        case GGML_OP_NONE: case GGML_OP_VIEW: case GGML_OP_RESHAPE: case GGML_OP_PERMUTE: case GGML_OP_TRANSPOSE:
            return 1;

        case GGML_OP_RMS_NORM: {
            // RMS_NORM -> MUL(norm, w): one kernel writing the MUL's output
            ggml_tensor * m = next(1);
            if (fuse && m && m->op == GGML_OP_MUL && single_use(st, n) && m->type == GGML_TYPE_F32 && m->nb[0] == 4) {
                const ggml_tensor * w = m->src[0] == n ? m->src[1] : (m->src[1] == n ? m->src[0] : nullptr);
                if (w && w->type == GGML_TYPE_F32 && ggml_abi_is_contiguous(w) && w->ne[0] == n->ne[0] && ggml_abi_nelements(w) == w->ne[0] && same_shape(m, n)) {
                    static const bool dbg_no_defer = getenv("GGML_MI355X_DBG_NO_DEFER_NORM") != nullptr;
                    static const char * dbg_only = getenv("GGML_MI355X_DBG_DEFER_ONLY");
                    if (c->opt.prologue && !dbg_no_defer && (!dbg_only || strstr(m->name, dbg_only)) && can_defer_norm(st, i, n, m, a, w)) {
                        st.deferred[m] = {a, w, ggml_abi_op_param_f32(n, 0), m};
                        c->st.fused_nodes += 2;
                        return 2;
                    }
                    if (c->opt.prologue && ggml_abi_nrows(m) > 1 && a->nb[0] == 4 && (a->nb[1] % 16) == 0 && a->ne[0] <= 16384 &&
                        !((((uintptr_t) a->data) | ((uintptr_t) w->data)) & 15) && quant_consumers_only(st, i + 1, m)) {
                        timed_scope ts(c, "rms_norm_mul_quantize", (double) ggml_abi_nbytes(a));
                        if (st.sk_dst == a) {  // the row is still split-K partial products: assemble it here
                            launch_splitk_rms_norm_mul_quantize(s, st.sk, (int) ggml_abi_nrows(a), (int) a->ne[0], (const float *) w->data, ggml_abi_op_param_f32(n, 0), (char *) c->ws + st.act_off);
                            st.sk_dst = nullptr;
                        } else
                        launch_rms_norm_mul_quantize(s, TD(a), (const float *) w->data, ggml_abi_op_param_f32(n, 0), (char *) c->ws + st.act_off);
                        mark_q8_cache(st, m);
                        c->st.kernel_launches++;
                        c->st.fused_nodes += 2;
                        return 2;
                    }
                    static const bool q80_producers = !getenv("GGML_MI355X_Q80_PRODUCERS") || atoi(getenv("GGML_MI355X_Q80_PRODUCERS")) != 0;
                    if (q80_producers && c->opt.prologue && ggml_abi_nrows(m) >= 9 && st.sk_dst != a && rms_norm_q80_panel_ok(TD(a), (const float *) w->data) && q80_panel_consumers_only(st, i + 1, m)) {
                        timed_scope ts(c, "rms_norm_mul_quantize_q8_0", (double) ggml_abi_nbytes(a));
                        launch_rms_norm_mul_q80_panel(s, TD(a), ggml_abi_op_param_f32(n, 0), (const float *) w->data, (char *) c->ws + st.act_off);
                        mark_q8_cache(st, m, MI_ACT_Q80_PANEL);
                        c->st.kernel_launches++;
                        c->st.fused_nodes += 2;
                        return 2;
                    }
                    flush_deferred_splitk(st);
                    const tdesc wd = TD(w);
                    timed_scope ts(c, "rms_norm_mul", (double) ggml_abi_nbytes(a) * 2);
                    launch_rms_norm(s, TD(a), TD(m), ggml_abi_op_param_f32(n, 0), &wd);
                    c->st.kernel_launches++;
                    c->st.fused_nodes++;
                    return 2;
                }
            }
            timed_scope ts(c, "rms_norm", (double) ggml_abi_nbytes(a) * 2);
            launch_rms_norm(s, TD(a), TD(n), ggml_abi_op_param_f32(n, 0), nullptr);
            c->st.kernel_launches++;
            return 1;
        }

        case GGML_OP_MUL_MAT: {
            if (mm_cache_image_ok(n) && !buffer_is_split(a->buffer)) {
                // K.q over a cache kept in a block format / bf16 (-fa off): the view goes through its f16 image, the product through the f16 kernel
                tdesc a16 = TD(a), none = TD(a);
                none.type = GGML_TYPE_F16;  // (launch_kv_images_f16 expands what is not f16: only `a16` here)
                const size_t off = (mul_mat_f_workspace_bytes(kv_image_desc(a16, nullptr), TD(b)) + 255) & ~(size_t) 255;
                timed_scope ts(c, "mul_mat_f_kv_image", (double) ggml_abi_nbytes(a));
                launch_kv_images_f16(s, a16, none, (char *) c->ws + st.aux_off + off);
                launch_mul_mat_f(s, a16, TD(b), TD(n), (float *) ((char *) c->ws + st.aux_off), off);
                c->st.kernel_launches += 2;
                c->st.kv_image_nodes++;
                return 1;
            }
            if (!is_quant(a->type) && fuse && c->opt.attn_nf && try_fuse_attn_nf(st, i)) return 1;
            if (!is_quant(a->type) && fuse && c->opt.attn_nf && try_fuse_attn_nf_mma(st, i)) return 1;
            if (!is_quant(a->type)) {
                timed_scope ts(c, "mul_mat_f", (double) ggml_abi_nbytes(a));
                launch_mul_mat_f(s, TD(a), TD(b), TD(n), (float *) ((char *) c->ws + st.aux_off), c->ws ? c->ws_size - st.aux_off : 0);
                c->st.kernel_launches++;
                return 1;
            }
            if (buffer_is_split(a->buffer)) {  // weights in the split buffer type (-sm row): split.cpp
                const split_tensor_info * si = split_info(a);
                if (!si) return -1;
                // the FFN on sharded weights: MUL_MAT(gate) MUL_MAT(up) GLU MUL_MAT(down) [ADD residual] as one sharded chain with ONE sum
                ggml_tensor * n2 = next(1), * n3 = next(2), * n4 = next(3);
                if (fuse && si->kind == 0 && n2 && n3 && n4 && n2->op == GGML_OP_MUL_MAT && n2->src[1] == b && buffer_is_split(n2->src[0]->buffer) &&
                    n3->op == GGML_OP_GLU && n3->op_params[0] == GGML_GLU_OP_SWIGLU && n3->op_params[1] == 0 && n3->src[0] == n && n3->src[1] == n2 &&
                    n4->op == GGML_OP_MUL_MAT && n4->src[1] == n3 && buffer_is_split(n4->src[0]->buffer) && single_use(st, n) && single_use(st, n2) && single_use(st, n3) &&
                    split_mul_mat_supported(n2) && ggml_abi_is_contiguous(n3) && ggml_abi_is_contiguous(n4) && n4->type == GGML_TYPE_F32 && split_ffn_applies(a, n2->src[0], n4->src[0])) {
                    ggml_tensor * a5 = next(4);
                    const ggml_tensor * o5 = (a5 && single_use(st, n4)) ? add_partner(a5, n4) : nullptr;
                    if (o5 && same_shape(o5, n4)) {
                        if (!run_split_ffn(c, a, n2->src[0], n4->src[0], b, a5, o5)) return -1;
                        c->st.fused_nodes += 4;
                        return 5;
                    }
                    if (!run_split_ffn(c, a, n2->src[0], n4->src[0], b, n4, nullptr)) return -1;
                    c->st.fused_nodes += 3;
                    return 4;
                }
                if (si->kind == 1) {  // row-parallel (attn_output, or ffn_down outside the chain above): K ranges out, partial sums back, + residual
                    ggml_tensor * a1 = next(1);
                    const ggml_tensor * o1 = (fuse && a1 && single_use(st, n)) ? add_partner(a1, n) : nullptr;
                    if (o1 && same_shape(o1, n)) {
                        if (!run_split_rowpar(c, a, b, a1, o1)) return -1;
                        c->st.fused_nodes += 1;
                        return 2;
                    }
                    if (!run_split_rowpar(c, a, b, n, nullptr)) return -1;
                    return 1;
                }
                if (!run_split_mul_mat(c, a, b, n)) return -1;  // column-parallel: broadcast, per-device rows, gather
                return 1;
            }
            const bool rowpar = tp_active(c) && buffer_is_rowpar(a->view_src ? a->view_src->buffer : a->buffer);
            const int64_t M = b->ne[1] * b->ne[2] * b->ne[3];
            if (fuse && c->opt.qkv && !rowpar && M == 1 && try_fuse_qkv(st, i)) return 1;
            if (fuse && !rowpar && M <= c->opt.mmvq_max_cols) {
                // gate/up/SwiGLU: MUL_MAT(Wg,x) MUL_MAT(Wu,x) GLU(g,u)
                ggml_tensor * n2 = next(1);
                ggml_tensor * n3 = next(2);
                if (n2 && n3 && quant_mm_ok(n2) && n2->src[1] == b && n2->src[0]->type == a->type && same_shape(n2->src[0], a) &&
                    n2->src[0]->nb[1] == a->nb[1] && n3->op == GGML_OP_GLU && n3->op_params[0] == GGML_GLU_OP_SWIGLU && n3->op_params[1] == 0 &&
                    n3->src[0] == n && n3->src[1] == n2 && single_use(st, n) && single_use(st, n2) && ggml_abi_is_contiguous(n3)) {
                    if (!run_mul_mat_q(st, a, n2->src[0], b, n3, nullptr, nullptr)) return -1;
                    c->st.fused_nodes += 2;
                    return 3;
                }
                // MUL_MAT -> ADD (bias or residual) [-> ADD]
                ggml_tensor * a1 = next(1);
                const ggml_tensor * o1 = (a1 && single_use(st, n)) ? add_partner(a1, n) : nullptr;
                if (o1) {
                    ggml_tensor * a2 = next(2);
                    const ggml_tensor * o2 = (a2 && single_use(st, a1)) ? add_partner(a2, a1) : nullptr;
                    if (o2) {
                        if (!run_mul_mat_q(st, a, nullptr, b, a2, o1, o2)) return -1;
                        c->st.fused_nodes += 2;
                        return 3;
                    }
                    if (!run_mul_mat_q(st, a, nullptr, b, a1, o1, nullptr)) return -1;
                    c->st.fused_nodes += 1;
                    return 2;
                }
            }
            if (fuse && !rowpar && c->opt.mm_merge && a->type == GGML_TYPE_Q8_0 && mmq_q80_skinny_supported(a->type, a->ne[0], a->ne[1], M)) {
                const int used = try_merge_q80_skinny(st, i);
                if (used != 0) return used;
            }
            if (fuse && !rowpar && c->opt.mm_merge && M >= mmq_min_cols_for(c, a->type) && c->opt.mmq_i8 && mmq_i8_supported(a->type, a->ne[0], a->ne[1], M)) {
                const int used = try_merge_mm_batch(st, i);
                if (used != 0) return used;
            }
            if (fuse && !rowpar && ((M >= mmq_min_cols_for(c, a->type) && c->opt.mmq_i8 && mmq_i8_supported(a->type, a->ne[0], a->ne[1], M)) ||
                                    (M >= c->opt.q80_min_cols && mmq_q80_supported(a->type, a->ne[0], a->ne[1], M)) || mmq_q80_skinny_supported(a->type, a->ne[0], a->ne[1], M))) {
                // batches: MUL_MAT -> ADD (bias row or residual) rides in the GEMM's store
                ggml_tensor * a1 = next(1);
                const ggml_tensor * o1 = (a1 && single_use(st, n)) ? add_partner(a1, n) : nullptr;
                if (o1) {
                    st.sk_next = i + 2;
                    const bool ok = run_mul_mat_q(st, a, nullptr, b, a1, o1, nullptr);
                    st.sk_next = -1;
                    if (!ok) return -1;
                    c->st.fused_nodes += 1;
                    return 2;
                }
            }
            st.sk_next = rowpar ? -1 : i + 1;  // (a row-parallel result is all-reduced right after: its split-K sum cannot wait for the next kernel)
            const bool ok_mm = run_mul_mat_q(st, a, nullptr, b, n, nullptr, nullptr);
            st.sk_next = -1;
            if (!ok_mm) return -1;
            if (rowpar) {
                timed_scope ts(c, "tp_all_reduce", (double) ggml_abi_nbytes(n));
                // sum over ranks -> + residual: the ADD that follows rides in the peer-to-peer launch (one decode token: its sum of squares too,
                // for the RMS_NORM prologue that reads the residual stream next)
                ggml_tensor * a1 = next(1);
                const ggml_tensor * o1 = (fuse && a1 && single_use(st, n)) ? add_partner(a1, n) : nullptr;
                static const bool dbg_no_ar_fused = getenv("GGML_MI355X_DBG_NO_AR_FUSED") != nullptr;
                if (o1 && same_shape(o1, n) && !dbg_no_ar_fused) {  // (the ADD may recycle the block of either operand: every thread reads its own elements before it writes them)
                    int ssn = 0;
                    const bool want_ss = c->opt.ss_partials && c->ss_buf != nullptr && M == 1;
                    if (st.ss_tensor != nullptr && ranges_overlap(a1, st.ss_tensor)) st.ss_tensor = nullptr;
                    if (tp_all_reduce_fused(c, (float *) n->data, (size_t) ggml_abi_nelements(n), (const float *) o1->data, (float *) a1->data, want_ss ? c->ss_buf : nullptr, &ssn)) {
                        c->st.allreduces++;
                        c->st.fused_nodes += 1;
                        if (want_ss) { st.ss_tensor = a1; st.ss_n = ssn; }
                        return 2;
                    }
                }
                if (!tp_all_reduce(c, (float *) n->data, (size_t) ggml_abi_nelements(n))) return -1;
                c->st.allreduces++;
            }
            return 1;
        }

        case GGML_OP_ADD: case GGML_OP_SUB: case GGML_OP_MUL: case GGML_OP_DIV: {
            timed_scope ts(c, "binary", (double) ggml_abi_nbytes(n) * 3);
            launch_binary(s, n->op, TD(a), TD(b), TD(n));
            c->st.kernel_launches++;
            return 1;
        }
        case GGML_OP_SCALE:
            launch_scale(s, TD(a), TD(n), ggml_abi_op_param_f32(n, 0), ggml_abi_op_param_f32(n, 1));
            c->st.kernel_launches++;
            return 1;
        case GGML_OP_UNARY:
            launch_unary(s, n->op_params[0], TD(a), TD(n));
            c->st.kernel_launches++;
            return 1;
        case GGML_OP_GLU: {
            const tdesc bd = b ? TD(b) : TD(a);
            if (fuse && c->opt.prologue && ggml_abi_nrows(n) > 1 && (a->nb[1] % 16) == 0 && (!b || (b->nb[1] % 16) == 0) && !(((uintptr_t) a->data) & 15) &&
                (!b || !(((uintptr_t) b->data) & 15)) && (b || ((n->ne[0] * 4) % 16) == 0) && quant_consumers_only(st, i, n)) {
                timed_scope ts(c, "swiglu_quantize", (double) ggml_abi_nbytes(n) * 2);
                launch_swiglu_quantize(s, TD(a), b ? &bd : nullptr, n->ne[0], n->op_params[1], (char *) c->ws + st.act_off);
                mark_q8_cache(st, n);
                c->st.kernel_launches++;
                c->st.fused_nodes++;
                return 1;
            }
            static const bool q80_producers = !getenv("GGML_MI355X_Q80_PRODUCERS") || atoi(getenv("GGML_MI355X_Q80_PRODUCERS")) != 0;
            if (q80_producers && fuse && c->opt.prologue && ggml_abi_nrows(n) >= 9 && swiglu_q80_panel_ok(TD(a), b ? &bd : nullptr, n->ne[0], n->op_params[1]) && q80_panel_consumers_only(st, i, n)) {
                timed_scope ts(c, "swiglu_quantize_q8_0", (double) ggml_abi_nbytes(n) * 2);
                launch_swiglu_q80_panel(s, TD(a), b ? &bd : nullptr, n->ne[0], n->op_params[1], (char *) c->ws + st.act_off);
                mark_q8_cache(st, n, MI_ACT_Q80_PANEL);
                c->st.kernel_launches++;
                c->st.fused_nodes++;
                return 1;
            }
            timed_scope ts(c, "swiglu", (double) ggml_abi_nbytes(n) * 3);
            launch_swiglu(s, TD(a), b ? &bd : nullptr, TD(n), n->op_params[1]);
            c->st.kernel_launches++;
            return 1;
        }
        case GGML_OP_CPY: case GGML_OP_DUP: case GGML_OP_CONT: {
            if (a->type == n->type && ggml_abi_is_contiguous(a) && ggml_abi_is_contiguous(n)) {
                if (hipMemcpyAsync(n->data, a->data, ggml_abi_nbytes(a), hipMemcpyDeviceToDevice, s) != hipSuccess) return -1;
            } else if (a->type == GGML_TYPE_Q8_0 || n->type == GGML_TYPE_Q8_0) {
                timed_scope ts(c, "cpy_q8_0", (double) ggml_abi_nbytes(n) + (double) ggml_abi_nbytes(a));
                launch_cpy_q8_0(s, a->data, n->data, ggml_abi_nelements(a), n->type == GGML_TYPE_Q8_0);
            } else if (kv_store_type(a->type) || kv_store_type(n->type)) {
                timed_scope ts(c, "cpy_kv_type", (double) ggml_abi_nbytes(n) + (double) ggml_abi_nbytes(a));
                launch_cpy_kv(s, kv_store_type(n->type) ? n->type : a->type, a->data, n->data, ggml_abi_nelements(a), kv_store_type(n->type));
            } else {
                timed_scope ts(c, "cpy", (double) ggml_abi_nbytes(n) * 2);
                launch_cpy(s, TD(a), TD(n));
            }
            c->st.kernel_launches++;
            return 1;
        }
        case GGML_OP_GET_ROWS: {
            // the head of a decode step (round 6): the token embeddings' GET_ROWS, the mask's F32 -> F16 cast (an input leaf cast once per graph) and the (cos, sin) table
            // the fused Q/K/V launches of every layer read are independent of each other — one launch instead of three at the dependent-launch floor.  The cast is
            // hoisted over the nodes between only when its result's memory is not touched by any of them; the table is the one ensure_rope_table would build at the
            // first fused Q/K/V launch (same positions, parameters, token count: it then finds it valid), and is simply rebuilt there if the guess was wrong
            static const bool head_on = !getenv("GGML_MI355X_STEP_HEAD") || atoi(getenv("GGML_MI355X_STEP_HEAD")) != 0;
            const int64_t n_rows = b->ne[0] * b->ne[1] * b->ne[2];
            if (head_on && fuse && !c->opt.timing && n_rows <= 64) {
                const ggml_tensor * cast = nullptr;
                int cast_at = -1;
                const ggml_tensor * rope = nullptr;
                const int lim = std::min(g->n_nodes, i + 48);
                for (int k = i + 1; k < lim && (!cast || !rope); ++k) {
                    const ggml_tensor * t = g->nodes[k];
                    if (st.done[k]) continue;
                    if (!cast && t->op == GGML_OP_CPY && t->src[0]->type == GGML_TYPE_F32 && t->type == GGML_TYPE_F16 && t->src[0]->op == GGML_OP_NONE &&
                        ggml_abi_is_contiguous(t->src[0]) && ggml_abi_is_contiguous(t) && same_shape(t->src[0], t) && ggml_abi_nelements(t) <= (int64_t) 1 << 22) {
                        // (neither the cast's result nor its SOURCE may be touched by anything that runs between here and the cast's place in the graph)
                        bool clean = !ranges_overlap(t, n) && !ranges_overlap(t, a) && !ranges_overlap(t, b) && !ranges_overlap(t->src[0], n);
                        for (int m = i + 1; m < k && clean; ++m) {
                            const ggml_tensor * u = g->nodes[m];
                            clean = !ranges_overlap(t, u) && !ranges_overlap(t->src[0], u);
                            for (int q = 0; q < GGML_MAX_SRC && clean; ++q)
                                if (u->src[q]) clean = !ranges_overlap(t, u->src[q]);
                        }
                        if (clean) { cast = t; cast_at = k; }
                    }
                    if (!rope && t->op == GGML_OP_ROPE && t->type == GGML_TYPE_F32 && !(t->op_params[2] & GGML_ROPE_TYPE_MROPE) && t->src[1] && t->src[1]->op == GGML_OP_NONE &&
                        t->src[1]->type == GGML_TYPE_I32 && (!t->src[2] || t->src[2]->op == GGML_OP_NONE))
                        rope = t;
                }
                rope_params rp{};
                int rope_m = 0;
                if (rope && c->rope_tab) {
                    rp.n_dims = rope->op_params[1];
                    rp.mode = rope->op_params[2];
                    rp.n_ctx_orig = rope->op_params[4];
                    rp.freq_base = ggml_abi_op_param_f32(rope, 5);
                    rp.freq_scale = ggml_abi_op_param_f32(rope, 6);
                    rp.ext_factor = ggml_abi_op_param_f32(rope, 7);
                    rp.attn_factor = ggml_abi_op_param_f32(rope, 8);
                    rp.beta_fast = ggml_abi_op_param_f32(rope, 9);
                    rp.beta_slow = ggml_abi_op_param_f32(rope, 10);
                    memset(rp.sections, 0, sizeof(rp.sections));
                    rope_m = (int) rope->ne[2];
                    static const bool tab_on = !getenv("GGML_MI355X_ROPE_TABLE") || atoi(getenv("GGML_MI355X_ROPE_TABLE")) != 0;
                    if (!tab_on || rope_m < 1 || rope_m > 64 || rp.n_dims < 2 || rp.n_dims > 256 || (int64_t) rope_m * rp.n_dims > (int64_t) backend_ctx::rope_tab_floats) rope_m = 0;
                }
                if (cast || rope_m > 0) {
                    const tdesc ta = TD(a), tb = TD(b), tn = TD(n);
                    launch_step_head(s, &ta, &tb, &tn, cast ? (const float *) cast->src[0]->data : nullptr, cast ? cast->data : nullptr, cast ? ggml_abi_nelements(cast) : 0,
                                     rope_m > 0 ? (const int32_t *) rope->src[1]->data : nullptr, rope_m > 0 && rope->src[2] ? (const float *) rope->src[2]->data : nullptr, &rp, rope_m, c->rope_tab);
                    c->st.kernel_launches++;
                    c->st.step_heads++;
                    if (cast) { mark_done(st, cast_at); c->st.fused_nodes++; }
                    if (rope_m > 0) {
                        st.rope_tab_pos = rope->src[1]->data;
                        st.rope_tab_ff = rope->src[2] ? rope->src[2]->data : nullptr;
                        st.rope_tab_p = rp;
                        st.rope_tab_m = rope_m;
                    }
                    return 1;
                }
            }
            timed_scope ts(c, "get_rows", (double) ggml_abi_nbytes(n));
            launch_get_rows(s, TD(a), TD(b), TD(n));
            c->st.kernel_launches++;
            return 1;
        }
        case GGML_OP_SET_ROWS: {
            timed_scope ts(c, "set_rows", (double) ggml_abi_nbytes(a));
            if (n->type == GGML_TYPE_Q8_0) launch_set_rows_q8_0(s, TD(a), TD(b), TD(n));
            else if (kv_store_type(n->type)) {
                // K and V rows of a step go into their caches through the same indices: SET_ROWS(k), [views], SET_ROWS(v) as ONE launch
                int j = i + 1;
                while (j < g->n_nodes && (st.done[j] || is_view_op(g->nodes[j]))) ++j;
                const ggml_tensor * n2 = j < g->n_nodes ? g->nodes[j] : nullptr;
                if (fuse && n2 && n2->op == GGML_OP_SET_ROWS && kv_store_type(n2->type) && n2->src[0]->type == GGML_TYPE_F32 && n2->src[1]->type == GGML_TYPE_I64 &&
                    n2->src[0] != n && !ranges_overlap(n2->src[0], n) && !ranges_overlap(n2, n) && supports_op(n2)) {
                    launch_set_rows_kv_pair(s, TD(a), TD(b), TD(n), TD(n2->src[0]), TD(n2->src[1]), TD(n2));
                    mark_done(st, j);
                    c->st.fused_nodes++;
                } else {
                    launch_set_rows_kv(s, TD(a), TD(b), TD(n));
                }
            }
            else launch_set_rows(s, TD(a), TD(b), TD(n));
            c->st.kernel_launches++;
            return 1;
        }
        case GGML_OP_SOFT_MAX: {
            const tdesc md = b ? TD(b) : TD(a);
            // a decode step on the non-flash path: the probabilities have one reader, MUL_MAT(V^T view, p) — made inside that product
            if (fuse && c->opt.softmax_mm && n->ne[1] == 1 && !n->src[2] && ggml_abi_op_param_f32(n, 1) == 0.0f && a->type == GGML_TYPE_F32 && ggml_abi_is_contiguous(a) &&
                single_use(st, n)) {
                int j = -1;
                for (int k = i + 1; k < std::min(g->n_nodes, i + 8) && j < 0; ++k) {
                    if (st.done[k]) continue;
                    const ggml_tensor * t = g->nodes[k];
                    if (t->op == GGML_OP_MUL_MAT && t->src[1] == n) j = k;
                    else if (!is_view_op(t)) break;
                }
                if (j >= 0) {
                    const ggml_tensor * mm = g->nodes[j];
                    const ggml_tensor * v = mm->src[0];
                    // workgroups store rows of the product while others still read the logits: no recycled memory between the two
                    if (v->type == GGML_TYPE_F16 && !buffer_is_split(v->buffer) && ggml_abi_is_contiguous(mm) && !ranges_overlap(mm, a) && !(b && ranges_overlap(mm, b))) {
                        // llama.cpp's non-flash graph ends the attention with cont(permute(kqv)): for ONE token the permuted view has kqv's own byte order, and the
                        // CONT was a 16 KB device-to-device copy per layer — a blit kernel of 4.9 us + its boundary, 10 % of a `-fa 0` decode step (round 6).  When
                        // kqv is read by nothing but that view chain, the product is written where the CONT would have put it and the CONT is done
                        int cont_at = -1;
                        const ggml_tensor * cont = nullptr;
                        static const bool elide_on = !getenv("GGML_MI355X_ELIDE_CONT") || atoi(getenv("GGML_MI355X_ELIDE_CONT")) != 0;
                        if (elide_on && use_count(st, mm) == 1 && !(mm->flags & GGML_TENSOR_FLAG_OUTPUT)) {
                            const ggml_tensor * cur = mm;
                            for (int k = j + 1; k < std::min(g->n_nodes, j + 6); ++k) {
                                const ggml_tensor * t = g->nodes[k];
                                if (st.done[k]) continue;
                                if (is_view_op(t) && t->src[0] == cur && t->data == mm->data && use_count(st, t) == 1 && !(t->flags & GGML_TENSOR_FLAG_OUTPUT)) { cur = t; continue; }
                                if ((t->op == GGML_OP_CONT || t->op == GGML_OP_CPY || t->op == GGML_OP_DUP) && t->src[0] == cur && cur != mm && t->type == mm->type && ggml_abi_is_contiguous(cur) &&
                                    ggml_abi_is_contiguous(t) && ggml_abi_nelements(t) == ggml_abi_nelements(mm) && !ranges_overlap(t, a) && !ranges_overlap(t, v) && !(b && ranges_overlap(t, b)) &&
                                    !ranges_overlap(t, mm)) {
                                    cont = t;
                                    cont_at = k;
                                }
                                break;
                            }
                        }
                        tdesc out = TD(mm);
                        if (cont) out.data = (char *) cont->data;
                        timed_scope ts(c, "soft_max_mul_mat_f", (double) ggml_abi_nbytes(v) + (double) ggml_abi_nbytes(n));
                        if (launch_soft_max_mul_mat_f16(s, TD(v), TD(a), b ? &md : nullptr, out, ggml_abi_op_param_f32(n, 0))) {
                            mark_done(st, j);
                            if (cont) { mark_done(st, cont_at); c->st.fused_nodes++; c->st.elided_conts++; }
                            c->st.kernel_launches++;
                            c->st.fused_nodes++;
                            return 1;
                        }
                    }
                }
            }
            timed_scope ts(c, "soft_max", (double) ggml_abi_nbytes(n) * 2);
            launch_soft_max(s, TD(a), b ? &md : nullptr, n->src[2] ? (const float *) n->src[2]->data : nullptr, TD(n), ggml_abi_op_param_f32(n, 0), ggml_abi_op_param_f32(n, 1));
            c->st.kernel_launches++;
            return 1;
        }
        case GGML_OP_ROPE: {
            if (fuse && try_fuse_rope_store(st, i)) return 1;
            rope_params p;
            p.n_dims = n->op_params[1];
            p.mode = n->op_params[2];
            p.n_ctx_orig = n->op_params[4];
            p.freq_base = ggml_abi_op_param_f32(n, 5);
            p.freq_scale = ggml_abi_op_param_f32(n, 6);
            p.ext_factor = ggml_abi_op_param_f32(n, 7);
            p.attn_factor = ggml_abi_op_param_f32(n, 8);
            p.beta_fast = ggml_abi_op_param_f32(n, 9);
            p.beta_slow = ggml_abi_op_param_f32(n, 10);
            memcpy(p.sections, n->op_params + 11, sizeof(p.sections));
            timed_scope ts(c, "rope", (double) ggml_abi_nbytes(n) * 2);
            launch_rope(s, TD(a), TD(b), n->src[2] ? (const float *) n->src[2]->data : nullptr, TD(n), p);
            c->st.kernel_launches++;
            return 1;
        }
        case GGML_OP_FLASH_ATTN_EXT: {
            const ggml_tensor * k = n->src[1];
            const ggml_tensor * v = n->src[2];
            const ggml_tensor * m = n->src[3];
            fattn_params p;
            p.scale = ggml_abi_op_param_f32(n, 0);
            p.max_bias = ggml_abi_op_param_f32(n, 1);
            p.logit_softcap = ggml_abi_op_param_f32(n, 2);
            const tdesc qd = TD(a);
            tdesc kd = TD(k), vd = TD(v);
            bool imaged = fa_route(n) == 3;
            bool native_kv = false;
            if (imaged) {  // decode-sized batches at the lane-parallel kernel's shapes read the cache IN PLACE (fattn.hip: the DQ form); everything else through the image
                fattn_params probe{};
                probe.logit_softcap = ggml_abi_op_param_f32(n, 2);
                probe.max_bias = ggml_abi_op_param_f32(n, 1);
                probe.n_splits = 1;
                native_kv = fattn_native_kv_ok(qd, kd, vd, probe);
                if (native_kv) {
                    imaged = false;
                    c->st.kv_native_nodes++;
                }
            }
            if (imaged) {
                // K / V kept in another type: expand the views to f16 once (kv_types.hip) and let the f16 kernels read the image
                char * img = (char *) c->ws + st.aux_off + fa_image_offset(qd, kd, vd);
                timed_scope ts(c, "kv_image_f16", (double) (ggml_abi_nbytes(k) + ggml_abi_nbytes(v)));
                launch_kv_images_f16(s, kd, vd, img);
                c->st.kernel_launches++;
                c->st.kv_image_nodes++;
            }
            const tdesc md0 = m ? TD(m) : qd;
            p.mask_sparse = fa_mask_hint(n);
            p.n_splits = std::min(64, c->opt.fa_splits > 0 ? c->opt.fa_splits : fattn_pick_splits(qd, kd, m ? &md0 : nullptr, p.mask_sparse));
            p.kv_type = (imaged || native_kv) ? (int) GGML_TYPE_F16 : (int) k->type;
            p.dq = native_kv ? 1 : 0;
            if (c->opt.fa_self_merge && c->fa_arrive) {
                p.arrive = c->fa_arrive;
                p.arrive_slots = backend_ctx::fa_arrive_slots;
            }
            const tdesc md = m ? TD(m) : qd;
            timed_scope ts(c, "flash_attn", (double) (k->ne[1] * k->ne[2] * k->ne[0] * 2 * 2));
            if (c->fa_lists) {
                const int tile = fattn_list_tile(qd, kd, m ? &md : nullptr, p, c->fa_lists_bytes);
                if (tile > 0) {
                    if (st.fa_list_mask != m->data || st.fa_list_tile != tile) {  // first attention node of this graph run: list the tiles
                        launch_fattn_tile_scan(s, md, (int) a->ne[1], (int) k->ne[1], tile, c->fa_lists);
                        c->st.kernel_launches++;
                        st.fa_list_mask = m->data;
                        st.fa_list_tile = tile;
                    }
                    p.lists = c->fa_lists;
                    c->st.fa_list_launches++;
                } else if (m && a->ne[1] >= fattn_mma_min_q() && a->ne[3] == 1 && m->ne[3] == 1 && m->type == GGML_TYPE_F16 && (m->nb[1] % 8) == 0 && !(((uintptr_t) m->data) & 7) &&
                           (k->ne[1] % 4) == 0 && fattn_vis_bytes(qd, kd) <= c->fa_lists_bytes && flash_attn_mma_applies(qd, kd, &md, n->src[4] ? (const float *) n->src[4]->data : nullptr, TD(n), p)) {
                    // prompt batches on the matrix-core kernel: which (query tile, kv tile) pairs hold anything visible — once per graph run
                    if (st.fa_list_mask != m->data || st.fa_list_tile != -1) {
                        launch_fattn_vis_scan(s, md, (int) a->ne[1], (int) k->ne[1], (uint8_t *) c->fa_lists);
                        c->st.kernel_launches++;
                        st.fa_list_mask = m->data;
                        st.fa_list_tile = -1;
                    }
                    p.tile_vis = (const uint8_t *) c->fa_lists;
                }
            }
            // one decode token whose attention result goes (through a reshape) straight into a quantised mat-vec — wo: few fat splits on
            // 8-wave workgroups, and the merge of their partial records is that mat-vec's prologue; the combine launch disappears
            if (fuse && c->opt.fa_wo && !native_kv && c->opt.prologue && c->opt.fa_splits == 0 && a->ne[1] == 1 && a->ne[3] == 1 && !p.lists && !p.tile_vis && use_count(st, n) == 1 &&
                !(n->flags & GGML_TENSOR_FLAG_OUTPUT) && ggml_abi_is_contiguous(n) && (((uintptr_t) n->data) & 15) == 0) {
                const int S = fattn_fat_splits(qd, kd, m ? &md : nullptr, n->src[4] ? (const float *) n->src[4]->data : nullptr, p);
                int j = i + 1;
                const ggml_tensor * view = nullptr;
                while (j < g->n_nodes && (st.done[j] || is_view_op(g->nodes[j]))) {
                    if (!st.done[j] && g->nodes[j]->src[0] == (view ? view : n) && g->nodes[j]->data == n->data) view = g->nodes[j];
                    ++j;
                }
                const ggml_tensor * mm = j < g->n_nodes ? g->nodes[j] : nullptr;
                const ggml_tensor * w = mm ? mm->src[0] : nullptr;
                const bool kq = w && (w->type == GGML_TYPE_Q4_K || w->type == GGML_TYPE_Q5_K || w->type == GGML_TYPE_Q6_K || w->type == GGML_TYPE_Q8_0) && w->ne[2] == 1 && w->ne[3] == 1 &&
                                rows_contig(w) && (w->ne[0] % 256) == 0 && !buffer_is_split(w->buffer);
                if (S > 0 && view && mm && mm->op == GGML_OP_MUL_MAT && mm->src[1] == view && kq && w->ne[0] == n->ne[0] * n->ne[1] && ggml_abi_nelements(view) == w->ne[0] &&
                    use_count(st, view) == 1 && !(view->flags & GGML_TENSOR_FLAG_OUTPUT) && ggml_abi_is_contiguous(mm) && mm->type == GGML_TYPE_F32 &&
                    !(tp_active(c) && buffer_is_rowpar(w->view_src ? w->view_src->buffer : w->buffer))) {
                    p.n_splits = S;
                    p.fat = 1;
                    timed_scope ts(c, "flash_attn_fat", (double) (k->ne[1] * k->ne[2] * k->ne[0] * 2 * 2));
                    launch_flash_attn(s, qd, kd, vd, m ? &md : nullptr, nullptr, TD(n), p, (char *) c->ws + st.aux_off);
                    st.fa_wo.b = view;
                    st.fa_wo.fa = n;
                    st.fa_wo.part = (const float *) ((char *) c->ws + st.aux_off);
                    st.fa_wo.splits = S;
                    c->st.kernel_launches++;
                    c->st.fused_nodes++;
                    return 1;
                }
            }
            // a batch's attention result read only by quantised mat-muls (wo), through the usual reshape: the combine pass writes
            // Q8_K blocks into the activation scratch instead of f32 (one launch less per layer)
            const ggml_tensor * q8_reader = nullptr;
            // (up to 128 tokens — launch-bound sizes — through the quantising form of the head-pair combine; bigger batches when the row-parallel
            // combine serves them: it quantises in registers)
            // (one decode token: measured in round 6 — the combine pass leaving Q8_K blocks and wo without a prologue is no faster: the quantising combine +0.3 us,
            // wo on pre-quantised blocks 7.2 us against ~5.6 with its f32 prologue; profiles/r06_lab_combine_q8k_to_wo_batch1.txt)
            if (fuse && c->opt.prologue && a->ne[1] > 1 &&
                (a->ne[1] <= 128 || (p.n_splits >= 2 && fattn_combine_rows_applies((int) k->ne[0], a->ne[1], a->ne[2], a->ne[3], p.n_splits, n->src[4] ? (const float *) n->src[4]->data : nullptr))) &&
                use_count(st, n) == 1 && !(n->flags & GGML_TENSOR_FLAG_OUTPUT) && ggml_abi_is_contiguous(n)) {
                for (int k = i + 1; k < std::min(g->n_nodes, i + 4) && !q8_reader; ++k) {
                    const ggml_tensor * t = g->nodes[k];
                    if ((t->op == GGML_OP_RESHAPE || t->op == GGML_OP_VIEW) && t->src[0] == n && t->data == n->data && ggml_abi_is_contiguous(t) && t->ne[0] == n->ne[0] * n->ne[1] &&
                        t->ne[1] == n->ne[2] && quant_consumers_only(st, k, t) &&
                        fattn_q8_out_ok(qd, kd, m ? &md : nullptr, n->src[4] ? (const float *) n->src[4]->data : nullptr, TD(n), p))
                        q8_reader = t;
                }
            }
            if (q8_reader) p.q8_out = (char *) c->ws + st.act_off;
            launch_flash_attn(s, qd, kd, vd, m ? &md : nullptr, n->src[4] ? (const float *) n->src[4]->data : nullptr, TD(n), p, (char *) c->ws + st.aux_off);
            if (q8_reader) {
                mark_q8_cache(st, q8_reader);
                c->st.fused_nodes++;
            }
            c->st.kernel_launches += p.n_splits > 1 ? 2 : 1;
            return 1;
        }
        case GGML_OP_ARGMAX:
            launch_argmax(s, TD(a), TD(n));
            c->st.kernel_launches++;
            return 1;
        default:
            MI_ERR("graph_compute: unsupported op %d (node %d '%s')", (int) n->op, i, n->name);
            return -1;
    }
}

static bool run_nodes(backend_ctx * c, ggml_cgraph * g, const ws_plan & wp) {
    exec_state st{};
    st.c = c;
    st.g = g;
    st.aux_off = wp.act_bytes;
    st.uses.reserve((size_t) g->n_nodes * 2);
    for (int i = 0; i < g->n_nodes; ++i)
        for (int s = 0; s < GGML_MAX_SRC; ++s)
            if (g->nodes[i]->src[s]) st.uses[g->nodes[i]->src[s]]++;
    c->q8_src = nullptr;  // inputs change between graph launches
    st.done.assign((size_t) g->n_nodes, 0);
    for (int i = 0; i < g->n_nodes;) {
        ggml_tensor * n = g->nodes[i];
        if (ggml_abi_nelements(n) == 0 || st.done[i]) { i++; continue; }
        // anything that writes memory invalidates a cached quantisation of that memory: every node executed by this call — the
        // `used` consecutive ones and those a multi-chain fusion ran out of order — is checked by address RANGE (a recycled block
        // may be written through a tensor that starts elsewhere in it)
        st.ooo.clear();
        const int used = run_node(st, i);
        if (used < 0) return false;
        for (int k = 0; k < used; ++k)  // (mask statistics describe uploaded bytes only: common.h)
            if (!is_view_op(g->nodes[i + k]) && g->nodes[i + k]->data) forget_mask_stats(g->nodes[i + k]->data, ggml_abi_nbytes(g->nodes[i + k]));
        if (!st.q8_fresh && c->q8_src) {
            const char * q0 = (const char *) c->q8_src, * q1 = q0 + c->q8_bytes;
            auto hits = [&](const ggml_tensor * t) {
                if (is_view_op(t) || !t->data) return false;  // (views write nothing)
                const char * t0 = (const char *) t->data;
                return t0 < q1 && q0 < t0 + ggml_abi_nbytes(t);
            };
            bool hit = false;
            for (int k = 0; k < used && !hit; ++k) hit = hits(g->nodes[i + k]);
            for (size_t k = 0; k < st.ooo.size() && !hit; ++k) hit = hits(g->nodes[st.ooo[k]]);
            if (hit) c->q8_src = nullptr;
        }
        st.q8_fresh = false;
        i += used;
    }
    return hipGetLastError() == hipSuccess;
}

// ------------------------------------------------------------------------------------------------ hipGraph cache
// A captured hipGraph stands for one graph KEY: every property of the ggml graph that decides which kernels run on which addresses
// with which arguments — per node {type, op, ne, nb, op_params, flags, buffer, data, which src slots are set, the mask-content hint}
// and per src {type, buffer, ne, nb, data} — laid out as 8-byte words (~20 per node + 11 per src: ~270 KB for a 32-layer decode graph).
// Round 4 hashed those bytes with a byte-serial FNV-1a before EVERY replay (4 cycles per byte on a dependent multiply chain: 100-370 us
// of host time per step during which the GPU idles) and trusted the 64-bit hash alone.  Now: the key words of the graph
// replayed last are compared in place (wide memcmp over the contiguous field runs of ggml_tensor, early exit at the first difference,
// nothing built, nothing hashed); only on a miss are the words materialised, hashed eight bytes per multiply on four independent lanes,
// and looked up — and a cached entry is used only if its stored words are EQUAL, so a hash collision can no longer replay another graph.
static_assert(offsetof(ggml_tensor, ne) == offsetof(ggml_tensor, buffer) + 8 && offsetof(ggml_tensor, nb) == offsetof(ggml_tensor, ne) + 32 &&
              offsetof(ggml_tensor, op) == offsetof(ggml_tensor, nb) + 32 && offsetof(ggml_tensor, op_params) == offsetof(ggml_tensor, op) + 4 &&
              offsetof(ggml_tensor, flags) == offsetof(ggml_tensor, op_params) + 64 && offsetof(ggml_tensor, src) == offsetof(ggml_tensor, flags) + 4 &&
              offsetof(ggml_tensor, buffer) % 8 == 0 && offsetof(ggml_tensor, src) % 8 == 0,
              "graph key: {buffer, ne, nb, op, op_params, flags} must be one padding-free run of ggml_tensor");
static constexpr size_t KEY_NODE_RUN = (offsetof(ggml_tensor, src) - offsetof(ggml_tensor, buffer)) / 8;  // buffer .. flags: 18 words
static constexpr size_t KEY_SRC_RUN = (offsetof(ggml_tensor, op) - offsetof(ggml_tensor, buffer)) / 8;    // buffer, ne, nb: 9 words

// walks the key words of `g` in their fixed order; `put(p, n)` receives runs of n words and returns false to stop
template <class Put>
static inline bool walk_key(const ggml_cgraph * g, Put && put) {
    uint64_t w[2];
    w[0] = (uint64_t) (uint32_t) g->n_nodes;
    if (!put(w, 1)) return false;
    for (int i = 0; i < g->n_nodes; ++i) {
        const ggml_tensor * n = g->nodes[i];
        uint32_t present = 0;
        for (int s = 0; s < GGML_MAX_SRC; ++s) present |= n->src[s] ? 1u << s : 0u;
        const bool hinted = n->op == GGML_OP_FLASH_ATTN_EXT || n->op == GGML_OP_SOFT_MAX;
        w[0] = (uint64_t) (uint32_t) n->type | (uint64_t) present << 32 | (uint64_t) (hinted ? fa_mask_hint(n) : 0) << 48;
        w[1] = (uint64_t) (uintptr_t) n->data;
        if (!put(w, 2) || !put(&n->buffer, KEY_NODE_RUN)) return false;
        for (int s = 0; s < GGML_MAX_SRC; ++s) {
            const ggml_tensor * t = n->src[s];
            if (!t) continue;
            w[0] = (uint64_t) (uint32_t) t->type;
            w[1] = (uint64_t) (uintptr_t) t->data;
            if (!put(w, 2) || !put(&t->buffer, KEY_SRC_RUN)) return false;
        }
    }
    return true;
}
static bool key_equals(const ggml_cgraph * g, const std::vector<uint64_t> & key) {
    const uint64_t * at = key.data(), * const end = at + key.size();
    const bool same = walk_key(g, [&](const void * p, size_t n) {
        if ((size_t) (end - at) < n || memcmp(at, p, n * 8) != 0) return false;
        at += n;
        return true;
    });
    return same && at == end;
}
static void key_build(const ggml_cgraph * g, std::vector<uint64_t> & key) {
    key.clear();
    key.reserve((size_t) g->n_nodes * 48);
    walk_key(g, [&](const void * p, size_t n) {
        const size_t at = key.size();
        key.resize(at + n);
        memcpy(key.data() + at, p, n * 8);
        return true;
    });
}
bool graph_key_equals(const ggml_cgraph * g, const std::vector<uint64_t> & key) { return key_equals(g, key); }
void graph_key_build(const ggml_cgraph * g, std::vector<uint64_t> & key) { key_build(g, key); }
static uint64_t key_hash(const std::vector<uint64_t> & key) {  // four independent multiply-fold lanes over 8-byte words
    uint64_t h[4] = {0x9E3779B97F4A7C15ull, 0xC2B2AE3D27D4EB4Full, 0x165667B19E3779F9ull, 0x27D4EB2F165667C5ull};
    auto fold = [](uint64_t a, uint64_t v) {
        const unsigned __int128 m = (unsigned __int128) (a ^ v) * 0xD6E8FEB86659FD93ull;
        return (uint64_t) m ^ (uint64_t) (m >> 64);
    };
    size_t i = 0;
    for (; i + 4 <= key.size(); i += 4)
        for (int l = 0; l < 4; ++l) h[l] = fold(h[l], key[i + l]);
    for (; i < key.size(); ++i) h[0] = fold(h[0], key[i]);
    return fold(fold(fold(h[0], h[1]), h[2]), h[3] ^ key.size());
}

void free_graph_cache(backend_ctx * c) {
    for (auto & kv : c->graphs) {
        if (kv.second.exec) (void) hipGraphExecDestroy(kv.second.exec);
        if (kv.second.graph) (void) hipGraphDestroy(kv.second.graph);
    }
    c->graphs.clear();
    c->last_graph = nullptr;
}

enum ggml_status graph_compute(backend_ctx * c, ggml_cgraph * g) {
    if (g->n_nodes == 0) return GGML_STATUS_SUCCESS;
    if (hip_failed()) {  // a set_tensor / get_tensor / copy / synchronize of this process failed earlier (HIP_SOFT): the inputs of this graph cannot be trusted
        MI_ERR("graph_compute: refused — an earlier HIP call of the data path failed (see the log above)");
        return GGML_STATUS_FAILED;
    }
    if (!tp_check(c)) return GGML_STATUS_FAILED;  // (one load of a host word; tp.cpp)
    const auto t_enter = std::chrono::steady_clock::now();
    struct host_clock {  // host time of this call (stats::graph_compute_host_ns: key comparison, planning, launches — the GPU may idle meanwhile)
        backend_ctx * c;
        std::chrono::steady_clock::time_point t0;
        ~host_clock() { c->st.graph_compute_host_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
    } clock{c, t_enter};
    c->tick++;
    {   // a decode copy was dropped since the cached graphs were captured (the host rewrote a weight, freed a weights buffer): they hold pointers to it
        const uint64_t ep = decode_copy_epoch();
        if (ep != c->decode_epoch) {
            if (!c->graphs.empty()) {
                (void) hipStreamSynchronize(c->stream);
                free_graph_cache(c);
            }
            c->decode_epoch = ep;
        }
    }
    auto replay = [&](cached_graph & cg) {
        cg.last_use = c->tick;
        cg.seen++;
        const auto t0 = std::chrono::steady_clock::now();
        c->st.graph_key_host_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(t0 - t_enter).count();
        if (hipGraphLaunch(cg.exec, c->stream) != hipSuccess) {
            (void) hipGetLastError();
            return GGML_STATUS_FAILED;
        }
        c->st.graph_launch_host_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        c->st.graph_launches++;
        c->st.allreduces += cg.allreduces;
        return GGML_STATUS_SUCCESS;
    };
    // the step llama-box repeats: the same graph as last time.  Its key is compared in place; an instantiated graph implies that the scratch it was
    // captured over is still the one in use (ensure_ws drops the cache when it regrows), so nothing else has to be planned for a replay.
    const bool may_replay = c->opt.graphs && !c->opt.timing && g->n_nodes >= 8;
    if (may_replay && c->last_graph && c->last_graph->exec && key_equals(g, c->last_graph->key)) {
        c->st.graph_key_fast_hits++;
        return replay(*c->last_graph);
    }
    // -sm row: a graph over row-split weights runs as tensor parallelism over the process's devices where the engine takes it (tp_inproc.cpp); a
    // backend that owns such an engine also tells it about every other graph (the K-shift works on the host's cache tensors)
    {
        bool handled = false;
        const enum ggml_status st_ip = ip_graph_compute(c, g, &handled);
        if (handled) return st_ip;
    }
    const ws_plan wp = plan_ws(c, g);
    if (!ensure_ws(c, wp.act_bytes + wp.aux_bytes + 256)) return GGML_STATUS_ALLOC_FAILED;
    bool has_split = false;  // multi-device launches + peer copies: executed eagerly (capture across devices is left for a box that has them)
    for (int i = 0; i < g->n_nodes && !has_split; ++i) has_split = g->nodes[i]->op == GGML_OP_MUL_MAT && buffer_is_split(g->nodes[i]->src[0]->buffer);
    // (graphs that launch on several devices' streams: captured only on request — GGML_MI355X_SPLIT_GRAPHS=1 — until that has run on a box
    // with more than one GPU; the fork / join over events is capturable and is exercised on logical devices by tests/test_gpu_split.py)
    static const bool split_graphs = getenv("GGML_MI355X_SPLIT_GRAPHS") && atoi(getenv("GGML_MI355X_SPLIT_GRAPHS")) != 0;
    const bool want_graph = may_replay && (!has_split || split_graphs);
    if (!want_graph) {
        c->st.eager_graphs++;
        return run_nodes(c, g, wp) ? GGML_STATUS_SUCCESS : GGML_STATUS_FAILED;
    }
    key_build(g, c->key_scratch);
    uint64_t fp = key_hash(c->key_scratch);
    auto it = c->graphs.find(fp);
    while (it != c->graphs.end() && it->second.key != c->key_scratch) {  // another graph with this hash (never seen): probe on
        c->st.graph_key_collisions++;
        fp = fp * 0x9E3779B97F4A7C15ull + 1;
        it = c->graphs.find(fp);
    }
    if (it == c->graphs.end()) {
        // bound the cache wherever a new entry appears (eager first sightings and captures at first sighting alike): drop what has not been
        // used for 256 graphs.  The entry replayed last (the predecessor a capture at first sighting patches) was used one tick ago and stays.
        if (c->graphs.size() > 64) {
            for (auto jt = c->graphs.begin(); jt != c->graphs.end();) {
                if (jt->second.last_use + 256 < c->tick) {
                    if (jt->second.exec) (void) hipGraphExecDestroy(jt->second.exec);
                    if (jt->second.graph) (void) hipGraphDestroy(jt->second.graph);
                    if (c->last_graph == &jt->second) c->last_graph = nullptr;
                    jt = c->graphs.erase(jt);
                    c->st.graph_evictions++;
                } else {
                    ++jt;
                }
            }
        }
        it = c->graphs.emplace(fp, cached_graph()).first;
        it->second.key = c->key_scratch;
        it->second.last_use = c->tick;
    }
    cached_graph & cg = it->second;
    // A graph seen for the first time whose predecessor — the graph replayed last — has the same number of nodes is the same step over a grown
    // cache (n_kv moves to the next multiple of 256: every 8th step of a -np 32 engine, llama-box/httpserver.hpp:3539-3623): its kernels have
    // all run before, so it is captured at once instead of after an eager run (profiles/r05_np32_ab_early_capture.txt).  Anything else keeps the
    // rule "first sighting eager" — one-off prompt graphs never pay for a capture, and a kernel's first launch never happens inside one.
    static const bool early_on = !getenv("GGML_MI355X_EARLY_CAPTURE") || atoi(getenv("GGML_MI355X_EARLY_CAPTURE")) != 0;
    const bool early = early_on && cg.seen == 0 && !cg.early_failed && c->last_graph && c->last_graph != &cg && c->last_graph->exec && c->last_graph->n_nodes == g->n_nodes &&
                       c->last_graph->last_use + 1 == c->tick && !has_split && !tp_active(c);
    cached_graph * const prev = early ? c->last_graph : nullptr;
    cg.n_nodes = g->n_nodes;
    c->last_graph = &cg;  // (unordered_map nodes stay where they are until erased; every erase below resets this)
    if (cg.exec) return replay(cg);
    cg.last_use = c->tick;
    cg.seen++;
    if (!early && (cg.seen < 2 || cg.seen > 1000000)) {  // first sighting: run eagerly (one-off prefill graphs never pay for capture)
        c->st.eager_graphs++;
        return run_nodes(c, g, wp) ? GGML_STATUS_SUCCESS : GGML_STATUS_FAILED;
    }
    // A capture at first sighting happens IN THE SHADOW of the step (round 6): the step's kernels are launched eagerly first — the GPU starts at once and is
    // ~3 ms behind the host for a -np 32 step — and the same walk is then repeated into a capture while they run (~1.2 ms of host-side launches + the
    // executable graph's update, during which the GPU used to idle: profiles/r06_ab_shadow_capture.txt).  Nothing is launched from that capture now; it
    // serves the NEXT step.  The second walk's counters are not launches and are taken back.
    const bool shadow = early && c->opt.shadow_capture != 0;
    auto now_ns = [] { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    int64_t t_mark = now_ns();
    if (shadow && !run_nodes(c, g, wp)) return GGML_STATUS_FAILED;
    if (shadow) c->st.graph_shadow_eager_ns += now_ns() - t_mark;
    const stats st_step = c->st;
    struct shadow_stats {  // (every exit below: the counters of the step as it ran, plus what the capture itself counts)
        backend_ctx * c; const stats & keep; bool on;
        ~shadow_stats() {
            if (!on) return;
            stats s = keep;
            s.graph_captures = c->st.graph_captures; s.graph_early_captures = c->st.graph_early_captures; s.graph_shadow_captures = c->st.graph_shadow_captures;
            s.graph_exec_updates = c->st.graph_exec_updates; s.graph_exec_update_failures = c->st.graph_exec_update_failures; s.graph_evictions = c->st.graph_evictions;
            s.graph_capture_walk_ns = c->st.graph_capture_walk_ns; s.graph_exec_update_ns = c->st.graph_exec_update_ns;
            c->st = s;
        }
    } shadow_guard{c, st_step, shadow};
    // second sighting: capture, instantiate, replay
    t_mark = now_ns();
    if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed) != hipSuccess) {
        (void) hipGetLastError();
        if (shadow) { cg.early_failed = true; return GGML_STATUS_SUCCESS; }
        c->st.eager_graphs++;
        return run_nodes(c, g, wp) ? GGML_STATUS_SUCCESS : GGML_STATUS_FAILED;
    }
    c->capturing = true;
    const int64_t red0 = c->st.allreduces;
    const bool ok = run_nodes(c, g, wp);
    c->capturing = false;
    const int64_t red_captured = c->st.allreduces - red0;
    hipGraph_t graph = nullptr;
    const hipError_t e_end = hipStreamEndCapture(c->stream, &graph);
    c->st.graph_capture_walk_ns += now_ns() - t_mark;
    t_mark = now_ns();
    if (!ok || e_end != hipSuccess || graph == nullptr) {
        (void) hipGetLastError();
        if (graph) (void) hipGraphDestroy(graph);
        if (shadow) { cg.early_failed = true; return GGML_STATUS_SUCCESS; }  // (the step has run; its next sighting is captured by the ordinary rule)
        if (early) {  // (the shortcut did not work for this graph: back to the ordinary rule, the next sighting tries again)
            cg.early_failed = true;
            c->st.eager_graphs++;
            return run_nodes(c, g, wp) ? GGML_STATUS_SUCCESS : GGML_STATUS_FAILED;
        }
        MI_INFO("hipGraph capture failed (ok=%d, err=%d); running this topology eagerly from now on", (int) ok, (int) e_end);
        cg.seen = 1000001;  // never try again
        c->st.eager_graphs++;
        return run_nodes(c, g, wp) ? GGML_STATUS_SUCCESS : GGML_STATUS_FAILED;
    }
    hipGraphExec_t exec = nullptr;
    // the same step over a grown cache: the predecessor's executable graph is PATCHED with this capture's kernel parameters (same kernels in the same
    // order, other extents / grid sizes) and moves to this entry — instantiating ~300 kernel nodes anew is the larger half of a re-capture.  Any
    // difference in topology (another attention kernel for the longer cache, another split count's combine pass) makes the update fail: then the
    // ordinary instantiation below.  Either way the predecessor gives its executable graph up: after a successful update it IS this entry's, after
    // a failed one it may be half patched (the runtime rewrites kernel nodes one by one until it meets the mismatch) and must never be launched
    // under the old key again — a -np 32 engine's n_kv shrinks when sequences finish, so that key does come back.  The predecessor's entry
    // returns to "seen once" and is captured afresh at its next sighting.  (exec_update 2 = the test hook: the update runs, then counts as failed.)
    if (early && c->opt.exec_update && prev && prev->exec) {
        hipGraphNode_t err_node = nullptr;
        hipGraphExecUpdateResult res = hipGraphExecUpdateError;
        const bool updated = hipGraphExecUpdate(prev->exec, graph, &err_node, &res) == hipSuccess && res == hipGraphExecUpdateSuccess && c->opt.exec_update != 2;
        if (updated) {
            exec = prev->exec;
            c->st.graph_exec_updates++;
        } else {
            (void) hipGetLastError();
            (void) hipGraphExecDestroy(prev->exec);
            c->st.graph_exec_update_failures++;
        }
        prev->exec = nullptr;
        if (prev->graph) (void) hipGraphDestroy(prev->graph);
        prev->graph = nullptr;
        prev->seen = 1;
    }
    if (exec == nullptr && (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess || exec == nullptr)) {
        (void) hipGetLastError();
        (void) hipGraphDestroy(graph);
        cg.seen = 1000001;
        if (shadow) return GGML_STATUS_SUCCESS;
        c->st.eager_graphs++;
        return run_nodes(c, g, wp) ? GGML_STATUS_SUCCESS : GGML_STATUS_FAILED;
    }
    c->st.graph_exec_update_ns += now_ns() - t_mark;
    cg.graph = graph;
    cg.exec = exec;
    cg.allreduces = red_captured;
    c->st.graph_captures++;
    c->st.graph_early_captures += early ? 1 : 0;
    if (shadow) {  // the step is already running; the executable graph waits for the next one
        c->st.graph_shadow_captures++;
        return GGML_STATUS_SUCCESS;
    }
    if (hipGraphLaunch(cg.exec, c->stream) != hipSuccess) {
        (void) hipGetLastError();
        return GGML_STATUS_FAILED;
    }
    c->st.graph_launches++;
    return GGML_STATUS_SUCCESS;
}

}  // namespace mi355x
