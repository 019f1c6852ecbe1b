// ggml_lite.cpp — own implementation of the ggml.h / ggml-backend.h / ggml-alloc.h entry points declared in
// ggml_lite.h (see that header for why this exists).  Nothing here is copied from ggml: the semantics of
// each constructor (result shape, op_params packing, src slots) are restated from the public API contract so
// that graphs built here are indistinguishable, to a backend, from graphs built by llama.cpp.
#include "ggml_lite.h"

#include <dlfcn.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <thread>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#define LITE_ASSERT(x)                                                                         \
    do {                                                                                       \
        if (!(x)) {                                                                            \
            fprintf(stderr, "ggml_lite: %s:%d: assertion failed: %s\n", __FILE__, __LINE__, #x); \
            abort();                                                                           \
        }                                                                                      \
    } while (0)

// allocator switch for per-node comparisons (tests, scripts/node_diff.py): keep every intermediate tensor in memory of its own;
// -1 = take it from the environment (GGML_LITE_NO_REUSE) on first use
static int g_no_reuse = -1;
static bool ggml_lite_no_reuse() {
    if (g_no_reuse < 0) g_no_reuse = getenv("GGML_LITE_NO_REUSE") != nullptr ? 1 : 0;
    return g_no_reuse != 0;
}
extern "C" void ggml_lite_set_no_reuse(int on) { g_no_reuse = on ? 1 : 0; }

struct ggml_context {
    std::vector<ggml_tensor *> tensors;
    std::vector<ggml_cgraph *> graphs;
};

extern "C" {

struct ggml_context * ggml_init(struct ggml_init_params) { return new ggml_context(); }

void ggml_free(struct ggml_context * ctx) {
    if (!ctx) return;
    for (auto * t : ctx->tensors) delete t;
    for (auto * g : ctx->graphs) {
        delete[] g->nodes;
        delete[] g->leafs;
        delete[] g->visited_hash_set.keys;
        delete[] g->visited_hash_set.used;
        delete[] g->use_counts;
        delete g;
    }
    delete ctx;
}

size_t ggml_type_size(enum ggml_type type) { return ggml_abi_type_size(type); }
int64_t ggml_blck_size(enum ggml_type type) { return ggml_abi_blck_size(type); }
size_t ggml_row_size(enum ggml_type type, int64_t ne) { return ggml_abi_row_size(type, ne); }
size_t ggml_nbytes(const struct ggml_tensor * t) { return ggml_abi_nbytes(t); }
int64_t ggml_nelements(const struct ggml_tensor * t) { return ggml_abi_nelements(t); }
int64_t ggml_nrows(const struct ggml_tensor * t) { return ggml_abi_nrows(t); }
bool ggml_is_contiguous(const struct ggml_tensor * t) { return ggml_abi_is_contiguous(t); }
bool ggml_is_quantized(enum ggml_type type) { return ggml_abi_blck_size(type) > 1; }

const char * ggml_type_name(enum ggml_type type) {
    switch (type) {
        case GGML_TYPE_F32: return "f32";
        case GGML_TYPE_F16: return "f16";
        case GGML_TYPE_BF16: return "bf16";
        case GGML_TYPE_Q8_0: return "q8_0";
        case GGML_TYPE_Q4_K: return "q4_K";
        case GGML_TYPE_Q5_K: return "q5_K";
        case GGML_TYPE_Q6_K: return "q6_K";
        case GGML_TYPE_Q8_K: return "q8_K";
        case GGML_TYPE_I32: return "i32";
        case GGML_TYPE_I64: return "i64";
        default: return "?";
    }
}

// GGML_OP_NAME (ggml.c): one entry per operator, in enum order — the backend checks its own numbering against this table at
// load time (csrc/backend.cpp: op_numbering_matches_host)
static const char * const k_op_name[GGML_OP_COUNT] = {
    "NONE", "DUP", "ADD", "ADD_ID", "ADD1", "ACC", "SUB", "MUL",
    "DIV", "SQR", "SQRT", "LOG", "SIN", "COS", "SUM", "SUM_ROWS",
    "MEAN", "ARGMAX", "COUNT_EQUAL", "REPEAT", "REPEAT_BACK", "CONCAT", "SILU_BACK", "NORM",
    "RMS_NORM", "RMS_NORM_BACK", "GROUP_NORM", "L2_NORM", "MUL_MAT", "MUL_MAT_ID", "OUT_PROD", "SCALE",
    "SET", "CPY", "CONT", "RESHAPE", "VIEW", "PERMUTE", "TRANSPOSE", "GET_ROWS",
    "GET_ROWS_BACK", "SET_ROWS", "DIAG", "DIAG_MASK_INF", "DIAG_MASK_ZERO", "SOFT_MAX", "SOFT_MAX_BACK", "ROPE",
    "ROPE_BACK", "CLAMP", "CONV_TRANSPOSE_1D", "IM2COL", "IM2COL_BACK", "CONV_2D", "CONV_2D_DW", "CONV_TRANSPOSE_2D",
    "POOL_1D", "POOL_2D", "POOL_2D_BACK", "UPSCALE", "PAD", "PAD_REFLECT_1D", "ROLL", "ARANGE",
    "TIMESTEP_EMBEDDING", "ARGSORT", "LEAKY_RELU", "FLASH_ATTN_EXT", "FLASH_ATTN_BACK", "SSM_CONV", "SSM_SCAN", "WIN_PART",
    "WIN_UNPART", "GET_REL_POS", "ADD_REL_POS", "RWKV_WKV6", "GATED_LINEAR_ATTN", "RWKV_WKV7", "UNARY", "MAP_CUSTOM1",
    "MAP_CUSTOM2", "MAP_CUSTOM3", "CUSTOM", "CROSS_ENTROPY_LOSS", "CROSS_ENTROPY_LOSS_BACK", "OPT_STEP_ADAMW", "OPT_STEP_SGD", "GLU",
};
const char * ggml_op_name(enum ggml_op op) { return (int) op >= 0 && op < GGML_OP_COUNT ? k_op_name[op] : "OP?"; }

// ------------------------------------------------------------------------------------------------ tensors
static ggml_tensor * new_tensor_impl(ggml_context * ctx, ggml_type type, int n_dims, const int64_t * ne, ggml_tensor * view_src, size_t view_offs) {
    LITE_ASSERT(ggml_abi_type_size(type) != 0);
    LITE_ASSERT(n_dims >= 1 && n_dims <= GGML_MAX_DIMS);
    if (view_src != nullptr && view_src->view_src != nullptr) {
        view_offs += view_src->view_offs;
        view_src = view_src->view_src;
    }
    ggml_tensor * t = new ggml_tensor();
    memset(t, 0, sizeof(*t));
    t->type = type;
    for (int i = 0; i < GGML_MAX_DIMS; ++i) t->ne[i] = i < n_dims ? ne[i] : 1;
    LITE_ASSERT(t->ne[0] % ggml_abi_blck_size(type) == 0);
    t->nb[0] = ggml_abi_type_size(type);
    t->nb[1] = t->nb[0] * (size_t) (t->ne[0] / ggml_abi_blck_size(type));
    for (int i = 2; i < GGML_MAX_DIMS; ++i) t->nb[i] = t->nb[i - 1] * (size_t) t->ne[i - 1];
    t->op = GGML_OP_NONE;
    t->view_src = view_src;
    t->view_offs = view_offs;
    if (view_src != nullptr && view_src->data != nullptr) t->data = (char *) view_src->data + view_offs;
    if (view_src != nullptr) t->buffer = view_src->buffer;
    ctx->tensors.push_back(t);
    return t;
}

struct ggml_tensor * ggml_new_tensor(struct ggml_context * ctx, enum ggml_type type, int n_dims, const int64_t * ne) {
    return new_tensor_impl(ctx, type, n_dims, ne, nullptr, 0);
}
struct ggml_tensor * ggml_new_tensor_1d(struct ggml_context * ctx, enum ggml_type type, int64_t ne0) { return ggml_new_tensor(ctx, type, 1, &ne0); }
struct ggml_tensor * ggml_new_tensor_2d(struct ggml_context * ctx, enum ggml_type type, int64_t ne0, int64_t ne1) {
    const int64_t ne[2] = {ne0, ne1};
    return ggml_new_tensor(ctx, type, 2, ne);
}
struct ggml_tensor * ggml_new_tensor_3d(struct ggml_context * ctx, enum ggml_type type, int64_t ne0, int64_t ne1, int64_t ne2) {
    const int64_t ne[3] = {ne0, ne1, ne2};
    return ggml_new_tensor(ctx, type, 3, ne);
}
struct ggml_tensor * ggml_new_tensor_4d(struct ggml_context * ctx, enum ggml_type type, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3) {
    const int64_t ne[4] = {ne0, ne1, ne2, ne3};
    return ggml_new_tensor(ctx, type, 4, ne);
}

struct ggml_tensor * ggml_set_name(struct ggml_tensor * t, const char * name) {
    snprintf(t->name, sizeof(t->name), "%s", name);
    return t;
}
static void format_name(ggml_tensor * t, const char * fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t->name, sizeof(t->name), fmt, ap);
    va_end(ap);
}
void ggml_set_input(struct ggml_tensor * t) { t->flags |= GGML_TENSOR_FLAG_INPUT; }
void ggml_set_output(struct ggml_tensor * t) { t->flags |= GGML_TENSOR_FLAG_OUTPUT; }

struct ggml_tensor * ggml_get_first_tensor(const struct ggml_context * ctx) { return ctx->tensors.empty() ? nullptr : ctx->tensors[0]; }
struct ggml_tensor * ggml_get_next_tensor(const struct ggml_context * ctx, struct ggml_tensor * t) {
    for (size_t i = 0; i + 1 < ctx->tensors.size(); ++i)
        if (ctx->tensors[i] == t) return ctx->tensors[i + 1];
    return nullptr;
}
struct ggml_tensor * ggml_get_tensor(struct ggml_context * ctx, const char * name) {
    for (auto * t : ctx->tensors)
        if (strcmp(t->name, name) == 0) return t;
    return nullptr;
}

static ggml_tensor * dup_tensor(ggml_context * ctx, const ggml_tensor * a) { return ggml_new_tensor(ctx, a->type, GGML_MAX_DIMS, a->ne); }

static ggml_tensor * view_tensor(ggml_context * ctx, ggml_tensor * a) {
    ggml_tensor * r = new_tensor_impl(ctx, a->type, GGML_MAX_DIMS, a->ne, a, 0);
    format_name(r, "%s (view)", a->name);
    for (int i = 0; i < GGML_MAX_DIMS; ++i) r->nb[i] = a->nb[i];
    return r;
}

// ------------------------------------------------------------------------------------------------ views
static ggml_tensor * view_impl(ggml_context * ctx, ggml_tensor * a, int n_dims, const int64_t * ne, size_t offset) {
    ggml_tensor * r = new_tensor_impl(ctx, a->type, n_dims, ne, a, offset);
    format_name(r, "%s (view)", a->name);
    memcpy(r->op_params, &offset, sizeof(offset));
    r->op = GGML_OP_VIEW;
    r->src[0] = a;
    return r;
}
struct ggml_tensor * ggml_view_1d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, size_t offset) { return view_impl(ctx, a, 1, &ne0, offset); }
struct ggml_tensor * ggml_view_2d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1, size_t nb1, size_t offset) {
    const int64_t ne[2] = {ne0, ne1};
    ggml_tensor * r = view_impl(ctx, a, 2, ne, offset);
    r->nb[1] = nb1;
    r->nb[2] = r->nb[1] * (size_t) ne1;
    r->nb[3] = r->nb[2];
    return r;
}
struct ggml_tensor * ggml_view_3d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1, int64_t ne2, size_t nb1, size_t nb2, size_t offset) {
    const int64_t ne[3] = {ne0, ne1, ne2};
    ggml_tensor * r = view_impl(ctx, a, 3, ne, offset);
    r->nb[1] = nb1;
    r->nb[2] = nb2;
    r->nb[3] = r->nb[2] * (size_t) ne2;
    return r;
}
struct ggml_tensor * ggml_view_4d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3, size_t nb1, size_t nb2, size_t nb3, size_t offset) {
    const int64_t ne[4] = {ne0, ne1, ne2, ne3};
    ggml_tensor * r = view_impl(ctx, a, 4, ne, offset);
    r->nb[1] = nb1;
    r->nb[2] = nb2;
    r->nb[3] = nb3;
    return r;
}

static ggml_tensor * reshape_impl(ggml_context * ctx, ggml_tensor * a, int n_dims, const int64_t * ne) {
    LITE_ASSERT(ggml_is_contiguous(a));
    int64_t n = 1;
    for (int i = 0; i < n_dims; ++i) n *= ne[i];
    LITE_ASSERT(n == ggml_nelements(a));
    ggml_tensor * r = new_tensor_impl(ctx, a->type, n_dims, ne, a, 0);
    format_name(r, "%s (reshaped)", a->name);
    r->op = GGML_OP_RESHAPE;
    r->src[0] = a;
    return r;
}
struct ggml_tensor * ggml_reshape_1d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0) { return reshape_impl(ctx, a, 1, &ne0); }
struct ggml_tensor * ggml_reshape_2d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1) {
    const int64_t ne[2] = {ne0, ne1};
    return reshape_impl(ctx, a, 2, ne);
}
struct ggml_tensor * ggml_reshape_3d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1, int64_t ne2) {
    const int64_t ne[3] = {ne0, ne1, ne2};
    return reshape_impl(ctx, a, 3, ne);
}
struct ggml_tensor * ggml_reshape_4d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3) {
    const int64_t ne[4] = {ne0, ne1, ne2, ne3};
    return reshape_impl(ctx, a, 4, ne);
}

struct ggml_tensor * ggml_permute(struct ggml_context * ctx, struct ggml_tensor * a, int axis0, int axis1, int axis2, int axis3) {
    const int ax[4] = {axis0, axis1, axis2, axis3};
    for (int i = 0; i < 4; ++i) {
        LITE_ASSERT(ax[i] >= 0 && ax[i] < GGML_MAX_DIMS);
        for (int j = 0; j < i; ++j) LITE_ASSERT(ax[i] != ax[j]);
    }
    ggml_tensor * r = view_tensor(ctx, a);
    format_name(r, "%s (permuted)", a->name);
    for (int i = 0; i < 4; ++i) {
        r->ne[ax[i]] = a->ne[i];
        r->nb[ax[i]] = a->nb[i];
    }
    r->op = GGML_OP_PERMUTE;
    r->src[0] = a;
    for (int i = 0; i < 4; ++i) r->op_params[i] = ax[i];
    return r;
}
struct ggml_tensor * ggml_transpose(struct ggml_context * ctx, struct ggml_tensor * a) {
    ggml_tensor * r = view_tensor(ctx, a);
    format_name(r, "%s (transposed)", a->name);
    r->ne[0] = a->ne[1];
    r->ne[1] = a->ne[0];
    r->nb[0] = a->nb[1];
    r->nb[1] = a->nb[0];
    r->op = GGML_OP_TRANSPOSE;
    r->src[0] = a;
    return r;
}

// ------------------------------------------------------------------------------------------------ ops
static bool can_repeat(const ggml_tensor * b, const ggml_tensor * a) {  // b broadcastable to a (ggml_can_repeat: an empty b only to an empty a)
    if (ggml_nelements(b) == 0) return ggml_nelements(a) == 0;
    for (int i = 0; i < GGML_MAX_DIMS; ++i)
        if (a->ne[i] % b->ne[i] != 0) return false;
    return true;
}
static void set_f32(ggml_tensor * t, int i, float v) { memcpy(&t->op_params[i], &v, 4); }

struct ggml_tensor * ggml_dup(struct ggml_context * ctx, struct ggml_tensor * a) {
    ggml_tensor * r = dup_tensor(ctx, a);
    r->op = GGML_OP_DUP;
    r->src[0] = a;
    return r;
}
struct ggml_tensor * ggml_cont(struct ggml_context * ctx, struct ggml_tensor * a) {
    ggml_tensor * r = dup_tensor(ctx, a);
    format_name(r, "%s (cont)", a->name);
    r->op = GGML_OP_CONT;
    r->src[0] = a;
    return r;
}
struct ggml_tensor * ggml_cont_2d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1) {
    LITE_ASSERT(ggml_nelements(a) == ne0 * ne1);
    ggml_tensor * r = ggml_new_tensor_2d(ctx, a->type, ne0, ne1);
    format_name(r, "%s (cont)", a->name);
    r->op = GGML_OP_CONT;
    r->src[0] = a;
    return r;
}
struct ggml_tensor * ggml_cpy(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b) {
    LITE_ASSERT(ggml_nelements(a) == ggml_nelements(b));
    ggml_tensor * r = view_tensor(ctx, b);
    if (b->name[0]) format_name(r, "%s (copy of %s)", b->name, a->name);
    else format_name(r, "%s (copy)", a->name);
    r->op = GGML_OP_CPY;
    r->src[0] = a;
    r->src[1] = b;
    return r;
}
struct ggml_tensor * ggml_cast(struct ggml_context * ctx, struct ggml_tensor * a, enum ggml_type type) {
    ggml_tensor * r = ggml_new_tensor(ctx, type, GGML_MAX_DIMS, a->ne);
    format_name(r, "%s (copy)", a->name);
    r->op = GGML_OP_CPY;
    r->src[0] = a;
    r->src[1] = r;
    return r;
}
static ggml_tensor * binary(ggml_context * ctx, ggml_op op, ggml_tensor * a, ggml_tensor * b) {
    LITE_ASSERT(can_repeat(b, a));
    ggml_tensor * r = dup_tensor(ctx, a);
    r->op = op;
    r->src[0] = a;
    r->src[1] = b;
    return r;
}
struct ggml_tensor * ggml_add(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b) { return binary(ctx, GGML_OP_ADD, a, b); }
struct ggml_tensor * ggml_sub(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b) { return binary(ctx, GGML_OP_SUB, a, b); }
struct ggml_tensor * ggml_mul(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b) { return binary(ctx, GGML_OP_MUL, a, b); }
struct ggml_tensor * ggml_div(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b) { return binary(ctx, GGML_OP_DIV, a, b); }

struct ggml_tensor * ggml_scale_bias(struct ggml_context * ctx, struct ggml_tensor * a, float s, float b) {
    ggml_tensor * r = dup_tensor(ctx, a);
    set_f32(r, 0, s);
    set_f32(r, 1, b);
    r->op = GGML_OP_SCALE;
    r->src[0] = a;
    return r;
}
struct ggml_tensor * ggml_scale(struct ggml_context * ctx, struct ggml_tensor * a, float s) { return ggml_scale_bias(ctx, a, s, 0.0f); }

struct ggml_tensor * ggml_rms_norm(struct ggml_context * ctx, struct ggml_tensor * a, float eps) {
    ggml_tensor * r = dup_tensor(ctx, a);
    set_f32(r, 0, eps);
    r->op = GGML_OP_RMS_NORM;
    r->src[0] = a;
    return r;
}

struct ggml_tensor * ggml_mul_mat(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b) {
    LITE_ASSERT(a->ne[0] == b->ne[0] && b->ne[2] % a->ne[2] == 0 && b->ne[3] % a->ne[3] == 0);
    LITE_ASSERT(a->nb[0] <= a->nb[1]);  // !ggml_is_transposed(a)
    const int64_t ne[4] = {a->ne[1], b->ne[1], b->ne[2], b->ne[3]};
    ggml_tensor * r = ggml_new_tensor(ctx, GGML_TYPE_F32, 4, ne);
    r->op = GGML_OP_MUL_MAT;
    r->src[0] = a;
    r->src[1] = b;
    return r;
}
void ggml_mul_mat_set_prec(struct ggml_tensor * a, enum ggml_prec prec) { a->op_params[0] = (int32_t) prec; }

struct ggml_tensor * ggml_get_rows(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b) {
    LITE_ASSERT(a->ne[2] == b->ne[1] && b->ne[3] == 1 && b->type == GGML_TYPE_I32);
    const ggml_type type = a->type == GGML_TYPE_I32 ? GGML_TYPE_I32 : GGML_TYPE_F32;
    ggml_tensor * r = ggml_new_tensor_4d(ctx, type, a->ne[0], b->ne[0], b->ne[1], b->ne[2]);
    r->op = GGML_OP_GET_ROWS;
    r->src[0] = a;
    r->src[1] = b;
    return r;
}
struct ggml_tensor * ggml_set_rows(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b, struct ggml_tensor * c) {
    LITE_ASSERT(a->ne[0] == b->ne[0] && a->ne[2] == b->ne[2] && a->ne[3] == b->ne[3]);
    LITE_ASSERT(b->ne[1] == c->ne[0] && b->ne[2] % c->ne[1] == 0 && b->ne[3] % c->ne[2] == 0 && c->ne[3] == 1);
    LITE_ASSERT(b->type == GGML_TYPE_F32 && c->type == GGML_TYPE_I64);
    ggml_tensor * r = view_tensor(ctx, a);
    r->op = GGML_OP_SET_ROWS;
    r->src[0] = b;
    r->src[1] = c;
    return r;
}

struct ggml_tensor * ggml_unary(struct ggml_context * ctx, struct ggml_tensor * a, enum ggml_unary_op op) {
    ggml_tensor * r = dup_tensor(ctx, a);
    r->op_params[0] = (int32_t) op;
    r->op = GGML_OP_UNARY;
    r->src[0] = a;
    return r;
}
struct ggml_tensor * ggml_silu(struct ggml_context * ctx, struct ggml_tensor * a) { return ggml_unary(ctx, a, GGML_UNARY_OP_SILU); }

static ggml_tensor * glu_impl(ggml_context * ctx, ggml_tensor * a, ggml_tensor * b, ggml_glu_op op, bool swapped) {
    int64_t ne[4] = {a->ne[0], a->ne[1], a->ne[2], a->ne[3]};
    if (b) {
        for (int i = 0; i < 4; ++i) LITE_ASSERT(a->ne[i] == b->ne[i]);
        LITE_ASSERT(a->type == b->type);
    } else {
        ne[0] = a->ne[0] / 2;
    }
    ggml_tensor * r = ggml_new_tensor(ctx, a->type, 4, ne);
    r->op_params[0] = (int32_t) op;
    r->op_params[1] = swapped ? 1 : 0;
    r->op = GGML_OP_GLU;
    r->src[0] = a;
    r->src[1] = b;
    return r;
}
struct ggml_tensor * ggml_swiglu(struct ggml_context * ctx, struct ggml_tensor * a) { return glu_impl(ctx, a, nullptr, GGML_GLU_OP_SWIGLU, false); }
struct ggml_tensor * ggml_swiglu_split(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b) { return glu_impl(ctx, a, b, GGML_GLU_OP_SWIGLU, false); }

struct ggml_tensor * ggml_soft_max_ext(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * mask, float scale, float max_bias) {
    LITE_ASSERT(ggml_is_contiguous(a));
    if (mask) {
        LITE_ASSERT(mask->type == GGML_TYPE_F16 || mask->type == GGML_TYPE_F32);
        LITE_ASSERT(mask->ne[0] == a->ne[0] && mask->ne[1] >= a->ne[1]);
        LITE_ASSERT(a->ne[2] % mask->ne[2] == 0 && a->ne[3] % mask->ne[3] == 0);
    }
    if (max_bias > 0.0f) LITE_ASSERT(mask);
    ggml_tensor * r = dup_tensor(ctx, a);
    set_f32(r, 0, scale);
    set_f32(r, 1, max_bias);
    r->op = GGML_OP_SOFT_MAX;
    r->src[0] = a;
    r->src[1] = mask;
    return r;
}
struct ggml_tensor * ggml_soft_max(struct ggml_context * ctx, struct ggml_tensor * a) { return ggml_soft_max_ext(ctx, a, nullptr, 1.0f, 0.0f); }
void ggml_soft_max_add_sinks(struct ggml_tensor * a, struct ggml_tensor * sinks) {
    LITE_ASSERT(a->op == GGML_OP_SOFT_MAX);
    a->src[2] = sinks;
}

static struct ggml_tensor * rope_impl(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b, struct ggml_tensor * c, int n_dims,
                                      int mode, int n_ctx_orig, float freq_base, float freq_scale, float ext_factor, float attn_factor,
                                      float beta_fast, float beta_slow, bool inplace) {
    LITE_ASSERT(b->type == GGML_TYPE_I32 && a->ne[2] == b->ne[0]);
    if (c) LITE_ASSERT(c->type == GGML_TYPE_F32 && c->ne[0] >= n_dims / 2);
    ggml_tensor * r = inplace ? view_tensor(ctx, a) : dup_tensor(ctx, a);
    int32_t params[15] = {/*n_past*/ 0, n_dims, mode, /*n_ctx*/ 0, n_ctx_orig};
    memcpy(params + 5, &freq_base, 4);
    memcpy(params + 6, &freq_scale, 4);
    memcpy(params + 7, &ext_factor, 4);
    memcpy(params + 8, &attn_factor, 4);
    memcpy(params + 9, &beta_fast, 4);
    memcpy(params + 10, &beta_slow, 4);
    memset(params + 11, 0, 16);
    memcpy(r->op_params, params, sizeof(params));
    r->op = GGML_OP_ROPE;
    r->src[0] = a;
    r->src[1] = b;
    r->src[2] = c;
    return r;
}
struct ggml_tensor * ggml_rope_multi(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b, struct ggml_tensor * c, int n_dims,
                                     int sections[4], int mode, int n_ctx_orig, float freq_base, float freq_scale, float ext_factor,
                                     float attn_factor, float beta_fast, float beta_slow) {
    // multimodal rotary embedding: four position ids per token (b is [4 * n_tokens]), sections = pairs per stream
    LITE_ASSERT((mode & 1) == 0 && b->type == GGML_TYPE_I32 && a->ne[2] * 4 == b->ne[0]);
    if (c) LITE_ASSERT(c->type == GGML_TYPE_F32 && c->ne[0] >= n_dims / 2);
    ggml_tensor * r = dup_tensor(ctx, a);
    int32_t params[15] = {/*n_past*/ 0, n_dims, mode, /*n_ctx*/ 0, n_ctx_orig};
    memcpy(params + 5, &freq_base, 4);
    memcpy(params + 6, &freq_scale, 4);
    memcpy(params + 7, &ext_factor, 4);
    memcpy(params + 8, &attn_factor, 4);
    memcpy(params + 9, &beta_fast, 4);
    memcpy(params + 10, &beta_slow, 4);
    memcpy(params + 11, sections, 16);
    memcpy(r->op_params, params, sizeof(params));
    r->op = GGML_OP_ROPE;
    r->src[0] = a;
    r->src[1] = b;
    r->src[2] = c;
    return r;
}
struct ggml_tensor * ggml_rope_ext(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b, struct ggml_tensor * c, int n_dims,
                                   int mode, int n_ctx_orig, float freq_base, float freq_scale, float ext_factor, float attn_factor,
                                   float beta_fast, float beta_slow) {
    return rope_impl(ctx, a, b, c, n_dims, mode, n_ctx_orig, freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow, false);
}
struct ggml_tensor * ggml_rope_ext_inplace(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b, struct ggml_tensor * c, int n_dims,
                                           int mode, int n_ctx_orig, float freq_base, float freq_scale, float ext_factor, float attn_factor,
                                           float beta_fast, float beta_slow) {
    return rope_impl(ctx, a, b, c, n_dims, mode, n_ctx_orig, freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow, true);
}

struct ggml_tensor * ggml_flash_attn_ext(struct ggml_context * ctx, struct ggml_tensor * q, struct ggml_tensor * k, struct ggml_tensor * v,
                                         struct ggml_tensor * mask, float scale, float max_bias, float logit_softcap) {
    LITE_ASSERT(k->ne[0] == q->ne[0] && k->ne[1] == v->ne[1]);
    LITE_ASSERT(q->ne[2] % k->ne[2] == 0 && q->ne[3] % k->ne[3] == 0);
    if (mask) {
        LITE_ASSERT(ggml_is_contiguous(mask) && mask->ne[1] >= q->ne[1]);
        LITE_ASSERT(q->ne[2] % mask->ne[2] == 0 && q->ne[3] % mask->ne[3] == 0);
    }
    if (max_bias > 0.0f) LITE_ASSERT(mask);
    const int64_t ne[4] = {v->ne[0], q->ne[2], q->ne[1], q->ne[3]};
    ggml_tensor * r = ggml_new_tensor(ctx, GGML_TYPE_F32, 4, ne);
    set_f32(r, 0, scale);
    set_f32(r, 1, max_bias);
    set_f32(r, 2, logit_softcap);
    r->op = GGML_OP_FLASH_ATTN_EXT;
    r->src[0] = q;
    r->src[1] = k;
    r->src[2] = v;
    r->src[3] = mask;
    return r;
}
void ggml_flash_attn_ext_set_prec(struct ggml_tensor * a, enum ggml_prec prec) {
    LITE_ASSERT(a->op == GGML_OP_FLASH_ATTN_EXT);
    a->op_params[3] = (int32_t) prec;
}
void ggml_flash_attn_ext_add_sinks(struct ggml_tensor * a, struct ggml_tensor * sinks) {
    LITE_ASSERT(a->op == GGML_OP_FLASH_ATTN_EXT);
    a->src[4] = sinks;
}

struct ggml_tensor * ggml_argmax(struct ggml_context * ctx, struct ggml_tensor * a) {
    LITE_ASSERT(a->ne[2] == 1 && a->ne[3] == 1);
    ggml_tensor * r = ggml_new_tensor_1d(ctx, GGML_TYPE_I32, a->ne[1]);
    r->op = GGML_OP_ARGMAX;
    r->src[0] = a;
    return r;
}

// ------------------------------------------------------------------------------------------------ graphs
struct ggml_cgraph * ggml_new_graph_custom(struct ggml_context * ctx, size_t size, bool) {
    ggml_cgraph * g = new ggml_cgraph();
    memset(g, 0, sizeof(*g));
    g->size = (int) size;
    g->nodes = new ggml_tensor *[size]();
    g->leafs = new ggml_tensor *[size]();
    g->visited_hash_set.size = size * 2 + 1;
    g->visited_hash_set.keys = new ggml_tensor *[g->visited_hash_set.size]();
    g->visited_hash_set.used = new ggml_bitset_t[(g->visited_hash_set.size + 31) / 32]();
    g->use_counts = new int32_t[g->visited_hash_set.size]();
    g->order = GGML_CGRAPH_EVAL_ORDER_LEFT_TO_RIGHT;
    ctx->graphs.push_back(g);
    return g;
}
struct ggml_cgraph * ggml_new_graph(struct ggml_context * ctx) { return ggml_new_graph_custom(ctx, GGML_DEFAULT_GRAPH_SIZE, false); }

// ggml_hash_find (ggml-impl.h): hash = pointer >> 4, linear probing over the `used` bitset
static size_t hash_find(const ggml_hash_set * hs, const ggml_tensor * key) {
    const size_t h = ((size_t) (uintptr_t) key >> 4) % hs->size;
    size_t i = h;
    while ((hs->used[i >> 5] & (1u << (i & 31))) && hs->keys[i] != key) {
        i = (i + 1) % hs->size;
        LITE_ASSERT(i != h && "graph hash set full");
    }
    return i;
}

// ggml_visit_parents: depth-first, and — as upstream since the fusion helpers (ggml_can_fuse) — counts every (node, src slot)
// reference in use_counts[hash slot of the operand]; graph views (ggml_graph_view, what ggml_backend_sched hands a backend
// for each split) share the parent's table, so a backend sees the WHOLE graph's counts for the nodes of its split
static size_t visit_parents(ggml_cgraph * g, ggml_tensor * node) {
    const size_t pos = hash_find(&g->visited_hash_set, node);
    if (g->visited_hash_set.used[pos >> 5] & (1u << (pos & 31))) return pos;
    g->visited_hash_set.keys[pos] = node;
    g->visited_hash_set.used[pos >> 5] |= 1u << (pos & 31);
    g->use_counts[pos] = 0;
    for (int i = 0; i < GGML_MAX_SRC; ++i)
        if (node->src[i]) {
            const size_t sp = visit_parents(g, node->src[i]);
            g->use_counts[sp]++;
        }
    if (node->op == GGML_OP_NONE && !(node->flags & GGML_TENSOR_FLAG_PARAM)) {
        LITE_ASSERT(g->n_leafs < g->size);
        if (node->name[0] == 0) format_name(node, "leaf_%d", g->n_leafs);
        g->leafs[g->n_leafs++] = node;
    } else {
        LITE_ASSERT(g->n_nodes < g->size);
        if (node->name[0] == 0) format_name(node, "node_%d", g->n_nodes);
        g->nodes[g->n_nodes++] = node;
    }
    return pos;
}
// ggml_graph_view (ggml.c): nodes [i0, i1) of cgraph0, returned by value; no leafs / grads; the hash set and the use counts
// are the parent's
struct ggml_cgraph ggml_graph_view(struct ggml_cgraph * cgraph0, int i0, int i1) {
    LITE_ASSERT(0 <= i0 && i0 <= i1 && i1 <= cgraph0->n_nodes);
    ggml_cgraph g;
    memset(&g, 0, sizeof(g));
    g.n_nodes = i1 - i0;
    g.nodes = cgraph0->nodes + i0;
    g.use_counts = cgraph0->use_counts;
    g.visited_hash_set = cgraph0->visited_hash_set;
    g.order = cgraph0->order;
    return g;
}
void ggml_build_forward_expand(struct ggml_cgraph * cgraph, struct ggml_tensor * tensor) { visit_parents(cgraph, tensor); }
int ggml_graph_n_nodes(struct ggml_cgraph * cgraph) { return cgraph->n_nodes; }
struct ggml_tensor * ggml_graph_node(struct ggml_cgraph * cgraph, int i) {
    if (i < 0) i += cgraph->n_nodes;
    LITE_ASSERT(i >= 0 && i < cgraph->n_nodes);
    return cgraph->nodes[i];
}

// ================================================================================================ backend API
ggml_backend_reg_t ggml_backend_load(const char * path) {
    void * h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        fprintf(stderr, "ggml_backend_load: failed to load %s: %s\n", path, dlerror());
        return nullptr;
    }
    auto score_fn = (ggml_backend_score_t) dlsym(h, "ggml_backend_score");
    if (score_fn && score_fn() == 0) {
        fprintf(stderr, "ggml_backend_load: backend %s is not supported on this system\n", path);
        dlclose(h);
        return nullptr;
    }
    auto init_fn = (ggml_backend_init_t) dlsym(h, "ggml_backend_init");
    if (!init_fn) {
        fprintf(stderr, "ggml_backend_load: failed to find ggml_backend_init in %s\n", path);
        dlclose(h);
        return nullptr;
    }
    ggml_backend_reg_t reg = init_fn();
    if (!reg || reg->api_version != GGML_BACKEND_API_VERSION) {
        if (!reg) fprintf(stderr, "ggml_backend_load: failed to initialize backend from %s: ggml_backend_init returned NULL\n", path);
        else fprintf(stderr, "ggml_backend_load: failed to initialize backend from %s: incompatible API version (backend: %d, current: %d)\n", path, reg->api_version, GGML_BACKEND_API_VERSION);
        dlclose(h);
        return nullptr;
    }
    return reg;  // the handle stays open for the life of the process
}
const char * ggml_backend_reg_name(ggml_backend_reg_t reg) { return reg->iface.get_name(reg); }
size_t ggml_backend_reg_dev_count(ggml_backend_reg_t reg) { return reg->iface.get_device_count(reg); }
ggml_backend_dev_t ggml_backend_reg_dev_get(ggml_backend_reg_t reg, size_t index) { return reg->iface.get_device(reg, index); }
void * ggml_backend_reg_get_proc_address(ggml_backend_reg_t reg, const char * name) {
    return reg->iface.get_proc_address ? reg->iface.get_proc_address(reg, name) : nullptr;
}
const char * ggml_backend_dev_name(ggml_backend_dev_t dev) { return dev->iface.get_name(dev); }
const char * ggml_backend_dev_description(ggml_backend_dev_t dev) { return dev->iface.get_description(dev); }
void ggml_backend_dev_memory(ggml_backend_dev_t dev, size_t * free, size_t * total) { dev->iface.get_memory(dev, free, total); }
enum ggml_backend_dev_type ggml_backend_dev_type(ggml_backend_dev_t dev) { return dev->iface.get_type(dev); }
void ggml_backend_dev_get_props(ggml_backend_dev_t dev, struct ggml_backend_dev_props * props) {
    memset(props, 0, sizeof(*props));
    dev->iface.get_props(dev, props);
}
ggml_backend_t ggml_backend_dev_init(ggml_backend_dev_t dev, const char * params) { return dev->iface.init_backend(dev, params); }
ggml_backend_buffer_type_t ggml_backend_dev_buffer_type(ggml_backend_dev_t dev) { return dev->iface.get_buffer_type(dev); }
ggml_backend_buffer_type_t ggml_backend_dev_host_buffer_type(ggml_backend_dev_t dev) {
    return dev->iface.get_host_buffer_type ? dev->iface.get_host_buffer_type(dev) : nullptr;
}
bool ggml_backend_dev_supports_op(ggml_backend_dev_t dev, const struct ggml_tensor * op) { return dev->iface.supports_op(dev, op); }
bool ggml_backend_dev_supports_buft(ggml_backend_dev_t dev, ggml_backend_buffer_type_t buft) { return dev->iface.supports_buft(dev, buft); }

const char * ggml_backend_buft_name(ggml_backend_buffer_type_t buft) { return buft->iface.get_name(buft); }
ggml_backend_buffer_t ggml_backend_buft_alloc_buffer(ggml_backend_buffer_type_t buft, size_t size) {
    if (size == 0) size = 1;  // upstream returns a dummy buffer for zero-sized allocations; keep callers simple
    return buft->iface.alloc_buffer(buft, size);
}
size_t ggml_backend_buft_get_alignment(ggml_backend_buffer_type_t buft) { return buft->iface.get_alignment(buft); }
size_t ggml_backend_buft_get_alloc_size(ggml_backend_buffer_type_t buft, const struct ggml_tensor * tensor) {
    if (buft->iface.get_alloc_size) {
        size_t s = buft->iface.get_alloc_size(buft, tensor);
        LITE_ASSERT(s >= ggml_nbytes(tensor));
        return s;
    }
    return ggml_nbytes(tensor);
}
bool ggml_backend_buft_is_host(ggml_backend_buffer_type_t buft) { return buft->iface.is_host ? buft->iface.is_host(buft) : false; }

ggml_backend_buffer_t ggml_backend_buffer_init(ggml_backend_buffer_type_t buft, struct ggml_backend_buffer_i iface, void * context, size_t size) {
    ggml_backend_buffer_t b = new ggml_backend_buffer{iface, buft, context, size, GGML_BACKEND_BUFFER_USAGE_ANY};
    return b;
}
void ggml_backend_buffer_free(ggml_backend_buffer_t buffer) {
    if (!buffer) return;
    if (buffer->iface.free_buffer) buffer->iface.free_buffer(buffer);
    delete buffer;
}
void * ggml_backend_buffer_get_base(ggml_backend_buffer_t buffer) { return buffer->iface.get_base(buffer); }
size_t ggml_backend_buffer_get_size(ggml_backend_buffer_t buffer) { return buffer->size; }
void ggml_backend_buffer_clear(ggml_backend_buffer_t buffer, uint8_t value) { buffer->iface.clear(buffer, value); }
void ggml_backend_buffer_set_usage(ggml_backend_buffer_t buffer, enum ggml_backend_buffer_usage usage) { buffer->usage = usage; }
bool ggml_backend_buffer_is_host(ggml_backend_buffer_t buffer) { return ggml_backend_buft_is_host(buffer->buft); }

const char * ggml_backend_name(ggml_backend_t backend) { return backend->iface.get_name(backend); }
void ggml_backend_free(ggml_backend_t backend) {
    if (backend) backend->iface.free(backend);
}
static ggml_backend_buffer_t tensor_buffer(const ggml_tensor * t) { return t->view_src ? t->view_src->buffer : t->buffer; }
void ggml_backend_tensor_set(struct ggml_tensor * tensor, const void * data, size_t offset, size_t size) {
    ggml_backend_buffer_t buf = tensor_buffer(tensor);
    if (size == 0) return;
    LITE_ASSERT(buf != nullptr && tensor->data != nullptr && offset + size <= ggml_nbytes(tensor));
    buf->iface.set_tensor(buf, tensor, data, offset, size);
}
void ggml_backend_tensor_get(const struct ggml_tensor * tensor, void * data, size_t offset, size_t size) {
    ggml_backend_buffer_t buf = tensor_buffer(tensor);
    if (size == 0) return;
    LITE_ASSERT(buf != nullptr && tensor->data != nullptr && offset + size <= ggml_nbytes(tensor));
    buf->iface.get_tensor(buf, tensor, data, offset, size);
}
void ggml_backend_tensor_memset(struct ggml_tensor * tensor, uint8_t value, size_t offset, size_t size) {
    ggml_backend_buffer_t buf = tensor_buffer(tensor);
    if (size == 0) return;
    LITE_ASSERT(buf != nullptr && tensor->data != nullptr && offset + size <= ggml_nbytes(tensor));
    buf->iface.memset_tensor(buf, tensor, value, offset, size);
}
void ggml_backend_tensor_set_async(ggml_backend_t backend, struct ggml_tensor * tensor, const void * data, size_t offset, size_t size) {
    if (backend->iface.set_tensor_async == nullptr) ggml_backend_tensor_set(tensor, data, offset, size);
    else backend->iface.set_tensor_async(backend, tensor, data, offset, size);
}
void ggml_backend_tensor_get_async(ggml_backend_t backend, const struct ggml_tensor * tensor, void * data, size_t offset, size_t size) {
    if (backend->iface.get_tensor_async == nullptr) ggml_backend_tensor_get(tensor, data, offset, size);
    else backend->iface.get_tensor_async(backend, tensor, data, offset, size);
}
void ggml_backend_synchronize(ggml_backend_t backend) {
    if (backend->iface.synchronize) backend->iface.synchronize(backend);
}
enum ggml_status ggml_backend_graph_compute_async(ggml_backend_t backend, struct ggml_cgraph * cgraph) { return backend->iface.graph_compute(backend, cgraph); }
enum ggml_status ggml_backend_graph_compute(ggml_backend_t backend, struct ggml_cgraph * cgraph) {
    enum ggml_status st = ggml_backend_graph_compute_async(backend, cgraph);
    ggml_backend_synchronize(backend);
    return st;
}
bool ggml_backend_supports_op(ggml_backend_t backend, const struct ggml_tensor * op) { return ggml_backend_dev_supports_op(backend->device, op); }

// ---- tensor copies between backends and events, as ggml-backend.cpp implements them (what ggml_backend_sched issues between the devices of a
// layer split: -sm layer, llama.cpp's default; /root/reference/llama-box/engine_param.hpp:900-916)
static bool same_layout(const struct ggml_tensor * a, const struct ggml_tensor * b) {
    if (a->type != b->type) return false;
    for (int i = 0; i < GGML_MAX_DIMS; ++i)
        if (a->ne[i] != b->ne[i] || a->nb[i] != b->nb[i]) return false;
    return true;
}
void ggml_backend_tensor_copy(struct ggml_tensor * src, struct ggml_tensor * dst) {
    LITE_ASSERT(same_layout(src, dst) && "cannot copy tensors with different layouts");
    if (src == dst) return;
    ggml_backend_buffer_t sb = src->view_src ? src->view_src->buffer : src->buffer;
    ggml_backend_buffer_t db = dst->view_src ? dst->view_src->buffer : dst->buffer;
    if (ggml_backend_buffer_is_host(sb)) {
        ggml_backend_tensor_set(dst, src->data, 0, ggml_nbytes(src));
    } else if (ggml_backend_buffer_is_host(db)) {
        ggml_backend_tensor_get(src, dst->data, 0, ggml_nbytes(src));
    } else if (!(db->iface.cpy_tensor && db->iface.cpy_tensor(db, src, dst))) {
        const size_t n = ggml_nbytes(src);  // (the slow way, through host memory)
        void * tmp = malloc(n);
        LITE_ASSERT(tmp);
        ggml_backend_tensor_get(src, tmp, 0, n);
        ggml_backend_tensor_set(dst, tmp, 0, n);
        free(tmp);
    }
}
void ggml_backend_tensor_copy_async(ggml_backend_t backend_src, ggml_backend_t backend_dst, struct ggml_tensor * src, struct ggml_tensor * dst) {
    LITE_ASSERT(same_layout(src, dst) && "cannot copy tensors with different layouts");
    if (src == dst) return;
    if (backend_dst->iface.cpy_tensor_async != nullptr && backend_dst->iface.cpy_tensor_async(backend_src, backend_dst, src, dst)) return;
    // an async copy would happen after everything queued on both backends: synchronise both, then copy
    ggml_backend_synchronize(backend_src);
    ggml_backend_synchronize(backend_dst);
    ggml_backend_tensor_copy(src, dst);
}
ggml_backend_event_t ggml_backend_event_new(ggml_backend_dev_t device) {
    if (device == nullptr || device->iface.event_new == nullptr) return nullptr;  // (a device without events: the scheduler synchronises instead)
    return device->iface.event_new(device);
}
void ggml_backend_event_free(ggml_backend_event_t event) {
    if (event == nullptr) return;
    event->device->iface.event_free(event->device, event);
}
void ggml_backend_event_record(ggml_backend_event_t event, ggml_backend_t backend) {
    LITE_ASSERT(backend->iface.event_record != nullptr);
    backend->iface.event_record(backend, event);
}
void ggml_backend_event_synchronize(ggml_backend_event_t event) {
    LITE_ASSERT(event->device->iface.event_synchronize);
    event->device->iface.event_synchronize(event->device, event);
}
void ggml_backend_event_wait(ggml_backend_t backend, ggml_backend_event_t event) {
    LITE_ASSERT(backend->iface.event_wait != nullptr);
    backend->iface.event_wait(backend, event);
}

// ---------------------------------------------------------------------------------- host (malloc) buffer type
static const char * cpu_buft_name(ggml_backend_buffer_type_t) { return "CPU"; }
static void cpu_buf_free(ggml_backend_buffer_t b) { free(b->context); }
static void * cpu_buf_base(ggml_backend_buffer_t b) { return b->context; }
static void cpu_buf_memset(ggml_backend_buffer_t, ggml_tensor * t, uint8_t v, size_t off, size_t sz) { memset((char *) t->data + off, v, sz); }
static void cpu_buf_set(ggml_backend_buffer_t, ggml_tensor * t, const void * d, size_t off, size_t sz) { memcpy((char *) t->data + off, d, sz); }
static void cpu_buf_get(ggml_backend_buffer_t, const ggml_tensor * t, void * d, size_t off, size_t sz) { memcpy(d, (const char *) t->data + off, sz); }
static void cpu_buf_clear(ggml_backend_buffer_t b, uint8_t v) { memset(b->context, v, b->size); }
// Large host buffers (the CPU copy of a model: bench.py's cpu_baseline, the oracle side of the model tests) are spread over the host's
// memory controllers before anything is written: MPOL_INTERLEAVE over every online NUMA node where the kernel lets us (mbind by raw
// syscall: no libnuma in the image), and in any case a first touch of the pages by many threads at once — striped 2 MiB apart — instead of
// by the one thread that later fills the weights.  A model whose pages all sit on the loading thread's node is read through ONE socket's
// memory channels whatever the thread count (53 GB/s on a host with > 400 GB/s).
static void spread_pages(char * p, size_t size) {
    if (size < ((size_t) 64 << 20)) return;
    unsigned long mask[16] = {0};
    int n_nodes = 0;
    for (int nd = 0; nd < 1024; ++nd) {
        char path[64];
        snprintf(path, sizeof(path), "/sys/devices/system/node/node%d", nd);
        if (access(path, F_OK) != 0) break;
        mask[nd / (8 * sizeof(unsigned long))] |= 1ul << (nd % (8 * sizeof(unsigned long)));
        n_nodes++;
    }
#if defined(SYS_mbind)
    if (n_nodes > 1) {
        char * a = (char *) (((uintptr_t) p + 4095) & ~(uintptr_t) 4095);
        // GGML_LITE_NUMA_NODE=n: everything on node n instead (MPOL_PREFERRED) — for a reader team pinned to that node's cores (bench.py's CPU leg on a
        // container whose CPU quota is a fraction of one socket: all-local reads instead of half of them crossing the socket link)
        const char * one = getenv("GGML_LITE_NUMA_NODE");
        if (one && atoi(one) >= 0 && atoi(one) < n_nodes) {
            unsigned long m1[16] = {0};
            m1[atoi(one) / (8 * sizeof(unsigned long))] = 1ul << (atoi(one) % (8 * sizeof(unsigned long)));
            (void) syscall(SYS_mbind, a, (size - (size_t) (a - p)) & ~(size_t) 4095, 1 /* MPOL_PREFERRED */, m1, sizeof(m1) * 8, 0);
        } else
            (void) syscall(SYS_mbind, a, (size - (size_t) (a - p)) & ~(size_t) 4095, 3 /* MPOL_INTERLEAVE */, mask, sizeof(mask) * 8, 0);  // (refused under some seccomp profiles: the touch below still spreads)
    }
#endif
    const unsigned T = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
    const size_t stripe = (size_t) 2 << 20;
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; ++t)
        th.emplace_back([=] {
            for (size_t s0 = (size_t) t * stripe; s0 < size; s0 += (size_t) T * stripe)
                for (size_t o = s0; o < std::min(size, s0 + stripe); o += 4096) ((volatile char *) p)[o] = 0;
        });
    for (auto & x : th) x.join();
}
static ggml_backend_buffer_t cpu_buft_alloc(ggml_backend_buffer_type_t buft, size_t size) {
    void * p = nullptr;
    if (posix_memalign(&p, 64, size + 64) != 0) return nullptr;
    spread_pages((char *) p, size + 64);
    ggml_backend_buffer_i iface = {cpu_buf_free, cpu_buf_base, nullptr, cpu_buf_memset, cpu_buf_set, cpu_buf_get, nullptr, cpu_buf_clear, nullptr};
    return ggml_backend_buffer_init(buft, iface, p, size);
}
static size_t cpu_buft_align(ggml_backend_buffer_type_t) { return 32; }
static bool cpu_buft_is_host(ggml_backend_buffer_type_t) { return true; }
ggml_backend_buffer_type_t ggml_backend_cpu_buffer_type(void) {
    static ggml_backend_buffer_type buft = {{cpu_buft_name, cpu_buft_alloc, cpu_buft_align, nullptr, nullptr, cpu_buft_is_host}, nullptr, nullptr};
    return &buft;
}

// ------------------------------------------------------------------------------------------------ allocators
static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static enum ggml_status init_tensor_in(ggml_backend_buffer_t buf, ggml_tensor * t) {
    t->buffer = buf;
    if (buf->iface.init_tensor) return buf->iface.init_tensor(buf, t);
    return GGML_STATUS_SUCCESS;
}

ggml_backend_buffer_t ggml_backend_alloc_ctx_tensors_from_buft(struct ggml_context * ctx, ggml_backend_buffer_type_t buft) {
    const size_t align = ggml_backend_buft_get_alignment(buft);
    size_t total = 0;
    for (auto * t : ctx->tensors)
        if (t->data == nullptr && t->view_src == nullptr) total = align_up(total, align) + ggml_backend_buft_get_alloc_size(buft, t);
    if (total == 0) return nullptr;
    ggml_backend_buffer_t buf = ggml_backend_buft_alloc_buffer(buft, total);
    if (!buf) return nullptr;
    char * base = (char *) ggml_backend_buffer_get_base(buf);
    size_t off = 0;
    for (auto * t : ctx->tensors) {
        if (t->data == nullptr && t->view_src == nullptr) {
            off = align_up(off, align);
            t->data = base + off;
            off += ggml_backend_buft_get_alloc_size(buft, t);
            init_tensor_in(buf, t);
        }
    }
    for (auto * t : ctx->tensors) {
        if (t->view_src != nullptr && t->data == nullptr && t->view_src->data != nullptr) {
            t->data = (char *) t->view_src->data + t->view_offs;
            init_tensor_in(t->view_src->buffer, t);
        }
    }
    return buf;
}

struct ggml_gallocr {
    ggml_backend_buffer_type_t buft = nullptr;
    ggml_backend_buffer_t buffer = nullptr;
    size_t buffer_size = 0;
    // address ranges this allocator has handed out (current buffer + any it outgrew): a tensor whose data
    // lies in one of them was placed by us and may be re-placed; anything else is externally owned
    std::vector<std::pair<const char *, const char *>> ranges;
    bool owns(const void * p) const {
        for (auto & r : ranges)
            if ((const char *) p >= r.first && (const char *) p < r.second) return true;
        return false;
    }
};

struct alloc_plan {
    std::vector<std::pair<ggml_tensor *, size_t>> placed;
    size_t total = 0;
};

// Liveness-based planner: a node output is released after the last node that reads it (directly or through a
// view); released blocks are re-used best-fit.  Deterministic, so identical graphs get identical addresses —
// which is what lets the backend replay a captured hipGraph across calls.
static alloc_plan plan_graph(ggml_gallocr_t ga, ggml_cgraph * g) {
    alloc_plan plan;
    const size_t align = ggml_backend_buft_get_alignment(ga->buft);
    auto root = [](ggml_tensor * t) { return t->view_src ? t->view_src : t; };
    auto needs_alloc = [&](ggml_tensor * t) { return t->view_src == nullptr && (t->data == nullptr || ga->owns(t->data)); };
    std::unordered_map<ggml_tensor *, int> last_use;
    for (int i = 0; i < g->n_nodes; ++i) {
        ggml_tensor * n = g->nodes[i];
        last_use[root(n)] = std::max(last_use.count(root(n)) ? last_use[root(n)] : -1, i);
        for (int s = 0; s < GGML_MAX_SRC; ++s)
            if (n->src[s]) last_use[root(n->src[s])] = i;
    }
    struct blk { size_t off, size; };
    std::vector<blk> free_list;
    std::unordered_map<ggml_tensor *, blk> live;
    auto take = [&](ggml_tensor * t) {
        const size_t need = align_up(ggml_backend_buft_get_alloc_size(ga->buft, t), align);
        int best = -1;
        for (size_t i = 0; i < free_list.size(); ++i)
            if (free_list[i].size >= need && (best < 0 || free_list[i].size < free_list[best].size)) best = (int) i;
        size_t off;
        if (best >= 0) {
            off = free_list[best].off;
            if (free_list[best].size > need) {
                free_list[best].off += need;
                free_list[best].size -= need;
            } else {
                free_list.erase(free_list.begin() + best);
            }
        } else {
            off = plan.total;
            plan.total += need;
        }
        live[t] = {off, need};
        plan.placed.push_back({t, off});
    };
    auto release = [&](ggml_tensor * t) {
        auto it = live.find(t);
        if (it == live.end()) return;
        blk b = it->second;
        live.erase(it);
        // coalesce with neighbours
        for (size_t i = 0; i < free_list.size();) {
            if (free_list[i].off + free_list[i].size == b.off) {
                b.off = free_list[i].off;
                b.size += free_list[i].size;
                free_list.erase(free_list.begin() + i);
            } else if (b.off + b.size == free_list[i].off) {
                b.size += free_list[i].size;
                free_list.erase(free_list.begin() + i);
            } else {
                ++i;
            }
        }
        free_list.push_back(b);
    };
    for (int i = 0; i < g->n_leafs; ++i)
        if (needs_alloc(g->leafs[i])) take(g->leafs[i]);  // inputs: never released
    // ggml-alloc runs element-wise / row-wise ops IN PLACE when the parent is not needed afterwards (ggml_op_can_inplace): the
    // result takes over the parent's block.  Mirrored here because it decides which tensors alias which in the graphs a backend
    // sees (and a backend must be correct with dst == src).
    auto can_inplace = [](const ggml_tensor * t) {
        switch (t->op) {
            case GGML_OP_SCALE: case GGML_OP_ADD: case GGML_OP_SUB: case GGML_OP_MUL: case GGML_OP_DIV: case GGML_OP_UNARY: case GGML_OP_ROPE:
            case GGML_OP_RMS_NORM: case GGML_OP_SOFT_MAX:
                return true;
            default:
                return false;
        }
    };
    auto same_layout = [](const ggml_tensor * a, const ggml_tensor * b) {
        if (a->type != b->type) return false;
        for (int d = 0; d < GGML_MAX_DIMS; ++d)
            if (a->ne[d] != b->ne[d] || a->nb[d] != b->nb[d]) return false;
        return true;
    };
    static const bool no_inplace = getenv("GGML_LITE_NO_INPLACE") != nullptr;
    for (int i = 0; i < g->n_nodes; ++i) {
        ggml_tensor * n = g->nodes[i];
        bool placed_inplace = false;
        if (needs_alloc(n) && can_inplace(n) && !no_inplace && !ggml_lite_no_reuse()) {  // (no-reuse keeps EVERY node's own result readable afterwards)
            for (int s = 0; s < GGML_MAX_SRC && !placed_inplace; ++s) {
                ggml_tensor * p = n->src[s];
                if (!p || !same_layout(n, p)) continue;
                ggml_tensor * r = root(p);
                auto it = live.find(r);
                if (it == live.end() || last_use[r] != i || (r->flags & (GGML_TENSOR_FLAG_OUTPUT | GGML_TENSOR_FLAG_INPUT)) || r->op == GGML_OP_NONE) continue;
                if (p != r && p->view_offs != 0) continue;                       // a view that does not start at its source
                if (ggml_nbytes(r) < ggml_nbytes(n)) continue;
                bool other_src = false;                                          // the parent feeds this node twice (x * x): keep it simple
                for (int s2 = 0; s2 < GGML_MAX_SRC; ++s2) other_src = other_src || (s2 != s && n->src[s2] && root(n->src[s2]) == r);
                if (other_src) continue;
                const blk b = it->second;
                live.erase(it);
                live[n] = b;
                plan.placed.push_back({n, b.off});
                placed_inplace = true;
            }
        }
        if (!placed_inplace && needs_alloc(n)) take(n);
        for (int s = 0; s < GGML_MAX_SRC; ++s) {
            if (!n->src[s]) continue;
            ggml_tensor * r = root(n->src[s]);
            if (r->op == GGML_OP_NONE || (r->flags & (GGML_TENSOR_FLAG_OUTPUT | GGML_TENSOR_FLAG_INPUT))) continue;
            if (last_use[r] == i && !ggml_lite_no_reuse()) release(r);
        }
    }
    return plan;
}

ggml_gallocr_t ggml_gallocr_new(ggml_backend_buffer_type_t buft) {
    ggml_gallocr_t ga = new ggml_gallocr();
    ga->buft = buft;
    return ga;
}
void ggml_gallocr_free(ggml_gallocr_t ga) {
    if (!ga) return;
    ggml_backend_buffer_free(ga->buffer);
    delete ga;
}
static bool ensure_buffer(ggml_gallocr_t ga, size_t size) {
    if (ga->buffer && ga->buffer_size >= size) return true;
    ggml_backend_buffer_free(ga->buffer);
    ga->buffer = ggml_backend_buft_alloc_buffer(ga->buft, size);
    ga->buffer_size = ga->buffer ? size : 0;
    if (ga->buffer) {
        ggml_backend_buffer_set_usage(ga->buffer, GGML_BACKEND_BUFFER_USAGE_COMPUTE);
        const char * b = (const char *) ggml_backend_buffer_get_base(ga->buffer);
        ga->ranges.push_back({b, b + size});
    }
    return ga->buffer != nullptr;
}
bool ggml_gallocr_reserve(ggml_gallocr_t ga, struct ggml_cgraph * graph) {
    alloc_plan plan = plan_graph(ga, graph);
    return ensure_buffer(ga, plan.total + 256);
}
bool ggml_gallocr_alloc_graph(ggml_gallocr_t ga, struct ggml_cgraph * graph) {
    alloc_plan plan = plan_graph(ga, graph);
    if (!ensure_buffer(ga, plan.total + 256)) return false;
    char * base = (char *) ggml_backend_buffer_get_base(ga->buffer);
    for (auto & p : plan.placed) {
        p.first->data = base + p.second;
        init_tensor_in(ga->buffer, p.first);
    }
    auto fix_view = [&](ggml_tensor * t) {
        if (t->view_src != nullptr) {
            LITE_ASSERT(t->view_src->data != nullptr);
            t->data = (char *) t->view_src->data + t->view_offs;
            init_tensor_in(t->view_src->buffer, t);
        }
    };
    for (int i = 0; i < graph->n_leafs; ++i) fix_view(graph->leafs[i]);
    for (int i = 0; i < graph->n_nodes; ++i) fix_view(graph->nodes[i]);
    return true;
}
size_t ggml_gallocr_get_buffer_size(ggml_gallocr_t ga, int) { return ga->buffer_size; }

}  // extern "C"
