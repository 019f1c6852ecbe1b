/*
 * ggml_lite.h — host-side mirror of the slice of the ggml public API that the hot path's callers use.
 *
 * The reference (llama-box -> llama.cpp) builds a ggml_cgraph with ggml.h's constructors and hands it
 * to the backend through ggml-backend.h (llama-box/rpcserver.hpp:1339-1393 rebuilds such a graph from
 * the wire and calls ggml_backend_graph_compute; llama-box/httpserver.hpp:3591 reaches it through
 * llama_decode).  ggml itself is absent from /root/reference (un-vendored submodule), so this file
 * provides our OWN implementation of the same entry points — same names, argument meaning and error
 * behaviour — so that (1) the backend .so can be exercised exactly the way ggml would exercise it and
 * (2) the parity tests read like upstream's test-backend-ops.  It is a driver/test harness: the
 * product is the backend library (csrc/), which depends on none of this.
 */
#ifndef GGML_LITE_H
#define GGML_LITE_H
#include "../../include/ggml_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

struct ggml_context;
struct ggml_init_params {
    size_t mem_size;   /* ignored: tensors are heap objects owned by the context */
    void * mem_buffer; /* ignored */
    bool no_alloc;     /* always treated as true: data is placed by the backend allocators below */
};

struct ggml_context * ggml_init(struct ggml_init_params params);
void ggml_free(struct ggml_context * ctx);

/* type helpers */
size_t ggml_type_size(enum ggml_type type);
int64_t ggml_blck_size(enum ggml_type type);
size_t ggml_row_size(enum ggml_type type, int64_t ne);
const char * ggml_type_name(enum ggml_type type);
const char * ggml_op_name(enum ggml_op op);
size_t ggml_nbytes(const struct ggml_tensor * t);
int64_t ggml_nelements(const struct ggml_tensor * t);
int64_t ggml_nrows(const struct ggml_tensor * t);
bool ggml_is_contiguous(const struct ggml_tensor * t);
bool ggml_is_quantized(enum ggml_type type);

/* tensors */
struct ggml_tensor * ggml_new_tensor(struct ggml_context * ctx, enum ggml_type type, int n_dims, const int64_t * ne);
struct ggml_tensor * ggml_new_tensor_1d(struct ggml_context * ctx, enum ggml_type type, int64_t ne0);
struct ggml_tensor * ggml_new_tensor_2d(struct ggml_context * ctx, enum ggml_type type, int64_t ne0, int64_t ne1);
struct ggml_tensor * ggml_new_tensor_3d(struct ggml_context * ctx, enum ggml_type type, int64_t ne0, int64_t ne1, int64_t ne2);
struct ggml_tensor * ggml_new_tensor_4d(struct ggml_context * ctx, enum ggml_type type, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3);
struct ggml_tensor * ggml_set_name(struct ggml_tensor * t, const char * name);
void ggml_set_input(struct ggml_tensor * t);
void ggml_set_output(struct ggml_tensor * t);
struct ggml_tensor * ggml_get_first_tensor(const struct ggml_context * ctx);
struct ggml_tensor * ggml_get_next_tensor(const struct ggml_context * ctx, struct ggml_tensor * t);
struct ggml_tensor * ggml_get_tensor(struct ggml_context * ctx, const char * name);

/* views (metadata only) */
struct ggml_tensor * ggml_view_1d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, size_t offset);
struct ggml_tensor * ggml_view_2d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1, size_t nb1, size_t offset);
struct ggml_tensor * ggml_view_3d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1, int64_t ne2, size_t nb1, size_t nb2, size_t offset);
struct ggml_tensor * ggml_view_4d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3, size_t nb1, size_t nb2, size_t nb3, size_t offset);
struct ggml_tensor * ggml_reshape_1d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0);
struct ggml_tensor * ggml_reshape_2d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1);
struct ggml_tensor * ggml_reshape_3d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1, int64_t ne2);
struct ggml_tensor * ggml_reshape_4d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3);
struct ggml_tensor * ggml_permute(struct ggml_context * ctx, struct ggml_tensor * a, int axis0, int axis1, int axis2, int axis3);
struct ggml_tensor * ggml_transpose(struct ggml_context * ctx, struct ggml_tensor * a);

/* ops */
struct ggml_tensor * ggml_dup(struct ggml_context * ctx, struct ggml_tensor * a);
struct ggml_tensor * ggml_cont(struct ggml_context * ctx, struct ggml_tensor * a);
struct ggml_tensor * ggml_cont_2d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1);
struct ggml_tensor * ggml_cpy(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b);
struct ggml_tensor * ggml_cast(struct ggml_context * ctx, struct ggml_tensor * a, enum ggml_type type);
struct ggml_tensor * ggml_add(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b);
struct ggml_tensor * ggml_sub(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b);
struct ggml_tensor * ggml_mul(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b);
struct ggml_tensor * ggml_div(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b);
struct ggml_tensor * ggml_scale(struct ggml_context * ctx, struct ggml_tensor * a, float s);
struct ggml_tensor * ggml_scale_bias(struct ggml_context * ctx, struct ggml_tensor * a, float s, float b);
struct ggml_tensor * ggml_rms_norm(struct ggml_context * ctx, struct ggml_tensor * a, float eps);
struct ggml_tensor * ggml_mul_mat(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b);
void ggml_mul_mat_set_prec(struct ggml_tensor * a, enum ggml_prec prec);
struct ggml_tensor * ggml_get_rows(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b);
struct ggml_tensor * ggml_set_rows(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b, struct ggml_tensor * c);
struct ggml_tensor * ggml_silu(struct ggml_context * ctx, struct ggml_tensor * a);
struct ggml_tensor * ggml_unary(struct ggml_context * ctx, struct ggml_tensor * a, enum ggml_unary_op op);
struct ggml_tensor * ggml_swiglu(struct ggml_context * ctx, struct ggml_tensor * a);
struct ggml_tensor * ggml_swiglu_split(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b);
struct ggml_tensor * ggml_soft_max(struct ggml_context * ctx, struct ggml_tensor * a);
struct ggml_tensor * ggml_soft_max_ext(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * mask, float scale, float max_bias);
void ggml_soft_max_add_sinks(struct ggml_tensor * a, struct ggml_tensor * sinks);
struct ggml_tensor * ggml_rope_ext(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b, struct ggml_tensor * c,
                                   int n_dims, int mode, int n_ctx_orig, float freq_base, float freq_scale, float ext_factor,
                                   float attn_factor, float beta_fast, float beta_slow);
struct ggml_tensor * ggml_rope_multi(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b, struct ggml_tensor * c,
                                     int n_dims, int sections[4], int mode, int n_ctx_orig, float freq_base, float freq_scale,
                                     float ext_factor, float attn_factor, float beta_fast, float beta_slow);
struct ggml_tensor * ggml_rope_ext_inplace(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b, struct ggml_tensor * c,
                                   int n_dims, int mode, int n_ctx_orig, float freq_base, float freq_scale, float ext_factor,
                                   float attn_factor, float beta_fast, float beta_slow);
struct ggml_tensor * ggml_flash_attn_ext(struct ggml_context * ctx, struct ggml_tensor * q, struct ggml_tensor * k, struct ggml_tensor * v,
                                         struct ggml_tensor * mask, float scale, float max_bias, float logit_softcap);
void ggml_flash_attn_ext_set_prec(struct ggml_tensor * a, enum ggml_prec prec);
void ggml_flash_attn_ext_add_sinks(struct ggml_tensor * a, struct ggml_tensor * sinks);
struct ggml_tensor * ggml_argmax(struct ggml_context * ctx, struct ggml_tensor * a);

/* graphs */
struct ggml_cgraph * ggml_new_graph(struct ggml_context * ctx);
struct ggml_cgraph * ggml_new_graph_custom(struct ggml_context * ctx, size_t size, bool grads);
void ggml_build_forward_expand(struct ggml_cgraph * cgraph, struct ggml_tensor * tensor);
void ggml_lite_set_no_reuse(int on);  // harness only: graph allocator keeps every intermediate (per-node comparisons)
struct ggml_cgraph ggml_graph_view(struct ggml_cgraph * cgraph, int i0, int i1);  // shares the parent's hash set + use counts
int ggml_graph_n_nodes(struct ggml_cgraph * cgraph);
struct ggml_tensor * ggml_graph_node(struct ggml_cgraph * cgraph, int i);

/* ------------------------- ggml-backend.h mirror (generic wrappers over the vtables) ------------------------- */
/* dlopen()s a backend library and calls its ggml_backend_init(); NULL (with a message on stderr) on failure:
 * missing symbol, api_version mismatch (same checks as ggml_backend_load in ggml-backend-reg.cpp). */
ggml_backend_reg_t ggml_backend_load(const char * path);
const char * ggml_backend_reg_name(ggml_backend_reg_t reg);
size_t ggml_backend_reg_dev_count(ggml_backend_reg_t reg);
ggml_backend_dev_t ggml_backend_reg_dev_get(ggml_backend_reg_t reg, size_t index);
void * ggml_backend_reg_get_proc_address(ggml_backend_reg_t reg, const char * name);
const char * ggml_backend_dev_name(ggml_backend_dev_t dev);
const char * ggml_backend_dev_description(ggml_backend_dev_t dev);
void ggml_backend_dev_memory(ggml_backend_dev_t dev, size_t * free, size_t * total);
enum ggml_backend_dev_type ggml_backend_dev_type(ggml_backend_dev_t dev);
void ggml_backend_dev_get_props(ggml_backend_dev_t dev, struct ggml_backend_dev_props * props);
ggml_backend_t ggml_backend_dev_init(ggml_backend_dev_t dev, const char * params);
ggml_backend_buffer_type_t ggml_backend_dev_buffer_type(ggml_backend_dev_t dev);
ggml_backend_buffer_type_t ggml_backend_dev_host_buffer_type(ggml_backend_dev_t dev);
bool ggml_backend_dev_supports_op(ggml_backend_dev_t dev, const struct ggml_tensor * op);
bool ggml_backend_dev_supports_buft(ggml_backend_dev_t dev, ggml_backend_buffer_type_t buft);

const char * ggml_backend_buft_name(ggml_backend_buffer_type_t buft);
ggml_backend_buffer_t ggml_backend_buft_alloc_buffer(ggml_backend_buffer_type_t buft, size_t size);
size_t ggml_backend_buft_get_alignment(ggml_backend_buffer_type_t buft);
size_t ggml_backend_buft_get_alloc_size(ggml_backend_buffer_type_t buft, const struct ggml_tensor * tensor);
bool ggml_backend_buft_is_host(ggml_backend_buffer_type_t buft);

/* what libggml-base provides to backends; exported here so the backend .so (which looks it up with
 * dlsym(RTLD_DEFAULT)) behaves exactly as it would inside a real ggml host */
ggml_backend_buffer_t ggml_backend_buffer_init(ggml_backend_buffer_type_t buft, struct ggml_backend_buffer_i iface, void * context, size_t size);
void ggml_backend_buffer_free(ggml_backend_buffer_t buffer);
void * ggml_backend_buffer_get_base(ggml_backend_buffer_t buffer);
size_t ggml_backend_buffer_get_size(ggml_backend_buffer_t buffer);
void ggml_backend_buffer_clear(ggml_backend_buffer_t buffer, uint8_t value);
void ggml_backend_buffer_set_usage(ggml_backend_buffer_t buffer, enum ggml_backend_buffer_usage usage);
bool ggml_backend_buffer_is_host(ggml_backend_buffer_t buffer);

const char * ggml_backend_name(ggml_backend_t backend);
void ggml_backend_free(ggml_backend_t backend);
void ggml_backend_tensor_set(struct ggml_tensor * tensor, const void * data, size_t offset, size_t size);
void ggml_backend_tensor_get(const struct ggml_tensor * tensor, void * data, size_t offset, size_t size);
void ggml_backend_tensor_memset(struct ggml_tensor * tensor, uint8_t value, size_t offset, size_t size);
void ggml_backend_tensor_set_async(ggml_backend_t backend, struct ggml_tensor * tensor, const void * data, size_t offset, size_t size);
void ggml_backend_tensor_get_async(ggml_backend_t backend, const struct ggml_tensor * tensor, void * data, size_t offset, size_t size);
void ggml_backend_synchronize(ggml_backend_t backend);
enum ggml_status ggml_backend_graph_compute(ggml_backend_t backend, struct ggml_cgraph * cgraph);
enum ggml_status ggml_backend_graph_compute_async(ggml_backend_t backend, struct ggml_cgraph * cgraph);
bool ggml_backend_supports_op(ggml_backend_t backend, const struct ggml_tensor * op);
/* copies between backends and events (ggml-backend.h): what ggml_backend_sched issues between the devices of a layer split */
void ggml_backend_tensor_copy(struct ggml_tensor * src, struct ggml_tensor * dst);
void ggml_backend_tensor_copy_async(ggml_backend_t backend_src, ggml_backend_t backend_dst, struct ggml_tensor * src, struct ggml_tensor * dst);
ggml_backend_event_t ggml_backend_event_new(ggml_backend_dev_t device);
void ggml_backend_event_free(ggml_backend_event_t event);
void ggml_backend_event_record(ggml_backend_event_t event, ggml_backend_t backend);
void ggml_backend_event_synchronize(ggml_backend_event_t event);
void ggml_backend_event_wait(ggml_backend_t backend, ggml_backend_event_t event);

/* plain host-memory buffer type (the role ggml_backend_cpu_buffer_type() plays upstream) */
ggml_backend_buffer_type_t ggml_backend_cpu_buffer_type(void);

/* ggml-alloc.h mirror */
/* allocates every tensor of ctx that has no data yet (views follow their view_src) in ONE buffer of buft */
ggml_backend_buffer_t ggml_backend_alloc_ctx_tensors_from_buft(struct ggml_context * ctx, ggml_backend_buffer_type_t buft);
typedef struct ggml_gallocr * ggml_gallocr_t;
ggml_gallocr_t ggml_gallocr_new(ggml_backend_buffer_type_t buft);
void ggml_gallocr_free(ggml_gallocr_t galloc);
/* sizes the compute buffer for `graph` (liveness-based reuse) without touching tensors */
bool ggml_gallocr_reserve(ggml_gallocr_t galloc, struct ggml_cgraph * graph);
/* places graph inputs and node outputs; re-uses the reserved buffer when it is large enough */
bool ggml_gallocr_alloc_graph(ggml_gallocr_t galloc, struct ggml_cgraph * graph);
size_t ggml_gallocr_get_buffer_size(ggml_gallocr_t galloc, int buffer_id);

#ifdef __cplusplus
}
#endif
#endif
