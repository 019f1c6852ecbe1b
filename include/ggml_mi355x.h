/*
 * ggml_mi355x.h — the C-ABI of libggml-mi355x.so, the MI355X (gfx950) ggml backend.
 *
 * This is the drop-in boundary.  The reference binds a device backend in exactly one way:
 *
 *   ggml_backend_load_all()                          /root/reference/llama-box/engine_param.hpp:542-545
 *     -> ggml_backend_load_best("hip", silent, dir)  /root/reference/llama-box/patches/llama.cpp/pure_cpu.patch:80-92
 *     -> $GGML_BACKEND_PATH                          /root/reference/llama-box/patches/llama.cpp/pure_cpu.patch:101-102
 *   each of which dlopen()s the library and resolves the two symbols below (what ggml's
 *   GGML_BACKEND_DL_IMPL / GGML_BACKEND_DL_SCORE_IMPL macros expand to in every stock backend).
 *
 * Everything else crosses the boundary through the vtables reachable from the returned registration
 * object (struct layouts: include/ggml_abi.h).  No torch / C++ types appear in any signature.
 */
#ifndef GGML_MI355X_H
#define GGML_MI355X_H
#include "ggml_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

#define GGML_MI355X_NAME "MI355X"
#define GGML_MI355X_MAX_DEVICES 16

/* Replaces: ggml_backend_init() exported by libggml-hip.so (ggml-backend-reg.cpp dl entry; call site
 * pure_cpu.patch:80-102).  Returns the backend registration (api_version == GGML_BACKEND_API_VERSION), or NULL
 * when no gfx950 device is visible. */
ggml_backend_reg_t ggml_backend_init(void);

/* Replaces: ggml_backend_score() (optional DL export used by ggml_backend_load_best to rank variants).
 * 0 = unusable on this system (no HIP device of arch gfx950), otherwise > 0. */
int ggml_backend_score(void);

/* Static-registration spelling of the same object (what ggml_backend_cuda_reg() is to the CUDA backend;
 * pure_cpu.patch:18-62 shows the static registry constructor that would call it). */
ggml_backend_reg_t ggml_backend_mi355x_reg(void);

/*
 * Names served by reg->iface.get_proc_address (pattern: llama-box/patches/llama.cpp/dynamic_link.patch:5-20,
 * llama-box/engine_param.hpp:99-101, llama-box/rpcserver.hpp:402-403):
 *
 *   "ggml_backend_get_features"        ggml_backend_get_features_t
 *   "ggml_backend_split_buffer_type"   ggml_backend_split_buffer_type_t   (row-split hook of -sm row: (main_device, tensor_split[]) ->
 *                                      buffer type whose matrices are cut by rows over the devices; llama-box/engine_param.hpp
 *                                      :821-842, :902-916; csrc/split.cpp)
 *   "ggml_backend_mi355x_split_rows"   ggml_backend_mi355x_split_rows_t   (the row plan of that buffer type, host arithmetic)
 *   "ggml_backend_mi355x_tp_init"      ggml_backend_mi355x_tp_init_t
 *   "ggml_backend_mi355x_tp_rowpar_buffer_type"   ggml_backend_mi355x_tp_rowpar_buffer_type_t
 *   "ggml_backend_mi355x_set_option"   ggml_backend_mi355x_set_option_t
 *   "ggml_backend_mi355x_get_stat"     ggml_backend_mi355x_get_stat_t
 *   "ggml_backend_mi355x_timing_report" ggml_backend_mi355x_timing_report_t
 *   "ggml_backend_mi355x_tp_get_unique_id" ggml_backend_mi355x_tp_get_unique_id_t
 *   "ggml_backend_mi355x_tp_p2p_export"   ggml_backend_mi355x_tp_p2p_export_t
 *   "ggml_backend_mi355x_tp_p2p_attach"   ggml_backend_mi355x_tp_p2p_attach_t
 *   "ggml_backend_mi355x_tp_all_reduce"   ggml_backend_mi355x_tp_all_reduce_t
 *   "ggml_backend_mi355x_graph_key_probe" ggml_backend_mi355x_graph_key_probe_t   (tests: is graph b the graph a captured hipGraph of a stands for?)
 *   "ggml_backend_mi355x_decode_copy_read" ggml_backend_mi355x_decode_copy_read_t (tests: the plane-layout decode copy of a weight matrix, csrc/repack.hip)
 */

/* Replaces: ggml_backend_cuda_split_buffer_type(int main_device, const float * tensor_split) of the stock GPU backends, as reached
 * through get_proc_address (upstream ggml-backend.h: ggml_backend_split_buffer_type_t).  tensor_split holds one proportion per
 * device of this registration (all zero = even).  Only the main device's backend supports_buft() the result. */
/* (typedef ggml_backend_split_buffer_type_t: include/ggml_abi.h, as in upstream's ggml-backend.h) */
/* row0[0 .. n_dev]: device d owns rows [row0[d], row0[d + 1]) of an nrows-row matrix (64-row granule). */
typedef void (*ggml_backend_mi355x_split_rows_t)(int64_t nrows, const float * tensor_split, int n_dev, int64_t * row0);

/* One-process-per-GPU tensor parallelism over RCCL/xGMI.  `unique_id` is the 128-byte ncclUniqueId produced by
 * rank 0 (ncclGetUniqueId) and distributed by the launcher (bench.py uses torch.distributed for that).  After a
 * successful call, MUL_MAT nodes whose src0 lives in the "rowpar" buffer type are followed by an in-stream
 * all-reduce(sum) of their f32 result across ranks.  Returns 0 on success. */
typedef int (*ggml_backend_mi355x_tp_init_t)(ggml_backend_t backend, int rank, int world_size, const void * unique_id, size_t unique_id_size);
/* One-shot peer-to-peer all-reduce for the latency-bound sums of tensor-parallel decode (csrc/tp_p2p.hip; no reference counterpart: llama.cpp
 * has no collective — SURVEY.md §2.5, §8e).  Two steps, both on every rank: export() allocates this rank's mailbox and returns its
 * hipIpcMemHandle_t (64 bytes) in handle_out; the launcher's control plane gathers the `world_size` handles in rank order; attach() maps the
 * peers' mailboxes.  From then on row-parallel sums of up to 256 KiB run as ONE launch that writes the partial into every peer's memory over
 * xGMI and adds the world_size mailboxes in rank order (bit-identical on all ranks); longer ones use RCCL when tp_init() has also been called,
 * else they go through the mailboxes in chunks.  Works without RCCL, and with several ranks on one GPU.  Return 0 on success. */
typedef int (*ggml_backend_mi355x_tp_p2p_export_t)(ggml_backend_t backend, int rank, int world_size, void * handle_out, size_t handle_size);
typedef int (*ggml_backend_mi355x_tp_p2p_attach_t)(ggml_backend_t backend, const void * handles, size_t handles_size);
/* The collective itself, for a launcher that wants to check its group before trusting it: in-place sum over the ranks of n f32 values at
 * device_ptr, enqueued in the backend's stream (ggml_backend_synchronize to wait) — exactly what graph_compute issues behind a row-parallel
 * mat-mul.  0 on success. */
typedef int (*ggml_backend_mi355x_tp_all_reduce_t)(ggml_backend_t backend, float * device_ptr, size_t n);
typedef int (*ggml_backend_mi355x_tp_get_unique_id_t)(void * unique_id_out, size_t unique_id_size);
typedef ggml_backend_buffer_type_t (*ggml_backend_mi355x_tp_rowpar_buffer_type_t)(int device);

/* The graph key (csrc/graph.cpp: walk_key) of `a`, built, then compared IN PLACE with graph `b` exactly as graph_compute compares the graph of a
 * llama_decode with the one it replayed last: 1 = same captured hipGraph serves both, 0 = not.  *n_words (may be NULL) = 8-byte words of the key.
 * Host arithmetic: works without a device. */
typedef int (*ggml_backend_mi355x_graph_key_probe_t)(const struct ggml_cgraph * a, const struct ggml_cgraph * b, int64_t * n_words);

/* The decode copy (round 6): K-quant weight matrices of a usage-WEIGHTS buffer are kept a second time in a line-aligned plane layout that the batch-1 mat-vec kernels
 * stream with non-temporal loads (csrc/repack.hip, csrc/mmvq_types.h; option "decode_copy", GGML_MI355X_DECODE_COPY).  The tensor the host uploaded stays as it is:
 * get_tensor returns its bytes, the batch kernels and GET_ROWS read it.  This entry returns the copy's bytes (tests compare them with the documented permutation):
 * the number of bytes written to `out` (or needed, when out is NULL / too small), 0 when the tensor has no copy, -1 on error. */
typedef int64_t (*ggml_backend_mi355x_decode_copy_read_t)(ggml_backend_t backend, const struct ggml_tensor * t, void * out, size_t size);

/* Runtime options (string key/value); 0 = accepted, -1 = unknown key, -2 = refused (e.g. "tp_p2p" = 0 in a group whose only transport the mailboxes are).  Keys (INTEGRATION.md 4b lists the defaults and the
 * environment variables that set the same things): "graphs", "fusion", "prologue", "qkv", "mm_merge", "mmq_i8", "mmq_bn",
 * "mmq_skinny", "skinny_rope", "softmax_mm", "attn_nf", "mmq_min_cols", "mmvq_max_cols", "fa_splits", "fa_wo", "fa_self_merge", "small_uploads", "small_downloads",
 * "exec_update", "shadow_capture", "decode_copy" (the K-quants' plane copy and the Q8_0 panel copy), "decode_copy_headroom_gib", "q80_min_cols", "timing", "tp_p2p", "tp_p2p_reset" (forget an all-reduce time-out: every rank, all idle), "clear_failure" (forget a remembered HIP failure of a status-less entry point). */
typedef int (*ggml_backend_mi355x_set_option_t)(ggml_backend_t backend, const char * key, const char * value);
/* Counters for tests/bench: "graph_launches", "graph_captures", "graph_early_captures", "graph_shadow_captures", "graph_capture_walk_ns", "graph_exec_update_ns", "graph_shadow_eager_ns", "graph_exec_updates", "eager_graphs", "kernel_launches", "fused_nodes",
 * "allreduces", "p2p_allreduces", "p2p_timeouts", "graph_launch_host_ns", "graph_key_host_ns", "graph_compute_host_ns", "graph_key_fast_hits", "graph_key_collisions",
 * "kernel_downloads", "graph_exec_update_failures", "graph_evictions", "graph_cache_size", "step_heads", "decode_copy_tensors", "decode_copy_bytes", "decode_copy_launches", "kv_image_nodes", "kv_native_nodes", "skinny_launches", "wide_launches", "tiled_launches", "rope_epilogues", "fa_list_launches"; in-process tensor parallel (-sm row, csrc/tp_inproc.cpp):
 * "ip_devices", "ip_graphs", "ip_declined", "ip_plans", "ip_input_copies", "ip_output_copies", "ip_kv_gathers", "ip_kv_scatters", "ip_worker_kernel_launches",
 * "ip_worker_graph_launches", "ip_worker_p2p_timeouts". */
typedef int64_t (*ggml_backend_mi355x_get_stat_t)(ggml_backend_t backend, const char * key);
/* Timing helper for bench.py (option "timing"="1": graphs off, every kernel class bracketed by hipEvents on the
 * backend's own stream — torch.cuda.Event cannot see that stream).  Writes lines "class count total_ms bytes". */
typedef int (*ggml_backend_mi355x_timing_report_t)(ggml_backend_t backend, char * buf, size_t size, int reset);

#ifdef __cplusplus
}
#endif
#endif /* GGML_MI355X_H */
