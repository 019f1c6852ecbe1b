/*
 * llama_lite.h — host-side driver that plays the role llama.cpp's llama_decode() plays for llama-box.
 *
 * llama-box's engine loop calls llama_decode(ctx, batch) (llama-box/httpserver.hpp:3591, :3615) and reads
 * logits with llama_get_logits_ith (llama-box/httpserver.hpp:442); llama.cpp turns that into a ggml graph
 * per micro-batch and hands it to the backend.  llama.cpp is an un-vendored
 * submodule, so this file provides our own driver with the same call shape — batch of (token, pos, seq_id,
 * want-logits) in, return code 0 / 1 (no KV slot) / -1 (bad batch) / -2 (compute failed) out
 * (llama-box/httpserver.hpp:3541-3545) — building the same op sequence llm_build_llama / llm_build_qwen2
 * produce, so the backend sees the graphs it would see under the real engine.
 * It is the harness for parity tests and bench.py; the product is the backend library.
 */
#ifndef LLAMA_LITE_H
#define LLAMA_LITE_H
#include "ggml_lite.h"

#ifdef __cplusplus
extern "C" {
#endif

enum llm_ftype {
    LLM_FTYPE_Q8_0 = 0,    /* everything Q8_0 (TinyLlama config 1) */
    LLM_FTYPE_Q4_K_M = 1,  /* Q4_K base, Q6_K for attn_v/ffn_down "more bits" layers + output (config 2-4) */
    LLM_FTYPE_Q5_K_M = 2,  /* Q5_K base, Q6_K "more bits" + output (config 5: Q6_K + Q5_K mix) */
    LLM_FTYPE_Q6_K = 3,    /* everything Q6_K */
    LLM_FTYPE_F16 = 4,     /* everything F16 */
    LLM_FTYPE_MIXED = 5,   /* test recipe: cycles Q4_K/Q5_K/Q6_K/Q8_0 over tensors so one tiny model hits every kernel */
};

struct llm_hparams {
    char arch[32]; /* "llama" | "qwen2" */
    int32_t n_layer, n_embd, n_head, n_head_kv, n_embd_head, n_ff, n_vocab, n_ctx_train;
    float rope_freq_base, rms_eps;
    int32_t rope_type; /* 0 = normal pairs (llama), GGML_ROPE_TYPE_NEOX = 2 (qwen2) */
    int32_t qkv_bias;
    int32_t ftype;
    int32_t attn_v_q5k_70b; /* 70B recipe: attn_v Q4_K -> Q5_K */
    int32_t peaked;         /* 0..8: that many eighths of the blocks of output.weight row r repeat the blocks of token_embd row
                               (r * 7919 + 13) mod n_vocab — the logits get a trained model's shape (one row far ahead of the rest, the
                               rest still decided by the layers) instead of the near-flat ones of independent random weights.  When the two
                               tensors differ in format (Q4_K_M / Q5_K_M: Q4_K / Q5_K embeddings, Q6_K output) the tied blocks are token_embd's
                               VALUES re-encoded as Q6_K */
    float branch_gain;      /* gain of attn_output / ffn_down (the residual branches); 0 = the default 0.25.  "-damped" presets: 0.08 / sqrt(2 n_layer) */
};

/* presets: "tinyllama-1.1b-q8_0", "tinyllama-1.1b-q8_0-peaked", "llama3-8b-q4_k_m", "llama3-70b-q4_k_m", "qwen2-7b-q5_k_m", "test-llama", "test-qwen2";
   any of them + "-damped" = peaked 3 + branch_gain 0.08 / sqrt(2 n_layer) */
int llm_preset(const char * name, struct llm_hparams * hp);

struct llm_model;
/* tp_size>1: this process holds shard tp_rank of a tensor-parallel model (column-parallel wq/wk/wv/gate/up,
 * row-parallel wo/down).  rowpar_buft (may be NULL when tp_size==1) is the backend's reducing buffer type. */
struct llm_model * llm_model_synth(const struct llm_hparams * hp, uint64_t seed, ggml_backend_buffer_type_t buft, int tp_rank, int tp_size,
                                   ggml_backend_buffer_type_t rowpar_buft);
/* -sm row: mat-mul weights in `split_buft` (the registry's "ggml_backend_split_buffer_type" result), everything else in `buft` */
struct llm_model * llm_model_synth_split(const struct llm_hparams * hp, uint64_t seed, ggml_backend_buffer_type_t buft, ggml_backend_buffer_type_t split_buft);
/* -sm layer (llama.cpp's default with several devices; llama-box/engine_param.hpp:900-916): device d's buffer type holds a contiguous range of layers */
struct llm_model * llm_model_synth_layer_split(const struct llm_hparams * hp, uint64_t seed, const ggml_backend_buffer_type_t * bufts, int n_dev);
int llm_synth_gguf(const struct llm_hparams * hp, uint64_t seed, const char * path);
struct llm_model * llm_model_load(const char * path, ggml_backend_buffer_type_t buft);
void llm_model_free(struct llm_model * m);
const struct llm_hparams * llm_model_hparams(const struct llm_model * m);
/* bytes of weights one decoded token streams (all matmul weights + norms + one token_embd row): SURVEY.md §8d */
uint64_t llm_model_stream_bytes(const struct llm_model * m);
uint64_t llm_model_total_bytes(const struct llm_model * m);
struct ggml_tensor * llm_model_tensor(struct llm_model * m, const char * name);

typedef enum ggml_status (*llm_compute_fn)(struct ggml_cgraph * graph, int n_threads);

#define LLM_KV_TYPE_F32 (-1)
struct llm_context_params {
    int32_t n_ctx;      /* KV cells */
    int32_t n_ubatch;   /* micro-batch (graph) size; batches are sliced into views of this many tokens */
    int32_t flash_attn; /* 1: FLASH_ATTN_EXT, 0: MUL_MAT/SOFT_MAX/MUL_MAT with transposed V cache */
    int32_t n_threads;  /* for the external compute function */
    int32_t graph_reuse; /* keep the built graph while the topology key is unchanged */
    int32_t type_k, type_v; /* ggml type of the K / V cache rows (llama-box -ctk / -ctv, engine_param.hpp:51-54): 0 or GGML_TYPE_F16 = f16; GGML_TYPE_Q8_0,
                               Q4_0, Q4_1, IQ4_NL, Q5_0, Q5_1, BF16; LLM_KV_TYPE_F32 = f32 (a V cache other than f16 needs flash_attn, as in llama.cpp) */
};
struct llm_context;
/* exactly one of backend / compute must be set: backend -> ggml_backend_graph_compute, else the callback
 * (tests pass the CPU oracle's oracle_graph_compute here; the product never does) */
struct llm_context * llm_context_new(struct llm_model * m, ggml_backend_t backend, llm_compute_fn compute, const struct llm_context_params * p);
/* one backend per device of a layer-split model: the graph is cut at the device boundaries and driven as ggml_backend_sched drives its splits
 * (blocking input copies, cpy_tensor_async of the residual stream, event_record / event_wait / event_synchronize per input-copy slot) */
struct llm_context * llm_context_new_layer_split(struct llm_model * m, const ggml_backend_t * backends, int n_dev, const struct llm_context_params * p);
/* [cpy_tensor_async calls between devices, blocking input copies between devices, events recorded, events waited for] since the context was made */
void llm_layer_split_stats(const struct llm_context * c, int64_t out[4]);
void llm_context_free(struct llm_context * c);
/* return codes as llama_decode: 0 ok, 1 no KV slot, -1 invalid batch, -2 compute/alloc failure */
int llm_decode(struct llm_context * c, int n_tokens, const int32_t * tokens, const int32_t * pos, const int32_t * seq_id, const int8_t * want_logits);
/* n_steps consecutive decode steps of n_par sequences (one token each, sequence ids 0..n_par-1, positions pos0+step): the
   engine loop's hot part without a scripting language between the steps — tokens is [n_steps][n_par].  Logits of every step
   are fetched to the host exactly as llm_decode does; after the call they hold the last step's.  llm_last_timings then
   reports the sums over all steps.  Returns 0 or the first failing llm_decode's code. */
int llm_decode_steps(struct llm_context * c, int n_steps, int n_par, const int32_t * tokens, int pos0);
// one step = llama_decode over n_par x (1 + n_draft) tokens (a slot's sampled token + its drafts, logits at every position), then the rejected
// drafts leave the cache (llama_memory_seq_rm): llama-box's speculative-decoding batch shape (httpserver.hpp:4042-4069, :4696-4768)
int llm_verify_steps(struct llm_context * c, int n_steps, int n_par, int n_draft, const int32_t * tokens, int pos0);
int llm_n_outputs(const struct llm_context * c);
float * llm_get_logits(struct llm_context * c);            /* [n_outputs][n_vocab], host memory */
/* llama_get_logits_ith semantics: i >= 0 = batch position (must have requested logits), i < 0 = output rows from the end; NULL otherwise */
float * llm_get_logits_ith(struct llm_context * c, int i);
/* host-side consumers of the logits: common_sampler_sample2 with a greedy chain, get_token_probabilities */
int32_t llm_sample_greedy(struct llm_context * c, int idx);
int llm_token_probabilities(struct llm_context * c, int idx, int top_n, int32_t * ids, float * probs);
void llm_kv_clear(struct llm_context * c);
int llm_kv_seq_rm(struct llm_context * c, int seq_id, int p0, int p1);
/* llama_memory_seq_add + the K-shift graph it triggers (context shift, llama-box httpserver.hpp:3453-3537): positions
   [p0, p1) of seq_id move by delta and the cached K rows are re-rotated by it (quantised caches: through an f32 copy) */
int llm_kv_seq_add(struct llm_context * c, int seq_id, int p0, int p1, int delta);
/* the graph of the last micro-batch (for inspection / per-node comparison in tests) */
struct ggml_cgraph * llm_last_graph(struct llm_context * c);
struct ggml_tensor * llm_context_cache_tensor(struct llm_context * c, int il, int which); /* K (0) / V (1) cache tensor of layer il */
/* host-side time split of the last llm_decode, microseconds: [build+alloc, set inputs, compute+sync, get logits] */
void llm_last_timings(const struct llm_context * c, double out[4]);

#ifdef __cplusplus
}
#endif
#endif
