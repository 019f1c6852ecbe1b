/*
 * oracle.h — CPU restatement of the ggml-cpu algorithms on the llama-box hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under llama_box_amd/ (the product) may include, link or call
 * this.  Allowed callers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
 *
 * PARITY UNPINNED: the reference snapshot (/root/reference) holds no tests, golden vectors or
 * known-answer files for this path and the library that holds the arithmetic
 * (github.com/ggml-org/llama.cpp, un-vendored submodule, commit unrecorded, ~Aug 2025 by the patch
 * pre-image blobs ggml-cpu/ops.cpp=854f1c2b4, ggml-cpu/ggml-cpu.cpp=8dacd3671) is absent and
 * cannot be built here.  This oracle restates the published *generic* (non-SIMD) ggml-cpu
 * algorithms and is pinned only by (1) hand-computed known-answer blocks,
 * (2) an independent NumPy restatement (tests/golden/make_golden.py) and (3) the three in-tree
 * hunks that do specify math: the soft_max zero-sum guard (llama-box/patches/llama.cpp/
 * ggml-cpu.patch:5-15), rope mode flags (mrope.patch:5-26) and scale = s*x+b (ggml-cuda.patch:8-22).
 */
#ifndef ORACLE_H
#define ORACLE_H
#include "../include/ggml_abi.h"
#ifdef __cplusplus
extern "C" {
#endif

float oracle_fp16_to_fp32(ggml_fp16_t h);
ggml_fp16_t oracle_fp32_to_fp16(float f);

void oracle_dequantize_row(enum ggml_type type, const void * x, float * y, int64_t k);
void oracle_quantize_row_q8_0(const float * x, block_q8_0 * y, int64_t k);
/* from_float of the types a KV-cache row may be stored in (f32, f16, bf16, q8_0, q8_1, q4_0, q4_1, q5_0, q5_1, iq4_nl): 1 done, 0 no such type here */
int oracle_quantize_row(enum ggml_type type, const float * x, void * y, int64_t k);
float oracle_kq_dot(enum ggml_type kt, int64_t n, const void * krow, const float * q); /* a K.q logit of a block-format cache row, as FLASH_ATTN_EXT forms it */
uint16_t oracle_fp32_to_bf16(float f);
float oracle_bf16_to_fp32(uint16_t h);
void oracle_quantize_row_q8_K(const float * x, block_q8_K * y, int64_t k);
float oracle_vec_dot_q8_0_q8_0(int64_t n, const block_q8_0 * x, const block_q8_0 * y);
float oracle_vec_dot_q4_K_q8_K(int64_t n, const block_q4_K * x, const block_q8_K * y);
float oracle_vec_dot_q5_K_q8_K(int64_t n, const block_q5_K * x, const block_q8_K * y);
float oracle_vec_dot_q6_K_q8_K(int64_t n, const block_q6_K * x, const block_q8_K * y);

/* executes one node / a whole graph whose tensors live in host memory (tensor->data is a host ptr) */
enum ggml_status oracle_compute_node(struct ggml_tensor * node, int n_threads);
enum ggml_status oracle_graph_compute(struct ggml_cgraph * graph, int n_threads);
/* 1 if oracle_compute_node implements this node */
int oracle_supports_op(const struct ggml_tensor * node);
int oracle_max_threads(void);
/* 0 = generic ggml-cpu; 1 / 2 / 3 = one-ulp sensitivity probes for the tests (reversed block order, f32 RMS_NORM sum,
 * expf one ulp up / down); 4 = block-format K rows against the unquantised query; 5 / 6 = FAULT INJECTION for the tests of the gates themselves
 * (the logits x 1.004 / every layer's FFN branch x 1.01) — see ggml_cpu_ref.c */
void oracle_set_variant(int v);
/* bench.py cpu_baseline ONLY: route the quantised block dots of MUL_MAT through the AVX2 restatement of ggml-cpu's x86 kernels
 * (same integers, another f32 accumulation order).  Returns the state actually set (0 when built without AVX2+FMA). */
int oracle_set_fast(int on);
float oracle_fast_vec_dot(enum ggml_type type, int64_t n, const void * x, const void * y);

#ifdef __cplusplus
}
#endif
#endif
