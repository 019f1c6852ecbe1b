// kernels.h — host-callable launchers of the hand-written gfx950 kernels (internal; not part of the C-ABI).
#pragma once
#include "common.h"

namespace mi355x {

// plain-data view of a ggml_tensor that can be passed to kernels by value
struct tdesc {
    char * data;
    int64_t ne[4];
    int64_t nb[4];  // bytes
    int type;
};
inline tdesc TD(const ggml_tensor * t) {
    tdesc d;
    d.data = (char *) t->data;
    for (int i = 0; i < 4; ++i) {
        d.ne[i] = t->ne[i];
        d.nb[i] = (int64_t) t->nb[i];
    }
    d.type = (int) t->type;
    return d;
}

// ---- activation quantisation (quantize.hip)
// src: f32 rows of K elements (dim0 contiguous), any strides on dims 1..3; rows are enumerated i1 fastest.
// kind: GGML_TYPE_Q8_K -> q8k_dev[rows][K/256]; GGML_TYPE_Q8_0 -> q80_dev[rows][K/32]
size_t quantized_act_bytes(int kind, int64_t K, int64_t rows);
void launch_quantize_act(hipStream_t s, int kind, const tdesc & src, void * dst);
// Q8_0 activations of up to 32 columns in PANEL order (mmq_q80.hip: k_mmq_q80_skinny): per block of 32 values [K half][column 0 .. 31][16 quants] + [column] f32 scales, 1152 B
#define MI_ACT_Q80_PANEL 1008
void launch_quantize_q80_panel(hipStream_t s, const tdesc & src, void * dst);
// producers fused with that quantisation (ops.hip; same arithmetic as the unfused kernels): RMS_NORM(x) * w, SwiGLU
bool rms_norm_q80_panel_ok(const tdesc & src, const float * w);
void launch_rms_norm_mul_q80_panel(hipStream_t s, const tdesc & src, float eps, const float * w, void * q80_panel);
bool swiglu_q80_panel_ok(const tdesc & a, const tdesc * b, int64_t nc, int swapped);
void launch_swiglu_q80_panel(hipStream_t s, const tdesc & a, const tdesc * b, int64_t nc, int swapped, void * q80_panel);

// ---- bandwidth-bound quantised mat-vec, 1..8 activation columns (mmvq.hip)
struct mmvq_args {
    const uint8_t * W;     // [N] rows, row stride w_nb1 bytes
    const uint8_t * W2;    // optional second matrix (fused gate/up): out = silu(W·x) * (W2·x)
    int64_t w_nb1;
    int type;              // ggml_type of W
    int K, N;
    int ncols;             // 1..8
    const void * act;      // quantised activations [ncols][K/blk]
    float * dst;           // dst[col * dst_stride + row]
    int64_t dst_stride;    // elements
    const float * add;     // optional: dst += add[col * add_stride + row]  (bias: add_stride = 0; residual: = dst_stride)
    int64_t add_stride;
    const float * add2;    // optional second addend (bias AND residual)
    int64_t add2_stride;
    // fused activation prologue (ncols == 1, K-quants): x != null -> the kernel quantises f32 x itself (act unused);
    // norm_w != null additionally applies RMS_NORM(x, eps) * norm_w first
    const float * x;
    const float * norm_w;
    float eps;
    // fa_part != null (instead of x): the row is the merge of `fa_splits` attention partial records per head ([head][split],
    // FA_REC floats apart: 128 values, max, sum — fattn.hip's 8-wave decode kernel); K = heads * 128.  x_out: optional f32 copy.
    const float * fa_part;
    int fa_splits;
    float * x_out;
    float * norm_out;  // norm_w != null: where RMS_NORM(x) * norm_w itself belongs (the graph's MUL node); written by workgroup 0 so that
                       // the fusion never leaves a tensor of the graph unwritten (readers in another split, or the host, may exist)
    int balance_tail;  // set by the launcher: spread the last, partial pass of rows evenly over the workgroups
    // sum of squares of the result row, handed from the mat-vec that WRITES a residual stream to the RMS_NORM prologue that READS it (round 4):
    //   ss_out != null (single column, f32 result, no SwiGLU): workgroup b writes the sum of (double) (v * v) over the rows it produced to ss_out[b];
    //     launch_mmvq_ss_count() tells how many workgroups that is
    //   ss_in != null (norm prologue): sum of squares of x = the ss_n partial sums in ss_in, added in a fixed order — the prologue then needs
    //     neither its pass over x nor the workgroup barrier of its own reduction (knock-out: gate/up 16.0 -> 14.8 us)
    double * ss_out;
    const double * ss_in;
    int ss_n;
    // the decode copies of W / W2 (plane layout, repack.hip) or nullptr: the streaming forms read these instead, with non-temporal loads (launch_mmvq decides)
    const uint8_t * Wp;
    const uint8_t * W2p;
};
int launch_mmvq_ss_count(const mmvq_args & a);  // workgroups (= partial sums) a launch with ss_out will write; 0 if this launch cannot publish them
#define FA_REC 132  // floats per attention partial record in the fat-split form (128 values + max + sum, padded to 16 bytes)
void launch_mmvq(hipStream_t s, const mmvq_args & a, int rows_per_wave);
// bench timing pass: when armed, the next streaming mat-vec launch records the kernel's own begin / end timestamps into
// (e0, e1) through hipExtLaunchKernelGGL instead of being bracketed by host-side hipEventRecord calls
struct launch_probe {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool armed = false, used = false;
};
extern thread_local launch_probe g_launch_probe;
// launch through the armed probe (once) or plainly: every launcher of a kernel class bench.py may quote in `roofline` goes through this
#define MI_LAUNCH_PROBED(kernel, grid, block, lds, stream, ...)                                                                  \
    do {                                                                                                                         \
        if (g_launch_probe.armed && !g_launch_probe.used) {                                                                      \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, g_launch_probe.e0, g_launch_probe.e1, 0, __VA_ARGS__);       \
            g_launch_probe.used = true;                                                                                          \
        } else {                                                                                                                 \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                                   \
        }                                                                                                                        \
    } while (0)

// ---- fused Q/K/V projection for one token (qkv.hip): up to three K-quant mat-vecs that share their input, with the
// activation prologue of mmvq (f32 or RMS_NORM*w), optional bias, rotary embedding and the KV-cache store in the epilogue
struct qkv_seg {
    const uint8_t * W;
    int64_t w_nb1;
    int N;
    uint8_t alt;          // 0 = first weight format of the launch, 1 = second      (alt / rope / store share one dword: see row_stride)
    uint8_t rope;         // rotate pairs of this segment
    uint8_t store;        // 0: f32 at out; 1: f16 / 2: blocks of type `kvt` / 4: bf16 into row `slot` of a cache tensor (out + slot*row_stride);
                          // 3: f16 element j at out + 2 idx[j] (the transposed V cache of the non-flash path: one row index per element)
    uint8_t kvt;          // store == 2: the cache's block format (GGML_TYPE_Q8_0, Q4_0, Q4_1, Q5_0, Q5_1, IQ4_NL)
    const float * bias;   // optional [N]
    char * out;
    int64_t row_stride;   // store == 3: the address of the int64 index vector instead (no extra field: the struct travels in SGPRs,
                          // and the two-format launch is at its register limit — one more pointer cost 12 bytes of spills)
};
struct qkv_args {
    qkv_seg seg[3];
    int nseg, K;
    const float * x;
    const float * norm_w;  // optional
    float eps;
    float * norm_out;      // as mmvq_args::norm_out
    int head_dim, neox;
    const int32_t * pos;
    const float * freq_factors;
    float theta_scale, freq_scale, ext_factor, attn_factor, corr0, corr1;
    const int64_t * slot;
    const float * rope_tab;  // optional: [head_dim / 2][cos, sin] for this token (launch_rope_table) instead of computing them in every workgroup's prologue
    int wg_a;  // set by launch_qkv: workgroups [0, wg_a) serve the alt == 0 segments, the rest the alt == 1 segments
    int wv_a;  // > 0 (two formats, f16 / bf16 / f32 stores — round 6): the split is by WAVES instead: waves [0, wv_a) of the launch (workgroup x waves + wave) serve the
               // alt == 0 segments, the rest the alt == 1 segments, so that 256 workgroups x 12 waves hold Llama-3-8B's 2560 + 512 row pairs with ONE unit per wave
    int planes;  // every segment's W is the matrix's decode copy (plane layout, repack.hip): the launch runs the plane forms of both formats
    const double * ss_in;  // as mmvq_args::ss_in / ss_n (norm prologue)
    int ss_n;
};
// ---- one-shot peer-to-peer all-reduce (tp_p2p.hip)
#define P2P_SLOT_FLOATS 65536  // values per (parity, source rank) mailbox: 256 KiB of f32 = 8 columns of Llama-3-70B's residual stream
#define P2P_MAX_BLOCKS 32
#define P2P_MAX_RANKS 16
struct p2p_args {
    float * data;                     // in: this rank's partial; out: the sum over ranks (in rank order)
    int n;                            // <= P2P_SLOT_FLOATS
    int rank, world;
    char * mbox[P2P_MAX_RANKS];       // every rank's mailbox as mapped HERE: [2 parities][world sources][P2P_SLOT_FLOATS] 8-byte granules
    unsigned * state;                 // device: [0] epoch of the last finished all-reduce, [1] arrivals of the running one, [2] time-outs
    unsigned max_spins;
    unsigned * err_host;              // host-mapped word (or null): raised together with state[2] so that the HOST sees a time-out without a device read —
                                      // graph_compute checks it on entry and reports GGML_STATUS_FAILED
    // fused epilogue (one launch of n <= P2P_SLOT_FLOATS only): out[i] = sum + add[i] (the residual ADD that follows a row-parallel mat-mul:
    // `add` rows of n values, or null) and, for ss_out != null, one partial sum of squares of `out` per workgroup (mmvq_args::ss_in's producer)
    const float * add;
    float * out;                      // null: in place
    double * ss_out;
};
void launch_p2p_all_reduce(hipStream_t s, const p2p_args & a);
int p2p_all_reduce_blocks(int n);  // workgroups (= ss_out partials) of a launch over n values
bool qkv_types_supported(int type_a, int type_b);
void launch_qkv(hipStream_t s, const qkv_args & a, int type_a, int type_b);

// ---- f16 / f32 weights (K cache, V cache, small dense) (mmf.hip): dst = src0 · src1 with ggml broadcasting;
// src1 rounded to f16 first when src0 is f16 (ggml-cpu vec_dot_type semantics)
void launch_mul_mat_f(hipStream_t s, const tdesc & src0, const tdesc & src1, const tdesc & dst, float * ws = nullptr, size_t ws_bytes = 0);  // ws: scratch for K-split partial tiles
size_t mul_mat_f_workspace_bytes(const tdesc & src0, const tdesc & src1);
// ---- non-flash attention chain of a small batch over position lists (attn_nf.hip)
size_t attn_nf_list_scratch_bytes(const tdesc & q, const tdesc & k, int * dq_out);
bool launch_attn_nf_list(hipStream_t s, const tdesc & q, const tdesc & k, const tdesc & v, const tdesc & mask, const tdesc & dst, const int * lists, int list_stride,
                         float * scratch, size_t scratch_bytes, float scale, void * q8_out = nullptr /* Q8_K blocks of the rows instead of f32 (one head-dimension slice, even group size) */);
bool launch_soft_max_mul_mat_f16(hipStream_t s, const tdesc & a, const tdesc & kq, const tdesc * mask, const tdesc & d, float scale);  // decode: SOFT_MAX folded into V^T.p (mmf.hip)

// ---- prefill: quantised weights x many columns through MFMA (mmq.hip)
bool mmq_supported(int type, int64_t K, int64_t N, int64_t M);
size_t mmq_workspace_bytes(int type, int64_t K, int64_t N, int64_t M, bool skinny);
// int8-matrix-core variant for Q4_K / Q5_K (mmq_i8.hip); force_bn: 0 = auto, 64 / 128 = weight-panel height
bool mmq_skinny_mix_ok(const int * types, const int64_t * N, const int64_t * w_nb1, int n, int64_t K, int64_t M);
struct mmq_mat_desc { const uint8_t * W; int64_t w_nb1; int N; float * dst; int64_t dst_stride; const float * add; int64_t add_stride;
                      int type = 0; };  // type: 0 = the launch's; another K-quant type makes it a two-format launch (2..32 columns only)
// Epilogue of the skinny kernel (2..32 columns, no K split) for the attention projections of a batch: the ROPE of q and k and both
// KV-cache stores happen where the projections' sums are complete — the rope + store launch of ops.hip (k_rope_qk_store) disappears.
// Adjacent-pair ("normal") rotation only: both elements of a pair sit in one 32-row tile.
struct mmq_epi {
    int kind[3];            // per matrix of the launch: 0 plain, 1 ROPE -> f32 rope(q) tensor, 2 ROPE -> f16 cache rows (k), 3 f16 cache rows (v),
                            // 4 f16 elements of the TRANSPOSED V cache (non-flash path): value n of token t -> element v_idx[t * N + n]
    char * out[3];          // kind 1: rope(q) data; 2 / 3: cache data
    int64_t nb1[3], nb2[3]; // kind 1: bytes per head / per token of rope(q); 2 / 3: nb1 = bytes per cache row
    const int32_t * pos;
    const float * ff;
    const int64_t * idx;    // cache row of each token
    const int64_t * v_idx;  // kind 4: one cache element index per (token, value)
    const float * tab;      // [token][n_dims / 2][cos, sin]: launch_rope_table's output for these positions and parameters
    float theta_scale, freq_scale, ext_factor, attn_factor, corr0, corr1;
    int head_dim, n_dims;
};
// up to three matrices of one type against the same activations in one launch (wq/wk/wv, ffn_gate/ffn_up of a batch)
// (both return which kernel served the launch: 0 the tiled GEMM of mmq_i8.hip, 1 the skinny weight-streaming kernel, 2 its wide form)
int launch_mmq_i8_multi(hipStream_t s, int type, int n_mat, const mmq_mat_desc * mats, int K, int M, const void * act_q8k, int force_bn, int ksplit, float * part, bool reduce = true, bool skinny = false,
                        const mmq_epi * epi = nullptr /* only with skinny-served launches and ksplit 1 (the caller checked mmq_skinny_supported) */);
void launch_splitk_reduce_mats(hipStream_t s, int n_mat, const mmq_mat_desc * mats, const float * part, int ks, int M);  // the reduce pass of launch_mmq_i8_multi(reduce = false), later
bool mmq_i8_supported(int type, int64_t K, int64_t N, int64_t M);
// ksplit > 1: the K range is split over that many workgroup rows, partials in `part` (mmq_workspace_bytes), summed in a fixed order
int mmq_pick_ksplit(int64_t K, int64_t N, int64_t M, bool skinny, int type);
void launch_splitk_reduce(hipStream_t s, const float * part, int ks, int M, int N, float * dst, int64_t dst_stride, const float * add, int64_t add_stride);
int launch_mmq_i8(hipStream_t s, int type, const uint8_t * W, int64_t w_nb1, int K, int N, int M, const void * act_q8k, float * dst, int64_t dst_stride, int force_bn,
                   int ksplit, float * part, const float * add /* optional bias row or residual */, int64_t add_stride, bool reduce = true, bool skinny = false, const mmq_epi * epi = nullptr);
// 2..32 columns (continuous-batching decode steps): the weight-streaming matrix-core kernel of mmq_skinny.hip, reached through
// launch_mmq_i8[_multi](skinny = true); the caller's activation area must hold 32 columns' worth of bytes (read, never used)
bool mmq_skinny_supported(int type, int64_t K, int64_t N, int64_t M, int64_t w_nb1);
int mmq_skinny_ksplit(int64_t K, int64_t n_total);
// Q8_0 weights x Q8_0 activations, one int8 MFMA per 32-value block + immediate f32 scale-accumulate (mmq_q80.hip)
bool mmq_q80_supported(int type, int64_t K, int64_t N, int64_t M);
void launch_mmq_q80(hipStream_t s, const uint8_t * W, const uint8_t * W_panels, int64_t w_nb1, int K, int N, int M, const void * act_q80, float * dst, int64_t dst_stride, const float * add, int64_t add_stride);
// 9 .. 32 columns: the weights streamed once, 32-row panels, K split over the waves of a workgroup (mmq_q80.hip; round 6)
bool mmq_q80_skinny_supported(int type, int64_t K, int64_t N, int64_t M);
struct mmq80s_desc { const uint8_t * W; const uint8_t * W_panels; int64_t w_nb1; int N; float * dst; int64_t dst_stride; const float * add; int64_t add_stride; };
void launch_mmq_q80_skinny_multi(hipStream_t s, int n, const mmq80s_desc * mats, int K, int M, const void * act_q80_panel);  // up to three matrices over the same activations
void launch_mmq_q80_skinny(hipStream_t s, const uint8_t * W, const uint8_t * W_panels, int64_t w_nb1, int K, int N, int M, const void * act_q80_panel, float * dst, int64_t dst_stride, const float * add, int64_t add_stride);
void launch_mmq(hipStream_t s, int type, const uint8_t * W, int64_t w_nb1, int K, int N, int M, const void * act_q8k, float * dst, int64_t dst_stride, int ksplit, float * part);

// ---- element-wise / normalisation / data movement (ops.hip)
void launch_rms_norm(hipStream_t s, const tdesc & src, const tdesc & dst, float eps, const tdesc * mul_w /* optional fused weight */);
void launch_binary(hipStream_t s, int op, const tdesc & a, const tdesc & b, const tdesc & dst);
void launch_scale(hipStream_t s, const tdesc & src, const tdesc & dst, float scale, float bias);
void launch_unary(hipStream_t s, int uop, const tdesc & src, const tdesc & dst);
void launch_swiglu(hipStream_t s, const tdesc & a, const tdesc * b, const tdesc & dst, int swapped);
// producers fused with the Q8_K activation quantisation (quantize.hip): the f32 intermediate is not written
// the same with the row ASSEMBLED first from split-K partials (+ epilogue add) — and written out as f32, for its other readers
struct splitk_src { const float * part; int ks; int64_t mn; const float * add; int64_t add_stride; float * out; int64_t out_stride; };
void launch_splitk_rms_norm_mul_quantize(hipStream_t s, const splitk_src & sk, int rows, int K, const float * w, float eps, void * dst);
void launch_rms_norm_mul_quantize(hipStream_t s, const tdesc & x, const float * w, float eps, void * dst_q8k);
void launch_swiglu_quantize(hipStream_t s, const tdesc & a, const tdesc * b, int64_t nc, int swapped, void * dst_q8k);
void launch_cpy(hipStream_t s, const tdesc & src, const tdesc & dst);
void launch_get_rows(hipStream_t s, const tdesc & src, const tdesc & idx, const tdesc & dst);
void launch_set_rows(hipStream_t s, const tdesc & src, const tdesc & idx, const tdesc & dst);
void launch_cpy_q8_0(hipStream_t s, const void * src, void * dst, int64_t n_values, bool to_q8);  // contiguous Q8_0 <-> F32 (K-shift of a quantised cache)
void launch_set_rows_q8_0(hipStream_t s, const tdesc & src, const tdesc & idx, const tdesc & dst);  // f32 rows -> block_q8_0 rows (quantised KV cache)
// kv_types.hip: a KV cache in one of the other types of -ctk / -ctv (q4_0, q4_1, q5_0, q5_1, iq4_nl, bf16; f32 for reading)
bool kv_type_is_block(int type);
bool kv_store_type(int type);  // SET_ROWS / CPY-from-f32 destination types served there
bool kv_image_type(int type);  // FLASH_ATTN_EXT reads K / V of this type through an f16 image in scratch memory
void launch_set_rows_kv(hipStream_t s, const tdesc & src, const tdesc & idx, const tdesc & dst);
void launch_set_rows_kv_pair(hipStream_t s, const tdesc & a0, const tdesc & i0, const tdesc & d0, const tdesc & a1, const tdesc & i1, const tdesc & d1);  // SET_ROWS(k) + SET_ROWS(v), one launch
void launch_cpy_kv(hipStream_t s, int type, const void * src, void * dst, int64_t n_values, bool to_type);  // contiguous type <-> F32 (K-shift)
size_t kv_image_bytes(const tdesc & t);
tdesc kv_image_desc(const tdesc & t, void * image);  // the f16 descriptor launch_kv_images_f16 leaves for this view (no launch)
void launch_kv_images_f16(hipStream_t s, tdesc & k, tdesc & v, void * image);  // one launch; rewrites the descriptors of the expanded ones
void launch_argmax(hipStream_t s, const tdesc & src, const tdesc & dst);
void launch_upload_multi(hipStream_t s, const upload_batch & b);  // up to 8 pinned-host -> device copies in one launch
void launch_upload_small(hipStream_t s, void * dst, const void * pinned_src, size_t n);
void launch_copy2d(hipStream_t s, void * dst, size_t dpitch, const void * src, size_t spitch, size_t width, size_t height);  // either side may be peer memory
void launch_rebase_row_index(hipStream_t s, int64_t * dst, const int64_t * src, int64_t n_tok, int64_t full, int64_t ext, int64_t o0, int64_t n_ctx);  // tp_inproc.cpp: transposed V cache
// dst[i] = (add ? add[i] : 0) + part[0][i] + part[1][i] + ... in device order (the partial products of a row-parallel mat-mul; the parts may
// live in peer devices' memory: read over xGMI by the main device's kernel)
struct reduce_parts { const float * part[GGML_MI355X_MAX_DEVICES]; int n; };
void launch_reduce_parts(hipStream_t s, const reduce_parts & p, const float * add, float * dst, int64_t n);


struct rope_params {
    int n_dims, mode, n_ctx_orig;
    float freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow;
    int sections[4];  // ggml_rope_multi (mode & GGML_ROPE_TYPE_MROPE): pairs per position stream (time, height, width, extra)
};
void rope_host_consts(const rope_params & p, float & theta_scale, float & corr0, float & corr1);
// (cos, sin) of every (token, rotation pair) of a small batch — rope_cos_sin's arithmetic, once per graph run instead of once per layer and element
void launch_rope_table(hipStream_t s, const int32_t * pos, const float * ff, const rope_params & p, int n_tok, float * tab);
// GET_ROWS + the mask's F32 -> F16 cast + the rotary (cos, sin) table of a decode step in one launch (ops.hip: k_step_head); absent parts: a == nullptr / cp_n == 0 / n_tok == 0
void launch_step_head(hipStream_t s, const tdesc * a, const tdesc * idx, const tdesc * d, const float * cp_src, void * cp_dst, int64_t cp_n,
                      const int32_t * pos, const float * ff, const rope_params * p, int n_tok, float * tab);
// batches: ROPE(q) + ROPE(k) + SET_ROWS(k) + SET_ROWS(v) in one launch (f32 sources [head_dim, heads, tokens], f16 cache rows)
struct rope_store_args {
    const char * q_src; char * q_dst; const char * k_src; const char * v_src;
    int64_t q_nb1, q_nb2, qd_nb1, qd_nb2, k_nb1, k_nb2, v_nb1, v_nb2;
    char * k_cache; char * v_cache;
    int64_t kc_nb1, vc_nb1;
    const int64_t * idx;
    const int64_t * v_idx;  // transposed V cache (non-flash path): one index per ELEMENT, [token][nkv * head_dim]; value j of the token -> f16 element v_idx[..] of v_cache
    const int32_t * pos;
    const float * ff;
    rope_params p;
    float theta_scale, corr0, corr1;
    int nh, nkv, head_dim;
    // a source that is still the split-K partial products of its projection ([ks][tokens][rows] floats + optional bias row): the kernel
    // sums them itself, in split order then bias — what the reduce pass would have written (part == nullptr: read *_src)
    struct { const float * part; int64_t mn; const float * bias; int n; } sk[3];  // q, k, v
    int ks;
};
void launch_rope_qk_store(hipStream_t s, rope_store_args a, int n_tokens);
// prompt micro-batches: the vectorised form over the per-run (cos, sin) table (ops.hip: k_rope_qk_store_vec)
bool rope_qk_store_vec_ok(const rope_store_args & a, int n_tokens);
void launch_rope_qk_store_vec(hipStream_t s, const rope_store_args & a, int n_tokens, const float * tab);
void launch_rope(hipStream_t s, const tdesc & src, const tdesc & pos, const float * freq_factors, const tdesc & dst, const rope_params & p);
void launch_soft_max(hipStream_t s, const tdesc & src, const tdesc * mask, const float * sinks, const tdesc & dst, float scale, float max_bias);

// ---- attention (fattn.hip)
struct fattn_params {
    float scale, max_bias, logit_softcap;
    int n_splits;  // KV splits per (token, kv-head group)
    int fat = 0;   // 1: single-token decode as a few fat splits on 8-wave workgroups, records FA_REC apart, NO combine pass (the wo
                   // mat-vec prologue merges them: mmvq_args::fa_part); see fattn_fat_splits()
    int kv_type;   // GGML_TYPE_F16 or GGML_TYPE_Q8_0 (K and V alike)
    int dq = 0;    // 1: K / V are read IN PLACE in another cache type (tdesc::type each: q4_0, q4_1, q5_0, q5_1, iq4_nl, bf16, q8_0, f16) by the lane-parallel kernel's
                   // DQ form — the caller asked fattn_native_kv_ok() first; kv_type is GGML_TYPE_F16 then (everything but the fetch is the f16 path)
    const int * lists = nullptr;  // per-token lists of visible cache positions (launch_fattn_tile_scan), or nullptr
    void * q8_out = nullptr;  // the result's only readers are quantised mat-muls (wo of a batch): leave it as Q8_K blocks here — honoured by
                              // the combine pass of the head_dim-128 kernels (n_splits > 1), see fattn_q8_out_ok()
    unsigned * arrive = nullptr;  // zeroed counters (arrive_slots of them): lets the split kernel of the head_dim-128 decode path merge its own
    int arrive_slots = 0;         // partial records (last workgroup to arrive) instead of a second launch; the kernel leaves them at zero
    const uint8_t * tile_vis = nullptr;  // matrix-core kernel: [q tile of 32][kv tile of 64] visibility bytes (launch_fattn_vis_scan), or nullptr
    int mask_sparse = 0;  // 1: the mask's CONTENT is known (it passed through set_tensor, possibly behind the host's F32 -> F16 cast node) and at most a
                          // quarter of its cells are visible (common.h: mask_sparse_hint, resolved by graph.cpp) — position lists beat dense tiles
};
// the non-flash chain (K.q -> SOFT_MAX -> V^T.p) of a prompt micro-batch on the matrix cores, two passes (fattn_mma.hip: k_attn_nf_mma)
bool attn_nf_mma_applies(const tdesc & q, const tdesc & k, const tdesc & vt, const tdesc & mask);
size_t attn_nf_mma_ws_bytes(const tdesc & q, int n_splits);
void launch_attn_nf_mma(hipStream_t s, const tdesc & q, const tdesc & k, const tdesc & vt, const tdesc & mask, const tdesc & dst, float scale, int n_splits, const uint8_t * tile_vis, void * workspace,
                        void * q8_out = nullptr);
bool fattn_native_kv_ok(const tdesc & q, const tdesc & k, const tdesc & v, const fattn_params & p);  // may launch_flash_attn read this K / V pair in place (p.dq = 1)?
bool fattn_combine_rows_applies(int D, int64_t n_q, int64_t n_head, int64_t n_batch, int n_splits, const float * sinks);  // the row-parallel combine pass (prompt micro-batches)
size_t fattn_vis_bytes(const tdesc & q, const tdesc & k);
void launch_fattn_vis_scan(hipStream_t s, const tdesc & mask, int n_q, int n_kv, uint8_t * vis);
// few query tokens over a large unified cache (continuous batching): tile size for the tile-list attention kernel, 0 = not applicable
int fattn_list_tile(const tdesc & q, const tdesc & k, const tdesc * mask, const fattn_params & p, size_t lists_bytes);
void launch_fattn_tile_scan(hipStream_t s, const tdesc & mask, int n_q, int n_kv, int tile, int * lists);
size_t fattn_workspace_bytes(const tdesc & q, const tdesc & k, const tdesc & v, int n_splits, int kv_type);
int fattn_mma_min_q();  // query tokens from which the matrix-core attention kernel takes over (env GGML_MI355X_FA_MMA_MIN_Q)
bool flash_attn_mma_applies(const tdesc & q, const tdesc & k, const tdesc * mask, const float * sinks, const tdesc & dst, const fattn_params & p);
int fattn_pick_splits(const tdesc & q, const tdesc & k, const tdesc * mask = nullptr, int mask_sparse = 0);
// 33+ query tokens normally run on the matrix-core kernel; a batch whose mask is KNOWN to be sparse (mask_sparse_hint: draft-verification
// batches of many sequences over a unified cache, llama-box/httpserver.hpp:4042-4069) of up to 256 tokens walks per-token position lists instead
bool fattn_prefers_lists(const tdesc & q, const tdesc * mask, int mask_sparse);
int fattn_fat_splits(const tdesc & q, const tdesc & k, const tdesc * mask, const float * sinks, const fattn_params & p);  // 0 = the fat-split form does not apply
bool fattn_q8_out_ok(const tdesc & q, const tdesc & k, const tdesc * mask, const float * sinks, const tdesc & dst, const fattn_params & p);  // will this launch end in the quantising combine?
void launch_flash_attn(hipStream_t s, const tdesc & q, const tdesc & k, const tdesc & v, const tdesc * mask, const float * sinks,
                       const tdesc & dst, const fattn_params & p, void * workspace);
// matrix-core variant for batches of >= 32 query tokens (fattn_mma.hip); false = does not apply
bool launch_flash_attn_mma(hipStream_t s, const tdesc & q, const tdesc & k, const tdesc & v, const tdesc * mask, const float * sinks, const tdesc & dst,
                           const fattn_params & p, void * workspace);
int fattn_mma_pick_splits(const tdesc & q, const tdesc & k);
void launch_flash_attn_combine(hipStream_t s, int D, const float * ws, const float * sinks, const tdesc & dst, int n_q, int n_head, int n_batch, int n_splits, void * q8_out = nullptr);

// ---- the decode copy (repack.hip): rows [r0, r0 + n_rows) of a K-quant [K, N] matrix from the block layout at `src` into the plane layout at `dst` (same row stride)
bool repack_supported(int type, int64_t K, int64_t nb1);
void launch_repack_planes(hipStream_t s, int type, const void * src, void * dst, int64_t K, int64_t nb1, int64_t r0, int64_t n_rows);
bool repack_q80_supported(int type, int64_t K, int64_t N, int64_t nb1);  // Q8_0: the panel copy of the 9 .. 32-column matrix-core kernel (repack.hip)
void launch_repack_q80_panels(hipStream_t s, const void * src, void * dst, int64_t K, int64_t N, int64_t nb1);

// ---- code-object preload (round 4).  The HIP runtime loads a translation unit's device code on the FIRST launch of one of its kernels
// (0.3 - 2 ms each, measured as idle gaps in front of the first prompt's kernels: 5.5 ms of a 64 ms prefill).  Every kernel file ends with
// MI_TU_TOUCH(name): an empty kernel + a launcher; init_backend launches them all once per device, so the cost moves to start-up.
#define MI_TU_TOUCH(name) \
    __global__ void k_tu_touch_##name() {} \
    void tu_touch_##name(hipStream_t s) { hipLaunchKernelGGL(k_tu_touch_##name, dim3(1), dim3(64), 0, s); }
void tu_touch_quantize(hipStream_t s);
void tu_touch_mmvq(hipStream_t s);
void tu_touch_qkv(hipStream_t s);
void tu_touch_mmf(hipStream_t s);
void tu_touch_attn_nf(hipStream_t s);
void tu_touch_mmq(hipStream_t s);
void tu_touch_mmq_i8(hipStream_t s);
void tu_touch_mmq_skinny(hipStream_t s);
void tu_touch_mmq_q80(hipStream_t s);
void tu_touch_ops(hipStream_t s);
void tu_touch_fattn(hipStream_t s);
void tu_touch_fattn_mma(hipStream_t s);
void tu_touch_tp_p2p(hipStream_t s);
void tu_touch_kv_types(hipStream_t s);
void tu_touch_repack(hipStream_t s);
inline void preload_kernel_files(hipStream_t s) {
    tu_touch_kv_types(s);
    tu_touch_repack(s);
    tu_touch_quantize(s);
    tu_touch_mmvq(s);
    tu_touch_qkv(s);
    tu_touch_mmf(s);
    tu_touch_attn_nf(s);
    tu_touch_mmq(s);
    tu_touch_mmq_i8(s);
    tu_touch_mmq_skinny(s);
    tu_touch_mmq_q80(s);
    tu_touch_ops(s);
    tu_touch_fattn(s);
    tu_touch_fattn_mma(s);
    tu_touch_tp_p2p(s);
}

}  // namespace mi355x
