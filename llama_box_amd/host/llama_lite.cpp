// llama_lite.cpp — see llama_lite.h.  Own driver: model container (synthetic or GGUF), unified KV cache,
// per-micro-batch graph construction in the op order llm_build_llama / llm_build_qwen2 emit,
// and a llama_decode-shaped entry point.  Harness code; depends only on ggml_lite.h (never on the oracle).
#include "llama_lite.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "gguf_lite.h"

#define LLM_ASSERT(x)                                                                           \
    do {                                                                                        \
        if (!(x)) {                                                                             \
            fprintf(stderr, "llama_lite: %s:%d: assertion failed: %s\n", __FILE__, __LINE__, #x); \
            abort();                                                                            \
        }                                                                                       \
    } while (0)

// ---------------------------------------------------------------------------------------------- utilities
static uint16_t f32_to_f16(float f) {  // IEEE binary16, round-to-nearest-even
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint16_t sign = (uint16_t) ((x >> 16) & 0x8000);
    const uint32_t ax = x & 0x7FFFFFFFu;
    if (ax >= 0x7F800000u) return (uint16_t) (sign | 0x7C00 | (ax > 0x7F800000u ? 0x200 : 0));
    if (ax >= 0x477FF000u) return (uint16_t) (sign | 0x7C00);
    if (ax < 0x33000001u) return sign;
    const int32_t e = (int32_t) (ax >> 23) - 127;
    uint32_t man = (ax & 0x7FFFFFu) | 0x800000u, shift = 13, base = 0;
    if (e < -14) shift = (uint32_t) (13 + (-14 - e));
    else { base = (uint32_t) (e + 15) << 10; man &= 0x7FFFFFu; }
    const uint32_t halfway = 1u << (shift - 1), rem = man & ((1u << shift) - 1);
    uint32_t q = man >> shift;
    if (rem > halfway || (rem == halfway && (q & 1))) q++;
    return (uint16_t) (sign | (base + q));
}

static inline uint64_t splitmix64(uint64_t & s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline float u01(uint64_t r) { return (float) ((r >> 40) * (1.0 / 16777216.0)); }

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ------------------------------------------------------------------------------ synthetic quant blocks
// Blocks are sampled directly in the packed domain (SURVEY.md §8d "Synthetic inputs"): quants uniform, 6-bit
// scales uniform with mins tied to scales so that dequantised weights are ~zero-mean with std ~ s/sqrt(K).
// Every block is a pure function of (seed, tensor id, GLOBAL row, GLOBAL block-in-row) so that a
// tensor-parallel shard generates exactly the bytes the unsharded model holds at the same coordinates.
static void synth_block(ggml_type type, uint64_t key, int64_t K, float wscale, uint8_t * out) {
    uint64_t s = key;
    const float scale = wscale * (0.5f + u01(splitmix64(s)));  // s in U(0.5, 1.5) x per-tensor gain
    const float rk = 1.0f / sqrtf((float) K);
    auto fill = [&](uint8_t * p, size_t n) {
        size_t i = 0;
        for (; i + 8 <= n; i += 8) {
            uint64_t r = splitmix64(s);
            memcpy(p + i, &r, 8);
        }
        if (i < n) {
            uint64_t r = splitmix64(s);
            memcpy(p + i, &r, n - i);
        }
    };
    switch (type) {
        case GGML_TYPE_Q8_0: {
            block_q8_0 * b = (block_q8_0 *) out;
            fill((uint8_t *) b->qs, 32);
            for (int i = 0; i < 32; ++i) if (b->qs[i] == -128) b->qs[i] = -127;
            b->d = f32_to_f16(scale * rk / 73.0f);
        } break;
        case GGML_TYPE_Q4_K:
        case GGML_TYPE_Q5_K: {
            uint8_t * p = out;
            uint8_t sc[8], mn[8];
            uint64_t r = splitmix64(s), r2 = splitmix64(s);
            for (int j = 0; j < 8; ++j) {
                sc[j] = (uint8_t) (8 + ((r >> (8 * j)) & 0xFF) % 56);
                int m = (int) sc[j] + (int) ((r2 >> (8 * j)) & 0xFF) % 9 - 4;
                mn[j] = (uint8_t) std::min(63, std::max(0, m));
            }
            const bool q5 = type == GGML_TYPE_Q5_K;
            const float qmean = q5 ? 15.5f : 7.5f, qstd = q5 ? 9.2f : 4.6f;
            const float d = scale * rk / (36.0f * qstd);
            const uint16_t hd = f32_to_f16(d), hm = f32_to_f16(d * qmean);
            memcpy(p, &hd, 2);
            memcpy(p + 2, &hm, 2);
            uint8_t * q = p + 4;
            for (int j = 0; j < 4; ++j) {
                q[j] = (uint8_t) ((sc[j] & 63) | ((sc[j + 4] >> 4) << 6));
                q[j + 4] = (uint8_t) ((mn[j] & 63) | ((mn[j + 4] >> 4) << 6));
                q[j + 8] = (uint8_t) ((sc[j + 4] & 0xF) | ((mn[j + 4] & 0xF) << 4));
            }
            fill(p + 16, q5 ? 32 + 128 : 128);
        } break;
        case GGML_TYPE_Q6_K: {
            block_q6_K * b = (block_q6_K *) out;
            fill(b->ql, 128 + 64);
            uint64_t r = splitmix64(s), r2 = splitmix64(s);
            for (int j = 0; j < 16; ++j) {
                uint64_t rr = j < 8 ? r : r2;
                int v = 16 + (int) ((rr >> (8 * (j & 7))) & 0x7F) % 112;  // 16..127
                if ((rr >> (8 * (j & 7) + 7)) & 1) v = -v;
                b->scales[j] = (int8_t) v;
            }
            b->d = f32_to_f16(scale * rk / (70.0f * 18.5f));
        } break;
        default: LLM_ASSERT(!"synth_block: unsupported type");
    }
}

// f16 -> f32 (harness-side: the tied output rows below are re-encoded from token_embd's dequantised values)
static float f16_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t) (h & 0x8000) << 16, e = (h >> 10) & 31, man = h & 0x3FF;
    uint32_t x;
    if (e == 0) {
        if (!man) x = sign;
        else {
            int sh = 0;
            uint32_t m = man;
            while (!(m & 0x400)) { m <<= 1; ++sh; }
            x = sign | (uint32_t) (127 - 15 - sh + 1) << 23 | (m & 0x3FF) << 13;
        }
    } else if (e == 31) x = sign | 0x7F800000u | man << 13;
    else x = sign | (e + 112) << 23 | man << 13;
    float f;
    memcpy(&f, &x, 4);
    return f;
}

// the 256 values a synthetic K-quant super-block stands for (block layouts: SURVEY.md Appendix A.1 / A.2)
static void block_values(ggml_type type, const uint8_t * p, float * y) {
    auto scale_min = [](int j, const uint8_t * q, int & sc, int & mn) {
        if (j < 4) { sc = q[j] & 63; mn = q[j + 4] & 63; }
        else { sc = (q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4); mn = (q[j + 4] >> 4) | ((q[j] >> 6) << 4); }
    };
    switch (type) {
        case GGML_TYPE_Q4_K:
        case GGML_TYPE_Q5_K: {
            const bool q5 = type == GGML_TYPE_Q5_K;
            uint16_t hd, hm;
            memcpy(&hd, p, 2);
            memcpy(&hm, p + 2, 2);
            const float d = f16_to_f32(hd), dmin = f16_to_f32(hm);
            const uint8_t * scales = p + 4, * qh = p + 16, * ql = p + (q5 ? 16 + 32 : 16);
            for (int j = 0; j < 4; ++j) {
                int s1, m1, s2, m2;
                scale_min(2 * j, scales, s1, m1);
                scale_min(2 * j + 1, scales, s2, m2);
                for (int l = 0; l < 32; ++l) {
                    const int lo = (ql[32 * j + l] & 0xF) + (q5 && (qh[l] >> (2 * j) & 1) ? 16 : 0);
                    const int hi = (ql[32 * j + l] >> 4) + (q5 && (qh[l] >> (2 * j + 1) & 1) ? 16 : 0);
                    y[64 * j + l] = d * (float) s1 * (float) lo - dmin * (float) m1;
                    y[64 * j + 32 + l] = d * (float) s2 * (float) hi - dmin * (float) m2;
                }
            }
        } break;
        case GGML_TYPE_Q6_K: {
            const block_q6_K * b = (const block_q6_K *) p;
            const float d = f16_to_f32(b->d);
            for (int n = 0; n < 2; ++n) {
                const uint8_t * ql = b->ql + 64 * n, * qh = b->qh + 32 * n;
                const int8_t * sc = b->scales + 8 * n;
                for (int l = 0; l < 32; ++l) {
                    const int is = l / 16;
                    const int q1 = (int) ((ql[l] & 0xF) | (((qh[l] >> 0) & 3) << 4)) - 32, q2 = (int) ((ql[l + 32] & 0xF) | (((qh[l] >> 2) & 3) << 4)) - 32;
                    const int q3 = (int) ((ql[l] >> 4) | (((qh[l] >> 4) & 3) << 4)) - 32, q4 = (int) ((ql[l + 32] >> 4) | (((qh[l] >> 6) & 3) << 4)) - 32;
                    y[128 * n + l] = d * (float) sc[is] * (float) q1;
                    y[128 * n + l + 32] = d * (float) sc[is + 2] * (float) q2;
                    y[128 * n + l + 64] = d * (float) sc[is + 4] * (float) q3;
                    y[128 * n + l + 96] = d * (float) sc[is + 6] * (float) q4;
                }
            }
        } break;
        default: LLM_ASSERT(!"block_values: unsupported type");
    }
}

// 256 values -> one Q6_K super-block (a plain absmax encoder: any valid block will do for a synthetic weight set — this is the model
// generator, not a restatement of ggml's quantize_row_q6_K)
static void pack_q6_K(const float * x, uint8_t * out) {
    block_q6_K * b = (block_q6_K *) out;
    float sub[16], smax = 0.0f;
    for (int j = 0; j < 16; ++j) {
        float a = 0.0f;
        for (int l = 0; l < 16; ++l) a = std::max(a, fabsf(x[16 * j + l]));
        sub[j] = a / 31.0f;
        smax = std::max(smax, sub[j]);
    }
    memset(b, 0, sizeof(*b));
    if (smax == 0.0f) return;
    const float d = f16_to_f32(f32_to_f16(smax / 127.0f));
    b->d = f32_to_f16(smax / 127.0f);
    int q[256];
    for (int j = 0; j < 16; ++j) {
        const int sc = std::max(1, std::min(127, (int) lrintf(sub[j] / d)));
        b->scales[j] = (int8_t) sc;
        for (int l = 0; l < 16; ++l) q[16 * j + l] = std::max(-32, std::min(31, (int) lrintf(x[16 * j + l] / (d * (float) sc)))) + 32;
    }
    for (int n = 0; n < 2; ++n)
        for (int l = 0; l < 32; ++l) {
            const int q1 = q[128 * n + l], q2 = q[128 * n + l + 32], q3 = q[128 * n + l + 64], q4 = q[128 * n + l + 96];
            b->ql[64 * n + l] = (uint8_t) ((q1 & 0xF) | ((q3 & 0xF) << 4));
            b->ql[64 * n + l + 32] = (uint8_t) ((q2 & 0xF) | ((q4 & 0xF) << 4));
            b->qh[32 * n + l] = (uint8_t) ((q1 >> 4) | ((q2 >> 4) << 2) | ((q3 >> 4) << 4) | ((q4 >> 4) << 6));
        }
}

struct synth_spec {
    uint64_t seed;
    int tensor_id;
    ggml_type type;
    int64_t K_global;       // full row length (elements)
    int64_t row_off;        // global index of local row 0
    int64_t blk_off;        // global block index (within a row) of local block 0
    int64_t blocks_per_row; // local
    int64_t blocks_per_row_global;
    int64_t n_rows;         // local
    float f32_lo, f32_hi;   // for F32 tensors
    float wscale;           // gain of quantised tensors
    int tie_id = -1;        // >= 0: `tie_eighths` of every 8 blocks repeat the block of tensor tie_id at row (row * 7919 + 13) % tie_rows (llm_hparams::peaked)
    int tie_eighths = 0;
    int64_t tie_rows = 0;
    ggml_type tie_type = GGML_TYPE_F32;  // the tied tensor's block format; when it is not `type`, its values are re-encoded (Q6_K)
    float tie_wscale = 1.0f;             // ... at the tied tensor's own gain, times tie_gain
    float tie_gain = 1.0f;
};

static void synth_rows(const synth_spec & sp, int64_t row0, int64_t row1, uint8_t * dst) {
    const size_t bs = ggml_abi_type_size(sp.type);
    if (sp.type == GGML_TYPE_F32 || sp.type == GGML_TYPE_F16) {
        const int64_t n = sp.blocks_per_row;  // elements per local row
        for (int64_t r = row0; r < row1; ++r) {
            for (int64_t i = 0; i < n; ++i) {
                uint64_t s = sp.seed ^ ((uint64_t) sp.tensor_id * 0xD1B54A32D192ED03ull) ^ ((uint64_t) ((r + sp.row_off) * sp.blocks_per_row_global + i + sp.blk_off) * 0x9E3779B97F4A7C15ull);
                const float v = sp.f32_lo + (sp.f32_hi - sp.f32_lo) * u01(splitmix64(s));
                if (sp.type == GGML_TYPE_F32) memcpy(dst + ((r - row0) * n + i) * 4, &v, 4);
                else { uint16_t h = f32_to_f16(v); memcpy(dst + ((r - row0) * n + i) * 2, &h, 2); }
            }
        }
        return;
    }
    for (int64_t r = row0; r < row1; ++r) {
        for (int64_t b = 0; b < sp.blocks_per_row; ++b) {
            const uint64_t gb = (uint64_t) ((r + sp.row_off) * sp.blocks_per_row_global + b + sp.blk_off);
            uint64_t key = sp.seed ^ ((uint64_t) sp.tensor_id * 0xD1B54A32D192ED03ull) ^ (gb * 0x9E3779B97F4A7C15ull);
            if (sp.tie_id >= 0) {
                uint64_t pick = key ^ 0xA5A5A5A55A5A5A5Aull;
                if ((int) (splitmix64(pick) & 7) < sp.tie_eighths) {
                    const uint64_t tr = (uint64_t) (((r + sp.row_off) * 7919 + 13) % sp.tie_rows);
                    key = sp.seed ^ ((uint64_t) sp.tie_id * 0xD1B54A32D192ED03ull) ^ ((tr * (uint64_t) sp.blocks_per_row_global + (uint64_t) (b + sp.blk_off)) * 0x9E3779B97F4A7C15ull);
                    if (sp.tie_type != sp.type) {  // Q4_K_M / Q5_K_M: token_embd is Q4_K / Q5_K, output.weight Q6_K — the tied block is token_embd's VALUES in Q6_K
                        LLM_ASSERT(sp.type == GGML_TYPE_Q6_K && ggml_abi_blck_size(sp.tie_type) == 256);
                        uint8_t src[256];
                        float v[256];
                        synth_block(sp.tie_type, key, sp.K_global, sp.tie_wscale, src);
                        block_values(sp.tie_type, src, v);
                        for (float & f : v) f *= sp.tie_gain;
                        pack_q6_K(v, dst + (size_t) ((r - row0) * sp.blocks_per_row + b) * bs);
                        continue;
                    }
                }
            }
            synth_block(sp.type, key, sp.K_global, sp.wscale, dst + (size_t) ((r - row0) * sp.blocks_per_row + b) * bs);
        }
    }
}

// ---------------------------------------------------------------------------------------------- presets
extern "C" int llm_preset(const char * name, struct llm_hparams * hp) {
    memset(hp, 0, sizeof(*hp));
    auto set = [&](const char * arch, int L, int E, int H, int HKV, int HD, int FF, int V, int CTX, float base, float eps, int rope, int bias, int ft) {
        snprintf(hp->arch, sizeof(hp->arch), "%s", arch);
        hp->n_layer = L; hp->n_embd = E; hp->n_head = H; hp->n_head_kv = HKV; hp->n_embd_head = HD; hp->n_ff = FF;
        hp->n_vocab = V; hp->n_ctx_train = CTX; hp->rope_freq_base = base; hp->rms_eps = eps; hp->rope_type = rope;
        hp->qkv_bias = bias; hp->ftype = ft;
    };
    std::string n = name;
    // "<preset>-damped": the same architecture and quantisation recipe with the two properties of a TRAINED network that independent random
    // weights lack — small residual branches (gain 0.08 / sqrt(2 n_layer): a flipped 8-bit activation rounding injects 1/127 of a block's range
    // however small the difference that caused it, and with branches of gain 0.25 every later quantiser re-amplifies it; measured on the oracle
    // against itself, Llama-3-8B: logits NMSE 3.1e-4 at gain 1/8, 4.4e-5 at 0.03, 9.9e-6 at 0.01 — scripts/lab/damped_probe.py) and an output matrix
    // tied to the embeddings (one logit far ahead of the rest): the weight sets north_star's 1e-3 / exact-id bar is testable on
    const bool damped = n.size() > 7 && n.compare(n.size() - 7, 7, "-damped") == 0;
    if (damped) {
        n.resize(n.size() - 7);
        if (llm_preset(n.c_str(), hp) != 0) return -1;
        hp->peaked = 3;
        hp->branch_gain = 0.08f / sqrtf(2.0f * (float) hp->n_layer);
        return 0;
    }
    if (n == "tinyllama-1.1b-q8_0") set("llama", 22, 2048, 32, 4, 64, 5632, 32000, 2048, 10000.0f, 1e-5f, 0, 0, LLM_FTYPE_Q8_0);
    else if (n == "tinyllama-1.1b-q8_0-peaked") { set("llama", 22, 2048, 32, 4, 64, 5632, 32000, 2048, 10000.0f, 1e-5f, 0, 0, LLM_FTYPE_Q8_0); hp->peaked = 3; }
    else if (n == "llama3-8b-q4_k_m") set("llama", 32, 4096, 32, 8, 128, 14336, 128256, 8192, 500000.0f, 1e-5f, 0, 0, LLM_FTYPE_Q4_K_M);
    else if (n == "llama3-70b-q4_k_m") { set("llama", 80, 8192, 64, 8, 128, 28672, 128256, 8192, 500000.0f, 1e-5f, 0, 0, LLM_FTYPE_Q4_K_M); hp->attn_v_q5k_70b = 1; }
    else if (n == "qwen2-7b-q5_k_m") set("qwen2", 28, 3584, 28, 4, 128, 18944, 152064, 32768, 1000000.0f, 1e-6f, GGML_ROPE_TYPE_NEOX, 1, LLM_FTYPE_Q5_K_M);
    // shapes of other families a llama-box user brings (round 6): multi-head attention (1 query head per KV head), 3 query heads per KV head on rows of 12 / 32
    // super-blocks, head_dim 64 with 4 query heads per KV head
    else if (n == "llama3-8b-q8_0") set("llama", 32, 4096, 32, 8, 128, 14336, 128256, 8192, 500000.0f, 1e-5f, 0, 0, LLM_FTYPE_Q8_0);
    else if (n == "llama2-7b-q4_k_m") set("llama", 32, 4096, 32, 32, 128, 11008, 32000, 4096, 10000.0f, 1e-5f, 0, 0, LLM_FTYPE_Q4_K_M);
    else if (n == "llama3.2-3b-q4_k_m") set("llama", 28, 3072, 24, 8, 128, 8192, 128256, 8192, 500000.0f, 1e-5f, 0, 0, LLM_FTYPE_Q4_K_M);
    else if (n == "llama3.2-1b-q4_k_m") set("llama", 16, 2048, 32, 8, 64, 8192, 128256, 8192, 500000.0f, 1e-5f, 0, 0, LLM_FTYPE_Q4_K_M);
    else if (n == "test-llama") set("llama", 3, 256, 4, 2, 64, 512, 512, 512, 10000.0f, 1e-5f, 0, 0, LLM_FTYPE_MIXED);
    else if (n == "test-llama-tp") set("llama", 2, 512, 8, 4, 64, 1024, 512, 512, 10000.0f, 1e-5f, 0, 0, LLM_FTYPE_MIXED);
    // four-way tensor split with head_dim 128: 8 heads on 4 KV heads (2 + 1 per rank), wo K slices of 256, ffn_down K slices of 512, 1024 vocab rows per rank
    else if (n == "test-llama-tp4") set("llama", 2, 1024, 8, 4, 128, 2048, 4096, 512, 10000.0f, 1e-5f, 0, 0, LLM_FTYPE_MIXED);
    else if (n == "test-qwen2") set("qwen2", 2, 256, 4, 2, 64, 768, 768, 512, 1000000.0f, 1e-6f, GGML_ROPE_TYPE_NEOX, 1, LLM_FTYPE_MIXED);
    else return -1;
    return 0;
}

// ---------------------------------------------------------------------------------------------- model
struct llm_layer {
    ggml_tensor *attn_norm = nullptr, *wq = nullptr, *wk = nullptr, *wv = nullptr, *wo = nullptr;
    ggml_tensor *bq = nullptr, *bk = nullptr, *bv = nullptr;
    ggml_tensor *ffn_norm = nullptr, *ffn_gate = nullptr, *ffn_up = nullptr, *ffn_down = nullptr;
};

struct tensor_plan {
    std::string name;
    ggml_type type;
    int64_t ne0, ne1;       // LOCAL shape
    int64_t K_global;       // global ne0
    int64_t row_off, k_off; // shard offsets (elements)
    int tensor_id;
    bool rowpar;            // lives in the reducing (row-parallel) buffer type
    float lo, hi;
    float wscale;
    int tie_id = -1, tie_eighths = 0;
    int64_t tie_rows = 0;
    ggml_type tie_type = GGML_TYPE_F32;
    float tie_wscale = 1.0f;
};

struct llm_model {
    llm_hparams hp{};
    int tp_rank = 0, tp_size = 1;
    int n_head_l = 0, n_head_kv_l = 0, n_ff_l = 0, n_vocab_l = 0;
    ggml_context * ctx = nullptr;
    std::vector<ggml_backend_buffer_t> buffers;
    ggml_tensor *tok_embd = nullptr, *output_norm = nullptr, *output = nullptr;
    std::vector<llm_layer> layers;
    uint64_t stream_bytes = 0, total_bytes = 0;
    int n_dev = 1;               // -sm layer: devices the layers are spread over (llm_model_synth_layer_split)
    std::vector<int> layer_dev;  // ... and which of them holds layer il
};

static bool use_more_bits(int i, int n) { return i < n / 8 || i >= 7 * n / 8 || (i - n / 8) % 3 == 2; }

static ggml_type pick_type(const llm_hparams & hp, const char * what, int il) {
    const bool more = use_more_bits(il, hp.n_layer);
    const std::string w = what;
    switch (hp.ftype) {
        case LLM_FTYPE_Q8_0: return GGML_TYPE_Q8_0;
        case LLM_FTYPE_Q6_K: return GGML_TYPE_Q6_K;
        case LLM_FTYPE_F16: return GGML_TYPE_F16;
        case LLM_FTYPE_Q4_K_M:
            if (w == "output") return GGML_TYPE_Q6_K;
            if (w == "attn_v") return more ? GGML_TYPE_Q6_K : (hp.attn_v_q5k_70b ? GGML_TYPE_Q5_K : GGML_TYPE_Q4_K);
            if (w == "ffn_down") return more ? GGML_TYPE_Q6_K : GGML_TYPE_Q4_K;
            return GGML_TYPE_Q4_K;
        case LLM_FTYPE_Q5_K_M:
            if (w == "output") return GGML_TYPE_Q6_K;
            if (w == "attn_v" || w == "ffn_down") return more ? GGML_TYPE_Q6_K : GGML_TYPE_Q5_K;
            return GGML_TYPE_Q5_K;
        case LLM_FTYPE_MIXED: {
            static const ggml_type cyc[4] = {GGML_TYPE_Q4_K, GGML_TYPE_Q5_K, GGML_TYPE_Q6_K, GGML_TYPE_Q8_0};
            unsigned h = (unsigned) il * 7u;
            for (char c : w) h = h * 31u + (unsigned char) c;
            return cyc[h % 4];
        }
        default: return GGML_TYPE_Q8_0;
    }
}

static std::vector<tensor_plan> make_plan(const llm_hparams & hp, int tp_rank, int tp_size, bool split_mm = false) {
    std::vector<tensor_plan> plan;
    int id = 0;
    const int64_t E = hp.n_embd, HD = hp.n_embd_head;
    LLM_ASSERT(hp.n_head % tp_size == 0 && hp.n_head_kv % tp_size == 0 && hp.n_ff % tp_size == 0 && hp.n_vocab % tp_size == 0);
    const int64_t nq = (int64_t) hp.n_head * HD, nkv = (int64_t) hp.n_head_kv * HD;
    const int64_t nq_l = nq / tp_size, nkv_l = nkv / tp_size, ff_l = hp.n_ff / tp_size, v_l = hp.n_vocab / tp_size;
    auto add = [&](const std::string & name, ggml_type type, int64_t ne0, int64_t ne1, int64_t Kg, int64_t row_off, int64_t k_off, bool rowpar, float lo = 0, float hi = 0) {
        // residual-branch output projections get a small gain, as in trained transformers: the residual stream stays
        // dominated by its own history, so one-ulp differences are not chaotically amplified layer over layer
        const bool branch_out = name.find("attn_output") != std::string::npos || name.find("ffn_down") != std::string::npos;
        // (llm_hparams::branch_gain: the "-damped" sets use ~1/sqrt(2 n_layer), the depth scaling trained residual networks are initialised with)
        const float bgain = hp.branch_gain > 0.0f ? hp.branch_gain : 0.25f;
        const float gain = name == "token_embd.weight" ? sqrtf((float) hp.n_embd) : (branch_out ? bgain : 1.0f);  // embeddings ~N(0,1)
        plan.push_back({name, type, ne0, ne1, Kg, row_off, k_off, id++, rowpar, lo, hi, gain});
    };
    add("token_embd.weight", pick_type(hp, "token_embd", 0), E, hp.n_vocab, E, 0, 0, false);
    for (int il = 0; il < hp.n_layer; ++il) {
        const std::string p = "blk." + std::to_string(il) + ".";
        add(p + "attn_norm.weight", GGML_TYPE_F32, E, 1, E, 0, 0, false, 0.5f, 1.5f);
        add(p + "attn_q.weight", pick_type(hp, "attn_q", il), E, nq_l, E, tp_rank * nq_l, 0, false);
        add(p + "attn_k.weight", pick_type(hp, "attn_k", il), E, nkv_l, E, tp_rank * nkv_l, 0, false);
        add(p + "attn_v.weight", pick_type(hp, "attn_v", il), E, nkv_l, E, tp_rank * nkv_l, 0, false);
        if (hp.qkv_bias) {
            add(p + "attn_q.bias", GGML_TYPE_F32, nq_l, 1, nq, 0, tp_rank * nq_l, false, -0.1f, 0.1f);
            add(p + "attn_k.bias", GGML_TYPE_F32, nkv_l, 1, nkv, 0, tp_rank * nkv_l, false, -0.1f, 0.1f);
            add(p + "attn_v.bias", GGML_TYPE_F32, nkv_l, 1, nkv, 0, tp_rank * nkv_l, false, -0.1f, 0.1f);
        }
        add(p + "attn_output.weight", pick_type(hp, "attn_output", il), nq_l, E, nq, 0, tp_rank * nq_l, tp_size > 1);
        add(p + "ffn_norm.weight", GGML_TYPE_F32, E, 1, E, 0, 0, false, 0.5f, 1.5f);
        add(p + "ffn_gate.weight", pick_type(hp, "ffn_gate", il), E, ff_l, E, tp_rank * ff_l, 0, false);
        add(p + "ffn_up.weight", pick_type(hp, "ffn_up", il), E, ff_l, E, tp_rank * ff_l, 0, false);
        add(p + "ffn_down.weight", pick_type(hp, "ffn_down", il), ff_l, E, hp.n_ff, 0, tp_rank * ff_l, tp_size > 1);
    }
    add("output_norm.weight", GGML_TYPE_F32, E, 1, E, 0, 0, false, 0.5f, 1.5f);
    add("output.weight", pick_type(hp, "output", 0), E, v_l, E, tp_rank * v_l, 0, false);
    if (hp.peaked > 0) {
        // the tied blocks are token_embd's: byte for byte (up to the gain) when the two tensors share a format, else its values re-encoded as Q6_K
        LLM_ASSERT(ggml_abi_blck_size(plan.back().type) > 1 && (plan.back().type == plan.front().type || (plan.back().type == GGML_TYPE_Q6_K && ggml_abi_blck_size(plan.front().type) == 256)));
        plan.back().tie_type = plan.front().type;
        plan.back().tie_wscale = plan.front().wscale;
        plan.back().tie_id = plan.front().tensor_id;
        plan.back().tie_eighths = std::min(8, hp.peaked);
        plan.back().tie_rows = hp.n_vocab;
    }
    if (split_mm)  // -sm row: llama.cpp allocates the layers' mat-mul weights and the output matrix in the split buffer type, everything else in the main GPU's
        for (auto & t : plan) {
            const bool mm = t.name.find("attn_q.weight") != std::string::npos || t.name.find("attn_k.weight") != std::string::npos || t.name.find("attn_v.weight") != std::string::npos ||
                            t.name.find("attn_output.weight") != std::string::npos || t.name.find("ffn_") != std::string::npos || t.name == "output.weight";
            t.rowpar = mm && t.name.find("norm") == std::string::npos && ggml_abi_blck_size(t.type) > 1;
        }
    for (auto & t : plan) {
        LLM_ASSERT(t.ne0 % ggml_abi_blck_size(t.type) == 0 && t.k_off % ggml_abi_blck_size(t.type) == 0);
    }
    return plan;
}

static synth_spec spec_of(const tensor_plan & t, uint64_t seed) {
    const int64_t blck = ggml_abi_blck_size(t.type);
    synth_spec sp{};
    sp.seed = seed;
    sp.tensor_id = t.tensor_id;
    sp.type = t.type;
    sp.K_global = t.K_global;
    sp.row_off = t.row_off;
    sp.blk_off = t.k_off / blck;
    sp.blocks_per_row = t.ne0 / blck;
    sp.blocks_per_row_global = t.K_global / blck;
    sp.n_rows = t.ne1;
    sp.f32_lo = t.lo;
    sp.f32_hi = t.hi;
    sp.wscale = t.wscale;
    sp.tie_id = t.tie_id;
    sp.tie_eighths = t.tie_eighths;
    sp.tie_rows = t.tie_rows;
    sp.tie_type = t.tie_id >= 0 ? t.tie_type : t.type;
    sp.tie_wscale = t.tie_wscale;
    sp.tie_gain = t.tie_id >= 0 ? t.wscale / t.tie_wscale : 1.0f;
    return sp;
}

static void bind_tensors(llm_model * m) {
    const llm_hparams & hp = m->hp;
    m->tok_embd = ggml_get_tensor(m->ctx, "token_embd.weight");
    m->output_norm = ggml_get_tensor(m->ctx, "output_norm.weight");
    m->output = ggml_get_tensor(m->ctx, "output.weight");
    if (!m->output) m->output = m->tok_embd;  // tied embeddings
    LLM_ASSERT(m->tok_embd && m->output_norm);
    m->layers.resize(hp.n_layer);
    for (int il = 0; il < hp.n_layer; ++il) {
        const std::string p = "blk." + std::to_string(il) + ".";
        auto g = [&](const char * s) { return ggml_get_tensor(m->ctx, (p + s).c_str()); };
        llm_layer & L = m->layers[il];
        L.attn_norm = g("attn_norm.weight"); L.wq = g("attn_q.weight"); L.wk = g("attn_k.weight"); L.wv = g("attn_v.weight");
        L.wo = g("attn_output.weight"); L.bq = g("attn_q.bias"); L.bk = g("attn_k.bias"); L.bv = g("attn_v.bias");
        L.ffn_norm = g("ffn_norm.weight"); L.ffn_gate = g("ffn_gate.weight"); L.ffn_up = g("ffn_up.weight"); L.ffn_down = g("ffn_down.weight");
        LLM_ASSERT(L.attn_norm && L.wq && L.wk && L.wv && L.wo && L.ffn_norm && L.ffn_gate && L.ffn_up && L.ffn_down);
    }
    m->n_head_l = hp.n_head / m->tp_size;
    m->n_head_kv_l = hp.n_head_kv / m->tp_size;
    m->n_ff_l = hp.n_ff / m->tp_size;
    m->n_vocab_l = (int) m->output->ne[1];
    m->stream_bytes = 0;
    m->total_bytes = 0;
    for (ggml_tensor * t = ggml_get_first_tensor(m->ctx); t; t = ggml_get_next_tensor(m->ctx, t)) {
        m->total_bytes += ggml_nbytes(t);
        if (t == m->tok_embd && m->output != m->tok_embd) m->stream_bytes += ggml_row_size(t->type, t->ne[0]);
        else m->stream_bytes += ggml_nbytes(t);
    }
}

static llm_model * model_synth_impl(const struct llm_hparams * hp, uint64_t seed, ggml_backend_buffer_type_t buft, int tp_rank, int tp_size,
                                    ggml_backend_buffer_type_t rowpar_buft, bool split_mm, const std::vector<ggml_backend_buffer_type_t> * layer_bufts = nullptr);
extern "C" struct llm_model * llm_model_synth(const struct llm_hparams * hp, uint64_t seed, ggml_backend_buffer_type_t buft, int tp_rank, int tp_size,
                                              ggml_backend_buffer_type_t rowpar_buft) {
    return model_synth_impl(hp, seed, buft, tp_rank, tp_size, rowpar_buft, false);
}
// -sm row (llama-box/engine_param.hpp:902-916): one process, every mat-mul weight in `split_buft` — what the backend registry's
// "ggml_backend_split_buffer_type" proc address returned for (main_gpu, tensor_split) — and the rest in the main device's buffer type
extern "C" struct llm_model * llm_model_synth_split(const struct llm_hparams * hp, uint64_t seed, ggml_backend_buffer_type_t buft, ggml_backend_buffer_type_t split_buft) {
    return model_synth_impl(hp, seed, buft, 0, 1, split_buft, true);
}
// -sm layer (llama.cpp's default with several devices; llama-box/engine_param.hpp:900-916): device d holds a contiguous range of layers in its own
// buffer type, token_embd goes with layer 0, output_norm / output with the last layer
static int layer_split_device(int il, int n_layer, int n_dev) { return std::min(n_dev - 1, il * n_dev / std::max(1, n_layer)); }
extern "C" struct llm_model * llm_model_synth_layer_split(const struct llm_hparams * hp, uint64_t seed, const ggml_backend_buffer_type_t * bufts, int n_dev) {
    if (n_dev < 1 || n_dev > hp->n_layer) return nullptr;
    std::vector<ggml_backend_buffer_type_t> v(bufts, bufts + n_dev);
    return model_synth_impl(hp, seed, bufts[0], 0, 1, nullptr, false, &v);
}
static llm_model * model_synth_impl(const struct llm_hparams * hp, uint64_t seed, ggml_backend_buffer_type_t buft, int tp_rank, int tp_size,
                                    ggml_backend_buffer_type_t rowpar_buft, bool split_mm, const std::vector<ggml_backend_buffer_type_t> * layer_bufts) {
    llm_model * m = new llm_model();
    m->hp = *hp;
    m->tp_rank = tp_rank;
    m->tp_size = tp_size;
    m->ctx = ggml_init({0, nullptr, true});
    std::vector<tensor_plan> plan = make_plan(*hp, tp_rank, tp_size, split_mm);
    std::vector<ggml_tensor *> ts;
    for (auto & t : plan) {
        ggml_tensor * x = ggml_new_tensor_2d(m->ctx, t.type, t.ne0, t.ne1);
        ggml_set_name(x, t.name.c_str());
        ts.push_back(x);
    }
    // allocation groups: [0] ordinary weights, [1] row-parallel weights (the backend's reducing / split buffer type), or one group per device of a layer split
    std::vector<ggml_backend_buffer_type_t> group_buft;
    std::vector<int> group_of(plan.size(), 0);
    if (layer_bufts) {
        group_buft = *layer_bufts;
        m->n_dev = (int) layer_bufts->size();
        m->layer_dev.resize((size_t) hp->n_layer);
        for (int il = 0; il < hp->n_layer; ++il) m->layer_dev[(size_t) il] = layer_split_device(il, hp->n_layer, m->n_dev);
        for (size_t i = 0; i < plan.size(); ++i) {
            int il = -1;
            if (sscanf(plan[i].name.c_str(), "blk.%d.", &il) == 1) group_of[i] = m->layer_dev[(size_t) il];
            else group_of[i] = plan[i].name == "token_embd.weight" ? 0 : m->n_dev - 1;
        }
    } else {
        group_buft = {buft, rowpar_buft};
        for (size_t i = 0; i < plan.size(); ++i) group_of[i] = (plan[i].rowpar && rowpar_buft) ? 1 : 0;
    }
    {
        std::vector<size_t> total(group_buft.size(), 0), off(group_buft.size(), 0);
        std::vector<ggml_backend_buffer_t> bufs(group_buft.size(), nullptr);
        for (size_t i = 0; i < plan.size(); ++i) {
            const int g = group_of[i];
            const size_t align = ggml_backend_buft_get_alignment(group_buft[(size_t) g]);
            total[(size_t) g] = (total[(size_t) g] + align - 1) / align * align + ggml_backend_buft_get_alloc_size(group_buft[(size_t) g], ts[i]);
        }
        for (size_t g = 0; g < group_buft.size(); ++g) {
            if (!total[g]) continue;
            bufs[g] = ggml_backend_buft_alloc_buffer(group_buft[g], total[g] + ggml_backend_buft_get_alignment(group_buft[g]));
            if (!bufs[g]) {
                fprintf(stderr, "llm_model_synth: failed to allocate %zu bytes of weights\n", total[g]);
                llm_model_free(m);
                return nullptr;
            }
            ggml_backend_buffer_set_usage(bufs[g], GGML_BACKEND_BUFFER_USAGE_WEIGHTS);
            m->buffers.push_back(bufs[g]);
        }
        for (size_t i = 0; i < plan.size(); ++i) {
            const size_t g = (size_t) group_of[i];
            const size_t align = ggml_backend_buft_get_alignment(group_buft[g]);
            ggml_backend_buffer_t bb = bufs[g];
            off[g] = (off[g] + align - 1) / align * align;
            ts[i]->data = (char *) ggml_backend_buffer_get_base(bb) + off[g];
            ts[i]->buffer = bb;
            if (bb->iface.init_tensor && bb->iface.init_tensor(bb, ts[i]) != GGML_STATUS_SUCCESS) {
                fprintf(stderr, "llm_model_synth: init_tensor failed for %s\n", plan[i].name.c_str());
                llm_model_free(m);
                return nullptr;
            }
            off[g] += ggml_backend_buft_get_alloc_size(group_buft[g], ts[i]);
        }
    }
    // generate + upload in chunks of rows, generation multi-threaded (each block is independent)
    const unsigned nthr = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    std::vector<uint8_t> stage;
    for (size_t i = 0; i < plan.size(); ++i) {
        const synth_spec sp = spec_of(plan[i], seed);
        const size_t row_bytes = ggml_row_size(plan[i].type, plan[i].ne0);
        // (a split buffer takes whole tensors only, like upstream's: the loader writes a tensor in one call)
        const int64_t rows_per_chunk = (split_mm && plan[i].rowpar) ? plan[i].ne1 : std::max<int64_t>(1, (int64_t) ((64u << 20) / row_bytes));
        for (int64_t r0 = 0; r0 < plan[i].ne1; r0 += rows_per_chunk) {
            const int64_t r1 = std::min<int64_t>(plan[i].ne1, r0 + rows_per_chunk);
            stage.resize((size_t) (r1 - r0) * row_bytes);
            std::vector<std::thread> th;
            const int64_t per = (r1 - r0 + nthr - 1) / nthr;
            for (unsigned t = 0; t < nthr; ++t) {
                const int64_t a = r0 + t * per, b = std::min(r1, a + per);
                if (a >= b) break;
                th.emplace_back([&, a, b]() { synth_rows(sp, a, b, stage.data() + (size_t) (a - r0) * row_bytes); });
            }
            for (auto & x : th) x.join();
            ggml_backend_tensor_set(ts[i], stage.data(), (size_t) r0 * row_bytes, stage.size());
        }
    }
    bind_tensors(m);
    return m;
}

struct gguf_fill_ctx {
    const std::vector<tensor_plan> * plan;
    uint64_t seed;
};
static void gguf_fill(const gguf_tensor_info & ti, void * dst, void * user) {
    gguf_fill_ctx * c = (gguf_fill_ctx *) user;
    for (auto & t : *c->plan) {
        if (t.name == ti.name) {
            synth_rows(spec_of(t, c->seed), 0, t.ne1, (uint8_t *) dst);
            return;
        }
    }
    LLM_ASSERT(!"gguf_fill: unknown tensor");
}

extern "C" int llm_synth_gguf(const struct llm_hparams * hp, uint64_t seed, const char * path) {
    std::vector<tensor_plan> plan = make_plan(*hp, 0, 1);
    gguf_writer w;
    const std::string a = hp->arch;
    w.set_str("general.architecture", a);
    w.set_str("general.name", "synthetic-" + a);
    w.set_u32("general.alignment", 32);
    w.set_u32("general.file_type", (uint32_t) hp->ftype);
    w.set_u32(a + ".block_count", (uint32_t) hp->n_layer);
    w.set_u32(a + ".context_length", (uint32_t) hp->n_ctx_train);
    w.set_u32(a + ".embedding_length", (uint32_t) hp->n_embd);
    w.set_u32(a + ".feed_forward_length", (uint32_t) hp->n_ff);
    w.set_u32(a + ".attention.head_count", (uint32_t) hp->n_head);
    w.set_u32(a + ".attention.head_count_kv", (uint32_t) hp->n_head_kv);
    w.set_f32(a + ".attention.layer_norm_rms_epsilon", hp->rms_eps);
    w.set_f32(a + ".rope.freq_base", hp->rope_freq_base);
    w.set_u32(a + ".rope.dimension_count", (uint32_t) hp->n_embd_head);
    w.set_u32(a + ".vocab_size", (uint32_t) hp->n_vocab);
    w.set_str("tokenizer.ggml.model", "no_vocab");
    for (auto & t : plan) {
        const int64_t ne[2] = {t.ne0, t.ne1};
        w.add_tensor(t.name, t.type, t.ne1 == 1 ? 1 : 2, ne);
    }
    gguf_fill_ctx fc{&plan, seed};
    return w.write(path, gguf_fill, &fc) ? 0 : -1;
}

extern "C" struct llm_model * llm_model_load(const char * path, ggml_backend_buffer_type_t buft) {
    gguf_file * f = gguf_open(path);
    if (!f) return nullptr;
    llm_model * m = new llm_model();
    llm_hparams & hp = m->hp;
    const std::string a = f->get_s("general.architecture", "llama");
    snprintf(hp.arch, sizeof(hp.arch), "%s", a.c_str());
    hp.n_layer = (int) f->get_u(a + ".block_count");
    hp.n_ctx_train = (int) f->get_u(a + ".context_length");
    hp.n_embd = (int) f->get_u(a + ".embedding_length");
    hp.n_ff = (int) f->get_u(a + ".feed_forward_length");
    hp.n_head = (int) f->get_u(a + ".attention.head_count");
    hp.n_head_kv = (int) f->get_u(a + ".attention.head_count_kv", hp.n_head);
    hp.rms_eps = (float) f->get_f(a + ".attention.layer_norm_rms_epsilon", 1e-5);
    hp.rope_freq_base = (float) f->get_f(a + ".rope.freq_base", 10000.0);
    hp.n_embd_head = (int) f->get_u(a + ".rope.dimension_count", hp.n_head ? hp.n_embd / hp.n_head : 0);
    hp.ftype = (int) f->get_u("general.file_type");
    hp.rope_type = a == "qwen2" ? GGML_ROPE_TYPE_NEOX : 0;
    const gguf_tensor_info * te = f->find("token_embd.weight");
    if (!te || hp.n_layer <= 0 || hp.n_embd <= 0 || hp.n_head <= 0) {
        fprintf(stderr, "llm_model_load: %s: missing hyper-parameters or token_embd.weight\n", path);
        delete f;
        delete m;
        return nullptr;
    }
    hp.n_vocab = (int) f->get_u(a + ".vocab_size", (uint64_t) te->ne[1]);
    hp.qkv_bias = f->find("blk.0.attn_q.bias") != nullptr;
    m->ctx = ggml_init({0, nullptr, true});
    for (auto & ti : f->tensors) {
        ggml_tensor * t = ggml_new_tensor(m->ctx, ti.type, ti.n_dims, ti.ne);
        ggml_set_name(t, ti.name.c_str());
    }
    ggml_backend_buffer_t buf = ggml_backend_alloc_ctx_tensors_from_buft(m->ctx, buft);
    if (!buf) {
        fprintf(stderr, "llm_model_load: failed to allocate weights\n");
        delete f;
        llm_model_free(m);
        return nullptr;
    }
    ggml_backend_buffer_set_usage(buf, GGML_BACKEND_BUFFER_USAGE_WEIGHTS);
    m->buffers.push_back(buf);
    for (auto & ti : f->tensors) ggml_backend_tensor_set(ggml_get_tensor(m->ctx, ti.name.c_str()), f->tensor_data(ti), 0, ti.size);
    delete f;
    bind_tensors(m);
    return m;
}

extern "C" void llm_model_free(struct llm_model * m) {
    if (!m) return;
    for (auto b : m->buffers) ggml_backend_buffer_free(b);
    ggml_free(m->ctx);
    delete m;
}
extern "C" const struct llm_hparams * llm_model_hparams(const struct llm_model * m) { return &m->hp; }
extern "C" uint64_t llm_model_stream_bytes(const struct llm_model * m) { return m->stream_bytes; }
extern "C" uint64_t llm_model_total_bytes(const struct llm_model * m) { return m->total_bytes; }
extern "C" struct ggml_tensor * llm_model_tensor(struct llm_model * m, const char * name) { return ggml_get_tensor(m->ctx, name); }

// ---------------------------------------------------------------------------------------------- context
// one row of the attention mask: `vis` where the cell holds a position of sequence s that is not after p, `hid` elsewhere (written to
// vectorise — no short-circuit — and compiled at -O3: the library is built at -O2, where this compiler does not try)
#pragma GCC push_options
#pragma GCC optimize("O3")
static void mask_row(float * __restrict__ row, const int n_kv, const int32_t * __restrict__ cpos, const int32_t * __restrict__ cseq, const int32_t s, const int32_t p,
                     const float vis, const float hid) {
    const uint32_t pu = (uint32_t) p;
    for (int j = 0; j < n_kv; ++j) {
        const bool v = (cseq[j] == s) & ((uint32_t) cpos[j] <= pu);
        row[j] = v ? vis : hid;
    }
}
#pragma GCC pop_options

struct kv_cell {
    int32_t pos = -1;
    int32_t seq = -1;
};

// the inputs a graph segment reads and the tensor it ends in.  One segment = the whole model on one backend; with -sm layer one segment per device,
// each with its own copies of the inputs and, from the second on, the residual stream of the previous device as an input (`resid`)
struct seg_io {
    ggml_tensor *tokens = nullptr, *pos = nullptr, *mask = nullptr, *k_idxs = nullptr, *v_idxs = nullptr, *out_ids = nullptr, *resid = nullptr;
    ggml_tensor * out = nullptr;  // l_out of the segment's last layer, or the logits
};

struct graph_key {
    int n_tokens = -1, n_kv = -1, n_outputs = -1;
    bool operator==(const graph_key & o) const { return n_tokens == o.n_tokens && n_kv == o.n_kv && n_outputs == o.n_outputs; }
};

struct llm_context {
    llm_model * model = nullptr;
    ggml_backend_t backend = nullptr;
    llm_compute_fn compute = nullptr;
    llm_context_params p{};
    ggml_backend_buffer_type_t buft = nullptr;
    // KV cache
    ggml_context * ctx_kv = nullptr;
    ggml_backend_buffer_t buf_kv = nullptr;
    std::vector<ggml_tensor *> k_l, v_l;
    std::vector<kv_cell> cells;
    std::vector<int32_t> mask_cpos, mask_cseq;  // the cells' fields as arrays (mask rows are filled from them)
    int kv_head = 0;
    int used_max = 0;           // one past the highest occupied cell; < 0: unknown (cells were freed), recounted by the next batch
    // compute graph
    ggml_gallocr_t galloc = nullptr;
    ggml_context * ctx_compute = nullptr;
    ggml_cgraph * gf = nullptr;
    graph_key key;
    ggml_tensor *inp_tokens = nullptr, *inp_pos = nullptr, *inp_mask = nullptr, *inp_k_idxs = nullptr, *inp_v_idxs = nullptr, *inp_out_ids = nullptr;
    ggml_tensor * t_logits = nullptr;
    // outputs
    std::vector<float> logits;
    float * logits_base = nullptr;  // where the last llm_decode's rows are: the pinned output area, or `logits`
    std::vector<int32_t> output_ids;  // batch position -> output row of the last llm_decode (-1: no logits requested there)
    int n_outputs = 0;
    double timings[4] = {0, 0, 0, 0};
    // pinned host staging, as llama.cpp does it (inputs are uploaded with tensor_set_async out of host-buffer-type memory,
    // logits come back with tensor_get_async into the pinned output buffer; ONE synchronize per micro-batch):
    // [ inputs of one micro-batch | logits of up to pin_out_rows outputs ]
    ggml_backend_buffer_t buf_pin = nullptr;
    char * pin = nullptr;
    size_t pin_in_bytes = 0;   // per input slot
    int pin_out_rows = 0;
    // micro-batches without outputs are not waited for (as llama.cpp: the wait happens when results are fetched); each
    // in-flight micro-batch keeps its own input slot, so the host builds and stages batch i+1 while the GPU runs batch i
    static constexpr int PIN_SLOTS = 4;
    int pin_slot = 0, in_flight = 0;
    // -sm layer (llm_context_new_layer_split): one graph segment per device, driven the way ggml_backend_sched drives the splits of a
    // layer-split model — input copies in LS_COPIES slots per device, cpy_tensor_async of the residual stream between neighbouring
    // devices, one event per (device, slot) recorded after the segment's graph and waited for before the slot is overwritten
    static constexpr int LS_COPIES = 2;
    struct ls_dev {
        ggml_backend_t be = nullptr;
        ggml_backend_buffer_type_t buft = nullptr;
        ggml_gallocr_t galloc = nullptr;
        int l0 = 0, l1 = 0;
        ggml_context * ctx_kv = nullptr;
        ggml_backend_buffer_t buf_kv = nullptr;
        ggml_context * ctx_in = nullptr;       // the input copies of all slots (+ nothing else): allocated once per graph key, outside the graph allocator
        ggml_backend_buffer_t buf_in = nullptr;
        ggml_context * ctx[LS_COPIES] = {nullptr, nullptr};
        ggml_cgraph * gf[LS_COPIES] = {nullptr, nullptr};
        seg_io io[LS_COPIES];
        ggml_backend_event_t ev[LS_COPIES] = {nullptr, nullptr};
    };
    std::vector<ls_dev> ls;
    int ls_copy = 0;
    int64_t ls_stats[4] = {0, 0, 0, 0};  // cpy_tensor_async calls between devices, blocking input copies, events recorded, events waited for
};

static ggml_tensor * named(ggml_tensor * t, const char * base, int il) {
    char buf[96];
    if (il >= 0) snprintf(buf, sizeof(buf), "%s-%d", base, il);
    else snprintf(buf, sizeof(buf), "%s", base);
    return ggml_set_name(t, buf);
}

// creates the input tensors of layers [l0, l1) (+ the output head if `head`) in `ctx`
static void make_inputs(llm_context * c, ggml_context * ctx, int l0, int l1, bool head, int n_tokens, int n_kv, int n_outputs, seg_io & io) {
    llm_model * m = c->model;
    const llm_hparams & hp = m->hp;
    const bool fa = c->p.flash_attn != 0;
    const int64_t n_embd_k = (int64_t) m->n_head_kv_l * hp.n_embd_head;
    if (l0 == 0) {
        io.tokens = ggml_new_tensor_1d(ctx, GGML_TYPE_I32, n_tokens);
        ggml_set_input(named(io.tokens, "inp_tokens", -1));
    } else {
        io.resid = ggml_new_tensor_2d(ctx, GGML_TYPE_F32, hp.n_embd, n_tokens);
        ggml_set_input(named(io.resid, "l_out_from_the_previous_device", -1));
    }
    io.pos = ggml_new_tensor_1d(ctx, GGML_TYPE_I32, n_tokens);
    ggml_set_input(named(io.pos, "inp_pos", -1));
    const int64_t n_tok_pad = (n_tokens + 63) / 64 * 64;  // GGML_KQ_MASK_PAD
    // llama.cpp's build_attn_inp_kv_unified: the mask is an F32 input filled on the host; with flash attention FLASH_ATTN_EXT is handed a
    // ggml_cast(F16) of it (self_kq_mask_cnv) — a CPY node in the graph, not an F16 upload
    io.mask = ggml_new_tensor_2d(ctx, GGML_TYPE_F32, n_kv, n_tok_pad);
    ggml_set_input(named(io.mask, "KQ_mask", -1));
    io.k_idxs = ggml_new_tensor_1d(ctx, GGML_TYPE_I64, n_tokens);
    ggml_set_input(named(io.k_idxs, "k_idxs", -1));
    if (!fa) {
        io.v_idxs = ggml_new_tensor_1d(ctx, GGML_TYPE_I64, (int64_t) n_tokens * n_embd_k);
        ggml_set_input(named(io.v_idxs, "v_idxs", -1));
    }
    if (l1 == hp.n_layer && n_outputs < n_tokens) {
        io.out_ids = ggml_new_tensor_1d(ctx, GGML_TYPE_I32, n_outputs);
        ggml_set_input(named(io.out_ids, "out_ids", -1));
    }
    (void) head;
}

// layers [l0, l1) (+ norm and output matrix if `head`) over the inputs `io`, appended to `gf`
static void build_segment(llm_context * c, ggml_context * ctx, ggml_cgraph * gf, int l0, int l1, bool head, int n_tokens, int n_kv, seg_io & io) {
    llm_model * m = c->model;
    const llm_hparams & hp = m->hp;
    const int64_t HD = hp.n_embd_head, NH = m->n_head_l, NKV = m->n_head_kv_l;
    const int64_t n_embd_k = NKV * HD, n_ctx = c->p.n_ctx;
    const bool fa = c->p.flash_attn != 0;
    const float kq_scale = 1.0f / sqrtf((float) HD);
    ggml_tensor * const mask_cnv = fa ? named(ggml_cast(ctx, io.mask, GGML_TYPE_F16), "KQ_mask_cnv", -1) : io.mask;

    ggml_tensor * inpL = l0 == 0 ? named(ggml_get_rows(ctx, m->tok_embd, io.tokens), "inp_embd", -1) : io.resid;
    for (int il = l0; il < l1; ++il) {
        const llm_layer & L = m->layers[il];
        ggml_tensor * inpSA = inpL;
        ggml_tensor * cur = named(ggml_rms_norm(ctx, inpL, hp.rms_eps), "norm", il);
        cur = named(ggml_mul(ctx, cur, L.attn_norm), "attn_norm", il);
        ggml_tensor * Qcur = named(ggml_mul_mat(ctx, L.wq, cur), "Qcur", il);
        if (L.bq) Qcur = named(ggml_add(ctx, Qcur, L.bq), "Qcur_b", il);
        ggml_tensor * Kcur = named(ggml_mul_mat(ctx, L.wk, cur), "Kcur", il);
        if (L.bk) Kcur = named(ggml_add(ctx, Kcur, L.bk), "Kcur_b", il);
        ggml_tensor * Vcur = named(ggml_mul_mat(ctx, L.wv, cur), "Vcur", il);
        if (L.bv) Vcur = named(ggml_add(ctx, Vcur, L.bv), "Vcur_b", il);
        Qcur = ggml_reshape_3d(ctx, Qcur, HD, NH, n_tokens);
        Kcur = ggml_reshape_3d(ctx, Kcur, HD, NKV, n_tokens);
        Vcur = ggml_reshape_3d(ctx, Vcur, HD, NKV, n_tokens);
        Qcur = named(ggml_rope_ext(ctx, Qcur, io.pos, nullptr, (int) HD, hp.rope_type, hp.n_ctx_train, hp.rope_freq_base, 1.0f, 0.0f, 1.0f, 32.0f, 1.0f), "Qcur_rope", il);
        Kcur = named(ggml_rope_ext(ctx, Kcur, io.pos, nullptr, (int) HD, hp.rope_type, hp.n_ctx_train, hp.rope_freq_base, 1.0f, 0.0f, 1.0f, 32.0f, 1.0f), "Kcur_rope", il);
        // llama.cpp's build_attn adds q, k and v to the graph together before the cache stores ("so that they are not reordered")
        ggml_build_forward_expand(gf, Qcur);
        ggml_build_forward_expand(gf, Kcur);
        ggml_build_forward_expand(gf, Vcur);
        // store K/V into the cache at the slots chosen for this micro-batch
        ggml_tensor * k_cache = c->k_l[il];
        ggml_tensor * v_cache = c->v_l[il];
        ggml_build_forward_expand(gf, named(ggml_set_rows(ctx, k_cache, ggml_reshape_2d(ctx, Kcur, n_embd_k, n_tokens), io.k_idxs), "k_store", il));
        if (fa) {
            ggml_build_forward_expand(gf, named(ggml_set_rows(ctx, v_cache, ggml_reshape_2d(ctx, Vcur, n_embd_k, n_tokens), io.k_idxs), "v_store", il));
        } else {
            // transposed V cache: every element is its own row of length 1
            ggml_tensor * v_view = ggml_reshape_2d(ctx, v_cache, 1, n_ctx * n_embd_k);
            ggml_tensor * v_src = ggml_reshape_2d(ctx, Vcur, 1, (int64_t) n_tokens * n_embd_k);
            ggml_build_forward_expand(gf, named(ggml_set_rows(ctx, v_view, v_src, io.v_idxs), "v_store", il));
        }
        ggml_tensor * q = ggml_permute(ctx, Qcur, 0, 2, 1, 3);  // [HD, n_tokens, NH]
        ggml_tensor * k = ggml_view_3d(ctx, k_cache, HD, n_kv, NKV, ggml_row_size(k_cache->type, n_embd_k), ggml_row_size(k_cache->type, HD), 0);
        if (fa) {
            ggml_tensor * v = ggml_view_3d(ctx, v_cache, HD, n_kv, NKV, ggml_row_size(v_cache->type, n_embd_k), ggml_row_size(v_cache->type, HD), 0);
            cur = ggml_flash_attn_ext(ctx, q, k, v, mask_cnv, kq_scale, 0.0f, 0.0f);
            ggml_flash_attn_ext_set_prec(cur, GGML_PREC_F32);
            named(cur, "fattn", il);
            cur = ggml_reshape_2d(ctx, cur, HD * NH, n_tokens);
        } else {
            ggml_tensor * v = ggml_view_3d(ctx, v_cache, n_kv, HD, NKV, ggml_row_size(v_cache->type, n_ctx), ggml_row_size(v_cache->type, n_ctx) * HD, 0);
            ggml_tensor * kq = named(ggml_mul_mat(ctx, k, q), "kq", il);  // [n_kv, n_tokens, NH]
            ggml_mul_mat_set_prec(kq, GGML_PREC_F32);
            kq = named(ggml_soft_max_ext(ctx, kq, mask_cnv, kq_scale, 0.0f), "kq_soft_max", il);
            ggml_tensor * kqv = named(ggml_mul_mat(ctx, v, kq), "kqv", il);  // [HD, n_tokens, NH]
            cur = ggml_permute(ctx, kqv, 0, 2, 1, 3);
            cur = named(ggml_cont_2d(ctx, cur, HD * NH, n_tokens), "kqv_out", il);
        }
        cur = named(ggml_mul_mat(ctx, L.wo, cur), "attn_out", il);
        if (il == hp.n_layer - 1 && io.out_ids) {
            cur = ggml_get_rows(ctx, cur, io.out_ids);
            inpSA = ggml_get_rows(ctx, inpSA, io.out_ids);
        }
        ggml_tensor * ffn_inp = named(ggml_add(ctx, cur, inpSA), "ffn_inp", il);
        cur = named(ggml_rms_norm(ctx, ffn_inp, hp.rms_eps), "norm_ffn", il);
        cur = named(ggml_mul(ctx, cur, L.ffn_norm), "ffn_norm", il);
        ggml_tensor * gate = named(ggml_mul_mat(ctx, L.ffn_gate, cur), "ffn_gate", il);
        ggml_tensor * up = named(ggml_mul_mat(ctx, L.ffn_up, cur), "ffn_up", il);
        cur = named(ggml_swiglu_split(ctx, gate, up), "ffn_swiglu", il);
        cur = named(ggml_mul_mat(ctx, L.ffn_down, cur), "ffn_out", il);
        cur = named(ggml_add(ctx, cur, ffn_inp), "l_out", il);
        inpL = cur;
    }
    ggml_tensor * cur = inpL;
    if (head) {
        cur = named(ggml_rms_norm(ctx, inpL, hp.rms_eps), "norm_final", -1);
        cur = named(ggml_mul(ctx, cur, m->output_norm), "result_norm", -1);
        cur = named(ggml_mul_mat(ctx, m->output, cur), "result_output", -1);
    }
    ggml_set_output(cur);
    ggml_build_forward_expand(gf, cur);
    io.out = cur;
}

static void build_graph(llm_context * c, int n_tokens, int n_kv, int n_outputs) {
    if (c->ctx_compute) ggml_free(c->ctx_compute);
    c->ctx_compute = ggml_init({0, nullptr, true});
    ggml_cgraph * gf = ggml_new_graph_custom(c->ctx_compute, 8192, false);
    seg_io io;
    make_inputs(c, c->ctx_compute, 0, c->model->hp.n_layer, true, n_tokens, n_kv, n_outputs, io);
    build_segment(c, c->ctx_compute, gf, 0, c->model->hp.n_layer, true, n_tokens, n_kv, io);
    c->inp_tokens = io.tokens;
    c->inp_pos = io.pos;
    c->inp_mask = io.mask;
    c->inp_k_idxs = io.k_idxs;
    c->inp_v_idxs = io.v_idxs;
    c->inp_out_ids = io.out_ids;
    c->t_logits = io.out;
    c->gf = gf;
}

extern "C" struct llm_context * llm_context_new(struct llm_model * m, ggml_backend_t backend, llm_compute_fn compute, const struct llm_context_params * p) {
    if ((backend == nullptr) == (compute == nullptr)) {
        fprintf(stderr, "llm_context_new: exactly one of backend / compute must be given\n");
        return nullptr;
    }
    llm_context * c = new llm_context();
    c->model = m;
    c->backend = backend;
    c->compute = compute;
    c->p = *p;
    if (c->p.n_ubatch <= 0) c->p.n_ubatch = 512;
    if (c->p.n_ctx <= 0) c->p.n_ctx = 512;
    c->p.n_ctx = (c->p.n_ctx + 255) / 256 * 256;
    c->buft = backend ? ggml_backend_dev_buffer_type(backend->device) : ggml_backend_cpu_buffer_type();
    const llm_hparams & hp = m->hp;
    const int64_t n_embd_k = (int64_t) m->n_head_kv_l * hp.n_embd_head;
    // -ctk / -ctv (llama-box/engine_param.hpp:51-54): f32, f16, bf16, q8_0, q4_0, q4_1, iq4_nl, q5_0, q5_1.  0 = the default (f16); f32 is ggml type 0
    // too, so it is asked for as LLM_KV_TYPE_F32 (-1)
    auto kv_type_of = [](int32_t t) { return t == 0 ? GGML_TYPE_F16 : t == LLM_KV_TYPE_F32 ? GGML_TYPE_F32 : (ggml_type) t; };
    auto kv_type_ok = [](ggml_type t) {
        return t == GGML_TYPE_F32 || t == GGML_TYPE_F16 || t == GGML_TYPE_BF16 || t == GGML_TYPE_Q8_0 || t == GGML_TYPE_Q4_0 || t == GGML_TYPE_Q4_1 || t == GGML_TYPE_IQ4_NL ||
               t == GGML_TYPE_Q5_0 || t == GGML_TYPE_Q5_1;
    };
    const ggml_type type_k = kv_type_of(c->p.type_k), type_v = kv_type_of(c->p.type_v);
    // (llama.cpp: a quantised V cache needs flash attention — the non-flash path keeps V transposed, one element per index; K may be of any type there:
    // K.q is then a MUL_MAT over the cache's blocks)
    if (!kv_type_ok(type_k) || !kv_type_ok(type_v) || (type_v != GGML_TYPE_F16 && !c->p.flash_attn) || (hp.n_embd_head % ggml_blck_size(type_k)) != 0 || (hp.n_embd_head % ggml_blck_size(type_v)) != 0) {
        fprintf(stderr, "llm_context_new: unsupported KV cache types %d/%d (f32, f16, bf16, q8_0, q4_0, q4_1, iq4_nl, q5_0, q5_1; a V cache other than f16 needs flash_attn)\n", (int) type_k, (int) type_v);
        delete c;
        return nullptr;
    }
    c->ctx_kv = ggml_init({0, nullptr, true});
    for (int il = 0; il < hp.n_layer; ++il) {
        ggml_tensor * k = ggml_new_tensor_2d(c->ctx_kv, type_k, n_embd_k, c->p.n_ctx);
        ggml_tensor * v = c->p.flash_attn ? ggml_new_tensor_2d(c->ctx_kv, type_v, n_embd_k, c->p.n_ctx)
                                          : ggml_new_tensor_2d(c->ctx_kv, GGML_TYPE_F16, c->p.n_ctx, n_embd_k);
        named(k, "cache_k_l", il);
        named(v, "cache_v_l", il);
        c->k_l.push_back(k);
        c->v_l.push_back(v);
    }
    c->buf_kv = ggml_backend_alloc_ctx_tensors_from_buft(c->ctx_kv, c->buft);
    if (!c->buf_kv) {
        fprintf(stderr, "llm_context_new: failed to allocate the KV cache\n");
        llm_context_free(c);
        return nullptr;
    }
    ggml_backend_buffer_clear(c->buf_kv, 0);
    c->cells.assign(c->p.n_ctx, kv_cell());
    c->used_max = 0;
    c->galloc = ggml_gallocr_new(c->buft);
    // reserve the worst-case graph (full micro-batch over the full cache) so later graphs re-use one buffer
    build_graph(c, std::min(c->p.n_ubatch, c->p.n_ctx), c->p.n_ctx, std::min(c->p.n_ubatch, c->p.n_ctx));
    if (!ggml_gallocr_reserve(c->galloc, c->gf)) {
        fprintf(stderr, "llm_context_new: failed to reserve the compute buffer\n");
        llm_context_free(c);
        return nullptr;
    }
    c->key = graph_key();
    if (backend) {
        ggml_backend_buffer_type_t hbuft = ggml_backend_dev_host_buffer_type(backend->device);
        if (hbuft) {
            const size_t nub = (size_t) std::min(c->p.n_ubatch, c->p.n_ctx);
            const size_t n_tok_pad = (nub + 63) / 64 * 64;
            size_t in_bytes = nub * (4 + 4 + 8 + 4) + n_tok_pad * (size_t) c->p.n_ctx * 4 + 256;
            if (!c->p.flash_attn) in_bytes += nub * (size_t) n_embd_k * 8;
            in_bytes = (in_bytes + 255) / 256 * 256;
            c->pin_out_rows = 64;
            c->buf_pin = ggml_backend_buft_alloc_buffer(hbuft, in_bytes * llm_context::PIN_SLOTS + (size_t) c->pin_out_rows * m->n_vocab_l * 4);
            if (c->buf_pin) {
                c->pin = (char *) ggml_backend_buffer_get_base(c->buf_pin);
                c->pin_in_bytes = in_bytes;
            }
        }
    }
    return c;
}

extern "C" void llm_context_free(struct llm_context * c) {
    if (!c) return;
    for (auto & d : c->ls) {
        if (d.be) ggml_backend_synchronize(d.be);
        for (int k = 0; k < llm_context::LS_COPIES; ++k) {
            if (d.ctx[k]) ggml_free(d.ctx[k]);
            if (d.ev[k]) ggml_backend_event_free(d.ev[k]);
        }
        if (d.galloc) ggml_gallocr_free(d.galloc);
        if (d.buf_in) ggml_backend_buffer_free(d.buf_in);
        if (d.ctx_in) ggml_free(d.ctx_in);
        if (d.buf_kv) ggml_backend_buffer_free(d.buf_kv);
        if (d.ctx_kv) ggml_free(d.ctx_kv);
    }
    if (c->ctx_compute) ggml_free(c->ctx_compute);
    if (c->galloc) ggml_gallocr_free(c->galloc);
    if (c->buf_pin) ggml_backend_buffer_free(c->buf_pin);
    if (c->buf_kv) ggml_backend_buffer_free(c->buf_kv);
    if (c->ctx_kv) ggml_free(c->ctx_kv);
    delete c;
}

extern "C" void llm_kv_clear(struct llm_context * c) {
    for (auto & x : c->cells) x = kv_cell();
    c->used_max = 0;
    c->kv_head = 0;
}
extern "C" int llm_kv_seq_rm(struct llm_context * c, int seq_id, int p0, int p1) {
    if (p0 < 0) p0 = 0;
    if (p1 < 0) p1 = INT32_MAX;
    for (auto & x : c->cells)
        if (x.pos >= p0 && x.pos < p1 && (seq_id < 0 || x.seq == seq_id)) { x = kv_cell(); c->used_max = -1; }
    c->kv_head = 0;
    return 1;
}

// K-shift: the graph llama.cpp's llama_kv_cache::build_rope_shift produces after llama_memory_seq_add (llama-box context
// shift, httpserver.hpp:3453-3537).  Every cell's K row is rotated by the cell's position delta: in place on an f16 cache;
// on a quantised cache through an f32 copy (cast -> rope -> cpy back, which re-quantises the rows).
static int kv_apply_shift(llm_context * c, const std::vector<int32_t> & delta) {
    llm_model * m = c->model;
    const llm_hparams & hp = m->hp;
    const int64_t HD = hp.n_embd_head, NKV = m->n_head_kv_l, n_ctx = c->p.n_ctx;
    if (c->backend && c->in_flight) { ggml_backend_synchronize(c->backend); c->in_flight = 0; }
    ggml_context * ctx = ggml_init({0, nullptr, true});
    ggml_cgraph * gf = ggml_new_graph_custom(ctx, 1024, false);
    ggml_tensor * shift = ggml_new_tensor_1d(ctx, GGML_TYPE_I32, n_ctx);
    ggml_set_input(named(shift, "k_shift", -1));
    for (int il = 0; il < hp.n_layer; ++il) {
        ggml_tensor * kc = c->k_l[il];
        ggml_tensor * k = ggml_view_3d(ctx, kc, HD, NKV, n_ctx, ggml_row_size(kc->type, HD), ggml_row_size(kc->type, NKV * HD), 0);
        ggml_tensor * r;
        if (kc->type == GGML_TYPE_BF16) {  // (llama.cpp ropes a non-quantised cache in place, and ROPE has no bf16 form: no context shift on such a cache)
            ggml_free(ctx);
            return -2;
        }
        if (ggml_blck_size(kc->type) > 1) {  // ggml_is_quantized
            ggml_tensor * tmp = ggml_cast(ctx, k, GGML_TYPE_F32);
            tmp = ggml_rope_ext_inplace(ctx, tmp, shift, nullptr, (int) HD, hp.rope_type, hp.n_ctx_train, hp.rope_freq_base, 1.0f, 0.0f, 1.0f, 32.0f, 1.0f);
            r = ggml_cpy(ctx, tmp, k);
        } else {
            r = ggml_rope_ext_inplace(ctx, k, shift, nullptr, (int) HD, hp.rope_type, hp.n_ctx_train, hp.rope_freq_base, 1.0f, 0.0f, 1.0f, 32.0f, 1.0f);
        }
        ggml_build_forward_expand(gf, named(r, "k_shifted", il));
    }
    ggml_gallocr_t ga = ggml_gallocr_new(c->buft);
    int rc = 0;
    if (!ga || !ggml_gallocr_alloc_graph(ga, gf)) rc = -2;
    if (rc == 0) {
        ggml_backend_tensor_set(shift, delta.data(), 0, (size_t) n_ctx * 4);
        const enum ggml_status st = c->backend ? ggml_backend_graph_compute(c->backend, gf) : c->compute(gf, c->p.n_threads);
        if (st != GGML_STATUS_SUCCESS) rc = -2;
    }
    if (ga) ggml_gallocr_free(ga);
    ggml_free(ctx);
    return rc;
}
// llama_memory_seq_add: positions [p0, p1) of a sequence move by `delta`; the cached K rows are re-rotated at once
// (llama.cpp defers that to the next llama_decode's memory update — same result).  Returns 0, or -2 if the shift graph failed.
extern "C" int llm_kv_seq_add(struct llm_context * c, int seq_id, int p0, int p1, int delta) {
    if (!c->ls.empty()) return -2;  // (the K-shift graph is not cut over devices in this harness)
    if (p0 < 0) p0 = 0;
    if (p1 < 0) p1 = INT32_MAX;
    if (delta == 0) return 0;
    std::vector<int32_t> d((size_t) c->p.n_ctx, 0);
    bool any = false;
    for (size_t i = 0; i < c->cells.size(); ++i) {
        kv_cell & x = c->cells[i];
        if (x.pos >= p0 && x.pos < p1 && (seq_id < 0 || x.seq == seq_id)) {
            x.pos += delta;
            d[i] = delta;
            any = true;
            if (x.pos < 0) { x = kv_cell(); c->used_max = -1; }  // shifted out of the sequence (llama.cpp frees such cells too)
        }
    }
    return any ? kv_apply_shift(c, d) : 0;
}

// ------------------------------------------------------------------------------------------------ -sm layer
// llama.cpp's default with several devices (llama-box/engine_param.hpp:900-916: split_mode stays LLAMA_SPLIT_MODE_LAYER unless -sm says
// otherwise; patches/llama.cpp/max_devices.patch:5-10 raises the device limit): every device holds a range of layers and their KV cache,
// ggml_backend_sched cuts the graph where the weights change device and moves the residual stream across with the destination backend's
// cpy_tensor_async, ordering the re-use of its input copies with event_record / event_wait / event_synchronize.  This is that driver for
// OUR devices (no scheduler exists in the snapshot): the calls a backend sees, in the order ggml_backend_sched_compute_splits issues them.
extern "C" struct llm_context * llm_context_new_layer_split(struct llm_model * m, const ggml_backend_t * backends, int n_dev, const struct llm_context_params * p) {
    if (n_dev < 1 || n_dev != m->n_dev || (int) m->layer_dev.size() != m->hp.n_layer) {
        fprintf(stderr, "llm_context_new_layer_split: the model was not built by llm_model_synth_layer_split for %d devices\n", n_dev);
        return nullptr;
    }
    llm_context * c = new llm_context();
    c->model = m;
    c->p = *p;
    if (c->p.n_ubatch <= 0) c->p.n_ubatch = 512;
    if (c->p.n_ctx <= 0) c->p.n_ctx = 512;
    c->p.n_ctx = (c->p.n_ctx + 255) / 256 * 256;
    const llm_hparams & hp = m->hp;
    const int64_t n_embd_k = (int64_t) m->n_head_kv_l * hp.n_embd_head;
    c->ls.resize((size_t) n_dev);
    c->k_l.assign((size_t) hp.n_layer, nullptr);
    c->v_l.assign((size_t) hp.n_layer, nullptr);
    for (int d = 0; d < n_dev; ++d) {
        llm_context::ls_dev & D = c->ls[(size_t) d];
        D.be = backends[d];
        D.buft = ggml_backend_dev_buffer_type(backends[d]->device);
        D.l0 = hp.n_layer;
        D.l1 = 0;
        for (int il = 0; il < hp.n_layer; ++il)
            if (m->layer_dev[(size_t) il] == d) { D.l0 = std::min(D.l0, il); D.l1 = std::max(D.l1, il + 1); }
        LLM_ASSERT(D.l0 < D.l1);
        D.ctx_kv = ggml_init({0, nullptr, true});
        for (int il = D.l0; il < D.l1; ++il) {
            ggml_tensor * k = ggml_new_tensor_2d(D.ctx_kv, GGML_TYPE_F16, n_embd_k, c->p.n_ctx);
            ggml_tensor * v = c->p.flash_attn ? ggml_new_tensor_2d(D.ctx_kv, GGML_TYPE_F16, n_embd_k, c->p.n_ctx) : ggml_new_tensor_2d(D.ctx_kv, GGML_TYPE_F16, c->p.n_ctx, n_embd_k);
            c->k_l[(size_t) il] = named(k, "cache_k_l", il);
            c->v_l[(size_t) il] = named(v, "cache_v_l", il);
        }
        D.buf_kv = ggml_backend_alloc_ctx_tensors_from_buft(D.ctx_kv, D.buft);
        if (!D.buf_kv) { llm_context_free(c); return nullptr; }
        ggml_backend_buffer_clear(D.buf_kv, 0);
        D.galloc = ggml_gallocr_new(D.buft);
        for (int k = 0; k < llm_context::LS_COPIES; ++k) D.ev[k] = ggml_backend_event_new(backends[d]->device);
        // reserve the worst-case segment so that later graphs re-use one buffer
        const int nub = std::min(c->p.n_ubatch, c->p.n_ctx);
        ggml_context * tmp = ggml_init({0, nullptr, true});
        ggml_cgraph * gf = ggml_new_graph_custom(tmp, 8192, false);
        seg_io io;
        make_inputs(c, tmp, D.l0, D.l1, d == n_dev - 1, nub, c->p.n_ctx, nub, io);
        build_segment(c, tmp, gf, D.l0, D.l1, d == n_dev - 1, nub, c->p.n_ctx, io);
        const bool ok = ggml_gallocr_reserve(D.galloc, gf);
        ggml_free(tmp);
        if (!ok) { llm_context_free(c); return nullptr; }
    }
    c->cells.assign((size_t) c->p.n_ctx, kv_cell());
    c->used_max = 0;
    c->key = graph_key();
    return c;
}
extern "C" void llm_layer_split_stats(const struct llm_context * c, int64_t out[4]) {
    for (int k = 0; k < 4; ++k) out[k] = c->ls_stats[k];
}

static void ls_sync_all(llm_context * c) {
    for (auto & d : c->ls) ggml_backend_synchronize(d.be);
    c->ls_copy = 0;  // ggml_backend_sched_synchronize: "always use copy 0 after a synchronization" — a decode step sees the same graph every time
    c->in_flight = 0;
}

static int decode_ubatch_ls(llm_context * c, int n_tokens, const int32_t * tokens, const int32_t * pos, const int32_t * seq_id, const std::vector<int> & slots,
                            const std::vector<int32_t> & out_ids, int n_kv, float * logits_out, int * n_out_acc, bool defer_sync) {
    llm_model * m = c->model;
    const llm_hparams & hp = m->hp;
    const int n_ctx = c->p.n_ctx, n_dev = (int) c->ls.size(), n_outputs = (int) out_ids.size();
    const int64_t n_embd_k = (int64_t) m->n_head_kv_l * hp.n_embd_head;
    graph_key key{n_tokens, n_kv, n_outputs};
    if (!(key == c->key)) {
        ls_sync_all(c);
        for (int d = 0; d < n_dev; ++d) {
            llm_context::ls_dev & D = c->ls[(size_t) d];
            if (D.buf_in) ggml_backend_buffer_free(D.buf_in);
            if (D.ctx_in) ggml_free(D.ctx_in);
            D.ctx_in = ggml_init({0, nullptr, true});
            for (int k = 0; k < llm_context::LS_COPIES; ++k) {
                D.io[k] = seg_io();
                make_inputs(c, D.ctx_in, D.l0, D.l1, d == n_dev - 1, n_tokens, n_kv, n_outputs, D.io[k]);
            }
            D.buf_in = ggml_backend_alloc_ctx_tensors_from_buft(D.ctx_in, D.buft);
            if (!D.buf_in) return -2;
            for (int k = 0; k < llm_context::LS_COPIES; ++k) {
                if (D.ctx[k]) ggml_free(D.ctx[k]);
                D.ctx[k] = ggml_init({0, nullptr, true});
                D.gf[k] = ggml_new_graph_custom(D.ctx[k], 8192, false);
                build_segment(c, D.ctx[k], D.gf[k], D.l0, D.l1, d == n_dev - 1, n_tokens, n_kv, D.io[k]);
                if (!ggml_gallocr_alloc_graph(D.galloc, D.gf[k])) return -2;  // (same topology, same plan: both slots' graphs share the compute buffer — they run one after the other on the device's stream)
            }
            // the padding rows of the masks (GGML_KQ_MASK_PAD) are never read; written once so that the cast never converts uninitialised memory
            for (int k = 0; k < llm_context::LS_COPIES; ++k) {
                std::vector<float> pad((size_t) D.io[k].mask->ne[0] * (size_t) D.io[k].mask->ne[1], -INFINITY);
                ggml_backend_tensor_set(D.io[k].mask, pad.data(), 0, pad.size() * 4);
            }
        }
        c->key = key;
    }
    const int cur = c->ls_copy;
    // host-side images of the user inputs
    std::vector<int64_t> kidx((size_t) n_tokens), vidx;
    for (int i = 0; i < n_tokens; ++i) kidx[(size_t) i] = slots[(size_t) i];
    if (!c->p.flash_attn) {
        vidx.resize((size_t) n_tokens * (size_t) n_embd_k);
        for (int i = 0; i < n_tokens; ++i)
            for (int64_t j = 0; j < n_embd_k; ++j) vidx[(size_t) i * n_embd_k + j] = j * n_ctx + slots[(size_t) i];
    }
    std::vector<int32_t> & cpos = c->mask_cpos, & cseq = c->mask_cseq;
    cpos.resize((size_t) n_kv);
    cseq.resize((size_t) n_kv);
    for (int j = 0; j < n_kv; ++j) { cpos[(size_t) j] = c->cells[(size_t) j].pos; cseq[(size_t) j] = c->cells[(size_t) j].seq; }
    std::vector<float> mask((size_t) n_kv * (size_t) n_tokens);
    for (int i = 0; i < n_tokens; ++i) mask_row(mask.data() + (size_t) i * n_kv, n_kv, cpos.data(), cseq.data(), seq_id ? seq_id[i] : 0, pos[i], 0.0f, -INFINITY);

    enum ggml_status st = GGML_STATUS_SUCCESS;
    for (int d = 0; d < n_dev && st == GGML_STATUS_SUCCESS; ++d) {
        llm_context::ls_dev & D = c->ls[(size_t) d];
        seg_io & io = D.io[cur];
        // --- user inputs (GGML_TENSOR_FLAG_INPUT): the split backend must have finished the graph that read this slot last, then a BLOCKING copy
        if (D.ev[cur]) ggml_backend_event_synchronize(D.ev[cur]);
        else ggml_backend_synchronize(D.be);
        if (d == 0) {
            // (llama.cpp fills its input tensors with ggml_backend_tensor_set — blocking — from host memory)
            if (io.tokens) ggml_backend_tensor_set(io.tokens, tokens, 0, (size_t) n_tokens * 4);
            ggml_backend_tensor_set(io.pos, pos, 0, (size_t) n_tokens * 4);
            ggml_backend_tensor_set(io.k_idxs, kidx.data(), 0, kidx.size() * 8);
            if (io.v_idxs) ggml_backend_tensor_set(io.v_idxs, vidx.data(), 0, vidx.size() * 8);
            ggml_backend_tensor_set(io.mask, mask.data(), 0, mask.size() * 4);
            if (io.out_ids) ggml_backend_tensor_set(io.out_ids, out_ids.data(), 0, out_ids.size() * 4);
        } else {
            // an input that lives on another device's backend: ggml_backend_tensor_copy -> the destination buffer's cpy_tensor
            seg_io & src = c->ls[0].io[cur];
            ggml_backend_tensor_copy(src.pos, io.pos);
            ggml_backend_tensor_copy(src.k_idxs, io.k_idxs);
            if (io.v_idxs) ggml_backend_tensor_copy(src.v_idxs, io.v_idxs);
            ggml_backend_tensor_copy(src.mask, io.mask);
            c->ls_stats[1] += 3 + (io.v_idxs ? 1 : 0);
            if (io.out_ids) ggml_backend_tensor_set(io.out_ids, out_ids.data(), 0, out_ids.size() * 4);  // (only the last device reads it: its own input)
            // --- the residual stream, produced by the previous device's graph: wait until this backend is done with the slot, then the async copy
            llm_context::ls_dev & P = c->ls[(size_t) d - 1];
            if (D.ev[cur]) { ggml_backend_event_wait(D.be, D.ev[cur]); c->ls_stats[3]++; }
            else ggml_backend_synchronize(D.be);
            // (harness addition, not in ggml_backend_sched: the copy is issued on the SOURCE stream — also order it behind the destination's last use of the slot)
            if (D.ev[cur]) { ggml_backend_event_wait(P.be, D.ev[cur]); c->ls_stats[3]++; }
            ggml_backend_tensor_copy_async(P.be, D.be, P.io[cur].out, io.resid);
            c->ls_stats[0]++;
        }
        st = ggml_backend_graph_compute_async(D.be, D.gf[cur]);
        if (D.ev[cur]) { ggml_backend_event_record(D.ev[cur], D.be); c->ls_stats[2]++; }
    }
    if (st != GGML_STATUS_SUCCESS) { ls_sync_all(c); return -2; }
    if (n_outputs > 0 || !defer_sync) {
        llm_context::ls_dev & Last = c->ls.back();
        if (n_outputs > 0) ggml_backend_tensor_get_async(Last.be, Last.io[cur].out, logits_out, 0, (size_t) n_outputs * m->n_vocab_l * 4);
        ls_sync_all(c);
        *n_out_acc += n_outputs;
        c->t_logits = Last.io[cur].out;
        c->gf = Last.gf[cur];
    } else {
        c->ls_copy = (cur + 1) % llm_context::LS_COPIES;  // the next micro-batch takes the other slot while this one is in flight
        c->in_flight++;
    }
    return 0;
}

static int decode_ubatch(llm_context * c, int n_tokens, const int32_t * tokens, const int32_t * pos, const int32_t * seq_id, const int8_t * want, float * logits_out, int * n_out_acc,
                         bool defer_sync) {
    llm_model * m = c->model;
    const llm_hparams & hp = m->hp;
    const int n_ctx = c->p.n_ctx;
    const double t0 = now_us();
    // find free cells (first fit from kv_head, wrapping once)
    std::vector<int> slots;
    for (int i = 0, at = c->kv_head; i < n_ctx && (int) slots.size() < n_tokens; ++i, at = (at + 1) % n_ctx)
        if (c->cells[at].pos < 0) slots.push_back(at);
    if ((int) slots.size() < n_tokens) return 1;
    for (int i = 0; i < n_tokens; ++i) {
        c->cells[slots[i]].pos = pos[i];
        c->cells[slots[i]].seq = seq_id ? seq_id[i] : 0;
    }
    c->kv_head = (slots.back() + 1) % n_ctx;
    // (kept up to date by the allocations; a full recount only after cells were freed)
    if (c->used_max < 0) {
        c->used_max = 0;
        for (int i = 0; i < n_ctx; ++i) if (c->cells[i].pos >= 0) c->used_max = i + 1;
    }
    for (int i = 0; i < n_tokens; ++i) c->used_max = std::max(c->used_max, slots[i] + 1);
    const int used_max = c->used_max;
    const int n_kv = std::min(n_ctx, (used_max + 255) / 256 * 256);
    std::vector<int32_t> out_ids;
    for (int i = 0; i < n_tokens; ++i) if (!want || want[i]) out_ids.push_back(i);
    const int n_outputs = (int) out_ids.size();
    // a micro-batch nobody wants logits from (every chunk of a prompt but the last): llama.cpp builds the graph with an EMPTY out_ids — the last
    // layer's tensors behind get_rows and the output head have zero rows and are skipped by the backends (ggml_is_empty)
    const int n_out_graph = n_outputs;
    if (!c->ls.empty()) {
        const int rc = decode_ubatch_ls(c, n_tokens, tokens, pos, seq_id, slots, out_ids, n_kv, logits_out, n_out_acc, defer_sync);
        if (rc != 0) {
            for (int i = 0; i < n_tokens; ++i) c->cells[slots[i]] = kv_cell();
            c->used_max = -1;
        }
        return rc;
    }

    graph_key key{n_tokens, n_kv, n_out_graph};
    bool rebuilt = false;
    if (!(c->p.graph_reuse && key == c->key && c->gf)) {
        rebuilt = true;
        build_graph(c, n_tokens, n_kv, n_out_graph);
        if (!ggml_gallocr_alloc_graph(c->galloc, c->gf)) return -2;
        c->key = key;
    }
    const double t1 = now_us();

    // inputs: filled straight into the pinned staging area and uploaded asynchronously on the backend's stream (the copies
    // and the graph are stream-ordered); without a device backend (CPU oracle) the plain blocking setter is used
    const bool async_io = c->backend != nullptr && c->pin != nullptr;
    if (async_io && c->in_flight >= llm_context::PIN_SLOTS - 1) {  // the slot about to be re-used may still be in flight
        ggml_backend_synchronize(c->backend);
        c->in_flight = 0;
    }
    char * const pin_in = c->pin ? c->pin + (size_t) c->pin_slot * c->pin_in_bytes : nullptr;
    char * const pin_out = c->pin ? c->pin + (size_t) llm_context::PIN_SLOTS * c->pin_in_bytes : nullptr;
    c->pin_slot = (c->pin_slot + 1) % llm_context::PIN_SLOTS;
    size_t pin_at = 0;
    std::vector<char> heap;  // CPU-oracle path
    auto upload = [&](ggml_tensor * t, size_t bytes, auto fill) {
        char * dst;
        if (async_io) {
            pin_at = (pin_at + 63) / 64 * 64;
            LLM_ASSERT(pin_at + bytes <= c->pin_in_bytes);
            dst = pin_in + pin_at;
            pin_at += bytes;
        } else {
            heap.resize(bytes);
            dst = heap.data();
        }
        fill(dst);
        if (async_io) ggml_backend_tensor_set_async(c->backend, t, dst, 0, bytes);
        else ggml_backend_tensor_set(t, dst, 0, bytes);
    };
    upload(c->inp_tokens, (size_t) n_tokens * 4, [&](char * d) { memcpy(d, tokens, (size_t) n_tokens * 4); });
    upload(c->inp_pos, (size_t) n_tokens * 4, [&](char * d) { memcpy(d, pos, (size_t) n_tokens * 4); });
    upload(c->inp_k_idxs, (size_t) n_tokens * 8, [&](char * d) {
        int64_t * k = (int64_t *) d;
        for (int i = 0; i < n_tokens; ++i) k[i] = slots[i];
    });
    if (c->inp_v_idxs) {
        const int64_t n_embd_k = (int64_t) m->n_head_kv_l * hp.n_embd_head;
        upload(c->inp_v_idxs, (size_t) n_tokens * n_embd_k * 8, [&](char * d) {
            int64_t * vidx = (int64_t *) d;
            for (int i = 0; i < n_tokens; ++i)
                for (int64_t j = 0; j < n_embd_k; ++j) vidx[(size_t) i * n_embd_k + j] = j * n_ctx + slots[i];
        });
    }
    if (c->inp_out_ids) upload(c->inp_out_ids, (size_t) n_out_graph * 4, [&](char * d) { memcpy(d, out_ids.data(), (size_t) n_out_graph * 4); });
    {
        // rows beyond n_tokens are padding (GGML_KQ_MASK_PAD) that no kernel reads: they are written once when the
        // graph is (re)built and skipped on re-use, which keeps the per-step upload at n_tokens * n_kv entries
        const int64_t n_tok_pad = c->inp_mask->ne[1];
        const int64_t rows = rebuilt ? n_tok_pad : n_tokens;
        // cell j is visible to token i iff it holds a position (>= 0) of i's sequence that is not after i's — the condition llama.cpp's
        // set_input_kq_mask evaluates per (token, cell); the cells' fields are gathered once per step so that the row loops vectorise
        // (an empty cell's position -1 compares as the largest unsigned value: never visible)
        std::vector<int32_t> & cpos = c->mask_cpos, & cseq = c->mask_cseq;
        cpos.resize((size_t) n_kv);
        cseq.resize((size_t) n_kv);
        for (int j = 0; j < n_kv; ++j) {
            cpos[j] = c->cells[j].pos;
            cseq[j] = c->cells[j].seq;
        }
        upload(c->inp_mask, (size_t) n_kv * rows * 4, [&](char * d) {
            float * mask = (float *) d;
            for (size_t e = (size_t) n_kv * n_tokens; e < (size_t) n_kv * rows; ++e) mask[e] = -INFINITY;
            for (int i = 0; i < n_tokens; ++i) mask_row(mask + (size_t) i * n_kv, n_kv, cpos.data(), cseq.data(), seq_id ? seq_id[i] : 0, pos[i], 0.0f, -INFINITY);
        });
    }
    const double t2 = now_us();
    const size_t logit_bytes = (size_t) n_outputs * m->n_vocab_l * 4;
    // (logits_out inside the pinned output area: llm_decode found room for ALL outputs of the call there — the device writes the
    // rows where llm_get_logits will read them, as llama.cpp's pinned output buffer works; otherwise staged + copied)
    const bool direct_out = c->pin != nullptr && (char *) logits_out >= pin_out && (char *) logits_out < pin_out + (size_t) c->pin_out_rows * m->n_vocab_l * 4;
    const bool async_out = async_io && n_outputs > 0 && (direct_out || n_outputs <= c->pin_out_rows);
    enum ggml_status st;
    if (async_io) {
        st = ggml_backend_graph_compute_async(c->backend, c->gf);
        if (st == GGML_STATUS_SUCCESS && async_out) ggml_backend_tensor_get_async(c->backend, c->t_logits, direct_out ? (void *) logits_out : (void *) pin_out, 0, logit_bytes);
        if (n_outputs > 0 || st != GGML_STATUS_SUCCESS || !defer_sync) {
            ggml_backend_synchronize(c->backend);
            c->in_flight = 0;
        } else {
            c->in_flight++;
        }
    } else {
        st = c->backend ? ggml_backend_graph_compute(c->backend, c->gf) : c->compute(c->gf, c->p.n_threads);
    }
    const double t3 = now_us();
    if (st != GGML_STATUS_SUCCESS) {
        for (int i = 0; i < n_tokens; ++i) c->cells[slots[i]] = kv_cell();  // roll the slots back
        c->used_max = -1;
        return -2;
    }
    if (n_outputs > 0) {
        if (async_out && !direct_out) memcpy(logits_out, pin_out, logit_bytes);
        else if (!async_out) ggml_backend_tensor_get(c->t_logits, logits_out, 0, logit_bytes);
        *n_out_acc += n_outputs;
    }
    const double t4 = now_us();
    c->timings[0] += t1 - t0;
    c->timings[1] += t2 - t1;
    c->timings[2] += t3 - t2;
    c->timings[3] += t4 - t3;
    return 0;
}

extern "C" int llm_decode(struct llm_context * c, int n_tokens, const int32_t * tokens, const int32_t * pos, const int32_t * seq_id, const int8_t * want_logits) {
    if (n_tokens <= 0 || !tokens || !pos) return -1;
    for (int i = 0; i < n_tokens; ++i)
        if (tokens[i] < 0 || tokens[i] >= c->model->hp.n_vocab || pos[i] < 0) return -1;
    int n_total_out = 0;
    for (int i = 0; i < n_tokens; ++i) n_total_out += (!want_logits || want_logits[i]) ? 1 : 0;
    // outputs land in the pinned area when they all fit (one sampled row per sequence: always, up to 64 sequences)
    const bool pinned_out = c->backend != nullptr && c->pin != nullptr && n_total_out <= c->pin_out_rows;
    if (pinned_out) {
        c->logits_base = (float *) (c->pin + (size_t) llm_context::PIN_SLOTS * c->pin_in_bytes);
    } else {
        c->logits.resize((size_t) std::max(1, n_total_out) * c->model->n_vocab_l);
        c->logits_base = c->logits.data();
    }
    c->n_outputs = 0;
    // batch position -> output row (llama.cpp's output_ids): what llama_get_logits_ith(ctx, i) resolves i >= 0 through
    c->output_ids.assign((size_t) n_tokens, -1);
    for (int i = 0, o = 0; i < n_tokens; ++i) if (!want_logits || want_logits[i]) c->output_ids[(size_t) i] = o++;
    for (int k = 0; k < 4; ++k) c->timings[k] = 0;
    for (int i0 = 0; i0 < n_tokens; i0 += c->p.n_ubatch) {
        const int n = std::min(c->p.n_ubatch, n_tokens - i0);
        int rc = decode_ubatch(c, n, tokens + i0, pos + i0, seq_id ? seq_id + i0 : nullptr, want_logits ? want_logits + i0 : nullptr,
                               c->logits_base + (size_t) c->n_outputs * c->model->n_vocab_l, &c->n_outputs, /*defer_sync=*/i0 + n < n_tokens);
        if (rc != 0) {
            if (c->backend && c->in_flight) { ggml_backend_synchronize(c->backend); c->in_flight = 0; }
            if (!c->ls.empty()) ls_sync_all(c);
            return rc;
        }
    }
    if (!c->ls.empty() && c->in_flight) ls_sync_all(c);
    if (c->backend && c->in_flight) {  // the call returns with everything finished
        ggml_backend_synchronize(c->backend);
        c->in_flight = 0;
    }
    return 0;
}
extern "C" int llm_decode_steps(struct llm_context * c, int n_steps, int n_par, const int32_t * tokens, int pos0) {
    if (n_steps <= 0 || n_par <= 0 || !tokens) return -1;
    std::vector<int32_t> pos((size_t) n_par), seq((size_t) n_par);
    for (int s = 0; s < n_par; ++s) seq[s] = s;
    double acc[4] = {0, 0, 0, 0};
    for (int i = 0; i < n_steps; ++i) {
        for (int s = 0; s < n_par; ++s) pos[s] = pos0 + i;
        const int rc = llm_decode(c, n_par, tokens + (size_t) i * n_par, pos.data(), seq.data(), nullptr);
        if (rc != 0) return rc;
        for (int k = 0; k < 4; ++k) acc[k] += c->timings[k];
    }
    for (int k = 0; k < 4; ++k) c->timings[k] = acc[k];
    return 0;
}
// Speculative decoding as llama-box drives it (httpserver.hpp:4042-4069: every slot's sampled token plus its drafts go into ONE batch, logits
// requested at every position; :4696-4768: the drafts are verified against them and the rejected tail is removed from the cache with
// llama_memory_seq_rm).  One step here = one llama_decode over n_par x (1 + n_draft) tokens — sequence s at positions p .. p + n_draft —
// followed by the WORST case of the verification: every draft rejected (seq_rm of p + 1 ..), so the next step starts at p + 1.
// tokens: [n_steps][n_par][1 + n_draft].
extern "C" int llm_verify_steps(struct llm_context * c, int n_steps, int n_par, int n_draft, const int32_t * tokens, int pos0) {
    if (n_steps <= 0 || n_par <= 0 || n_draft < 0 || !tokens) return -1;
    const int T1 = 1 + n_draft, n = n_par * T1;
    std::vector<int32_t> pos((size_t) n), seq((size_t) n);
    for (int s = 0; s < n_par; ++s)
        for (int j = 0; j < T1; ++j) seq[(size_t) s * T1 + j] = s;
    double acc[4] = {0, 0, 0, 0};
    for (int i = 0; i < n_steps; ++i) {
        for (int s = 0; s < n_par; ++s)
            for (int j = 0; j < T1; ++j) pos[(size_t) s * T1 + j] = pos0 + i + j;
        const int rc = llm_decode(c, n, tokens + (size_t) i * n, pos.data(), seq.data(), nullptr);
        if (rc != 0) return rc;
        for (int k = 0; k < 4; ++k) acc[k] += c->timings[k];
        if (n_draft > 0)
            for (int s = 0; s < n_par; ++s) llm_kv_seq_rm(c, s, pos0 + i + 1, -1);
    }
    for (int k = 0; k < 4; ++k) c->timings[k] = acc[k];
    return 0;
}
extern "C" int llm_n_outputs(const struct llm_context * c) { return c->n_outputs; }
extern "C" float * llm_get_logits(struct llm_context * c) { return c->logits_base; }
// llama_get_logits_ith (include/llama.h; the reference reads through it at llama-box/httpserver.hpp:442 and inside
// common_sampler_sample2 -> set_logits, patches/llama.cpp/sampling.patch:58-81): i >= 0 is a POSITION OF THE BATCH and must be one
// whose logits were requested; i < 0 counts output rows from the end (-1 = the last one).  NULL for anything else.
extern "C" float * llm_get_logits_ith(struct llm_context * c, int i) {
    int row;
    if (i < 0) row = c->n_outputs + i;
    else if ((size_t) i < c->output_ids.size()) row = c->output_ids[(size_t) i];
    else return nullptr;
    if (row < 0 || row >= c->n_outputs) return nullptr;
    return c->logits_base + (size_t) row * c->model->n_vocab_l;
}
// common_sampler_sample2 with the greedy chain llama-box builds for temperature 0 (llama_sampler_init_greedy: the FIRST maximal
// logit wins): the host-side consumer of the logits rows the backend delivers.  -1 if position idx has no logits.
extern "C" int32_t llm_sample_greedy(struct llm_context * c, int idx) {
    const float * lg = llm_get_logits_ith(c, idx);
    if (!lg) return -1;
    const int nv = c->model->n_vocab_l;
    int best = 0;
    for (int t = 1; t < nv; ++t) if (lg[t] > lg[best]) best = t;
    return best;
}
// get_token_probabilities (llama-box/httpserver.hpp:440-467): tokens sorted by logit, soft-max over the whole vocabulary; writes the
// top_n leading (id, p) pairs.  Returns the number written, -1 if position idx has no logits.
extern "C" int llm_token_probabilities(struct llm_context * c, int idx, int top_n, int32_t * ids, float * probs) {
    const float * lg = llm_get_logits_ith(c, idx);
    if (!lg) return -1;
    const int nv = c->model->n_vocab_l;
    std::vector<int32_t> order((size_t) nv);
    for (int t = 0; t < nv; ++t) order[(size_t) t] = t;
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return lg[a] > lg[b]; });
    const float max_l = lg[order[0]];
    float cum = 0.0f;
    std::vector<float> p((size_t) nv);
    for (int t = 0; t < nv; ++t) { p[(size_t) t] = expf(lg[order[(size_t) t]] - max_l); cum += p[(size_t) t]; }
    const int n = std::min(top_n, nv);
    for (int t = 0; t < n; ++t) { ids[t] = order[(size_t) t]; probs[t] = p[(size_t) t] / cum; }
    return n;
}
extern "C" struct ggml_cgraph * llm_last_graph(struct llm_context * c) { return c->gf; }
// the K (which = 0) or V (1) cache tensor of layer il: what llama.cpp's state save / slot save reads with ggml_backend_tensor_get
extern "C" struct ggml_tensor * llm_context_cache_tensor(struct llm_context * c, int il, int which) {
    if (il < 0 || il >= (int) c->k_l.size()) return nullptr;
    return which ? c->v_l[(size_t) il] : c->k_l[(size_t) il];
}
extern "C" void llm_last_timings(const struct llm_context * c, double out[4]) {
    for (int k = 0; k < 4; ++k) out[k] = c->timings[k];
}
