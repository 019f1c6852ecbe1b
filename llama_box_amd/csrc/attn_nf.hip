// attn_nf.hip — the non-flash attention chain of a small batch over a unified cache, as ONE launch.
//
// llama-box runs with -fa off unless asked (engine_param.hpp:772-779), so a `-np` decode step reaches the backend as
//   kq = MUL_MAT(K view [D, n_kv, NKV] f16, q [D, T, NH])  ->  p = SOFT_MAX(kq, mask, scale)  ->  MUL_MAT(V^T view [n_kv, D, NKV] f16, p)
// over ALL cells of the cache although token t can see only its own sequence's: dense, that is 32 x the arithmetic and two 30 MB
// intermediates per layer at -np 32 (94 us in three launches).  Masked cells contribute exactly nothing in ggml-cpu — the logit is
// -inf, expf gives 0, the running max and the double sum are unchanged, and 0 x V adds 0 — so walking the token's list of VISIBLE
// positions (k_fattn_pos_scan, once per graph run: the mask is the same tensor in every layer) computes the same numbers:
//   1. logits of the listed cells for the G query heads of a KV head (K rows are gathered, 256 B each; q rounded to f16 as
//      from_float does; f32 accumulation), kept compactly in scratch;
//   2. ggml_compute_forward_soft_max_f32 over them: w = s scale + mask, max, e = expf(w - max), sum in double, llama-box's zero-sum
//      guard (ggml-cpu.patch:5-15), p = e (1 / sum), rounded to f16 where the CPU's second MUL_MAT rounds its src1;
//   3. V^T.p: a wave takes rows of the TRANSPOSED cache (one per head dimension), lanes take listed cells — the prompt's cells are
//      contiguous in a row, the decode cells of 32 interleaved sequences are 64 B apart — and the G sums are wave-reduced.
// A workgroup is (token, KV head, slice of the head dimensions); slices exist so that a handful of tokens still fill the chip, and
// each repeats steps 1-2 for itself (cheap next to step 3).  Sums run in another order than the dense kernels' (and the CPU's).
#include <algorithm>
#include <type_traits>

#include "dev_util.h"
#include "kernels.h"

namespace mi355x {

#define NF_CAP 1024  // visible cells per token served from registers and LDS
struct __attribute__((aligned(4))) u32x4a { uint32_t x, y, z, w; };  // a 16-byte load that only promises dword alignment
template <int G>
__global__ void __launch_bounds__(256) k_attn_nf_list(const tdesc q, const tdesc k, const tdesc v, const tdesc mask, const tdesc dst, const int * __restrict__ lists,
                                                      const int list_stride, float * __restrict__ scratch, const float scale, const int dq_n, q8k_dev * __restrict__ q8, const int nf_groups_on) {
    // q8 != null (only with dq_n == 1 and an even G): the result's only readers are quantised mat-muls (wo) — it leaves the kernel as the
    // Q8_K blocks quantize_row_q8_K builds of the token's row (a block = two heads), through the q area of the LDS, and not as f32
    constexpr int D = 128;
    const bool nf_groups = nf_groups_on;
    extern __shared__ __attribute__((aligned(16))) float nf_smem[];  // q [G][128] | positions [NF_CAP] | probabilities [G][NF_CAP]
    __shared__ float shf[4];
    __shared__ double shd[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tok = blockIdx.x, kvh = blockIdx.y, dq = blockIdx.z;
    const int n_kv = (int) k.ne[1], nkv_heads = (int) k.ne[2];
    const int * lt = lists + (int64_t) tok * list_stride;
    const int cnt = lt[0];
    const int * pos = lt + 1;
    float * sc = scratch + (((int64_t) tok * nkv_heads + kvh) * dq_n + dq) * (int64_t) G * n_kv;  // [G][n_kv], the first cnt of each row in use
    const int rows_per = D / dq_n, d0 = dq * rows_per;
    auto out_ptr = [&](const int g, const int d) { return (float *) (dst.data + (int64_t) d * dst.nb[0] + (int64_t) tok * dst.nb[1] + (int64_t) (kvh * G + g) * dst.nb[2]); };
    auto put = [&](const int g, const int d, const float t) {
        if (q8) nf_smem[g * D + d] = t;
        else *out_ptr(g, d) = t;
    };
    auto finish = [&]() {  // (q8) every wave has stored its rows: quantise head pairs
        if (!q8) return;
        __syncthreads();
        if (wave < G / 2) {
            const float4 t4 = ((const float4 *) (nf_smem + wave * 256))[lane];
            const float t[4] = {t4.x, t4.y, t4.z, t4.w};
            wave_quantize_q8_K(t, lane, q8 + (int64_t) tok * (nkv_heads * G * D / 256) + (kvh * G) / 2 + wave);
        }
    };
    if (cnt == 0) {  // a token that sees nothing: the CPU's row is expf(-inf - -inf) = NaN throughout
        for (int e = tid; e < G * rows_per; e += 256) put(e / rows_per, d0 + e % rows_per, __builtin_nanf(""));
        finish();
        return;
    }
    const char * mrow = mask.data + (int64_t) tok * mask.nb[1];
    const bool m16 = mask.type == GGML_TYPE_F16;
    const char * vbase = v.data + (int64_t) kvh * v.nb[2];
    // ---- up to NF_CAP visible cells (every -np decode step): everything stays on chip.  A thread owns cells tid, tid + 256, ..: it reads
    // their K rows whole (16 independent 16-byte loads, the q rows come from LDS by broadcast), so a logit needs no cross-lane sum and
    // the soft-max runs on registers; probabilities and positions go to LDS for step 3.  (The first version walked the list with one
    // dependent load pair per trip in every phase: 102 us per launch, all of it memory latency.)
    if (cnt <= NF_CAP) {
        float * qs = nf_smem;                       // [G][128] q rounded to f16
        int * pos_s = (int *) (nf_smem + G * D);    // [NF_CAP]
        float * p_s = nf_smem + G * D + NF_CAP;     // [G][NF_CAP]
        for (int e = tid; e < G * D; e += 256) {
            const int g = e / D, i = e % D;
            qs[e] = h2f(f2h(*((const float *) (q.data + (int64_t) tok * q.nb[1] + (int64_t) (kvh * G + g) * q.nb[2]) + i)));
        }
        for (int c = tid; c < cnt; c += 256) pos_s[c] = pos[c];
        __syncthreads();
        float mx[G];
#pragma unroll
        for (int g = 0; g < G; ++g) mx[g] = -INFINITY;
        const char * kbase = k.data + (int64_t) kvh * k.nb[2];
#pragma unroll 1
        for (int c = tid; c < cnt; c += 256) {
            const int pc = pos_s[c];
            const uint4 * kr = (const uint4 *) (kbase + (int64_t) pc * k.nb[1]);
            uint4 kv[16];  // the K row: 128 halves, all 16 loads in flight
#pragma unroll
            for (int u = 0; u < 16; ++u) kv[u] = kr[u];
            const float mv = m16 ? h2f(((const uint16_t *) mrow)[pc]) : ((const float *) mrow)[pc];
            // one head at a time (a real loop: unrolled over G, the scheduler reads all 128 G values of q ahead — 512 registers and spills
            // at G = 4); the logit goes straight to its LDS slot
#pragma unroll 1
            for (int g = 0; g < G; ++g) {
                const float * qg = qs + g * D;
                float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const uint32_t uu[4] = {kv[u].x, kv[u].y, kv[u].z, kv[u].w};
                    const float4 q0 = *(const float4 *) (qg + 8 * u), q1 = *(const float4 *) (qg + 8 * u + 4);
                    s0 = fmaf(h2f((uint16_t) (uu[0] & 0xFFFF)), q0.x, s0); s1 = fmaf(h2f((uint16_t) (uu[0] >> 16)), q0.y, s1);
                    s0 = fmaf(h2f((uint16_t) (uu[1] & 0xFFFF)), q0.z, s0); s1 = fmaf(h2f((uint16_t) (uu[1] >> 16)), q0.w, s1);
                    s0 = fmaf(h2f((uint16_t) (uu[2] & 0xFFFF)), q1.x, s0); s1 = fmaf(h2f((uint16_t) (uu[2] >> 16)), q1.y, s1);
                    s0 = fmaf(h2f((uint16_t) (uu[3] & 0xFFFF)), q1.z, s0); s1 = fmaf(h2f((uint16_t) (uu[3] >> 16)), q1.w, s1);
                }
                float t = (s0 + s1) * scale;
                t += mv;
                p_s[g * NF_CAP + c] = t;  // (this thread's own slot until the probabilities are published below)
            }
        }
        for (int c = tid; c < cnt; c += 256) {
#pragma unroll
            for (int g = 0; g < G; ++g) mx[g] = fmaxf(mx[g], p_s[g * NF_CAP + c]);
        }
        __shared__ float shm[4][G];
        __shared__ double shs[4][G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float t = wave_max(mx[g]);
            if (lane == 0) shm[wave][g] = t;
        }
        __syncthreads();
        double sum[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            mx[g] = fmaxf(fmaxf(shm[0][g], shm[1][g]), fmaxf(shm[2][g], shm[3][g]));
            sum[g] = 0.0;
        }
        for (int c = tid; c < cnt; c += 256) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float e = expf(p_s[g * NF_CAP + c] - mx[g]);
                p_s[g * NF_CAP + c] = e;
                sum[g] += (double) e;
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const double t = wave_sum_d(sum[g]);
            if (lane == 0) shs[wave][g] = t;
        }
        __syncthreads();
#pragma unroll
        for (int g = 0; g < G; ++g) {
            double t = ((shs[0][g] + shs[1][g]) + shs[2][g]) + shs[3][g];
            if (isnan(t) || t == 0.0) t = -INFINITY;
            const float inv = (float) (1.0 / t);
            for (int c = tid; c < cnt; c += 256) p_s[g * NF_CAP + c] = h2f(f2h(p_s[g * NF_CAP + c] * inv));
        }
        __syncthreads();
        // 3. V^T.p: lanes take cells (4 per lane and group of 256), a wave takes RB rows of the transposed cache at a time — all RB x 4 two-byte
        // loads of a block are requested before the first is used (one row pair at a time, each trip waited for its own loads: 16 round
        // trips per wave were 30 of the launch's 52 us)
        auto vtp = [&](auto rb_tag) {
        constexpr int RB = decltype(rb_tag)::value;
        for (int r0 = wave * RB; r0 < rows_per; r0 += 4 * RB) {
            float acc[RB][G];
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int g = 0; g < G; ++g) acc[r][g] = 0.0f;
            const char * vr0 = vbase + (int64_t) (d0 + r0) * v.nb[1];
            for (int cg = 0; cg < cnt; cg += 256) {
                int64_t o[4];
                float pp[G][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = cg + lane + 64 * j;
                    const bool in = c < cnt;
                    o[j] = in ? (int64_t) pos_s[c] * 2 : (int64_t) pos_s[0] * 2;
#pragma unroll
                    for (int g = 0; g < G; ++g) pp[g][j] = in ? p_s[g * NF_CAP + c] : 0.0f;
                }
                uint16_t xv[RB][4];
#pragma unroll
                for (int r = 0; r < RB; ++r) {
                    const char * vr = vr0 + (int64_t) min(r, rows_per - 1 - r0) * v.nb[1];
#pragma unroll
                    for (int j = 0; j < 4; ++j) xv[r][j] = *(const uint16_t *) (vr + o[j]);
                }
#pragma unroll
                for (int r = 0; r < RB; ++r)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float x = h2f(xv[r][j]);
#pragma unroll
                        for (int g = 0; g < G; ++g) acc[r][g] = fmaf(x, pp[g][j], acc[r][g]);
                    }
            }
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const float t = wave_sum(acc[r][g]);
                    if (lane == 0 && r0 + r < rows_per) put(g, d0 + r0 + r, t);
                }
        }
        };
        // rows per wave and block: as many as keep all four waves busy with this workgroup's slice of the head dimensions
        // 3'. The same product with LANES on the rows of V^T and the list walked in GROUPS of eight entries.  A token's prompt is a run of
        // consecutive cells (the unified cache hands cells out first-fit): eight consecutive cells of a row are ONE 16-byte load, and only
        // the scattered rest — the decode cells of interleaved sequences — is gathered two bytes at a time.  The launch is bound by the
        // address pipeline (every lane of these loads sits in another cache line: ~65 clocks per wave-instruction), so the count of load
        // instructions is what matters: at 128 + 72 cells a wave issues 8 + 36 of them instead of 128, and a lane owns its row's sums (no
        // 64-lane reductions).  Waves 0 / 1 take the even groups of rows 0..63 / 64..127, waves 2 / 3 the odd groups; the halves meet in LDS.
        // One slice of the head dimensions and at most 512 cells (64 groups: one ballot); the form above serves the rest.
        if (dq_n == 1 && cnt <= 512 && nf_groups) {
            const int ngr = (cnt + 7) / 8, gi_l = lane;
            const bool in = gi_l < ngr;
            const int p_first = pos_s[min(8 * gi_l, cnt - 1)], p_last = pos_s[min(8 * gi_l + 7, cnt - 1)];
            const bool is_run = in && 8 * gi_l + 7 < cnt && p_last == p_first + 7 && (p_first & 1) == 0;  // ascending list: consecutive; even start: dword-aligned
            const int wv = __builtin_amdgcn_readfirstlane(wave);  // (wave-uniform for the compiler too: the masks and group indices below are scalars)
            const unsigned long long par = (wv >> 1) ? 0xAAAAAAAAAAAAAAAAull : 0x5555555555555555ull;
            unsigned long long m_run = __ballot(is_run) & par, m_sc = __ballot(in && !is_run) & par;
            const int row = 64 * (wv & 1) + lane;
            const char * vr = vbase + (int64_t) row * v.nb[1];
            float acc[G];
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g] = 0.0f;
            auto take = [](unsigned long long & m) { const int i = __builtin_ctzll(m); m &= m - 1; return i; };
            // runs: up to eight 16-byte loads in flight
            while (m_run) {
                int gi[8];
                u32x4a x[8];
                int n = 0;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    gi[u] = m_run ? take(m_run) : -1;
                    if (gi[u] >= 0) { x[u] = *(const u32x4a *) (vr + (int64_t) pos_s[8 * gi[u]] * 2); n = u + 1; }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (u >= n) break;
                    const uint32_t w4[4] = {x[u].x, x[u].y, x[u].z, x[u].w};
                    float xf[8];
#pragma unroll
                    for (int k = 0; k < 4; ++k) { xf[2 * k] = h2f((uint16_t) (w4[k] & 0xFFFF)); xf[2 * k + 1] = h2f((uint16_t) (w4[k] >> 16)); }
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const float4 pa = *(const float4 *) (p_s + g * NF_CAP + 8 * gi[u]), pb = *(const float4 *) (p_s + g * NF_CAP + 8 * gi[u] + 4);
                        const float pv[8] = {pa.x, pa.y, pa.z, pa.w, pb.x, pb.y, pb.z, pb.w};
#pragma unroll
                        for (int k = 0; k < 8; ++k) acc[g] = fmaf(xf[k], pv[k], acc[g]);
                    }
                }
            }
            // scattered groups: the gathers of up to five groups (40 two-byte loads) in flight together
            while (m_sc) {
                int gi[5];
                uint16_t xs[5][8];
#pragma unroll
                for (int u = 0; u < 5; ++u) {
                    gi[u] = m_sc ? take(m_sc) : -1;
                    if (gi[u] >= 0) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) xs[u][k] = *(const uint16_t *) (vr + (int64_t) pos_s[min(8 * gi[u] + k, cnt - 1)] * 2);
                    }
                }
#pragma unroll
                for (int u = 0; u < 5; ++u) {
                    if (gi[u] < 0) break;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        if (8 * gi[u] + k >= cnt) break;
                        const float xv = h2f(xs[u][k]);
#pragma unroll
                        for (int g = 0; g < G; ++g) acc[g] = fmaf(xv, p_s[g * NF_CAP + 8 * gi[u] + k], acc[g]);
                    }
                }
            }
            __syncthreads();  // every wave is done with the probabilities: their area carries the odd groups' sums over
            if (wv >> 1) {
#pragma unroll
                for (int g = 0; g < G; ++g) p_s[g * D + row] = acc[g];
            }
            __syncthreads();
            if (!(wv >> 1)) {
#pragma unroll
                for (int g = 0; g < G; ++g) put(g, row, acc[g] + p_s[g * D + row]);
            }
            finish();
            return;
        }
        if (G <= 4 && rows_per >= 64) vtp(std::integral_constant<int, 16>{});
        else if (rows_per >= 32) vtp(std::integral_constant<int, 8>{});
        else vtp(std::integral_constant<int, 4>{});
        finish();
        return;
    }
    // ---- longer lists (a token that sees thousands of cells): the same three steps through scratch memory
    // ---- 1. logits: 16 lanes per K row (8 dims each), 4 rows per wave and trip
    {
        const int sub = lane >> 4, sl = lane & 15;
        float qf[G][8];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float * qp = (const float *) (q.data + (int64_t) tok * q.nb[1] + (int64_t) (kvh * G + g) * q.nb[2]) + 8 * sl;
            const float4 a = *(const float4 *) qp, b = *(const float4 *) (qp + 4);
            const float t[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int i = 0; i < 8; ++i) qf[g][i] = h2f(f2h(t[i]));
        }
        const char * kbase = k.data + (int64_t) kvh * k.nb[2] + sl * 16;
        for (int c0 = wave * 4; c0 < cnt; c0 += 16) {
            const int c = min(c0 + sub, cnt - 1);
            const uint4 kr = *(const uint4 *) (kbase + (int64_t) pos[c] * k.nb[1]);
            const uint32_t u[4] = {kr.x, kr.y, kr.z, kr.w};
            float kf[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                kf[2 * i] = h2f((uint16_t) (u[i] & 0xFFFF));
                kf[2 * i + 1] = h2f((uint16_t) (u[i] >> 16));
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float s = 0.0f;
#pragma unroll
                for (int i = 0; i < 8; ++i) s = fmaf(kf[i], qf[g][i], s);
                s = row16_sum(s);
                if (sl == 0 && c0 + sub < cnt) sc[(int64_t) g * n_kv + c0 + sub] = s;
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    // ---- 2. soft-max of each head's row (all four waves per head: the reductions are block-wide, as in ops.hip)
#pragma unroll 1
    for (int g = 0; g < G; ++g) {
        float * row = sc + (int64_t) g * n_kv;
        float mx = -INFINITY;
        for (int c = tid; c < cnt; c += 256) {
            float w = row[c] * scale;
            w += m16 ? h2f(((const uint16_t *) mrow)[pos[c]]) : ((const float *) mrow)[pos[c]];
            row[c] = w;
            mx = fmaxf(mx, w);
        }
        mx = wave_max(mx);
        __syncthreads();
        if (lane == 0) shf[wave] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(shf[0], shf[1]), fmaxf(shf[2], shf[3]));
        double sum = 0.0;
        for (int c = tid; c < cnt; c += 256) {
            const float e = expf(row[c] - mx);
            row[c] = e;
            sum += (double) e;
        }
        sum = wave_sum_d(sum);
        __syncthreads();
        if (lane == 0) shd[wave] = sum;
        __syncthreads();
        sum = ((shd[0] + shd[1]) + shd[2]) + shd[3];
        if (isnan(sum) || sum == 0.0) sum = -INFINITY;
        const float inv = (float) (1.0 / sum);
        for (int c = tid; c < cnt; c += 256) row[c] = h2f(f2h(row[c] * inv));
    }
    __threadfence_block();
    __syncthreads();
    // ---- 3. V^T.p: wave w takes rows d0 + w, d0 + w + 4, ..; lanes take listed cells
    for (int r = wave; r < rows_per; r += 4) {
        const int d = d0 + r;
        const char * vrow = vbase + (int64_t) d * v.nb[1];
        float acc[G];
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g] = 0.0f;
        for (int c = lane; c < cnt; c += 64) {
            const float vv = h2f(*(const uint16_t *) (vrow + (int64_t) pos[c] * 2));
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g] = fmaf(vv, sc[(int64_t) g * n_kv + c], acc[g]);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float t = wave_sum(acc[g]);
            if (lane == 0) put(g, d, t);
        }
    }
    finish();
}

// scratch floats the launch needs (0: the shape is not served)
size_t attn_nf_list_scratch_bytes(const tdesc & q, const tdesc & k, int * dq_out) {
    const int64_t T = q.ne[1], NH = q.ne[2], NKV = k.ne[2], n_kv = k.ne[1];
    if (T < 2 || T > 32 || NKV <= 0 || NH % NKV != 0 || q.ne[3] != 1 || k.ne[3] != 1 || k.ne[0] != 128 || q.ne[0] != 128) return 0;
    const int64_t G = NH / NKV;
    if (G < 1 || G > 8) return 0;
    // The lists live on the device, so the host cannot know how many cells a token sees; it knows the average.  Tokens beyond NF_CAP cells
    // take the scratch-memory path of the kernel, which is correct but slow (one dependent load pair per trip): the chain is only fused
    // when the cache holds at most NF_CAP cells per token of the batch — the decode steps of many sequences — and a few tokens over
    // long contexts stay on the dense kernels.
    if (n_kv > (int64_t) NF_CAP * T) return 0;
    int dq = 1;
    while (dq < 8 && T * NKV * dq < 192) dq *= 2;  // slices of the head dimensions: enough workgroups for the chip when the tokens are few
    if (dq_out) *dq_out = dq;
    return (size_t) (T * NKV * dq * G * n_kv) * sizeof(float);
}
// q: the permuted view [D, T, NH] MUL_MAT reads as src1; k: [D, n_kv, NKV] f16; v: the transposed view [n_kv, D, NKV] f16; dst: the
// second MUL_MAT's result [D, T, NH] f32.  false: not served (the caller runs the three nodes one by one)
bool launch_attn_nf_list(hipStream_t s, const tdesc & q, const tdesc & k, const tdesc & v, const tdesc & mask, const tdesc & dst, const int * lists, int list_stride,
                         float * scratch, size_t scratch_bytes, float scale, void * q8_out) {
    int dq = 1;
    const size_t need = attn_nf_list_scratch_bytes(q, k, &dq);
    if (need == 0 || need > scratch_bytes || !scratch || !lists) return false;
    if (k.type != GGML_TYPE_F16 || v.type != GGML_TYPE_F16 || q.type != GGML_TYPE_F32 || k.nb[0] != 2 || v.nb[0] != 2 || q.nb[0] != 4 || dst.nb[0] != 4) return false;
    if (v.ne[0] != k.ne[1] || v.ne[1] != 128 || v.ne[2] != k.ne[2] || v.ne[3] != 1) return false;
    if ((k.nb[1] % 16) || (k.nb[2] % 16) || (((uintptr_t) k.data) & 15) || (q.nb[1] % 16) || (q.nb[2] % 16) || (((uintptr_t) q.data) & 15)) return false;
    if ((mask.type != GGML_TYPE_F16 && mask.type != GGML_TYPE_F32) || mask.ne[0] < k.ne[1] || mask.ne[1] < q.ne[1] || mask.ne[2] != 1 || mask.ne[3] != 1) return false;
    const int G = (int) (q.ne[2] / k.ne[2]);
    if (q8_out && (dq != 1 || (G & 1))) return false;  // Q8_K blocks are head pairs of whole rows
    static const int groups_on = !getenv("GGML_MI355X_NF_GROUPS") || atoi(getenv("GGML_MI355X_NF_GROUPS")) != 0;
    dim3 grid((unsigned) q.ne[1], (unsigned) k.ne[2], (unsigned) dq);
#define NF(G_) case G_: hipLaunchKernelGGL(k_attn_nf_list<G_>, grid, dim3(256), (size_t) (G_ * 128 + NF_CAP + G_ * NF_CAP) * 4, s, q, k, v, mask, dst, lists, list_stride, scratch, scale, dq, (q8k_dev *) q8_out, groups_on); return true;
    switch (G) { NF(1) NF(2) NF(3) NF(4) NF(5) NF(6) NF(7) NF(8) default: break; }
#undef NF
    return false;
}

MI_TU_TOUCH(attn_nf)

}  // namespace mi355x
