// attn_probe.hip — where do the microseconds of the non-flash decode attention go?  K.q (grouped-head short-row mat-vec), soft-max
// over 32 rows x 2304 cells and V^T.p, each launched back to back on one stream, as shipped and with one ingredient removed at a
// time (VAR).  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I llama_box_amd/csrc
// scripts/ubench/attn_probe.hip -o /tmp/attn_probe
#include "../../llama_box_amd/csrc/mmf.hip"
#include "../../llama_box_amd/csrc/ops.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
namespace mi355x {
int log_level() { return 0; }
// (stand-in for mmq_i8.hip's reduce pass, same access pattern: float4 slots, partials added in slice order)
__global__ void __launch_bounds__(256) p_reduce(const float * part, int ks, int64_t mn, float * dst) {
    const int64_t e = ((int64_t) blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= mn) return;
    float4 acc = *(const float4 *) (part + e);
    for (int k = 1; k < ks; ++k) {
        const float4 v = *(const float4 *) (part + (int64_t) k * mn + e);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *(float4 *) (dst + e) = acc;
}
void launch_splitk_reduce(hipStream_t s, const float * part, int ks, int M, int N, float * dst, int64_t, const float *, int64_t) {
    const int64_t mn = (int64_t) M * N;
    hipLaunchKernelGGL(p_reduce, dim3((unsigned) ((mn / 4 + 255) / 256)), dim3(256), 0, s, part, ks, mn, dst);
}
}
using namespace mi355x;

// copies of the shipped kernels with knock-outs
template <int VAR> __global__ void __launch_bounds__(256) p_short(const tdesc a, const tdesc b, const tdesc d, const int lpr) {
    constexpr int G = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rpw = 64 / lpr, sub = lane / lpr, sl = lane % lpr;
    const int64_t i02 = blockIdx.y, K = a.ne[0];
    const int64_t base = (int64_t) blockIdx.x * 16 * rpw + wave * rpw + sub;
    const bool kin = (int64_t) sl * 8 < K;
    float x[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (kin && VAR != 1) mmf_x8(b.data + (i02 * G + g) * b.nb[2] + (int64_t) sl * 32, x[g]);
        else
#pragma unroll
            for (int i = 0; i < 8; ++i) x[g][i] = 1.0f + lane;
    }
    float w[4][8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t row = std::min<int64_t>(base + 4 * rpw * r, a.ne[1] - 1);
        if (kin && VAR != 2) mmf_w8(a.data + row * a.nb[1] + i02 * a.nb[2] + (int64_t) sl * 16, w[r]);
        else
#pragma unroll
            for (int i = 0; i < 8; ++i) w[r][i] = 0.5f * lane;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t row = base + 4 * rpw * r;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc = fmaf(w[r][i], x[g][i], acc);
            if (VAR == 5) acc = group_sum<16>(acc);
            else if (VAR != 4) for (int o = lpr >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
            if (VAR == 3) { if (acc == 12345.678f) *(float *) d.data = acc; }
            else if (sl == 0 && row < a.ne[1]) *(float *) (d.data + row * d.nb[0] + (i02 * G + g) * d.nb[2]) = acc;
        }
    }
}
template <int VAR> __global__ void __launch_bounds__(256) p_soft(const tdesc a, const tdesc m, const tdesc d, const float scale) {
    constexpr int NR = 12;
    __shared__ double shd[4];
    __shared__ float shf[4];
    const int64_t row = blockIdx.x;
    const int64_t i1 = row % a.ne[1], i2 = (row / a.ne[1]) % a.ne[2], i3 = row / (a.ne[1] * a.ne[2]);
    const float * x = (const float *) (a.data + i1 * a.nb[1] + i2 * a.nb[2] + i3 * a.nb[3]);
    float * y = (float *) (d.data + i1 * d.nb[1] + i2 * d.nb[2] + i3 * d.nb[3]);
    const char * mp = VAR != 1 ? m.data + i1 * m.nb[1] : nullptr;
    const int n = (int) a.ne[0];
    float w[NR];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int i = threadIdx.x + 256 * j;
        w[j] = -INFINITY;
        if (i < n) {
            float v = x[i] * scale;
            if (mp) v += h2f(((const uint16_t *) mp)[i]);
            w[j] = v;
            mx = fmaxf(mx, v);
        }
    }
    if (VAR != 5) mx = block_max_f(mx, shf);
    double sum = 0.0;
    float fsum = 0.0f;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        if ((int) threadIdx.x + 256 * j < n) {
            w[j] = VAR == 3 ? w[j] - mx : expf(w[j] - mx);
            if (VAR == 2) fsum += w[j];
            else sum += (double) w[j];
        }
    }
    float inv;
    if (VAR == 2) {
        fsum = wave_sum(fsum);
        inv = 1.0f / fsum;
    } else {
        if (VAR != 5) sum = block_sum_d(sum, shd);
        if (isnan(sum) || sum == 0.0) sum = -INFINITY;
        inv = (float) (1.0 / sum);
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int i = threadIdx.x + 256 * j;
        if (VAR == 4) { if (w[j] * inv == 12345.678f) y[i] = w[j]; }
        else if (i < n) y[i] = w[j] * inv;
    }
}
__global__ void p_empty(int * p) { if (p && threadIdx.x == 1234567) *p = 1; }

static tdesc mk(void * data, int type, int64_t n0, int64_t n1, int64_t n2, int64_t b0, int64_t b1, int64_t b2) {
    tdesc t;
    t.data = (char *) data; t.type = type;
    t.ne[0] = n0; t.ne[1] = n1; t.ne[2] = n2; t.ne[3] = 1;
    t.nb[0] = b0; t.nb[1] = b1; t.nb[2] = b2; t.nb[3] = b2 * n2;
    return t;
}
template <typename F> static void timeit(const char * name, F f) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    const int it = 400;
    for (int i = 0; i < it; ++i) f();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-64s %8.2f us per call\n", name, ms * 1000.0f / it);
}
int main() {
    const int HD = 128, NKV = 8, NH = 32, NCTX = 2304, n_kv = 2304;
    char *kc, *vc, *q, *kq, *p, *mask, *out;
    CK(hipMalloc(&kc, (size_t) NCTX * NKV * HD * 2)); CK(hipMalloc(&vc, (size_t) NCTX * NKV * HD * 2));
    CK(hipMalloc(&q, NH * HD * 4)); CK(hipMalloc(&kq, (size_t) n_kv * NH * 4)); CK(hipMalloc(&p, (size_t) n_kv * NH * 4));
    CK(hipMalloc(&mask, (size_t) n_kv * 64 * 2)); CK(hipMalloc(&out, NH * HD * 4));
    CK(hipMemset(kc, 0x11, (size_t) NCTX * NKV * HD * 2)); CK(hipMemset(vc, 0x11, (size_t) NCTX * NKV * HD * 2));
    CK(hipMemset(q, 0, NH * HD * 4)); CK(hipMemset(kq, 0, (size_t) n_kv * NH * 4)); CK(hipMemset(mask, 0, (size_t) n_kv * 64 * 2));
    const tdesc k = mk(kc, GGML_TYPE_F16, HD, n_kv, NKV, 2, NKV * HD * 2, HD * 2);
    const tdesc v = mk(vc, GGML_TYPE_F16, n_kv, HD, NKV, 2, NCTX * 2, (int64_t) NCTX * 2 * HD);
    const tdesc tq = mk(q, GGML_TYPE_F32, HD, 1, NH, 4, HD * NH * 4, HD * 4);
    const tdesc tkq = mk(kq, GGML_TYPE_F32, n_kv, 1, NH, 4, n_kv * 4, n_kv * 4);
    const tdesc tp = mk(p, GGML_TYPE_F32, n_kv, 1, NH, 4, n_kv * 4, n_kv * 4);
    const tdesc tm = mk(mask, GGML_TYPE_F16, n_kv, 64, 1, 2, n_kv * 2, n_kv * 2 * 64);
    const tdesc to = mk(out, GGML_TYPE_F32, HD, 1, NH, 4, HD * 4, HD * 4);
    timeit("empty kernel (1 workgroup)", [&] { p_empty<<<1, 64>>>(nullptr); });
    timeit("empty kernel (288 workgroups x 256)", [&] { p_empty<<<288, 256>>>(nullptr); });
    timeit("shipped K.q   (launch_mul_mat_f)", [&] { launch_mul_mat_f(0, k, tq, tkq); });
    timeit("shipped soft_max", [&] { launch_soft_max(0, tkq, &tm, nullptr, tp, 0.088f, 0.0f); });
    timeit("shipped V^T.p (launch_mul_mat_f)", [&] { launch_mul_mat_f(0, v, tp, to); });
    timeit("shipped SOFT_MAX folded into V^T.p", [&] { launch_soft_max_mul_mat_f16(0, v, tkq, &tm, to, 0.088f); });
    timeit("shipped sequence K.q, folded soft_max + V^T.p", [&] { launch_mul_mat_f(0, k, tq, tkq); launch_soft_max_mul_mat_f16(0, v, tkq, &tm, to, 0.088f); });
    timeit("shipped sequence of the three", [&] { launch_mul_mat_f(0, k, tq, tkq); launch_soft_max(0, tkq, &tm, nullptr, tp, 0.088f, 0.0f); launch_mul_mat_f(0, v, tp, to); });
    const dim3 gs((n_kv + 63) / 64, NKV);
    timeit("K.q copy, shuffle butterfly (as first shipped)", [&] { p_short<0><<<gs, 256>>>(k, tq, tkq, 16); });
    timeit("K.q without the q loads", [&] { p_short<1><<<gs, 256>>>(k, tq, tkq, 16); });
    timeit("K.q without the K loads", [&] { p_short<2><<<gs, 256>>>(k, tq, tkq, 16); });
    timeit("K.q without the stores", [&] { p_short<3><<<gs, 256>>>(k, tq, tkq, 16); });
    timeit("K.q with DPP group sums", [&] { p_short<5><<<gs, 256>>>(k, tq, tkq, 16); });
    timeit("K.q without the shuffles", [&] { p_short<4><<<gs, 256>>>(k, tq, tkq, 16); });
    timeit("soft_max copy, as shipped", [&] { p_soft<0><<<NH, 256>>>(tkq, tm, tp, 0.088f); });
    timeit("soft_max without mask", [&] { p_soft<1><<<NH, 256>>>(tkq, tm, tp, 0.088f); });
    timeit("soft_max float sum, wave only", [&] { p_soft<2><<<NH, 256>>>(tkq, tm, tp, 0.088f); });
    timeit("soft_max without expf", [&] { p_soft<3><<<NH, 256>>>(tkq, tm, tp, 0.088f); });
    timeit("soft_max without stores", [&] { p_soft<4><<<NH, 256>>>(tkq, tm, tp, 0.088f); });
    timeit("soft_max without block reductions", [&] { p_soft<5><<<NH, 256>>>(tkq, tm, tp, 0.088f); });
    // ---- a -np 32 decode batch on the non-flash path: 32 tokens against 7296 cells
    {
        const int T = 32, ncell = 7296;
        char *kc2, *vc2, *q2, *kq2, *p2, *out2, *ws2;
        CK(hipMalloc(&kc2, (size_t) ncell * NKV * HD * 2)); CK(hipMalloc(&vc2, (size_t) ncell * NKV * HD * 2));
        CK(hipMalloc(&q2, (size_t) T * NH * HD * 4)); CK(hipMalloc(&kq2, (size_t) ncell * T * NH * 4)); CK(hipMalloc(&p2, (size_t) ncell * T * NH * 4));
        CK(hipMalloc(&out2, (size_t) T * NH * HD * 4)); CK(hipMalloc(&ws2, 64u << 20));
        CK(hipMemset(kc2, 0x11, (size_t) ncell * NKV * HD * 2)); CK(hipMemset(vc2, 0x11, (size_t) ncell * NKV * HD * 2));
        CK(hipMemset(q2, 0, (size_t) T * NH * HD * 4)); CK(hipMemset(kq2, 0, (size_t) ncell * T * NH * 4)); CK(hipMemset(p2, 0, (size_t) ncell * T * NH * 4));
        const tdesc k2 = mk(kc2, GGML_TYPE_F16, HD, ncell, NKV, 2, NKV * HD * 2, HD * 2);
        const tdesc v2 = mk(vc2, GGML_TYPE_F16, ncell, HD, NKV, 2, ncell * 2, (int64_t) ncell * 2 * HD);
        const tdesc tq2 = mk(q2, GGML_TYPE_F32, HD, T, NH, 4, HD * NH * 4, HD * 4);
        const tdesc tkq2 = mk(kq2, GGML_TYPE_F32, ncell, T, NH, 4, ncell * 4, (int64_t) ncell * 4 * T);
        const tdesc tp2 = mk(p2, GGML_TYPE_F32, ncell, T, NH, 4, ncell * 4, (int64_t) ncell * 4 * T);
        const tdesc to2 = mk(out2, GGML_TYPE_F32, HD, T, NH, 4, HD * 4, HD * 4 * T);
        const tdesc tm2 = mk(mask, GGML_TYPE_F16, n_kv, 64, 1, 2, n_kv * 2, n_kv * 2 * 64);
        (void) tm2;
        timeit("batch 32 x 7296: K.Q (launch_mul_mat_f)", [&] { launch_mul_mat_f(0, k2, tq2, tkq2); });
        timeit("batch 32 x 7296: K.Q tile kernel (no resident columns)", [&] { dim3 grid((ncell + 63) / 64, 1, NH); hipLaunchKernelGGL(k_mul_mat_f16_mma, grid, dim3(256), 0, 0, k2, tq2, tkq2); });
        timeit("batch 32 x 7296: soft_max (no mask)", [&] { launch_soft_max(0, tkq2, nullptr, nullptr, tp2, 0.088f, 0.0f); });
        timeit("batch 32 x 7296: V^T.p with scratch (K cut over workgroups + reduce)", [&] { launch_mul_mat_f(0, v2, tp2, to2, (float *) ws2, 64u << 20); });
        timeit("batch 32 x 7296: V^T.p without scratch (16 x 16 tiles, K over waves)", [&] { launch_mul_mat_f(0, v2, tp2, to2); });
        timeit("batch 32 x 7296: fill of the logits' size (30 MB memset)", [&] { CK(hipMemsetAsync(kq2, 0, (size_t) ncell * T * NH * 4, 0)); });
    }
    return 0;
}
