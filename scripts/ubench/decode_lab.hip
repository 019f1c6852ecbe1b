// decode_lab.hip — times the REAL decode mat-vec launchers (linked from llama_box_amd/build/*.o) in dependent chains inside a
// hipGraph, the way a decode step runs them, and checks the LDS-DMA kernels against round 1's register-streaming kernels bit
// for bit on the same inputs.  Build + run: scripts/ubench/run_lab.sh (through gpurun).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include "../../llama_box_amd/csrc/kernels.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
namespace mi355x { int log_level() { return 1; } int mask_sparse_hint(const void *) { return 0; } /* (backend.cpp's mask statistics: none in the lab) */ bool launch_mmvq_dma(hipStream_t s, const mmvq_args & a); /* scripts/ubench/experiments/mmvq_dma.hip */ }
static int g_mode = 0;
static void mmvq_set_dma(int on) { g_mode = on; }
static void lab_launch(hipStream_t s, const mi355x::mmvq_args & a) { if (!g_mode || !mi355x::launch_mmvq_dma(s, a)) mi355x::launch_mmvq(s, a, 1); }
using namespace mi355x;

__global__ void k_fill(uint32_t * p, size_t n, uint32_t seed) {
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    const size_t T = (size_t) gridDim.x * blockDim.x;
    for (; i < n; i += T) {
        uint32_t x = (uint32_t) i * 2654435761u ^ seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        p[i] = x;
    }
}
// sane f16 scales in every block header: d (and dmin) = small positive halves so that results stay finite
__global__ void k_fix(uint8_t * w, size_t nblocks, int bytes, int d_off, int has_dmin) {
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nblocks) return;
    uint16_t * d = (uint16_t *) (w + i * bytes + d_off);
    d[0] = (uint16_t) (0x1C00 + (i & 0xFF));  // ~0.004
    if (has_dmin) d[1] = (uint16_t) (0x1800 + (i & 0x7F));
}
__global__ void k_fillf(float * p, size_t n, uint32_t seed) {
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x = (uint32_t) i * 2654435761u ^ seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15;
    p[i] = ((float) (x & 0xFFFF) / 32768.0f - 1.0f);
}

struct wt { int type, blk, bytes, d_off, dmin; };
static wt WT(int type) {
    switch (type) {
        case GGML_TYPE_Q4_K: return {type, 256, 144, 0, 1};
        case GGML_TYPE_Q5_K: return {type, 256, 176, 0, 1};
        case GGML_TYPE_Q6_K: return {type, 256, 210, 208, 0};
        default: return {GGML_TYPE_Q8_0, 32, 34, 0, 0};
    }
}
static size_t mat_bytes(int type, int K, int N) { wt t = WT(type); return (size_t) N * (K / t.blk) * t.bytes; }

struct op { const char * name; int type, K, N; bool glu, norm, res; };

int main(int argc, char ** argv) {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const size_t pool_bytes = (size_t) 5 << 30;
    uint8_t * pool; CK(hipMalloc(&pool, pool_bytes));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, s, (uint32_t *) pool, pool_bytes / 4, 12345u);
    CK(hipStreamSynchronize(s));
    float *xa, *xb, *nw, *r0; const size_t VMAX = 160000;
    CK(hipMalloc(&xa, VMAX * 4)); CK(hipMalloc(&xb, VMAX * 4)); CK(hipMalloc(&nw, VMAX * 4)); CK(hipMalloc(&r0, VMAX * 4));
    hipLaunchKernelGGL(k_fillf, dim3((VMAX + 255) / 256), dim3(256), 0, s, xa, VMAX, 1u);
    hipLaunchKernelGGL(k_fillf, dim3((VMAX + 255) / 256), dim3(256), 0, s, xb, VMAX, 2u);
    hipLaunchKernelGGL(k_fillf, dim3((VMAX + 255) / 256), dim3(256), 0, s, nw, VMAX, 3u);
    hipLaunchKernelGGL(k_fillf, dim3((VMAX + 255) / 256), dim3(256), 0, s, r0, VMAX, 4u);
    CK(hipStreamSynchronize(s));
    std::vector<op> ops = {
        {"wo        q4_K 4096x4096  +res f32pro", GGML_TYPE_Q4_K, 4096, 4096, false, false, true},
        {"gate/up   q4_K 4096x14336 glu normpro", GGML_TYPE_Q4_K, 4096, 14336, true, true, false},
        {"down      q4_K 14336x4096 +res f32pro", GGML_TYPE_Q4_K, 14336, 4096, false, false, true},
        {"down      q6_K 14336x4096 +res f32pro", GGML_TYPE_Q6_K, 14336, 4096, false, false, true},
        {"output    q6_K 4096x128256 normpro   ", GGML_TYPE_Q6_K, 4096, 128256, false, true, false},
        {"tl ffn    q8_0 2048x5632  glu normpro", GGML_TYPE_Q8_0, 2048, 5632, true, true, false},
        {"tl down   q8_0 5632x2048  +res f32pro", GGML_TYPE_Q8_0, 5632, 2048, false, false, true},
        {"70b gate  q4_K 8192x28672 glu normpro", GGML_TYPE_Q4_K, 8192, 28672, true, true, false},
        {"70b down  q6_K 28672x8192 +res f32pro", GGML_TYPE_Q6_K, 28672, 8192, false, false, true},
        {"q2 gate   q5_K 3584x18944 glu normpro", GGML_TYPE_Q5_K, 3584, 18944, true, true, false},
    };
    auto make_args = [&](const op & o, const uint8_t * W, const float * x, float * dst) {
        mmvq_args a{};
        a.W = W; a.W2 = o.glu ? W + mat_bytes(o.type, o.K, o.N) : nullptr;
        a.w_nb1 = (int64_t) (o.K / WT(o.type).blk) * WT(o.type).bytes;
        a.type = o.type; a.K = o.K; a.N = o.N; a.ncols = 1;
        a.dst = dst; a.dst_stride = o.N;
        a.add = o.res ? r0 : nullptr;
        a.x = x; a.norm_w = o.norm ? nw : nullptr; a.eps = 1e-5f; a.norm_out = nullptr;
        return a;
    };
    for (const op & o : ops) {
        const wt t = WT(o.type);
        const size_t mb = mat_bytes(o.type, o.K, o.N) * (o.glu ? 2 : 1);
        const int chain = (int) std::min<size_t>(32, pool_bytes / mb);
        // header fix-up over the whole pool for this format
        const size_t nblocks = pool_bytes / t.bytes;
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, s, (uint32_t *) pool, pool_bytes / 4, 777u + (uint32_t) o.type);
        hipLaunchKernelGGL(k_fix, dim3((unsigned) ((nblocks + 255) / 256)), dim3(256), 0, s, pool, nblocks, t.bytes, t.d_off, t.dmin);
        CK(hipStreamSynchronize(s));
        // ---- correctness: DMA kernel vs register-streaming kernel, same inputs, bitwise
        float * d0, * d1; CK(hipMalloc(&d0, (size_t) o.N * 4)); CK(hipMalloc(&d1, (size_t) o.N * 4));
        CK(hipMemsetAsync(d0, 0xFF, (size_t) o.N * 4, s)); CK(hipMemsetAsync(d1, 0xEE, (size_t) o.N * 4, s));
        mmvq_set_dma(0); lab_launch(s, make_args(o, pool, xa, d0));
        mmvq_set_dma(1); lab_launch(s, make_args(o, pool, xa, d1));
        CK(hipStreamSynchronize(s));
        std::vector<uint32_t> h0(o.N), h1(o.N);
        CK(hipMemcpy(h0.data(), d0, (size_t) o.N * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), d1, (size_t) o.N * 4, hipMemcpyDeviceToHost));
        size_t bad = 0; int first = -1;
        for (int i = 0; i < o.N; ++i) if (h0[i] != h1[i]) { if (first < 0) first = i; ++bad; }
        CK(hipFree(d0)); CK(hipFree(d1));
        // ---- timing: chain of `chain` launches over distinct weights, each reading what the previous one wrote
        double us[2] = {0, 0};
        for (int mode = 0; mode < 2; ++mode) {
            mmvq_set_dma(mode);
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
            for (int i = 0; i < chain; ++i) lab_launch(s, make_args(o, pool + (size_t) i * mb, (i & 1) ? xb : xa, (i & 1) ? xa : xb));
            CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
            const int reps = 10;
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            us[mode] = ms * 1e3 / reps / chain;
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
            // restore finite activations for the next mode
            hipLaunchKernelGGL(k_fillf, dim3((VMAX + 255) / 256), dim3(256), 0, s, xa, VMAX, 1u);
            hipLaunchKernelGGL(k_fillf, dim3((VMAX + 255) / 256), dim3(256), 0, s, xb, VMAX, 2u);
            CK(hipStreamSynchronize(s));
        }
        printf("%s  %7.1f MB | reg-stream %7.2f us %5.2f TB/s | lds-dma %7.2f us %5.2f TB/s | bitwise mismatches %zu/%d (first %d)\n", o.name, mb / 1048576.0,
               us[0], mb / us[0] / 1e6, us[1], mb / us[1] / 1e6, bad, o.N, first);
        fflush(stdout);
    }
    // ---- fused Q/K/V + rope + cache store (qkv.hip) and decode attention (fattn.hip) in the same kind of dependent chain
    {
        const int K = 4096, NQ = 4096, NKV = 1024, HD = 128, NCTX = 2304;
        uint16_t * kc, * vc; CK(hipMalloc(&kc, (size_t) NCTX * NKV * 2)); CK(hipMalloc(&vc, (size_t) NCTX * NKV * 2));
        hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, s, (uint32_t *) kc, (size_t) NCTX * NKV / 2, 5u);
        hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, s, (uint32_t *) vc, (size_t) NCTX * NKV / 2, 6u);
        int32_t hpos = 2100; int64_t hslot = 2100;
        int32_t * dpos; int64_t * dslot; CK(hipMalloc(&dpos, 4)); CK(hipMalloc(&dslot, 8));
        CK(hipMemcpy(dpos, &hpos, 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dslot, &hslot, 8, hipMemcpyHostToDevice));
        for (int tv : {GGML_TYPE_Q4_K, GGML_TYPE_Q6_K}) {
            const wt ta = WT(GGML_TYPE_Q4_K), tb = WT(tv);
            const size_t bq = mat_bytes(GGML_TYPE_Q4_K, K, NQ), bk = mat_bytes(GGML_TYPE_Q4_K, K, NKV), bv = mat_bytes(tv, K, NKV);
            const size_t mb = bq + bk + bv;
            hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, s, (uint32_t *) pool, pool_bytes / 4, 31u + (uint32_t) tv);
            hipLaunchKernelGGL(k_fix, dim3((unsigned) ((pool_bytes / ta.bytes + 255) / 256)), dim3(256), 0, s, pool, pool_bytes / ta.bytes, ta.bytes, ta.d_off, ta.dmin);
            CK(hipStreamSynchronize(s));
            const int chain = 32;
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
            for (int i = 0; i < chain; ++i) {
                qkv_args a{};
                const uint8_t * base = pool + (size_t) i * mb;
                a.seg[0] = {base, (int64_t) (K / 256) * ta.bytes, NQ, 0, 1, 0, 0, nullptr, (char *) ((i & 1) ? xa : xb), 0};
                a.seg[1] = {base + bq, (int64_t) (K / 256) * ta.bytes, NKV, 0, 1, 1, 0, nullptr, (char *) kc, (int64_t) NKV * 2};
                a.seg[2] = {base + bq + bk, (int64_t) (K / 256) * tb.bytes, NKV, (uint8_t) (tv == GGML_TYPE_Q4_K ? 0 : 1), 0, 1, 0, nullptr, (char *) vc, (int64_t) NKV * 2};
                a.nseg = 3; a.K = K; a.x = (i & 1) ? xb : xa; a.norm_w = nw; a.eps = 1e-5f; a.norm_out = nullptr;
                a.head_dim = HD; a.neox = 0; a.pos = dpos; a.freq_factors = nullptr;
                rope_params rp{HD, 0, 8192, 500000.0f, 1.0f, 0.0f, 1.0f, 32.0f, 1.0f, {0, 0, 0, 0}};
                rope_host_consts(rp, a.theta_scale, a.corr0, a.corr1);
                a.freq_scale = 1.0f; a.ext_factor = 0.0f; a.attn_factor = 1.0f; a.slot = dslot;
                launch_qkv(s, a, GGML_TYPE_Q4_K, tv);
            }
            CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < 10; ++r) CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / 10 / chain;
            printf("qkv+rope+store q4_K/%s 4096 -> 4096+1024+1024  %7.1f MB | %7.2f us %5.2f TB/s\n", tv == GGML_TYPE_Q4_K ? "q4_K" : "q6_K", mb / 1048576.0, us, mb / us / 1e6);
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
            hipLaunchKernelGGL(k_fillf, dim3((VMAX + 255) / 256), dim3(256), 0, s, xa, VMAX, 1u);
            hipLaunchKernelGGL(k_fillf, dim3((VMAX + 255) / 256), dim3(256), 0, s, xb, VMAX, 2u);
        }
        // attention: q [128, 1, 32] f32 (written by the previous launch of the chain), K/V f16 views [128, n_kv, 8]
        for (int n_kv : {2304, 8192}) {
            uint16_t * kc2, * vc2; CK(hipMalloc(&kc2, (size_t) n_kv * NKV * 2 * 8)); CK(hipMalloc(&vc2, (size_t) n_kv * NKV * 2 * 8));
            hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, s, (uint32_t *) kc2, (size_t) n_kv * NKV * 8 / 2, 7u);
            hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, s, (uint32_t *) vc2, (size_t) n_kv * NKV * 8 / 2, 8u);
            uint16_t * mask; CK(hipMalloc(&mask, (size_t) n_kv * 64 * 2)); CK(hipMemset(mask, 0, (size_t) n_kv * 64 * 2));
            void * ws; CK(hipMalloc(&ws, 64u << 20));
            for (int splits : {0, 16, 32, 64}) {
                const int chain = 8;
                hipGraph_t g; hipGraphExec_t ge;
                CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
                for (int i = 0; i < chain; ++i) {
                    tdesc q{(char *) ((i & 1) ? xb : xa), {HD, 1, 32, 1}, {4, HD * 32 * 4, HD * 4, HD * 32 * 4}, GGML_TYPE_F32};
                    tdesc k{(char *) (kc2 + (size_t) i * n_kv * NKV), {HD, n_kv, 8, 1}, {2, NKV * 2, HD * 2, (int64_t) n_kv * NKV * 2}, GGML_TYPE_F16};
                    tdesc v{(char *) (vc2 + (size_t) i * n_kv * NKV), {HD, n_kv, 8, 1}, {2, NKV * 2, HD * 2, (int64_t) n_kv * NKV * 2}, GGML_TYPE_F16};
                    tdesc m{(char *) mask, {n_kv, 64, 1, 1}, {2, (int64_t) n_kv * 2, (int64_t) n_kv * 128, (int64_t) n_kv * 128}, GGML_TYPE_F16};
                    tdesc d{(char *) ((i & 1) ? xa : xb), {HD, 32, 1, 1}, {4, HD * 4, HD * 32 * 4, HD * 32 * 4}, GGML_TYPE_F32};
                    fattn_params p{};
                    p.scale = 0.0883883f; p.max_bias = 0.0f; p.logit_softcap = 0.0f; p.kv_type = GGML_TYPE_F16;
                    p.n_splits = splits ? splits : fattn_pick_splits(q, k);
                    launch_flash_attn(s, q, k, v, &m, nullptr, d, p, ws);
                }
                CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
                CK(hipEventRecord(e0, s));
                for (int r = 0; r < 10; ++r) CK(hipGraphLaunch(ge, s));
                CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double us = ms * 1e3 / 10 / chain, mbv = (double) n_kv * NKV * 2 * 2 / 1048576.0;
                tdesc q0{(char *) xa, {HD, 1, 32, 1}, {4, HD * 32 * 4, HD * 4, HD * 32 * 4}, GGML_TYPE_F32};
                tdesc k0{(char *) kc2, {HD, n_kv, 8, 1}, {2, NKV * 2, HD * 2, (int64_t) n_kv * NKV * 2}, GGML_TYPE_F16};
                printf("attention d128 32/8 heads n_kv=%5d splits=%2d (split + combine launches)  %6.1f MB | %7.2f us %5.2f TB/s\n", n_kv, splits ? splits : fattn_pick_splits(q0, k0), mbv, us, mbv * 1.048576 / us);
                CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
                hipLaunchKernelGGL(k_fillf, dim3((VMAX + 255) / 256), dim3(256), 0, s, xa, VMAX, 1u);
                hipLaunchKernelGGL(k_fillf, dim3((VMAX + 255) / 256), dim3(256), 0, s, xb, VMAX, 2u);
            }
            CK(hipFree(kc2)); CK(hipFree(vc2)); CK(hipFree(mask)); CK(hipFree(ws));
        }
    }
    return 0;
}
