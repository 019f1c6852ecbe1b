// mmvq_lab2.hip — round 4: the product's batch-1 mat-vec (k_mmvq_stream, linked from llama_box_amd/build/mmvq.o) against the lab's
// block-column kernel (experiments/mmvq_v2.hip) in dependent chains inside a hipGraph, the way a decode step runs them; results compared
// by NMSE (the two sum a row's super-blocks in different orders).  Build + run: scripts/ubench/run_lab2.sh (through gpurun).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include "../../llama_box_amd/csrc/kernels.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
namespace mi355x { int log_level() { return 1; } bool launch_mmvq_v2(hipStream_t s, const mmvq_args & a); /* scripts/ubench/experiments/mmvq_v2.hip */ }
static int g_mode = 0;
static void mmvq_set_dma(int on) { g_mode = on; }
static void lab_launch(hipStream_t s, const mi355x::mmvq_args & a) { if (!g_mode || !mi355x::launch_mmvq_v2(s, a)) mi355x::launch_mmvq(s, a, 1); }
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
        {"q2 down   q6_K 18944x3584 +res f32pro", GGML_TYPE_Q6_K, 18944, 3584, false, false, true},
        {"tp8 wo    q4_K 1024x8192  +res f32pro", GGML_TYPE_Q4_K, 1024, 8192, false, false, true},
        {"tp8 gate  q4_K 8192x3584  glu normpro", GGML_TYPE_Q4_K, 8192, 3584, true, true, false},
        {"tp8 down  q4_K 3584x8192  +res f32pro", GGML_TYPE_Q4_K, 3584, 8192, false, false, true},
        {"small     q4_K 2048x1000  plain      ", GGML_TYPE_Q4_K, 2048, 1000, false, false, false},
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
        std::vector<float> h0(o.N), h1(o.N);
        CK(hipMemcpy(h0.data(), d0, (size_t) o.N * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), d1, (size_t) o.N * 4, hipMemcpyDeviceToHost));
        size_t bad = 0; int first = -1;
        double num = 0, den = 0;
        for (int i = 0; i < o.N; ++i) {
            const double df = (double) h0[i] - (double) h1[i];
            num += df * df; den += (double) h0[i] * (double) h0[i];
            if (!(fabs(df) <= 1e-4 * (fabs((double) h0[i]) + 1e-3))) { if (first < 0) first = i; ++bad; }
        }
        const double nmse = num / (den + 1e-30);
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
        printf("%s  %7.1f MB | reg-stream %7.2f us %5.2f TB/s | v2-cols %7.2f us %5.2f TB/s | nmse %.2e, outliers %zu/%d (first %d)\n", o.name, mb / 1048576.0,
               us[0], mb / us[0] / 1e6, us[1], mb / us[1] / 1e6, nmse, bad, o.N, first);
        fflush(stdout);
    }
    return 0;
}
