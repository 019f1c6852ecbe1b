// prologue_probe.hip — precise timeline of the activation prologue every decode mat-vec launch starts with (RMS_NORM * w -> Q8_K
// in LDS), inside a dependent chain of launches: each launch reads the 16 KB row the previous one wrote, exactly like a decode layer.
// Stamps are s_memrealtime (100 MHz) taken by inline asm that consumes / produces the values around it, so their position in the
// instruction stream is fixed.  Variants: NW = 16-byte weight loads per lane put in flight right after the x loads (0 / 6 / 12),
// HOLD = issue those weight loads only after x has arrived.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../llama_box_amd/csrc/dev_util.h"
#include "../../llama_box_amd/csrc/common.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
namespace mi355x { int log_level() { return 1; } }
using namespace mi355x;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned long long now() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
template <int NW, bool HOLD>
__global__ void __launch_bounds__(1024) k_probe(const float * __restrict__ x, const float * __restrict__ nw, float * __restrict__ out, const u32x4 * __restrict__ W,
                                                unsigned long long * __restrict__ st, const int slot) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long t[8];
    t[0] = now();
    const float4 * x4 = (const float4 *) x;
    const float4 * w4 = (const float4 *) nw;
    float4 v = x4[wave * 64 + lane], g = w4[wave * 64 + lane];
    u32x4 wv[NW > 0 ? NW : 1];
    const u32x4 * wp = W + ((size_t) blockIdx.x * 1024 + tid) * (NW > 0 ? NW : 1);
    if constexpr (!HOLD && NW > 0) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NW; ++i) wv[i] = __builtin_nontemporal_load(wp + i);
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w), "v"(g.x), "v"(g.y), "v"(g.z), "v"(g.w));  // x and w have arrived
    t[1] = now();
    if constexpr (HOLD && NW > 0) {
#pragma unroll
        for (int i = 0; i < NW; ++i) wv[i] = __builtin_nontemporal_load(wp + i);
        __builtin_amdgcn_sched_barrier(0);
    }
    double ss = (double) (v.x * v.x) + (double) (v.y * v.y) + (double) (v.z * v.z) + (double) (v.w * v.w);
    ss = wave_sum_d(ss);
    double * red = (double *) (smem + 16 * sizeof(q8k_dev));
    if (lane == 0) red[wave] = ss;
    asm volatile("" ::"v"(ss));
    t[2] = now();
    __syncthreads();
    t[3] = now();
    double tot = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) tot += red[i];
    const float mean = (float) (tot / 4096.0);
    const float scale = 1.0f / sqrtf(mean + 1e-5f);
    asm volatile("" ::"v"(scale));
    t[4] = now();
    float tt[4] = {(v.x * scale) * g.x, (v.y * scale) * g.y, (v.z * scale) * g.z, (v.w * scale) * g.w};
    wave_quantize_q8_K(tt, lane, (q8k_dev *) smem + wave);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    t[5] = now();
    __syncthreads();
    t[6] = now();
    uint32_t acc = ((const uint32_t *) smem)[tid];
    if constexpr (NW > 0) {
#pragma unroll
        for (int i = 0; i < NW; ++i) acc ^= wv[i].x ^ wv[i].y ^ wv[i].z ^ wv[i].w;
    }
    asm volatile("" ::"v"(acc));
    t[7] = now();
    if (blockIdx.x == 0 && wave < 16) out[tid * 4 + (acc & 3)] = v.x + (float) (acc & 1);  // next launch's x depends on this one (16 KB written)
    if (blockIdx.x == 0) { out[tid * 4 + 1] = v.y; out[tid * 4 + 2] = v.z; out[tid * 4 + 3] = v.w; out[tid * 4] = v.x; }
    if (lane == 0 && (wave == 0 || wave == 15))
        for (int k = 0; k < 8; ++k) st[(((size_t) slot * 256 + blockIdx.x) * 2 + (wave ? 1 : 0)) * 8 + k] = t[k];
}

template <int NW, bool HOLD> static void run(hipStream_t s, const char * name, float * xa, float * xb, float * nw, const u32x4 * W, unsigned long long * st) {
    const int CH = 10;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
    for (int i = 0; i < CH; ++i)
        hipLaunchKernelGGL((k_probe<NW, HOLD>), dim3(256), dim3(1024), 16 * sizeof(q8k_dev) + 256, s, (i & 1) ? xb : xa, nw, (i & 1) ? xa : xb,
                           W + (size_t) i * 256 * 1024 * (NW > 0 ? NW : 1), st, i);
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < 10; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t) CH * 256 * 2 * 8);
    CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
    // launch 5: per stamp min / median / max over (workgroup, wave 0|15), relative to the earliest entry
    const char * nm[8] = {"entry", "x-arrived", "ss-reduced", "barrier1", "scale", "quantised", "barrier2", "weights-arrived"};
    printf("%-34s %6.2f us/launch |", name, ms * 1e3 / 10 / CH);
    const int L = 5;
    unsigned long long t0 = ~0ull, pe = 0;
    for (int b = 0; b < 512; ++b) { t0 = std::min(t0, h[((size_t) L * 512 + b) * 8]); pe = std::max(pe, h[((size_t) (L - 1) * 512 + b) * 8 + 7]); }
    printf(" gap-after-prev %.2f |", (double) ((long long) t0 - (long long) pe) / 100.0);
    for (int k = 0; k < 8; ++k) {
        std::vector<double> c;
        for (int b = 0; b < 512; ++b) c.push_back((double) (h[((size_t) L * 512 + b) * 8 + k] - t0) / 100.0);
        std::sort(c.begin(), c.end());
        printf(" %s %.2f/%.2f/%.2f |", nm[k], c.front(), c[c.size() / 2], c.back());
    }
    printf("\n");
}

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float *xa, *xb, *nw; CK(hipMalloc(&xa, 65536)); CK(hipMalloc(&xb, 65536)); CK(hipMalloc(&nw, 65536));
    std::vector<float> h(16384);
    for (int i = 0; i < 16384; ++i) h[i] = (float) ((i * 7919) % 1000) / 500.0f - 1.0f;
    CK(hipMemcpy(xa, h.data(), 65536, hipMemcpyHostToDevice)); CK(hipMemcpy(xb, h.data(), 65536, hipMemcpyHostToDevice)); CK(hipMemcpy(nw, h.data(), 65536, hipMemcpyHostToDevice));
    u32x4 * W; CK(hipMalloc(&W, (size_t) 10 * 256 * 1024 * 12 * 16 + 4096)); CK(hipMemset(W, 1, (size_t) 10 * 256 * 1024 * 12 * 16));
    unsigned long long * st; CK(hipMalloc(&st, (size_t) 10 * 256 * 2 * 8 * 8)); CK(hipMemset(st, 0, (size_t) 10 * 256 * 2 * 8 * 8));
    run<0, false>(s, "prologue only", xa, xb, nw, W, st);
    run<3, false>(s, "prologue + 3x16B/lane weights", xa, xb, nw, W, st);
    run<6, false>(s, "prologue + 6x16B/lane weights", xa, xb, nw, W, st);
    run<12, false>(s, "prologue + 12x16B/lane weights", xa, xb, nw, W, st);
    run<6, true>(s, "weights held until x arrived (6)", xa, xb, nw, W, st);
    run<12, true>(s, "weights held until x arrived (12)", xa, xb, nw, W, st);
    return 0;
}
