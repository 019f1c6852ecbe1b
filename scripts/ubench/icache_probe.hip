// icache_probe.hip — round 6: what does executed CODE SIZE cost a short dependent kernel?  The decode step is ~197 launches of 5-25 KB kernels that run once
// through most of their code (one KV trip, one prologue, one epilogue).  Same number of executed VALU instructions as straight-line code of KB kilobytes
// vs. as a loop over a 512-byte body; launches chained in a hipGraph (each reads what the previous wrote), 256 workgroups of WAVES waves.
// ALT: the chain alternates two distinct kernels of the same size (does the instruction cache survive a launch of the same kernel?).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// N fma instructions (8 bytes each as VOP3 with a literal-free encoding: operands are registers)
template <int N, bool LOOP, int TAG> __global__ void __launch_bounds__(512) k_code(float * __restrict__ p, const float c0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = p[i] + (float) (j + TAG);
    if constexpr (LOOP) {
#pragma unroll 1
        for (int r = 0; r < N / 64; ++r) {
#pragma unroll
            for (int k = 0; k < 64; ++k) x[k & 7] = __builtin_fmaf(x[k & 7], c0, x[(k + 3) & 7]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < N; ++k) x[k & 7] = __builtin_fmaf(x[k & 7], c0, x[(k + 3) & 7]);
    }
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += x[j];
    p[i] = s * 1e-30f;
}

template <int N, bool LOOP> static double run(hipStream_t s, float * d, int waves, bool alt) {
    const int n = 200;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
    for (int i = 0; i < n; ++i) {
        if (alt && (i & 1)) hipLaunchKernelGGL((k_code<N, LOOP, 1>), dim3(256), dim3(64 * waves), 0, s, d, 0.999f);
        else hipLaunchKernelGGL((k_code<N, LOOP, 0>), dim3(256), dim3(64 * waves), 0, s, d, 0.999f);
    }
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    double best = 1e30;
    for (int r = 0; r < 12; ++r) {
        const double t0 = now_us();
        CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        const double t = (now_us() - t0) / n;
        if (r >= 2 && t < best) best = t;
    }
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return best;
}

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float * d; CK(hipMalloc(&d, 256 * 512 * 4)); CK(hipMemset(d, 0, 256 * 512 * 4));
    printf("per launch, us (200 chained launches in a hipGraph, 256 workgroups); N fma = N x 8 bytes of code\n");
    printf("%8s %6s | %10s %10s %10s | %10s\n", "code KB", "waves", "straight", "loop", "delta", "alt-straight");
#define ROW(N)                                                                                              \
    for (int w : {1, 8}) {                                                                                  \
        const double a = run<N, false>(s, d, w, false), b = run<N, true>(s, d, w, false), c = run<N, false>(s, d, w, true); \
        printf("%8.1f %6d | %10.2f %10.2f %10.2f | %10.2f\n", N * 8 / 1024.0, w, a, b, a - b, c);        \
    }
    ROW(64) ROW(256) ROW(512) ROW(1024) ROW(2048) ROW(4096)
    return 0;
}
