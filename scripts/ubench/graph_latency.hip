// graph_latency.hip — launch-to-completion latency on an IDLE stream: one kernel launched directly vs the same kernel(s) as
// a hipGraph of n nodes (what a decode step pays once per token before its first kernel runs).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__global__ void k_tiny(float * p) { if (threadIdx.x == 0) p[blockIdx.x] += 1.0f; }
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float * d; CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
    const int reps = 200;
    {
        double tot = 0, tl = 0;
        for (int r = 0; r < reps + 10; ++r) {
            const double t0 = now_us();
            hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, d);
            const double t1 = now_us();
            CK(hipStreamSynchronize(s));
            const double t2 = now_us();
            if (r >= 10) { tot += t2 - t0; tl += t1 - t0; }
        }
        printf("direct launch  1 kernel : launch call %.1f us, launch->done %.1f us\n", tl / reps, tot / reps);
    }
    for (int n : {1, 8, 32, 200}) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, d);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        double tot = 0, tl = 0;
        for (int r = 0; r < reps + 10; ++r) {
            const double t0 = now_us();
            CK(hipGraphLaunch(ge, s));
            const double t1 = now_us();
            CK(hipStreamSynchronize(s));
            const double t2 = now_us();
            if (r >= 10) { tot += t2 - t0; tl += t1 - t0; }
        }
        printf("graph of %3d kernels    : launch call %.1f us, launch->done %.1f us (%.2f us/kernel)\n", n, tl / reps, tot / reps, tot / reps / n);
    }
    {   // eager chain of 200 for comparison
        double tot = 0, tl = 0;
        for (int r = 0; r < 30; ++r) {
            const double t0 = now_us();
            for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, d);
            const double t1 = now_us();
            CK(hipStreamSynchronize(s));
            const double t2 = now_us();
            if (r >= 10) { tot += t2 - t0; tl += t1 - t0; }
        }
        printf("eager 200 kernels       : launch calls %.1f us, launch->done %.1f us (%.2f us/kernel)\n", tl / 20, tot / 20, tot / 20 / 200);
    }
    return 0;
}
