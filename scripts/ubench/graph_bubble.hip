// graph_bubble.hip — does hipGraphLaunch make the GPU wait for ALL of a graph's packets before the first kernel runs?
// Graphs of n kernels that each spin for ~10 us: (launch -> done) - n * 10 us is what the launch adds; if it grows with n, a
// decode step (~200 kernels) pays that bubble once per token, and launching the step as a short graph followed by the rest
// (whose launch then overlaps the short one's execution) removes it.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__global__ void k_spin(float * p, long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (threadIdx.x == 0) p[blockIdx.x] += 1.0f;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static hipGraphExec_t make(hipStream_t s, float * d, int n, long long ticks) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, d, ticks);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    return ge;
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float * d; CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
    const long long ticks = 1000;  // wall_clock64 runs at 100 MHz: 10 us
    const int reps = 40;
    double t_kernel = 0;
    for (int n : {1, 8, 32, 200}) {
        hipGraphExec_t ge = make(s, d, n, ticks);
        double tot = 0;
        for (int r = 0; r < reps + 5; ++r) {
            const double t0 = now_us();
            CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
            if (r >= 5) tot += now_us() - t0;
        }
        tot /= reps;
        if (n == 1) t_kernel = 10.0;
        printf("graph of %3d x 10 us kernels: launch->done %8.1f us, minus kernels %7.1f us\n", n, tot, tot - n * t_kernel);
    }
    {   // 200 kernels as 8 + 32 + 160
        hipGraphExec_t a = make(s, d, 8, ticks), b = make(s, d, 32, ticks), c = make(s, d, 160, ticks);
        double tot = 0;
        for (int r = 0; r < reps + 5; ++r) {
            const double t0 = now_us();
            CK(hipGraphLaunch(a, s)); CK(hipGraphLaunch(b, s)); CK(hipGraphLaunch(c, s));
            CK(hipStreamSynchronize(s));
            if (r >= 5) tot += now_us() - t0;
        }
        tot /= reps;
        printf("200 kernels as graphs of 8 + 32 + 160: launch->done %8.1f us, minus kernels %7.1f us\n", tot, tot - 200 * 10.0);
    }
    {   // eager
        double tot = 0;
        for (int r = 0; r < reps + 5; ++r) {
            const double t0 = now_us();
            for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, d, ticks);
            CK(hipStreamSynchronize(s));
            if (r >= 5) tot += now_us() - t0;
        }
        tot /= reps;
        printf("200 kernels launched one by one      : launch->done %8.1f us, minus kernels %7.1f us\n", tot, tot - 200 * 10.0);
    }
    return 0;
}
