// ubench.hip — micro-benchmarks that bound what a decode-step kernel can cost on MI355X (run through gpurun).
//   empty      : chain of empty kernels inside a hipGraph                      -> pure boundary cost
//   vec        : chain of 1-WG kernels, each reads the 16 KB the previous wrote -> dependent round trip + boundary
//   stream B U : chain of 256-WG x 1024-thread kernels, each streams B bytes of DISTINCT weights (16-byte loads, U in
//                flight per lane), depends on the previous kernel's 16 KB output and writes 16 KB -> what an ideal
//                weight-streaming launch of that size costs inside a dependent chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void k_empty() {}
__global__ void __launch_bounds__(1024) k_vec(const float4 * __restrict__ in, float4 * __restrict__ out) {
    float4 v = in[threadIdx.x];
    v.x += 1.0f;
    out[threadIdx.x] = v;
}
template <int U>
__global__ void __launch_bounds__(1024) k_stream(const uint4 * __restrict__ W, size_t n16, const float4 * __restrict__ x, float4 * __restrict__ out) {
    const size_t T = (size_t) gridDim.x * 1024;
    size_t i = (size_t) blockIdx.x * 1024 + threadIdx.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = (i + u * T) < n16 ? W[i + u * T] : make_uint4(0, 0, 0, 0);
    const float4 xv = x[threadIdx.x];  // dependency on the previous kernel
    for (;;) {
        const size_t ni = i + U * T;
        uint4 nv[U];
        const bool more = ni < n16;
        if (more) {
#pragma unroll
            for (int u = 0; u < U; ++u) nv[u] = (ni + u * T) < n16 ? W[ni + u * T] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x ^= v[u].x; acc.y += v[u].y; acc.z ^= v[u].z; acc.w += v[u].w; }
        if (!more) break;
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = nv[u];
        i = ni;
    }
    if (blockIdx.x == 0) {
        float4 o = xv;
        o.x += (float) (acc.x & 1) + (float) (acc.y & 1) + (float) (acc.z & 1) + (float) (acc.w & 1);
        out[threadIdx.x] = o;
    } else if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) {
        out[threadIdx.x] = xv;  // never (keeps the loads alive)
    }
}

static double run_graph(hipStream_t s, hipGraphExec_t ge, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / reps;
}

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int CH = 64;
    float4 *a, *b; CK(hipMalloc(&a, 16384)); CK(hipMalloc(&b, 16384)); CK(hipMemset(a, 0, 16384)); CK(hipMemset(b, 0, 16384));
    const size_t pool_bytes = (size_t) 6 << 30;
    uint4 * pool; CK(hipMalloc(&pool, pool_bytes)); CK(hipMemset(pool, 1, pool_bytes));
    auto capture = [&](auto body) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
        body();
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        return ge;
    };
    {
        auto ge = capture([&] { for (int i = 0; i < CH; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); });
        printf("empty chain            : %.2f us/kernel\n", run_graph(s, ge, 20) / CH);
    }
    {
        auto ge = capture([&] { for (int i = 0; i < CH; ++i) hipLaunchKernelGGL(k_vec, dim3(1), dim3(1024), 0, s, (i & 1) ? b : a, (i & 1) ? a : b); });
        printf("dependent 16KB vec     : %.2f us/kernel\n", run_graph(s, ge, 20) / CH);
    }
    const size_t sizes[] = {(size_t) 2400 << 10, (size_t) 9437184, (size_t) 33 << 20, (size_t) 66 << 20, (size_t) 430 << 20};
    for (size_t B : sizes) {
        const int nk = B > ((size_t) 100 << 20) ? 8 : CH;
        for (int U : {1, 2, 4, 8}) {
            auto ge = capture([&] {
                size_t off = 0;
                for (int i = 0; i < nk; ++i) {
                    if (off + B > pool_bytes) off = 0;
                    const uint4 * W = pool + off / 16;
                    off += B;
                    const float4 * x = (i & 1) ? b : a; float4 * o = (i & 1) ? a : b;
                    for (int grid : {256}) {
                        if (U == 1) hipLaunchKernelGGL(k_stream<1>, dim3(grid), dim3(1024), 0, s, W, B / 16, x, o);
                        if (U == 2) hipLaunchKernelGGL(k_stream<2>, dim3(grid), dim3(1024), 0, s, W, B / 16, x, o);
                        if (U == 4) hipLaunchKernelGGL(k_stream<4>, dim3(grid), dim3(1024), 0, s, W, B / 16, x, o);
                        if (U == 8) hipLaunchKernelGGL(k_stream<8>, dim3(grid), dim3(1024), 0, s, W, B / 16, x, o);
                    }
                }
            });
            const double us = run_graph(s, ge, 5) / nk;
            printf("stream %7.1f MB U=%d    : %.2f us/kernel  -> %.2f TB/s\n", B / 1048576.0, U, us, B / us / 1e6);
        }
    }
    // the same kernel over a working set that is re-used: 192 MB cycle = resident in the 256 MB Infinity Cache after the first
    // pass, 24 MB cycle = resident in the 8 x 4 MB L2s (what a weight prefetch one layer ahead could buy)
    for (size_t B : {(size_t) 9437184, (size_t) 33 << 20, (size_t) 66 << 20}) {
        for (size_t cyc : {pool_bytes, (size_t) 192 << 20, (size_t) 24 << 20}) {
            if (cyc < B) continue;
            auto ge = capture([&] {
                size_t off = 0;
                for (int i = 0; i < CH; ++i) {
                    if (off + B > cyc) off = 0;
                    const uint4 * W = pool + off / 16;
                    off += B;
                    const float4 * x = (i & 1) ? b : a; float4 * o = (i & 1) ? a : b;
                    hipLaunchKernelGGL(k_stream<4>, dim3(256), dim3(1024), 0, s, W, B / 16, x, o);
                }
            });
            const double us = run_graph(s, ge, 8) / CH;
            printf("reuse  %7.1f MB cycle %6.0f MB : %.2f us/kernel  -> %.2f TB/s\n", B / 1048576.0, cyc / 1048576.0, us, B / us / 1e6);
        }
    }
    return 0;
}
