// grid_barrier.hip — what does a device-wide barrier inside one persistent kernel cost on MI355X (256 workgroups, one per CU),
// compared with the ~4.5 us a dependent kernel boundary costs?  Counter barrier with agent-scope atomics; a release/acquire pair
// around it (what a producer/consumer hand-over needs).  Spins are bounded: nothing can hang the GPU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <bool FENCE, bool DATA>
__global__ void __launch_bounds__(1024) k_barriers(unsigned * ctr, unsigned * err, float * buf, int iters, int nwg) {
    const int tid = threadIdx.x;
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
        if (DATA) {  // every workgroup writes 64 B, after the barrier reads what another workgroup wrote (cross-XCD hand-over)
            if (tid < 16) buf[(size_t) (it & 1) * nwg * 16 + blockIdx.x * 16 + tid] = (float) (it + blockIdx.x);
        }
        __syncthreads();
        if (tid == 0) {
            if (FENCE) __atomic_thread_fence(__ATOMIC_RELEASE);  // agent scope by default on amdgcn for __atomic builtins? use hip scope below
            __hip_atomic_fetch_add(ctr, 1u, FENCE ? __ATOMIC_RELEASE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned) (it + 1) * (unsigned) nwg;
            int spins = 0;
            while (__hip_atomic_load(ctr, FENCE ? __ATOMIC_ACQUIRE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > (1 << 16)) { atomicAdd(err, 1u); break; }
            }
        }
        __syncthreads();
        if (DATA) {
            const int other = (blockIdx.x + 37) % nwg;
            if (tid < 16) acc += __builtin_nontemporal_load(&buf[(size_t) (it & 1) * nwg * 16 + other * 16 + tid]);
        }
    }
    if (DATA && tid < 16 && acc == -1.0f) buf[0] = acc;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    unsigned * ctr; float * buf;
    CK(hipMalloc(&ctr, 256)); CK(hipMalloc(&buf, 2 * 256 * 16 * 4));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    for (int nwg : {64, 256}) {
        for (int mode = 0; mode < 3; ++mode) {
            float best = 1e30f; unsigned nerr = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemsetAsync(ctr, 0, 256, s));
                CK(hipEventRecord(e0, s));
                if (mode == 0) hipLaunchKernelGGL((k_barriers<false, false>), dim3(nwg), dim3(1024), 0, s, ctr, ctr + 1, buf, iters, nwg);
                if (mode == 1) hipLaunchKernelGGL((k_barriers<true, false>), dim3(nwg), dim3(1024), 0, s, ctr, ctr + 1, buf, iters, nwg);
                if (mode == 2) hipLaunchKernelGGL((k_barriers<true, true>), dim3(nwg), dim3(1024), 0, s, ctr, ctr + 1, buf, iters, nwg);
                CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
                CK(hipMemcpy(&nerr, ctr + 1, 4, hipMemcpyDeviceToHost));
            }
            printf("%3d workgroups  %-38s: %.2f us per barrier (spin timeouts %u)\n", nwg,
                   mode == 0 ? "relaxed counter barrier" : mode == 1 ? "release/acquire barrier" : "release/acquire + 64 B hand-over", best * 1e3f / iters, nerr);
        }
    }
    return 0;
}
