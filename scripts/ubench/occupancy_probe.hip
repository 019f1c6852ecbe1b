// occupancy_probe.hip — do TWO workgroups of a given shape really share a CU?  (round 4: k_mmq_wide with a 76 KB workgroup was given 2 blocks/CU by
// hipOccupancyMaxActiveBlocksPerMultiprocessor, yet PMC showed one resident at a time.)  A census: every workgroup arrives at a counter and waits (bounded)
// until all gridDim.x have arrived; the largest count any workgroup saw = the number resident together.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int THREADS, int WPE, int REGS>
__global__ void __launch_bounds__(THREADS, WPE) k_census(unsigned * cnt, unsigned * seen, const int lds_bytes) {
    extern __shared__ char smem[];
    // (a clobbered high register makes the kernel's allocation reach it: REGS = 0 few, 1 -> v167, 2 -> v127, 3 -> v239)
    if (REGS == 1) asm volatile("" ::: "v167");
    if (REGS == 2) asm volatile("" ::: "v127");
    if (REGS == 3) asm volatile("" ::: "v239");
    if (threadIdx.x == 0) {
        smem[lds_bytes - 1] = 1;
        // live count: + 1 on arrival, - 1 on leaving; a workgroup waits (bounded) until everybody is there, then leaves
        unsigned c = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        unsigned mx = c;
        for (int spin = 0; spin < 20000 && __hip_atomic_load(cnt + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0; ++spin) {
            c = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            mx = c > mx ? c : mx;
            if (c >= gridDim.x) { __hip_atomic_store(cnt + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            __builtin_amdgcn_s_sleep(8);
        }
        __hip_atomic_fetch_sub(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        c = mx;
        atomicMax(seen, c);
    }
    __syncthreads();
}

template <int THREADS, int WPE, int REGS> static void run(const char * name, int lds, int grid) {
    unsigned * d; CK(hipMalloc(&d, 64)); CK(hipMemset(d, 0, 64));
    const void * fn = (const void *) k_census<THREADS, WPE, REGS>;
    CK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    int nb = -1; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, THREADS, lds));
    hipFuncAttributes fa{}; CK(hipFuncGetAttributes(&fa, fn));
    hipLaunchKernelGGL((k_census<THREADS, WPE, REGS>), dim3(grid), dim3(THREADS), lds, 0, d, d + 1, lds);
    CK(hipDeviceSynchronize());
    unsigned h[2]; CK(hipMemcpy(h, d, 8, hipMemcpyDeviceToHost));
    printf("%-28s threads %4d regs %3d lds %6d grid %4d: API %d blocks/CU, resident together %u\n", name, THREADS, fa.numRegs, lds, grid, nb, h[1]);
    CK(hipFree(d));
}

int main() {
    for (int lds : {16384, 81920})  {
        run<384, 3, 0>("6 waves, few regs", lds, 512);
        run<320, 3, 0>("5 waves, few regs", lds, 512);
    }
    for (int lds : {16384, 76288})  {
        run<384, 3, 1>("6 waves, 168 regs", lds, 512);
        run<320, 3, 1>("5 waves, 168 regs", lds, 512);
        run<256, 3, 1>("4 waves, 168 regs", lds, 512);
        run<384, 4, 2>("6 waves, 128 regs", lds, 512);
        run<512, 2, 3>("8 waves, 240 regs", lds, 512);
        run<256, 2, 3>("4 waves, 240 regs", lds, 512);
        run<384, 2, 3>("6 waves, 240 regs", lds, 512);
    }
    return 0;
}
