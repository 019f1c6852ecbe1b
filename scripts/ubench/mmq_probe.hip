// mmq_probe.hip — times the prefill mat-mul kernels alone on Llama-3-8B shapes (random bytes; values do not matter).
// Built once per MMQ_PROBE value by scripts/ubench/build_probe.sh; MMQ_PROBE != 0 removes one stage (see mmq_i8.hip).
#ifndef MMQ_SRC
#define MMQ_SRC "../../llama_box_amd/csrc/mmq_i8.hip"
#endif
#include MMQ_SRC
#ifndef MMQ_PROBE
#define MMQ_PROBE 0
#endif
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
using namespace mi355x;
int main(int argc, char ** argv) {
    const int mt = argc > 1 ? atoi(argv[1]) : 0;
    const int ks = argc > 2 ? atoi(argv[2]) : 1;  // split-K factor (partials + fixed-order reduce)  // forwarded as the kernel's force_* argument (0 = auto)
    struct shape { int N, K, M; const char * name; } shapes[] = {{14336, 4096, 512, "gate/up"}, {4096, 4096, 512, "wq/wo"}, {4096, 14336, 512, "down"}, {1024, 4096, 512, "wk"}, {14336, 4096, 2048, "gate M=2048"}};
    hipStream_t s; CK(hipStreamCreate(&s));
    for (auto & sh : shapes) {
        const size_t wb = (size_t) sh.N * (sh.K / 256) * 144, ab = (size_t) sh.M * (sh.K / 256) * sizeof(q8k_dev), ob = (size_t) sh.M * sh.N * 4;
        uint8_t * W; q8k_dev * A; float * O; float * P;
        CK(hipMalloc(&W, wb)); CK(hipMalloc(&A, ab)); CK(hipMalloc(&O, ob)); CK(hipMalloc(&P, ob * (size_t) (ks > 1 ? ks : 1)));
        std::vector<uint8_t> h(wb); for (size_t i = 0; i < wb; ++i) h[i] = (uint8_t) (i * 2654435761u >> 13);
        for (size_t b = 0; b < wb / 144; ++b) { h[b * 144] = 0; h[b * 144 + 1] = 0x3c; h[b * 144 + 2] = 0; h[b * 144 + 3] = 0x38; }
        CK(hipMemcpy(W, h.data(), wb, hipMemcpyHostToDevice));
        std::vector<uint8_t> ha(ab); for (size_t i = 0; i < ab; ++i) ha[i] = (uint8_t) (i * 40503u >> 7);
        CK(hipMemcpy(A, ha.data(), ab, hipMemcpyHostToDevice));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 3; ++i) launch_mmq_i8(s, GGML_TYPE_Q4_K, W, (int64_t) (sh.K / 256) * 144, sh.K, sh.N, sh.M, A, O, sh.N, mt, ks, P, nullptr, 0);
        CK(hipStreamSynchronize(s));
        const int reps = 20;
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) launch_mmq_i8(s, GGML_TYPE_Q4_K, W, (int64_t) (sh.K / 256) * 144, sh.K, sh.N, sh.M, A, O, sh.N, mt, ks, P, nullptr, 0);
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps, tf = 2.0 * sh.N * sh.K * sh.M / (us * 1e-6) / 1e12;
        printf("probe=%d mt=%d ks=%d %-12s N=%5d K=%5d M=%4d  %8.1f us  %7.1f TFLOP/s (%.1f%% of 2.5 PF)\n", MMQ_PROBE, mt, ks, sh.name, sh.N, sh.K, sh.M, us, tf, tf / 25.0);
        CK(hipFree(W)); CK(hipFree(A)); CK(hipFree(O)); CK(hipFree(P));
    }
    return 0;
}
