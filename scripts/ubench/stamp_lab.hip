// stamp_lab.hip — where does the fixed cost of a decode mat-vec launch go?  Runs chains of the real k_mmvq_stream launches
// (private copy of mmvq.hip built with -DMI_LAB_STAMPS) and prints, per launch, the wall-clock (100 MHz) of: kernel entry,
// first weight loads issued, x arrived + reduced, prologue barrier passed, first item consumed, wave 0 finished — as
// min / median / max over the workgroups, relative to the earliest entry of that launch; plus the gap to the previous launch.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../llama_box_amd/csrc/kernels.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
namespace mi355x { int log_level() { return 1; } void lab_set_stamps(unsigned long long * p); }
using namespace mi355x;
__global__ void k_fill(uint32_t * p, size_t n, uint32_t seed) {
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; const size_t T = (size_t) gridDim.x * blockDim.x;
    for (; i < n; i += T) { uint32_t x = (uint32_t) i * 2654435761u ^ seed; x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; p[i] = x & 0x3C003C00u ? x & 0x1FFF1FFFu : x; }
}
struct op { const char * name; int type, blk, bytes, K, N; bool glu, norm, res; };
int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const size_t pool_bytes = (size_t) 3 << 30;
    uint8_t * pool; CK(hipMalloc(&pool, pool_bytes));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, s, (uint32_t *) pool, pool_bytes / 4, 99u);
    float *xa, *xb, *nw; CK(hipMalloc(&xa, 1 << 20)); CK(hipMalloc(&xb, 1 << 20)); CK(hipMalloc(&nw, 1 << 20));
    CK(hipMemsetAsync(xa, 0, 1 << 20, s)); CK(hipMemsetAsync(xb, 0, 1 << 20, s)); CK(hipMemsetAsync(nw, 0, 1 << 20, s));
    const int CH = 12;
    unsigned long long * st; CK(hipMalloc(&st, (size_t) CH * 256 * 8 * 8));
    lab_set_stamps(st);
    std::vector<op> ops = {
        {"wo q4_K 4096x4096 f32pro", GGML_TYPE_Q4_K, 256, 144, 4096, 4096, false, false, true},
        {"gate/up q4_K 4096x14336 glu normpro", GGML_TYPE_Q4_K, 256, 144, 4096, 14336, true, true, false},
        {"down q4_K 14336x4096 f32pro", GGML_TYPE_Q4_K, 256, 144, 14336, 4096, false, false, true},
        {"down q6_K 14336x4096 f32pro", GGML_TYPE_Q6_K, 256, 210, 14336, 4096, false, false, true},
    };
    for (const op & o : ops) {
        const size_t mb = (size_t) o.N * (o.K / o.blk) * o.bytes * (o.glu ? 2 : 1);
        CK(hipMemsetAsync(st, 0, (size_t) CH * 256 * 8 * 8, s));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
        for (int i = 0; i < CH; ++i) {
            mmvq_args a{};
            a.W = pool + (size_t) i * mb; a.W2 = o.glu ? a.W + mb / 2 : nullptr;
            a.w_nb1 = (int64_t) (o.K / o.blk) * o.bytes; a.type = o.type; a.K = o.K; a.N = o.N; a.ncols = 1;
            a.dst = (i & 1) ? xa : xb; a.dst_stride = i /* stamp slot */; a.add = o.res ? nw : nullptr;
            a.x = (i & 1) ? xb : xa; a.norm_w = o.norm ? nw : nullptr; a.eps = 1e-5f;
            launch_mmvq(s, a, 1);
        }
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        CK(hipMemsetAsync(st, 0, (size_t) CH * 256 * 8 * 8, s));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        std::vector<unsigned long long> h((size_t) CH * 256 * 8);
        CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
        printf("== %s (%.1f MB per launch); times in us relative to the launch's earliest workgroup entry: min / median / max over workgroups\n", o.name, mb / 1048576.0);
        unsigned long long prev_end = 0;
        for (int i = 2; i < CH; ++i) {
            std::vector<double> col[6];
            unsigned long long t0 = ~0ull, tend = 0;
            int nb = 0;
            for (int b = 0; b < 256; ++b) { const unsigned long long * r = &h[((size_t) i * 256 + b) * 8]; if (r[0]) { t0 = std::min(t0, r[0]); tend = std::max(tend, r[5]); nb++; } }
            for (int b = 0; b < 256; ++b) { const unsigned long long * r = &h[((size_t) i * 256 + b) * 8]; if (!r[0]) continue; for (int k = 0; k < 6; ++k) if (r[k]) col[k].push_back((double) (r[k] - t0) / 100.0); }
            printf("launch %2d (%3d WGs) gap-from-prev-end %6.2f |", i, nb, prev_end ? (double) ((long long) t0 - (long long) prev_end) / 100.0 : 0.0);
            const char * nm[6] = {"entry", "loads-issued", "x+reduce", "prologue-done", "first-item", "wave0-done"};
            for (int k = 0; k < 6; ++k) {
                if (col[k].empty()) continue;
                std::sort(col[k].begin(), col[k].end());
                printf(" %s %.2f/%.2f/%.2f |", nm[k], col[k].front(), col[k][col[k].size() / 2], col[k].back());
            }
            printf("\n");
            prev_end = tend;
        }
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
