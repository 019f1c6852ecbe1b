// stride_probe.hip — does the ORDER in which a tile kernel walks a row-major quantised matrix decide its HBM rate?
// The matrix: 28672 rows x 2304 bytes (Llama-3-8B gate + up as Q4_K: 16 super-blocks of 144 bytes per row), 66 MB, a fresh copy per
// launch out of a 2 GB pool (nothing is served by L2 / the 256 MB memory-side cache).  A wave owns an item = (32 rows) x (1 / KP of
// the row bytes) and fetches it in steps of 4 KB (4 x global_load_dwordx4 per lane), U steps requested before the first is consumed.
//   mode 0  block columns : a step = 128 bytes of each of the 32 rows (what a per-super-block K walk does: 144-byte runs, row stride)
//   mode 1  4 blocks x 8 rows : a step = 512 bytes of each of 8 rows
//   mode 2  row major     : a step = 2 rows x 2048 bytes
//   mode 3  contiguous    : the item's bytes are one contiguous range (upper bound: the mat-vec kernels' order)
// (rows are 2048 payload bytes at a stride of 2304 in modes 0..2: same run lengths and page behaviour as 144-byte blocks, clean piece counts)
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/stride_probe.hip -o scripts/ubench/stride_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int ROWS = 28672, STRIDE = 2304, PAY = 2048;

template <int MODE, int U>
__global__ void __launch_bounds__(1024) k_walk(const char * __restrict__ W, const int kp, const int n_items, int * sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int steps = 16 / kp;  // 4 KB steps per item: 32 rows x 2048 / kp bytes
    uint32_t acc = 0;
    for (int item = blockIdx.x * nw + wave; item < n_items; item += gridDim.x * nw) {
        const int tile = item / kp, part = item % kp;
        const char * base = W + (size_t) tile * 32 * STRIDE;
        const int span = PAY / kp;  // payload bytes of a row that belong to this item
        for (int s0 = 0; s0 < steps; s0 += U) {
            uint4 v[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int s = s0 + u;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int p = lane + 64 * i;  // piece of the step, 0..255
                    size_t off;
                    if (MODE == 0) off = (size_t) (p >> 3) * STRIDE + part * span + s * 128 + (p & 7) * 16;
                    else if (MODE == 1) { const int per_row = span / 512; /* 512-byte chunks per row */ const int rg = s / per_row, bq = s % per_row; off = (size_t) (8 * (rg & 3) + (p >> 5)) * STRIDE + part * span + bq * 512 + (p & 31) * 16; }
                    else if (MODE == 2) { const int lin = s * 4096 + p * 16; off = (size_t) (lin / span) * STRIDE + part * span + lin % span; }
                    else off = (size_t) part * (32 * PAY / kp) + s * 4096 + p * 16;
                    v[u][i] = s < steps ? *(const uint4 *) (base + off) : make_uint4(0, 0, 0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc += v[u][i].x ^ v[u][i].y ^ v[u][i].z ^ v[u][i].w;
        }
    }
    if (acc == 0x7fffffffu) *sink = (int) acc;
}

template <int MODE, int U> static float run(const char * pool, size_t pool_bytes, int wgs, int waves, int kp, int * sink) {
    const size_t mat = (size_t) ROWS * STRIDE;
    const int copies = (int) (pool_bytes / mat);
    const int n_items = ROWS / 32 * kp;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        for (int c = 0; c < copies; ++c) hipLaunchKernelGGL((k_walk<MODE, U>), dim3(wgs), dim3(waves * 64), 0, 0, pool + (size_t) c * mat, kp, n_items, sink);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms / copies < best) best = ms / copies;
    }
    return best * 1e3f;  // us per launch
}

int main() {
    const size_t pool_bytes = (size_t) 2 << 30;
    char * pool; int * sink;
    CK(hipMalloc(&pool, pool_bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(pool, 1, pool_bytes));
    const double mb = (double) ROWS * PAY / 1e6;  // payload actually read
    struct { int wgs, waves, kp; } shapes[] = {{224, 4, 1}, {224, 8, 2}, {256, 8, 2}, {256, 8, 4}, {256, 16, 4}};
    printf("payload %.1f MB per launch; us per launch (TB/s)\n", mb);
    for (auto sh : shapes) {
        printf("grid %d x %d waves, K parts %d:\n", sh.wgs, sh.waves, sh.kp);
#define ROW(MODE, name)                                                                                                  \
        {                                                                                                                \
            const float a = run<MODE, 1>(pool, pool_bytes, sh.wgs, sh.waves, sh.kp, sink);                               \
            const float b = run<MODE, 2>(pool, pool_bytes, sh.wgs, sh.waves, sh.kp, sink);                               \
            const float c = run<MODE, 4>(pool, pool_bytes, sh.wgs, sh.waves, sh.kp, sink);                               \
            printf("  %-22s U=1 %6.1f (%.2f)  U=2 %6.1f (%.2f)  U=4 %6.1f (%.2f)\n", name, a, mb / a, b, mb / b, c, mb / c); \
        }
        ROW(0, "block columns")
        ROW(1, "4 blocks x 8 rows")
        ROW(2, "row major")
        ROW(3, "contiguous")
    }
    return 0;
}
