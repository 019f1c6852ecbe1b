// ta_probe.hip — what does the vector-memory front end (TA / L1 tag lookup) charge for 16-byte loads whose lanes do NOT form
// contiguous runs?  Decides how the skinny (M <= 32) matrix-core kernel fetches its operands: straight into the MFMA operand
// layout (lane = (row, k-group): every lane in another row, 16 B each) or as row-contiguous runs staged through LDS.
// All patterns read an L2-resident region (2 MB, re-walked), every CU busy with one 1024-thread workgroup, 8 loads in flight
// per lane.  Output: bytes per clock per CU (2.4 GHz assumed) for each pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(2))) u64_a2 { unsigned x, y; };
struct __attribute__((packed, aligned(2))) u128_a2 { unsigned x, y, z, w; };

constexpr size_t REGION = 2u << 20;

// PAT 0: wave-contiguous 1 KB · 1: lane = (row, half), rows 2304 B apart (Q4_K, K = 4096), 16 B · 2: rows 5120 B apart (Q8_K
// activations of 32 tokens) · 3: runs of 144 B (9 lanes) per row · 4: as 1 with 8-byte loads at 2-byte alignment (Q6_K rows of
// 3360 B) · 5: as 1 with 16-byte loads at 2-byte alignment · 6: runs of 4 lanes (64 B) per row
template <int PAT> __global__ void __launch_bounds__(1024) k_probe(const char * __restrict__ base, unsigned * out, const int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    size_t off;
    size_t step;
    if (PAT == 0) { off = (size_t) lane * 16; step = 1024; }
    else if (PAT == 1) { off = (size_t) (lane & 31) * 2304 + (lane >> 5) * 16; step = 32; }
    else if (PAT == 2) { off = (size_t) (lane & 31) * 5120 + (lane >> 5) * 16; step = 32; }
    else if (PAT == 3) { off = (size_t) (lane / 9) * 2304 + (lane % 9) * 16; step = 2304 * 8; }
    else if (PAT == 4) { off = (size_t) (lane & 31) * 3360 + (lane >> 5) * 8 + 2; step = 16; }
    else if (PAT == 5) { off = (size_t) (lane & 31) * 3360 + (lane >> 5) * 16 + 2; step = 32; }
    else { off = (size_t) (lane >> 2) * 2304 + (lane & 3) * 16; step = 64; }
    const size_t wave_base = ((size_t) blockIdx.x * 16 + wave) * 73728 % REGION;  // waves start in different places
    unsigned acc = 0;
    for (int it = 0; it < iters; it += 8) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            // rows of a pattern are walked 32 (or `step`) bytes at a time; after 4 steps move on by the span of the pattern
            const size_t o = (wave_base + off + (size_t) ((it + u) & 3) * step + (size_t) ((it + u) >> 2) * 147456) % (REGION - 32);
            if (PAT == 4) {
                const u64_a2 t = *(const u64_a2 *) (base + (o & ~(size_t) 1));
                v[u] = u32x4{t.x, t.y, 0, 0};
            } else if (PAT == 5) {
                const u128_a2 t = *(const u128_a2 *) (base + (o & ~(size_t) 1));
                v[u] = u32x4{t.x, t.y, t.z, t.w};
            } else
                v[u] = *(const u32x4 *) (base + (o & ~(size_t) 15));
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

template <int PAT> static void run(const char * name, const char * base, unsigned * out, const int bytes_per_lane) {
    const int iters = 2048;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    k_probe<PAT><<<256, 1024>>>(base, out, iters);
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0));
        k_probe<PAT><<<256, 1024>>>(base, out, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    const double bytes_per_cu = (double) iters * 1024.0 * bytes_per_lane;  // 1024 lanes per CU
    const double clk = best * 1e-3 * 2.4e9;
    printf("%-58s %8.3f ms  %6.1f B/clk/CU  %6.1f clk per wave-instruction\n", name, best, bytes_per_cu / clk, clk / iters / 16.0);
}

int main() {
    char * base;
    unsigned * out;
    CK(hipMalloc(&base, REGION + 4096));
    CK(hipMemset(base, 1, REGION + 4096));
    CK(hipMalloc(&out, 4096));
    run<0>("0 contiguous 1 KB per wave, 16 B/lane", base, out, 16);
    run<1>("1 lane=(row,half) rows 2304 B apart, 16 B/lane", base, out, 16);
    run<2>("2 lane=(tok,half) rows 5120 B apart, 16 B/lane", base, out, 16);
    run<3>("3 runs of 9 lanes (144 B) per row, 16 B/lane", base, out, 16);
    run<6>("6 runs of 4 lanes (64 B) per row, 16 B/lane", base, out, 16);
    run<4>("4 lane=(row,half) rows 3360 B apart, 8 B/lane @2B align", base, out, 8);
    run<5>("5 lane=(row,half) rows 3360 B apart, 16 B/lane @2B align", base, out, 16);
    return 0;
}
