// chain_probe.hip — VERDICT r01 "Next round" #2: is a persistent decode kernel (phases of one layer separated by a grid barrier,
// each workgroup issuing the next phase's first weight loads BEFORE it arrives at the barrier) faster than one launch per phase?
// Phases = one Llama-3-8B layer by weight bytes: QKV 14.2 MB, attention (KV) 8.6 MB, wo 9.4 MB, gate+up 66 MB, ffn_down 33 MB.
// Every phase is the real thing in the parts that decide the timing: it reads the 16 KB activation row the PREVIOUS phase wrote
// (RMS_NORM * w -> Q8_K into LDS: the product's wave_quantize_q8_K, double-precision sum of squares, two workgroup barriers),
// streams its slice of distinct weights with 16-byte loads, two items of 48 B per lane in flight, does ~the ALU work of the
// Q4_K dot per item (16 v_dot4 + unpacking against the LDS activations), reduces per wave and writes 16 KB of f32 results.
//   variant L: 5 launches per layer inside one hipGraph (what the product does today)
//   variant P: ONE launch per `LAYERS` layers; phases separated by a device-wide barrier (flat counter, or XCD-hierarchical:
//              per-XCC counter -> leader -> top counter, MI355X_MICROARCH.md "barrier-xcd"); the next phase's first item is
//              requested before arriving (weights do not depend on activations)
// Output: us per layer for each variant, and for P the time spent between arrive and release.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../llama_box_amd/csrc/dev_util.h"
#include "../../llama_box_amd/csrc/common.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
namespace mi355x { int log_level() { return 1; } }
using namespace mi355x;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct phase { const u32x4 * W; size_t n16; };  // weights of the phase, in 16-byte units (multiple of 256 * 1024 * 3)
struct prog { phase ph[5]; };
struct alignas(256) padded_u32 { unsigned v; unsigned pad[63]; };  // one counter per 256-byte line: pollers of one do not slow arrivals at another
struct sync_area { padded_u32 flat, top, xcd_cnt[8], xcd_gen[8]; unsigned pop[8]; unsigned npop; unsigned fail; unsigned pad[54]; };

__device__ __forceinline__ unsigned ld_relaxed(const unsigned * p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// device-wide barrier, called by all threads after their results are stored.  `prefetch` (the next phase's first weight loads) is
// run by waves 1..15 as soon as the workgroup has arrived and by wave 0 once the barrier has released it, so that the polling
// lane's loads never queue behind its own prefetch.  MODE 0 flat counter · 1 XCD-hierarchical · 2 flat without fences (invalid,
// timing only) · 3 no barrier (invalid, timing only) · 4 flat, prefetch only after the barrier
template <int MODE, typename F> __device__ __forceinline__ void grid_barrier(sync_area * sa, const unsigned epoch, const int xcc, F prefetch) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's result stores have reached the XCD's L2
    __syncthreads();
    if (MODE == 3) { prefetch(); return; }
    const int wave = threadIdx.x >> 6;
    if (wave != 0 && MODE != 4) prefetch();
    if (threadIdx.x == 0) {
        int spins = 0;
        if (MODE == 0 || MODE == 2 || MODE == 4) {
            if (MODE != 2) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __hip_atomic_fetch_add(&sa->flat.v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = epoch * gridDim.x;
            while (ld_relaxed(&sa->flat.v) < target) { __builtin_amdgcn_s_sleep(4); if (++spins > 2000000) { sa->fail = 1; break; } }
            if (MODE != 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        } else {
            // every workgroup's stores are in ITS XCD's L2 (vmcnt drained above); the XCD's last arriver writes that L2 back
            const unsigned mine = sa->pop[xcc];
            const unsigned old = __hip_atomic_fetch_add(&sa->xcd_cnt[xcc].v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1 == epoch * mine) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add(&sa->top.v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned target = epoch * sa->npop;
                while (ld_relaxed(&sa->top.v) < target) { __builtin_amdgcn_s_sleep(2); if (++spins > 2000000) { sa->fail = 2; break; } }
                __hip_atomic_store(&sa->xcd_gen[xcc].v, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                while (ld_relaxed(&sa->xcd_gen[xcc].v) < epoch) { __builtin_amdgcn_s_sleep(4); if (++spins > 2000000) { sa->fail = 3; break; } }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }
    if (wave == 0 || MODE == 4) prefetch();
    __syncthreads();
}

struct item { u32x4 w[3]; };
__device__ __forceinline__ void load_item(const u32x4 * W, const size_t n16, const size_t idx, item & it) {
#pragma unroll
    for (int u = 0; u < 3; ++u) it.w[u] = idx + (size_t) u * 64 < n16 ? __builtin_nontemporal_load(W + idx + (size_t) u * 64) : (u32x4) (0u);
}
// ~ the integer work of one Q4_K (super-block, chunk) pair: 16 dot4 against LDS activations + unpack + scale math
__device__ __forceinline__ float eat_item(const item & it, const q8k_dev * y, const int lane) {
    const q8k_dev * yb = y + (lane >> 2);
    const uint4 * yq = (const uint4 *) (yb->qs + 64 * (lane & 3));
    const uint4 y0 = yq[0], y1 = yq[1], y2 = yq[2], y3 = yq[3];
    const uint32_t yl[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w}, yh[8] = {y2.x, y2.y, y2.z, y2.w, y3.x, y3.y, y3.z, y3.w};
    const uint32_t q[8] = {it.w[1].x, it.w[1].y, it.w[1].z, it.w[1].w, it.w[2].x, it.w[2].y, it.w[2].z, it.w[2].w};
    int s_lo = 0, s_hi = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        s_lo = dot4((int) (q[k] & 0x0F0F0F0Fu), (int) yl[k], s_lo);
        s_hi = dot4((int) ((q[k] >> 4) & 0x0F0F0F0Fu), (int) yh[k], s_hi);
    }
    const int sc0 = (int) (it.w[0].y & 63), sc1 = (int) ((it.w[0].y >> 8) & 63), m0 = (int) (it.w[0].z & 63), m1 = (int) ((it.w[0].z >> 8) & 63);
    const float d = h2f((uint16_t) (it.w[0].x & 0x3FFF)), dmin = h2f((uint16_t) ((it.w[0].x >> 16) & 0x3FFF));
    return yb->d * (d * (float) (sc0 * s_lo + sc1 * s_hi) - dmin * (float) (m0 * yb->bs32[0] + m1 * yb->bs32[1]));
}

// one phase for this workgroup.  `first` was requested by the caller (before the barrier in the persistent variant).
// GRAN (round 4, VERDICT r02 / r03 "(d)"): no barrier at all between the phases.  A wave's result leaves as ONE naturally aligned 8-byte granule
// {value, tag = index of the phase that reads it} written with an agent-scope (sc1, write-through) store; the next phase's prologue polls the
// granules of its 256-value chunk (4 per lane, all requested together, only the late ones again) — Guideline 16 R2, "the data IS the flag".  The
// vectors are double-buffered by phase parity (a workgroup cannot publish phase k + 2 before it has gathered every workgroup's phase k + 1, and
// those were published after their owners had finished reading phase k).  gin: granules this phase reads (tag `tag`), gout: those it writes (tag + 1).
#ifndef GRAN_SLEEP
#define GRAN_SLEEP 1
#endif
__device__ __forceinline__ unsigned long long ld_gran(const unsigned long long * p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <bool COHERENT, bool GRAN = false>
__device__ __forceinline__ void run_phase(const phase & ph, const float * x, const float * nw, float * out, char * smem, item first,
                                          const unsigned long long * gin = nullptr, unsigned long long * gout = nullptr, const unsigned tag = 0, unsigned * fail = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // ---- prologue: RMS_NORM * w -> Q8_K (K = 4096: one 256-value chunk per wave)
    float4 v, g;
    if (GRAN) {
        const unsigned long long * gp = gin + (size_t) (wave * 64 + lane) * 4;
        unsigned long long q[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = ld_gran(gp + i);
        for (int spins = 0;; ++spins) {
            bool ok = true;
#pragma unroll
            for (int i = 0; i < 4; ++i) ok = ok && (unsigned) (q[i] >> 32) == tag;
            if (__all(ok)) break;
            if (spins > 4000000) { if (fail) *fail = 7; break; }
            __builtin_amdgcn_s_sleep(GRAN_SLEEP);
#pragma unroll
            for (int i = 0; i < 4; ++i) if ((unsigned) (q[i] >> 32) != tag) q[i] = ld_gran(gp + i);
        }
        v = make_float4(__builtin_bit_cast(float, (unsigned) q[0]), __builtin_bit_cast(float, (unsigned) q[1]), __builtin_bit_cast(float, (unsigned) q[2]), __builtin_bit_cast(float, (unsigned) q[3]));
    } else if (COHERENT) {  // written by other workgroups of THIS launch: L2-coherent loads (the acquire fence of the barrier invalidated L1)
        v = ((const float4 *) x)[wave * 64 + lane];
    } else v = ((const float4 *) x)[wave * 64 + lane];
    g = ((const float4 *) nw)[wave * 64 + lane];
    double ss = (double) (v.x * v.x) + (double) (v.y * v.y) + (double) (v.z * v.z) + (double) (v.w * v.w);
    ss = wave_sum_d(ss);
    double * red = (double *) (smem + 16 * sizeof(q8k_dev));
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) tot += red[i];
    const float scale = 1.0f / sqrtf((float) (tot / 4096.0) + 1e-5f);
    float t[4] = {(v.x * scale) * g.x, (v.y * scale) * g.y, (v.z * scale) * g.z, (v.w * scale) * g.w};
    wave_quantize_q8_K(t, lane, (q8k_dev *) smem + wave);
    __syncthreads();
    const q8k_dev * y = (const q8k_dev *) smem;
    // ---- stream: this wave's items, interleaved over the grid (item i of wave w of block b = chunk ((i * G + b) * 16 + w))
    const size_t stride = (size_t) gridDim.x * 16 * 192;
    size_t idx = ((size_t) blockIdx.x * 16 + wave) * 192 + lane;
    float acc = 0.0f;
    item cur = first;
    while (idx < ph.n16) {
        item nxt;
        const size_t ni = idx + stride;
        if (ni < ph.n16) load_item(ph.W, ph.n16, ni, nxt);
        acc += eat_item(cur, y, lane);
        cur = nxt;
        idx = ni;
    }
    const float r = wave_sum(acc);
    __syncthreads();  // (LDS activations are rewritten by the next phase)
    const float res = r * 1e-6f + (float) (blockIdx.x & 3);
    if (GRAN) {
        if (lane == 0) __hip_atomic_store(gout + blockIdx.x * 16 + wave, ((unsigned long long) (tag + 1) << 32) | __builtin_bit_cast(unsigned, res), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (lane == 0) out[blockIdx.x * 16 + wave] = res;
}

__global__ void __launch_bounds__(1024) k_phase(const phase ph, const float * x, const float * nw, float * out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    item first;
    load_item(ph.W, ph.n16, ((size_t) blockIdx.x * 16 + (threadIdx.x >> 6)) * 192 + (threadIdx.x & 63), first);
    run_phase<false>(ph, x, nw, out, smem, first);
}

template <int MODE> __global__ void __launch_bounds__(1024) k_persist(const prog * __restrict__ progs, const int n_layers, float * xa, float * xb, const float * nw, sync_area * sa,
                                                                     unsigned long long * wait_ticks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int xcc = 0;
    if (MODE == 1) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc = (int) (id & 7);
        // population census + one flat barrier (epoch 1 of the flat counter) so that every workgroup knows how many share its XCD
        if (threadIdx.x == 0) {
            const unsigned old = __hip_atomic_fetch_add(&sa->pop[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == 0) __hip_atomic_fetch_add(&sa->npop, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        grid_barrier<0>(sa, 1, xcc, [] {});
    }
    unsigned epoch = 0;
    unsigned long long waited = 0;
    const size_t my = ((size_t) blockIdx.x * 16 + (threadIdx.x >> 6)) * 192 + (threadIdx.x & 63);
    item first;
    load_item(progs[0].ph[0].W, progs[0].ph[0].n16, my, first);
    int k = 0;
    for (int l = 0; l < n_layers; ++l) {
        for (int p = 0; p < 5; ++p, ++k) {
            const phase ph = progs[l].ph[p];
            if (MODE == 5) {
                // granule vectors live behind the two activation buffers' first 32 KB each (4096 x 8 bytes), zeroed by the memset node: tag 0 = phase 0's input
                unsigned long long * ga = (unsigned long long *) (xa + 16384), * gb = (unsigned long long *) (xb + 16384);
                run_phase<true, true>(ph, nullptr, nw, nullptr, smem, first, (k & 1) ? gb : ga, (k & 1) ? ga : gb, (unsigned) k, &sa->fail);
                const bool last5 = l == n_layers - 1 && p == 4;
                const phase nx5 = last5 ? ph : (p < 4 ? progs[l].ph[p + 1] : progs[l + 1].ph[0]);
                if (!last5) load_item(nx5.W, nx5.n16, my, first);  // the next phase's weights are requested before its prologue polls for its activations
                continue;
            }
            run_phase<true>(ph, (k & 1) ? xb : xa, nw, (k & 1) ? xa : xb, smem, first);
            // request the next phase's first item, THEN arrive: the weights stream while the barrier and the next prologue run
            const bool last = l == n_layers - 1 && p == 4;
            const phase nx = last ? ph : (p < 4 ? progs[l].ph[p + 1] : progs[l + 1].ph[0]);
            const unsigned long long t0 = wall_clock64();
            if (!last) grid_barrier<MODE>(sa, ++epoch, xcc, [&] { load_item(nx.W, nx.n16, my, first); });
            waited += wall_clock64() - t0;
        }
    }
    if (threadIdx.x == 0) wait_ticks[blockIdx.x] = waited;
}

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int LAYERS = 8;
    const double mb[5] = {14.2, 8.6, 9.4, 66.0, 33.0};
    const size_t unit = (size_t) 256 * 16 * 192;  // 16-byte units per full grid pass (786432 = 12.6 MB)
    std::vector<prog> hp(LAYERS);
    size_t total16 = 0;
    for (int l = 0; l < LAYERS; ++l) for (int p = 0; p < 5; ++p) { size_t n = (size_t) (mb[p] * 1048576.0 / 16.0); n = (n + 63) / 64 * 64; hp[l].ph[p].n16 = n; total16 += n; }
    u32x4 * W; CK(hipMalloc(&W, total16 * 16 + 4096)); CK(hipMemset(W, 0x11, total16 * 16));
    size_t off = 0;
    for (int l = 0; l < LAYERS; ++l) for (int p = 0; p < 5; ++p) { hp[l].ph[p].W = W + off; off += hp[l].ph[p].n16; }
    (void) unit;
    prog * dp; CK(hipMalloc(&dp, sizeof(prog) * LAYERS)); CK(hipMemcpy(dp, hp.data(), sizeof(prog) * LAYERS, hipMemcpyHostToDevice));
    float *xa, *xb, *nw; CK(hipMalloc(&xa, 65536 + 32768)); CK(hipMalloc(&xb, 65536 + 32768)); CK(hipMalloc(&nw, 65536));
    std::vector<float> h(16384);
    for (int i = 0; i < 16384; ++i) h[i] = (float) ((i * 7919) % 1000) / 500.0f - 1.0f;
    CK(hipMemcpy(xa, h.data(), 65536, hipMemcpyHostToDevice)); CK(hipMemcpy(xb, h.data(), 65536, hipMemcpyHostToDevice)); CK(hipMemcpy(nw, h.data(), 65536, hipMemcpyHostToDevice));
    sync_area * sa; CK(hipMalloc(&sa, sizeof(sync_area)));
    unsigned long long * wt; CK(hipMalloc(&wt, 256 * 8));
    const size_t lds = 16 * sizeof(q8k_dev) + 256;
    double layer_bytes = 0; for (int p = 0; p < 5; ++p) layer_bytes += (double) hp[0].ph[p].n16 * 16;
    auto time_graph = [&](hipGraphExec_t ge, int reps) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1e3 / reps;
    };
    {   // variant L
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
        int k = 0;
        for (int l = 0; l < LAYERS; ++l) for (int p = 0; p < 5; ++p, ++k) hipLaunchKernelGGL(k_phase, dim3(256), dim3(1024), lds, s, hp[l].ph[p], (k & 1) ? xb : xa, nw, (k & 1) ? xa : xb);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        const double us = time_graph(ge, 10);
        printf("L  five launches per layer        : %7.2f us/layer  (%.1f MB/layer -> %.2f TB/s)\n", us / LAYERS, layer_bytes / 1048576.0, layer_bytes * LAYERS / us / 1e6);
    }
    for (int mode = 0; mode < 6; ++mode) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
        CK(hipMemsetAsync(sa, 0, sizeof(sync_area), s));
        if (mode == 5) { CK(hipMemsetAsync(xa + 16384, 0, 32768, s)); CK(hipMemsetAsync(xb + 16384, 0, 32768, s)); }
        if (mode == 5) hipLaunchKernelGGL(k_persist<5>, dim3(256), dim3(1024), lds, s, dp, LAYERS, xa, xb, nw, sa, wt);
        else if (mode == 0) hipLaunchKernelGGL(k_persist<0>, dim3(256), dim3(1024), lds, s, dp, LAYERS, xa, xb, nw, sa, wt);
        else if (mode == 1) hipLaunchKernelGGL(k_persist<1>, dim3(256), dim3(1024), lds, s, dp, LAYERS, xa, xb, nw, sa, wt);
        else if (mode == 2) hipLaunchKernelGGL(k_persist<2>, dim3(256), dim3(1024), lds, s, dp, LAYERS, xa, xb, nw, sa, wt);
        else if (mode == 3) hipLaunchKernelGGL(k_persist<3>, dim3(256), dim3(1024), lds, s, dp, LAYERS, xa, xb, nw, sa, wt);
        else hipLaunchKernelGGL(k_persist<4>, dim3(256), dim3(1024), lds, s, dp, LAYERS, xa, xb, nw, sa, wt);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        const double us = time_graph(ge, 10);
        sync_area hs; CK(hipMemcpy(&hs, sa, sizeof(hs), hipMemcpyDeviceToHost));
        std::vector<unsigned long long> hw(256); CK(hipMemcpy(hw.data(), wt, 256 * 8, hipMemcpyDeviceToHost));
        std::sort(hw.begin(), hw.end());
        printf("P%d persistent, %s barrier : %7.2f us/layer  (%.2f TB/s); time in barrier per phase: min %.2f median %.2f max %.2f us; fail=%u xcd populations %u %u %u %u %u %u %u %u\n", mode,
               mode == 1 ? "XCD-hierarchical" : mode == 0 ? "flat counter    " : mode == 2 ? "flat, NO fences " : mode == 3 ? "NO barrier      " : mode == 5 ? "GRANULE hand-off (no barrier)" : "flat, no prefetch", us / LAYERS, layer_bytes * LAYERS / us / 1e6, hw[0] / 100.0 / (5 * LAYERS - 1), hw[128] / 100.0 / (5 * LAYERS - 1),
               hw[255] / 100.0 / (5 * LAYERS - 1), hs.fail, hs.pop[0], hs.pop[1], hs.pop[2], hs.pop[3], hs.pop[4], hs.pop[5], hs.pop[6], hs.pop[7]);
    }
    return 0;
}
