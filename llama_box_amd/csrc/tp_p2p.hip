// tp_p2p.hip — one-shot peer-to-peer all-reduce for the latency-bound messages of tensor-parallel decode (round 4).
//
// What it serves: the two sums per layer of a --tensor-split run (llama-box/engine_param.hpp:821-842): n_embd x M f32
// partial products, 16 KiB (8B) / 32 KiB (70B) at batch 1.  A ring collective pays 2 (N - 1) hops for them; xGMI is point to point and
// every GPU reaches every other directly, so here each rank WRITES its partial straight into a mailbox in every peer's memory and then
// sums the N mailboxes of its own memory in rank order — one hop, one launch, and every rank adds the same numbers in the same order
// (the result is bit-identical on all ranks and from run to run).
//
// Protocol (MI355X_MICROARCH.md, "handoff-1to1" / Guideline 16 R2: the data IS the flag):
//   * a value travels as ONE naturally aligned 8-byte granule {f32 value, u32 tag}, written with one system-scope store (untorn);
//     tag = the epoch of this all-reduce, so a reader polling a granule knows the value is the one it waits for — no flag, no fence;
//   * mailboxes are double-buffered by epoch parity: a rank can be at most one all-reduce ahead of a peer (it cannot finish all-reduce
//     k + 1 without the peer's contribution to it, which the peer sends only after it has finished reading all-reduce k);
//   * the epoch lives in DEVICE memory and is advanced by the kernel itself (the last workgroup to finish writes it; the next launch on the
//     stream starts after this one has ended): a hipGraph replay re-runs the same kernel nodes with frozen arguments, so nothing per-launch
//     may come from the host;
//   * every spin is bounded: on a time-out the kernel raises an error word in device memory (later launches stop waiting) AND a host-mapped
//     word: the sums of a broken group are garbage, so the next graph_compute of this backend returns GGML_STATUS_FAILED (tp.cpp: tp_check)
//     until the host resets the group (set_option("tp_p2p_reset", 1) on every rank) or, with a RCCL communicator attached, from then on
//     sums through RCCL.
// The mailbox memory is allocated uncached / fine-grained and exported with hipIpcGetMemHandle; peers map it with hipIpcOpenMemHandle
// (tp.cpp).  Two ranks on ONE GPU work the same way (that is how the tests run it on a one-GPU box).
#include <hip/hip_runtime.h>

#include "common.h"
#include "dev_util.h"
#include "kernels.h"

namespace mi355x {

// values per (parity, source) mailbox; messages up to this size are one launch, longer ones go in chunks
static_assert(P2P_SLOT_FLOATS % 256 == 0, "slot size");

__device__ __forceinline__ void st_sys(unsigned long long * p, const unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ unsigned long long ld_sys(const unsigned long long * p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// grid: up to P2P_MAX_BLOCKS workgroups of 256 threads; thread t of block b owns values b * 256 + t, + grid * 256, ...
__global__ void __launch_bounds__(256) k_p2p_all_reduce(const p2p_args a) {
    const int tid = threadIdx.x;
    __shared__ unsigned s_epoch;
    __shared__ unsigned s_dead;
    if (tid == 0) {
        s_epoch = a.state[0] + 1u;  // (written by the previous all-reduce launch of this stream, which has ended)
        s_dead = a.state[2];        // an earlier all-reduce of this group timed out: the group is broken, nobody waits any more (the host
    }                               // reads the word — stat "p2p_timeouts" — and falls back; waiting again would cost seconds per launch)
    __syncthreads();
    const unsigned epoch = s_epoch;
    const unsigned max_spins = s_dead ? 0u : a.max_spins;
    const unsigned par = epoch & 1u;
    const size_t slot = (size_t) P2P_SLOT_FLOATS;
    // ---- send: my partial into slot [par][rank] of every rank's mailbox (my own included)
    for (int i = (int) blockIdx.x * 256 + tid; i < a.n; i += (int) gridDim.x * 256) {
        const unsigned long long g = ((unsigned long long) epoch << 32) | (unsigned long long) __builtin_bit_cast(unsigned, a.data[i]);
#pragma unroll 1
        for (int p = 0; p < a.world; ++p) {
            unsigned long long * mb = (unsigned long long *) a.mbox[p] + ((size_t) par * a.world + a.rank) * slot;
            st_sys(mb + i, g);
        }
    }
    // ---- receive: the N contributions to my values, summed in rank order.  All N granules are requested together and only the
    // late ones again: one memory round trip when the peers are on time, not N dependent ones
    bool failed = false;
    double ssq = 0.0;
    for (int i = (int) blockIdx.x * 256 + tid; i < a.n; i += (int) gridDim.x * 256) {
        const unsigned long long * mb = (const unsigned long long *) a.mbox[a.rank] + (size_t) par * a.world * slot + i;
        unsigned long long g[P2P_MAX_RANKS];
#pragma unroll
        for (int s = 0; s < P2P_MAX_RANKS; ++s) g[s] = s < a.world ? ld_sys(mb + (size_t) s * slot) : ((unsigned long long) epoch << 32);
        unsigned spins = 0;
        for (;;) {
            bool all = true;
#pragma unroll
            for (int s = 0; s < P2P_MAX_RANKS; ++s) all = all && (unsigned) (g[s] >> 32) == epoch;
            if (all) break;
            if (++spins > max_spins) { failed = true; break; }
            __builtin_amdgcn_s_sleep(1);
#pragma unroll
            for (int s = 0; s < P2P_MAX_RANKS; ++s)
                if (s < a.world && (unsigned) (g[s] >> 32) != epoch) g[s] = ld_sys(mb + (size_t) s * slot);
        }
        float sum = 0.0f;
#pragma unroll
        for (int s = 0; s < P2P_MAX_RANKS; ++s) sum += s < a.world ? __builtin_bit_cast(float, (unsigned) g[s]) : 0.0f;
        if (a.add) sum += a.add[i];
        (a.out ? a.out : a.data)[i] = sum;
        ssq += (double) (sum * sum);
    }
    if (a.ss_out) {  // (uniform) the row's sum of squares for the RMS_NORM prologue that reads it next: one partial per workgroup
        __shared__ double s_ss[4];
        ssq = wave_sum_d(ssq);
        if ((tid & 63) == 0) s_ss[tid >> 6] = ssq;
        __syncthreads();
        if (tid == 0) a.ss_out[blockIdx.x] = ((s_ss[0] + s_ss[1]) + s_ss[2]) + s_ss[3];
    }
    if (failed) {
        atomicAdd(a.state + 2, 1u);
        if (a.err_host) __hip_atomic_store(a.err_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    if (tid == 0) {  // the last workgroup to get here publishes the epoch for the next launch
        const unsigned done = __hip_atomic_fetch_add(a.state + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == gridDim.x - 1u) {
            __hip_atomic_store(a.state + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.state, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

int p2p_all_reduce_blocks(int n) { return std::max(1, std::min(P2P_MAX_BLOCKS, (n + 255) / 256)); }
void launch_p2p_all_reduce(hipStream_t s, const p2p_args & a) {
    hipLaunchKernelGGL(k_p2p_all_reduce, dim3((unsigned) p2p_all_reduce_blocks(a.n)), dim3(256), 0, s, a);
}

MI_TU_TOUCH(tp_p2p)

}  // namespace mi355x
