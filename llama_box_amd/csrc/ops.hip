// ops.hip — normalisation, element-wise, RoPE, soft-max and data-movement kernels of the hot path.
//
// Each kernel restates the ggml-cpu routine named in its comment with the same
// arithmetic per element (this file is compiled with -ffp-contract=off, so `a*b - c` is two roundings exactly as
// in the CPU reference) and a wave64 mapping: rows are reduced with 64-lane butterflies, f32 is moved as float4.
// These ops are launch-bound at decode sizes; graph.cpp fuses the common chains so most of them disappear.
#include <algorithm>

#include "dev_util.h"
#include "kernels.h"

namespace mi355x {

// block-wide reductions over 256 threads (4 waves)
__device__ __forceinline__ double block_sum_d(double v, double * sh) {
    v = wave_sum_d(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double t = 0.0;
    const int nw = blockDim.x >> 6;
    for (int i = 0; i < nw; ++i) t += sh[i];
    return t;
}
__device__ __forceinline__ float block_max_f(float v, float * sh) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    float t = sh[0];
    const int nw = blockDim.x >> 6;
    for (int i = 1; i < nw; ++i) t = fmaxf(t, sh[i]);
    return t;
}

// ------------------------------------------------------------------------------------------------ RMS_NORM (+MUL)
// ggml_compute_forward_rms_norm_f32: sum of squares accumulated in double, scale = 1/sqrtf(mean + eps), y = x*scale;
// optional fused `* w` (the MUL node that always follows in llm_build_*)
// one 32-value block of a row, held four values a lane by eight consecutive lanes, quantised as k_quantize_q8_0 does it and written in PANEL order (quantize.hip;
// mmq_q80.hip: the 9 .. 128-column kernel's activations): row = column index, blk = block index along K
__device__ __forceinline__ void q80_panel_store(const float r0, const float r1, const float r2, const float r3, const int w8, const int64_t row, const int64_t blk, const int64_t nblk, char * __restrict__ dst) {
    float amax = fmaxf(fmaxf(fabsf(r0), fabsf(r1)), fmaxf(fabsf(r2), fabsf(r3)));
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
    const float d = amax / 127.0f;
    const float id = d != 0.0f ? 1.0f / d : 0.0f;
    const float v[4] = {r0, r1, r2, r3};
    uint32_t packed = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) packed |= (uint32_t) ((int) roundf(v[k] * id) & 0xFF) << (8 * k);
    char * tile = dst + ((size_t) (row >> 5) * (size_t) nblk + (size_t) blk) * 1152;
    const int cr = (int) (row & 31);
    *(uint32_t *) (tile + (w8 >> 2) * 512 + cr * 16 + (w8 & 3) * 4) = packed;
    if (w8 == 0) *(float *) (tile + 1024 + cr * 4) = h2f(f2h(d));
}
// Q80P (round 6): the row leaves as Q8_0 blocks in panel order instead of f32 — every reader is a Q8_0 mat-mul of 9 .. 128 columns (graph.cpp: q80_panel_consumers_only);
// the vector path only (the launcher checks), same arithmetic as the two kernels it replaces
template <bool Q80P>
__global__ void __launch_bounds__(256) k_rms_norm(const tdesc a, const tdesc d, const float eps, const float * __restrict__ w, char * __restrict__ q80 = nullptr) {
    __shared__ double sh[4];
    const int64_t row = blockIdx.x;
    const int64_t i1 = row % a.ne[1], i2 = (row / a.ne[1]) % a.ne[2], i3 = row / (a.ne[1] * a.ne[2]);
    const float * __restrict__ x = (const float *) (a.data + i1 * a.nb[1] + i2 * a.nb[2] + i3 * a.nb[3]);
    float * __restrict__ y = (float *) (d.data + i1 * d.nb[1] + i2 * d.nb[2] + i3 * d.nb[3]);
    const int64_t n = a.ne[0];
    const bool vec = (n & 3) == 0 && ((((uintptr_t) x) | ((uintptr_t) y) | ((uintptr_t) w)) & 15) == 0;
    double s = 0.0;
    if (vec) {
        // up to 4 independent 16-byte loads per thread per trip: a 4096-wide row is ONE round trip to L2
        const float4 * x4 = (const float4 *) x;
        const int64_t n4 = n >> 2;
        for (int64_t i0 = threadIdx.x; i0 < n4; i0 += 1024) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = (i0 + 256 * u) < n4 ? x4[i0 + 256 * u] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 4; ++u) s += (double) (v[u].x * v[u].x) + (double) (v[u].y * v[u].y) + (double) (v[u].z * v[u].z) + (double) (v[u].w * v[u].w);
        }
    } else {
        for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
            const float v = x[i];
            s += (double) (v * v);
        }
    }
    s = block_sum_d(s, sh);
    const float mean = (float) (s / (double) n);
    const float scale = 1.0f / sqrtf(mean + eps);
    if (vec) {
        const float4 * x4 = (const float4 *) x;
        const float4 * w4 = (const float4 *) w;
        float4 * y4 = (float4 *) y;
        const int64_t n4 = n >> 2;
        for (int64_t i0 = threadIdx.x; i0 < n4; i0 += 1024) {
            float4 v[4], g[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i = i0 + 256 * u;
                if (i < n4) { v[u] = x4[i]; if (w) g[u] = w4[i]; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i = i0 + 256 * u;
                if (i < n4) {
                    float4 r;
                    r.x = v[u].x * scale; r.y = v[u].y * scale; r.z = v[u].z * scale; r.w = v[u].w * scale;
                    if (w) { r.x = r.x * g[u].x; r.y = r.y * g[u].y; r.z = r.z * g[u].z; r.w = r.w * g[u].w; }
                    if constexpr (Q80P) q80_panel_store(r.x, r.y, r.z, r.w, (int) (i & 7), row, i >> 3, n >> 5, q80);
                    else y4[i] = r;
                }
            }
        }
    } else {
        for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
            float v = x[i] * scale;
            if (w) v = v * w[i];
            y[i] = v;
        }
    }
}
void launch_rms_norm(hipStream_t s, const tdesc & src, const tdesc & dst, float eps, const tdesc * mul_w) {
    const int64_t rows = src.ne[1] * src.ne[2] * src.ne[3];
    hipLaunchKernelGGL(k_rms_norm<false>, dim3((unsigned) rows), dim3(256), 0, s, src, dst, eps, mul_w ? (const float *) mul_w->data : nullptr, (char *) nullptr);
}
bool rms_norm_q80_panel_ok(const tdesc & src, const float * w) {  // the kernel's vector path, whole Q8_0 blocks
    return (src.ne[0] % 32) == 0 && src.nb[0] == 4 && ((((uintptr_t) src.data) | ((uintptr_t) w) | (uintptr_t) src.nb[1] | (uintptr_t) src.nb[2] | (uintptr_t) src.nb[3]) & 15) == 0;
}
void launch_rms_norm_mul_q80_panel(hipStream_t s, const tdesc & src, float eps, const float * w, void * q80_panel) {
    const int64_t rows = src.ne[1] * src.ne[2] * src.ne[3];
    hipLaunchKernelGGL(k_rms_norm<true>, dim3((unsigned) rows), dim3(256), 0, s, src, src, eps, w, (char *) q80_panel);
}

// ------------------------------------------------------------------------------------------------ ADD/SUB/MUL/DIV
__global__ void __launch_bounds__(256) k_binary(const int op, const tdesc a, const tdesc b, const tdesc d) {
    const int64_t row = blockIdx.x;
    const int64_t i1 = row % a.ne[1], i2 = (row / a.ne[1]) % a.ne[2], i3 = row / (a.ne[1] * a.ne[2]);
    const char * pa = a.data + i1 * a.nb[1] + i2 * a.nb[2] + i3 * a.nb[3];
    const char * pb = b.data + (i1 % b.ne[1]) * b.nb[1] + (i2 % b.ne[2]) * b.nb[2] + (i3 % b.ne[3]) * b.nb[3];
    char * pd = d.data + i1 * d.nb[1] + i2 * d.nb[2] + i3 * d.nb[3];
    for (int64_t i0 = threadIdx.x; i0 < a.ne[0]; i0 += blockDim.x) {
        const float x = *(const float *) (pa + i0 * a.nb[0]);
        const float y = *(const float *) (pb + (i0 % b.ne[0]) * b.nb[0]);
        float r;
        switch (op) {
            case GGML_OP_ADD: r = x + y; break;
            case GGML_OP_SUB: r = x - y; break;
            case GGML_OP_MUL: r = x * y; break;
            default: r = x / y; break;
        }
        *(float *) (pd + i0 * d.nb[0]) = r;
    }
}
__global__ void __launch_bounds__(256) k_reduce_parts(const reduce_parts p, const float * __restrict__ add, float * __restrict__ dst, const int64_t n4) {
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 acc = add ? ((const float4 *) add)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int d = 0; d < p.n; ++d) {
        const float4 v = ((const float4 *) p.part[d])[i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    ((float4 *) dst)[i] = acc;
}
void launch_reduce_parts(hipStream_t s, const reduce_parts & p, const float * add, float * dst, int64_t n) {
    const int64_t n4 = n / 4;  // (row lengths are multiples of 4: checked by the caller)
    hipLaunchKernelGGL(k_reduce_parts, dim3((unsigned) ((n4 + 255) / 256)), dim3(256), 0, s, p, add, dst, n4);
}

void launch_binary(hipStream_t s, int op, const tdesc & a, const tdesc & b, const tdesc & d) {
    const int64_t rows = a.ne[1] * a.ne[2] * a.ne[3];
    hipLaunchKernelGGL(k_binary, dim3((unsigned) rows), dim3(256), 0, s, op, a, b, d);
}

// ------------------------------------------------------------------------------------------------ SCALE / UNARY / GLU
// scale: dst = scale*x + bias (llama-box/patches/llama.cpp/ggml-cuda.patch:8-15)
__global__ void __launch_bounds__(256) k_scale(const float * __restrict__ x, float * __restrict__ y, const int64_t n, const float sc, const float bias) {
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t) gridDim.x * blockDim.x) {
        const float p = x[i] * sc;
        y[i] = p + bias;
    }
}
void launch_scale(hipStream_t s, const tdesc & src, const tdesc & dst, float scale, float bias) {
    const int64_t n = src.ne[0] * src.ne[1] * src.ne[2] * src.ne[3];
    const unsigned grid = (unsigned) std::min<int64_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(k_scale, dim3(grid), dim3(256), 0, s, (const float *) src.data, (float *) dst.data, n, scale, bias);
}

__global__ void __launch_bounds__(256) k_unary(const int uop, const float * __restrict__ x, float * __restrict__ y, const int64_t n) {
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t) gridDim.x * blockDim.x) {
        const float v = x[i];
        float r;
        switch (uop) {
            case GGML_UNARY_OP_SILU: r = silu_f(v); break;
            case GGML_UNARY_OP_RELU: r = v > 0.f ? v : 0.f; break;
            case GGML_UNARY_OP_NEG: r = -v; break;
            case GGML_UNARY_OP_EXP: r = expf(v); break;
            case GGML_UNARY_OP_TANH: r = tanhf(v); break;
            default: r = 1.f / (1.f + expf(-v)); break;  // SIGMOID
        }
        y[i] = r;
    }
}
void launch_unary(hipStream_t s, int uop, const tdesc & src, const tdesc & dst) {
    const int64_t n = src.ne[0] * src.ne[1] * src.ne[2] * src.ne[3];
    const unsigned grid = (unsigned) std::min<int64_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(k_unary, dim3(grid), dim3(256), 0, s, uop, (const float *) src.data, (float *) dst.data, n);
}

// swiglu: y = silu(a) * b (ggml_compute_forward_swiglu_f32); split form (two tensors) or the two halves of one row
// (grid.x = row, grid.y = chunk of 1024 values: a 32-row batch of Llama-3-8B's 14336 columns used to run on 32 workgroups, 56 scalar elements per thread — 23 us;
// VEC: rows and row starts 16-byte aligned, four values per thread as one float4)
template <bool VEC, bool Q80P = false>
__global__ void __launch_bounds__(256) k_swiglu(const char * __restrict__ pa, const char * __restrict__ pb, char * __restrict__ pd, const int64_t nc,
                                                const int64_t nba1, const int64_t nbb1, const int64_t nbd1) {
    const int64_t row = blockIdx.x;
    const float * a = (const float *) (pa + row * nba1);
    const float * b = (const float *) (pb + row * nbb1);
    float * y = (float *) (pd + row * nbd1);
    const int64_t i0 = ((int64_t) blockIdx.y * 256 + threadIdx.x) * 4;
    if constexpr (VEC) {
        if (i0 < nc) {  // (nc % 4 == 0)
            const float4 av = *(const float4 *) (a + i0), bv = *(const float4 *) (b + i0);
            if constexpr (Q80P) q80_panel_store(silu_f(av.x) * bv.x, silu_f(av.y) * bv.y, silu_f(av.z) * bv.z, silu_f(av.w) * bv.w, (int) ((i0 >> 2) & 7), row, i0 >> 5, nc >> 5, pd);  // (nc % 32 == 0)
            else
            *(float4 *) (y + i0) = make_float4(silu_f(av.x) * bv.x, silu_f(av.y) * bv.y, silu_f(av.z) * bv.z, silu_f(av.w) * bv.w);
        }
    } else {
        for (int64_t i = i0; i < i0 + 4 && i < nc; ++i) y[i] = silu_f(a[i]) * b[i];
    }
}
bool swiglu_q80_panel_ok(const tdesc & a, const tdesc * b, int64_t nc, int swapped) {
    (void) swapped;
    const char * pb = b ? b->data : a.data + nc * 4;
    return (nc % 32) == 0 && (((uintptr_t) a.data | (uintptr_t) pb | (uintptr_t) a.nb[1] | (uintptr_t) (b ? b->nb[1] : a.nb[1])) & 15) == 0;
}
void launch_swiglu_q80_panel(hipStream_t s, const tdesc & a, const tdesc * b, int64_t nc, int swapped, void * q80_panel) {
    const int64_t rows = a.ne[1] * a.ne[2] * a.ne[3];
    const char * pa = a.data;
    const char * pb = b ? b->data : a.data;
    if (!b) {
        pa += swapped ? nc * 4 : 0;
        pb += swapped ? 0 : nc * 4;
    }
    hipLaunchKernelGGL((k_swiglu<true, true>), dim3((unsigned) rows, (unsigned) ((nc + 1023) / 1024)), dim3(256), 0, s, pa, pb, (char *) q80_panel, nc, a.nb[1], b ? b->nb[1] : a.nb[1], (int64_t) 0);
}
void launch_swiglu(hipStream_t s, const tdesc & a, const tdesc * b, const tdesc & d, int swapped) {
    const int64_t rows = a.ne[1] * a.ne[2] * a.ne[3];
    const int64_t nc = d.ne[0];
    const char * pa = a.data;
    const char * pb = b ? b->data : a.data;
    if (!b) {
        pa += swapped ? nc * 4 : 0;
        pb += swapped ? 0 : nc * 4;
    }
    const int64_t nbb1 = b ? b->nb[1] : a.nb[1];
    const bool vec = (nc % 4) == 0 && (((uintptr_t) pa | (uintptr_t) pb | (uintptr_t) d.data | (uintptr_t) a.nb[1] | (uintptr_t) nbb1 | (uintptr_t) d.nb[1]) & 15) == 0;
    const dim3 grid((unsigned) rows, (unsigned) ((nc + 1023) / 1024));
    if (vec) hipLaunchKernelGGL(k_swiglu<true>, grid, dim3(256), 0, s, pa, pb, d.data, nc, a.nb[1], nbb1, d.nb[1]);
    else hipLaunchKernelGGL(k_swiglu<false>, grid, dim3(256), 0, s, pa, pb, d.data, nc, a.nb[1], nbb1, d.nb[1]);
}

// ------------------------------------------------------------------------------------------------ CPY / DUP / CONT
// ggml_compute_forward_dup: same logical element order on both sides, f32/f16 conversion (RNE)
template <typename TS, typename TDST> __device__ __forceinline__ TDST conv(TS v);
template <> __device__ __forceinline__ float conv<float, float>(float v) { return v; }
template <> __device__ __forceinline__ uint16_t conv<float, uint16_t>(float v) { return f2h(v); }
template <> __device__ __forceinline__ float conv<uint16_t, float>(uint16_t v) { return h2f(v); }
template <> __device__ __forceinline__ uint16_t conv<uint16_t, uint16_t>(uint16_t v) { return v; }
template <> __device__ __forceinline__ int32_t conv<int32_t, int32_t>(int32_t v) { return v; }

template <typename TS, typename TDST> __global__ void __launch_bounds__(256) k_cpy(const tdesc a, const tdesc d, const int64_t n) {
    for (int64_t e = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t) gridDim.x * blockDim.x) {
        int64_t r = e;
        const int64_t s0 = r % a.ne[0]; r /= a.ne[0];
        const int64_t s1 = r % a.ne[1]; r /= a.ne[1];
        const int64_t s2 = r % a.ne[2]; r /= a.ne[2];
        const int64_t s3 = r;
        r = e;
        const int64_t d0 = r % d.ne[0]; r /= d.ne[0];
        const int64_t d1 = r % d.ne[1]; r /= d.ne[1];
        const int64_t d2 = r % d.ne[2]; r /= d.ne[2];
        const int64_t d3 = r;
        const TS v = *(const TS *) (a.data + s0 * a.nb[0] + s1 * a.nb[1] + s2 * a.nb[2] + s3 * a.nb[3]);
        *(TDST *) (d.data + d0 * d.nb[0] + d1 * d.nb[1] + d2 * d.nb[2] + d3 * d.nb[3]) = conv<TS, TDST>(v);
    }
}
void launch_cpy(hipStream_t s, const tdesc & a, const tdesc & d) {
    const int64_t n = a.ne[0] * a.ne[1] * a.ne[2] * a.ne[3];
    const unsigned grid = (unsigned) std::min<int64_t>((n + 255) / 256, 4096);
    const int st = a.type, dt = d.type;
    if (st == GGML_TYPE_F32 && dt == GGML_TYPE_F32) hipLaunchKernelGGL((k_cpy<float, float>), dim3(grid), dim3(256), 0, s, a, d, n);
    else if (st == GGML_TYPE_F32 && dt == GGML_TYPE_F16) hipLaunchKernelGGL((k_cpy<float, uint16_t>), dim3(grid), dim3(256), 0, s, a, d, n);
    else if (st == GGML_TYPE_F16 && dt == GGML_TYPE_F32) hipLaunchKernelGGL((k_cpy<uint16_t, float>), dim3(grid), dim3(256), 0, s, a, d, n);
    else if (st == GGML_TYPE_F16 && dt == GGML_TYPE_F16) hipLaunchKernelGGL((k_cpy<uint16_t, uint16_t>), dim3(grid), dim3(256), 0, s, a, d, n);
    else if (st == GGML_TYPE_I32 && dt == GGML_TYPE_I32) hipLaunchKernelGGL((k_cpy<int32_t, int32_t>), dim3(grid), dim3(256), 0, s, a, d, n);
    else { MI_ERR("launch_cpy: unsupported types %d -> %d", st, dt); abort(); }
}

// ------------------------------------------------------------------------------------------------ GET_ROWS
// dequantize_row_{q8_0,q4_K,q5_K,q6_K} element formulas: one workgroup per gathered row
__device__ __forceinline__ void k4_scale_min(int j, const uint8_t * q, int & sc, int & m) {
    if (j < 4) { sc = q[j] & 63; m = q[j + 4] & 63; }
    else { sc = (q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4); m = (q[j + 4] >> 4) | ((q[j] >> 6) << 4); }
}
__device__ __forceinline__ float dequant_elem(const int type, const uint8_t * __restrict__ row, const int64_t i) {
    switch (type) {
        case GGML_TYPE_F32: return ((const float *) row)[i];
        case GGML_TYPE_F16: return h2f(((const uint16_t *) row)[i]);
        case GGML_TYPE_Q8_0: {
            const uint8_t * b = row + (i >> 5) * 34;
            return (float) (int8_t) b[2 + (i & 31)] * h2f(ld16(b));
        }
        case GGML_TYPE_Q4_K:
        case GGML_TYPE_Q5_K: {
            const bool q5 = type == GGML_TYPE_Q5_K;
            const uint8_t * b = row + (i >> 8) * (q5 ? 176 : 144);
            const int e = (int) (i & 255), is = e >> 5, l = e & 31, j = e >> 6;
            const float d = h2f(ld16(b)), mn = h2f(ld16(b + 2));
            int sc, m;
            k4_scale_min(is, b + 4, sc, m);
            const uint8_t * qs = b + (q5 ? 48 : 16) + 32 * j;
            int q = (is & 1) ? (qs[l] >> 4) : (qs[l] & 0xF);
            if (q5) q += ((b[16 + l] >> is) & 1) ? 16 : 0;
            const float d1 = d * (float) sc, m1 = mn * (float) m;
            return d1 * (float) q - m1;
        }
        case GGML_TYPE_Q6_K: {
            const uint8_t * b = row + (i >> 8) * 210;
            const int e = (int) (i & 255), h = e >> 7, r = e & 127, k = r >> 5, l = r & 31;
            const uint8_t * ql = b + 64 * h, * qh = b + 128 + 32 * h;
            const int8_t * sc = (const int8_t *) (b + 192 + 8 * h);
            const int lo = (k & 1) ? ql[l + 32] : ql[l];
            const int nib = (k & 2) ? (lo >> 4) : (lo & 0xF);
            const int q = (int) (int8_t) (nib | (((qh[l] >> (2 * k)) & 3) << 4)) - 32;
            const float d = h2f(ld16(b + 208));
            return d * (float) sc[(l >> 4) + 2 * k] * (float) q;
        }
        default: return 0.0f;
    }
}
// (a decode step gathers ONE row: 1024 threads and independent iterations keep its chain of dependent byte loads short)
__global__ void __launch_bounds__(1024) k_get_rows(const tdesc a, const tdesc idx, const tdesc d) {
    const int64_t r = blockIdx.x;
    const int64_t i10 = r % idx.ne[0], i11 = (r / idx.ne[0]) % idx.ne[1], i12 = r / (idx.ne[0] * idx.ne[1]);
    const int32_t i01 = *(const int32_t *) (idx.data + i10 * idx.nb[0] + i11 * idx.nb[1] + i12 * idx.nb[2]);
    const uint8_t * row = (const uint8_t *) (a.data + (int64_t) i01 * a.nb[1] + i11 * a.nb[2] + i12 * a.nb[3]);
    float * y = (float *) (d.data + i10 * d.nb[1] + i11 * d.nb[2] + i12 * d.nb[3]);
#pragma unroll 4
    for (int64_t i = threadIdx.x; i < a.ne[0]; i += blockDim.x) y[i] = dequant_elem(a.type, row, i);
}
void launch_get_rows(hipStream_t s, const tdesc & a, const tdesc & idx, const tdesc & d) {
    const int64_t rows = idx.ne[0] * idx.ne[1] * idx.ne[2];
    hipLaunchKernelGGL(k_get_rows, dim3((unsigned) rows), dim3((unsigned) std::min<int64_t>(1024, std::max<int64_t>(64, (a.ne[0] + 63) / 64 * 64))), 0, s, a, idx, d);
}

// ------------------------------------------------------------------------------------------------ SET_ROWS
// ggml_compute_forward_set_rows_f32: dst row idx[i] = convert(src row i); I64 indices; f32 -> f32 / f16 (RNE)
__global__ void __launch_bounds__(256) k_set_rows(const tdesc a, const tdesc idx, const tdesc d) {
    const int64_t nc = a.ne[0];
    if (nc == 1) {  // element scatter (transposed V cache): one thread per row
        const int64_t total = a.ne[1] * a.ne[2] * a.ne[3];
        for (int64_t r = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; r < total; r += (int64_t) gridDim.x * blockDim.x) {
            const int64_t i01 = r % a.ne[1], i02 = (r / a.ne[1]) % a.ne[2], i03 = r / (a.ne[1] * a.ne[2]);
            const int64_t i1 = *(const int64_t *) (idx.data + i01 * idx.nb[0] + (i02 % idx.ne[1]) * idx.nb[1] + (i03 % idx.ne[2]) * idx.nb[2]);
            const float v = *(const float *) (a.data + i01 * a.nb[1] + i02 * a.nb[2] + i03 * a.nb[3]);
            char * p = d.data + i1 * d.nb[1] + i02 * d.nb[2] + i03 * d.nb[3];
            if (d.type == GGML_TYPE_F16) *(uint16_t *) p = f2h(v); else *(float *) p = v;
        }
        return;
    }
    const int64_t r = blockIdx.x;
    const int64_t i01 = r % a.ne[1], i02 = (r / a.ne[1]) % a.ne[2], i03 = r / (a.ne[1] * a.ne[2]);
    const int64_t i1 = *(const int64_t *) (idx.data + i01 * idx.nb[0] + (i02 % idx.ne[1]) * idx.nb[1] + (i03 % idx.ne[2]) * idx.nb[2]);
    const float * __restrict__ x = (const float *) (a.data + i01 * a.nb[1] + i02 * a.nb[2] + i03 * a.nb[3]);
    char * __restrict__ p = d.data + i1 * d.nb[1] + i02 * d.nb[2] + i03 * d.nb[3];
    if ((nc & 3) == 0 && ((((uintptr_t) x) | ((uintptr_t) p)) & 15) == 0) {
        const int64_t n4 = nc >> 2;
        for (int64_t i0 = threadIdx.x; i0 < n4; i0 += 1024) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) if (i0 + 256 * u < n4) v[u] = ((const float4 *) x)[i0 + 256 * u];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i = i0 + 256 * u;
                if (i < n4) {
                    if (d.type == GGML_TYPE_F16) {
                        uint2 h;
                        h.x = (uint32_t) f2h(v[u].x) | ((uint32_t) f2h(v[u].y) << 16);
                        h.y = (uint32_t) f2h(v[u].z) | ((uint32_t) f2h(v[u].w) << 16);
                        ((uint2 *) p)[i] = h;
                    } else {
                        ((float4 *) p)[i] = v[u];
                    }
                }
            }
        }
        return;
    }
    if (d.type == GGML_TYPE_F16) for (int64_t i = threadIdx.x; i < nc; i += blockDim.x) ((uint16_t *) p)[i] = f2h(x[i]);
    else for (int64_t i = threadIdx.x; i < nc; i += blockDim.x) ((float *) p)[i] = x[i];
}
void launch_set_rows(hipStream_t s, const tdesc & a, const tdesc & idx, const tdesc & d) {
    const int64_t rows = a.ne[1] * a.ne[2] * a.ne[3];
    const unsigned grid = a.ne[0] == 1 ? (unsigned) std::min<int64_t>((rows + 255) / 256, 4096) : (unsigned) rows;
    hipLaunchKernelGGL(k_set_rows, dim3(grid), dim3(256), 0, s, a, idx, d);
}

// ------------------------------------------------------------------------------------------------ ARGMAX
__global__ void __launch_bounds__(256) k_argmax(const tdesc a, const tdesc d) {
    __shared__ float smax[4];
    __shared__ int sidx[4];
    const int64_t row = blockIdx.x;
    const float * x = (const float *) (a.data + row * a.nb[1]);
    float mx = -INFINITY;
    int mi = 0x7FFFFFFF;
    for (int64_t i = threadIdx.x; i < a.ne[0]; i += blockDim.x) {
        const float v = x[i];
        if (v > mx) { mx = v; mi = (int) i; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(mx, o, 64);
        const int oi = __shfl_xor(mi, o, 64);
        if (om > mx || (om == mx && oi < mi)) { mx = om; mi = oi; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { smax[wave] = mx; sidx[wave] = mi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) if (smax[w] > mx || (smax[w] == mx && sidx[w] < mi)) { mx = smax[w]; mi = sidx[w]; }
        ((int32_t *) d.data)[row] = mi == 0x7FFFFFFF ? 0 : mi;
    }
}
void launch_argmax(hipStream_t s, const tdesc & a, const tdesc & d) {
    hipLaunchKernelGGL(k_argmax, dim3((unsigned) a.ne[1]), dim3(256), 0, s, a, d);
}

// ------------------------------------------------------------------------------------------------ ROPE
// ggml_compute_forward_rope_f32/_f16 (patch->ggml-cpu/ops.cpp:6204,:6390; llama-box/patches/llama.cpp/mrope.patch:5-26).
// theta for pair i is produced by the SAME chain of f32 multiplies as the CPU's cache loop (theta *= theta_scale),
// so the angle is bit-identical; cosf/sinf are the accurate device versions.  normal: pairs (2i, 2i+1); neox: (i, i+n/2).
__device__ __forceinline__ float yarn_ramp(const float low, const float high, const int i0) {
    const float y = ((float) (i0 / 2) - low) / fmaxf(0.001f, high - low);
    return 1.0f - fminf(1.0f, fmaxf(0.0f, y));
}
template <bool F16IO> __global__ void __launch_bounds__(64) k_rope(const tdesc a, const tdesc pos, const float * __restrict__ ff, const tdesc d, const rope_params p,
                                                               const float theta_scale, const float corr0, const float corr1) {
    // grid: (head i1, token i2, batch i3); lanes = rotation pairs, so all loads of a launch are independent
    const int64_t i1 = blockIdx.x, i2 = blockIdx.y, i3 = blockIdx.z;
    const int n_pairs = p.n_dims / 2;
    const float pos_f = (float) *(const int32_t *) (pos.data + i2 * pos.nb[0]);
    const char * __restrict__ src = a.data + i1 * a.nb[1] + i2 * a.nb[2] + i3 * a.nb[3];
    char * __restrict__ dst = d.data + i1 * d.nb[1] + i2 * d.nb[2] + i3 * d.nb[3];
    const bool neox = (p.mode & GGML_ROPE_TYPE_NEOX) != 0;
    for (int ip = threadIdx.x; ip < n_pairs; ip += blockDim.x) {
        const int ia = neox ? ip : 2 * ip, ib = neox ? ip + n_pairs : 2 * ip + 1;
        float x0, x1;
        if (F16IO) { x0 = h2f(((const uint16_t *) src)[ia]); x1 = h2f(((const uint16_t *) src)[ib]); }
        else { x0 = ((const float *) src)[ia]; x1 = ((const float *) src)[ib]; }
        float cs, sn;
        rope_cos_sin(ip, pos_f, ff, rope_consts{theta_scale, p.freq_scale, p.ext_factor, p.attn_factor, corr0, corr1}, cs, sn);
        if (F16IO) {
            ((uint16_t *) dst)[ia] = f2h(x0 * cs - x1 * sn);
            ((uint16_t *) dst)[ib] = f2h(x0 * sn + x1 * cs);
        } else {
            ((float *) dst)[ia] = x0 * cs - x1 * sn;
            ((float *) dst)[ib] = x0 * sn + x1 * cs;
        }
    }
    for (int64_t i0 = p.n_dims + threadIdx.x; i0 < a.ne[0]; i0 += blockDim.x) {  // pass-through tail beyond n_dims
        if (F16IO) ((uint16_t *) dst)[i0] = ((const uint16_t *) src)[i0];
        else ((float *) dst)[i0] = ((const float *) src)[i0];
    }
}
// ggml_rope_multi (ggml_mrope_cache_init): four position streams, one per section of the rotation pairs; every stream's angle
// advances by theta_scale per pair, and in vision mode restarts from its position when its section begins.  A lane replays
// that recurrence up to its own pair — a few hundred multiplies, and exactly the CPU's chain including its corner cases
// (empty sections, sections that do not fill a cycle).  Pairs are (ic, ic + n_dims/2), vision: (ic, ic + n_dims) over the row.
template <bool F16IO> __global__ void __launch_bounds__(64) k_rope_multi(const tdesc a, const int32_t * __restrict__ pos, const float * __restrict__ ff, const tdesc d,
                                                                     const rope_params p, const float theta_scale, const float corr0, const float corr1) {
    const int64_t i1 = blockIdx.x, i2 = blockIdx.y, i3 = blockIdx.z, ne2 = a.ne[2];
    const bool vision = p.mode == GGML_ROPE_TYPE_VISION;
    const int n_pairs = vision ? (int) (a.ne[0] / 2) : p.n_dims / 2;
    const int off = vision ? p.n_dims : p.n_dims / 2;
    const float base_t = (float) pos[i2], base_h = (float) pos[i2 + ne2], base_w = (float) pos[i2 + 2 * ne2], base_e = (float) pos[i2 + 3 * ne2];
    const int s0 = p.sections[0], sec_w = s0 + p.sections[1], sec_e = sec_w + p.sections[2], sect_dims = sec_e + p.sections[3];
    const char * __restrict__ src = a.data + i1 * a.nb[1] + i2 * a.nb[2] + i3 * a.nb[3];
    char * __restrict__ dst = d.data + i1 * d.nb[1] + i2 * d.nb[2] + i3 * d.nb[3];
    for (int ic = threadIdx.x; ic < n_pairs; ic += blockDim.x) {
        float tt = base_t, th = base_h, tw = base_w, te = base_e, theta = 0.0f;
        for (int k = 0; k <= ic; ++k) {
            const int sector = k % sect_dims;
            if (vision) {
                if (sector == 0) tt = base_t;
                else if (sector == s0) th = base_h;
                else if (sector == sec_w) tw = base_w;
                else if (sector == sec_e) te = base_e;
            }
            if (k == ic) {
                theta = tt;
                if (sector >= s0 && sector < sec_w) theta = th;
                else if (sector >= sec_w && sector < sec_e) theta = tw;
                else if (sector >= sec_e) theta = te;
            }
            tt *= theta_scale; tw *= theta_scale; th *= theta_scale; te *= theta_scale;
        }
        float cs, sn;
        rope_yarn_dev(theta, ic, ff, rope_consts{theta_scale, p.freq_scale, p.ext_factor, p.attn_factor, corr0, corr1}, cs, sn);
        float x0, x1;
        if (F16IO) { x0 = h2f(((const uint16_t *) src)[ic]); x1 = h2f(((const uint16_t *) src)[ic + off]); }
        else { x0 = ((const float *) src)[ic]; x1 = ((const float *) src)[ic + off]; }
        if (F16IO) {
            ((uint16_t *) dst)[ic] = f2h(x0 * cs - x1 * sn);
            ((uint16_t *) dst)[ic + off] = f2h(x0 * sn + x1 * cs);
        } else {
            ((float *) dst)[ic] = x0 * cs - x1 * sn;
            ((float *) dst)[ic + off] = x0 * sn + x1 * cs;
        }
    }
    if (!vision) {
        for (int64_t i0 = p.n_dims + threadIdx.x; i0 < a.ne[0]; i0 += blockDim.x) {
            if (F16IO) ((uint16_t *) dst)[i0] = ((const uint16_t *) src)[i0];
            else ((float *) dst)[i0] = ((const float *) src)[i0];
        }
    }
}
__global__ void __launch_bounds__(256) k_rope_table(const int32_t * __restrict__ pos, const float * __restrict__ ff, const rope_consts rc, const int half, float * __restrict__ tab) {
    const int tok = blockIdx.x, ip = threadIdx.x;
    if (ip >= half) return;
    float cs, sn;
    rope_cos_sin(ip, (float) pos[tok], ff, rc, cs, sn);
    *(float2 *) (tab + ((size_t) tok * half + ip) * 2) = make_float2(cs, sn);
}
// ---- the head of a decode step in ONE launch (round 6): GET_ROWS of the token embeddings, the F32 -> F16 cast of the attention mask (llama.cpp builds the mask in
// F32 and casts it for flash attention) and the (cos, sin) table of the step's positions used to be three dependent launches at the ~4.5 us floor each, in front of
// layer 0.  They do not depend on each other: blocks [0, n_rows) gather a row each (k_get_rows), the next n_cpy blocks cast (k_cpy's contiguous case), the last
// n_tok blocks fill the rotary table (k_rope_table).  Element formulas are the three kernels' own.
struct step_head_args {
    tdesc gr_a, gr_idx, gr_d;
    int n_rows;
    const float * cp_src;
    uint16_t * cp_dst;
    int64_t cp_n;
    int n_cpy;
    const int32_t * pos;
    const float * ff;
    rope_consts rc;
    int half, n_tok;
    float * tab;
};
__global__ void __launch_bounds__(1024) k_step_head(const step_head_args h) {
    int b = (int) blockIdx.x;
    if (b < h.n_rows) {
        const tdesc & a = h.gr_a, & idx = h.gr_idx, & d = h.gr_d;
        const int64_t r = b;
        const int64_t i10 = r % idx.ne[0], i11 = (r / idx.ne[0]) % idx.ne[1], i12 = r / (idx.ne[0] * idx.ne[1]);
        const int32_t i01 = *(const int32_t *) (idx.data + i10 * idx.nb[0] + i11 * idx.nb[1] + i12 * idx.nb[2]);
        const uint8_t * row = (const uint8_t *) (a.data + (int64_t) i01 * a.nb[1] + i11 * a.nb[2] + i12 * a.nb[3]);
        float * y = (float *) (d.data + i10 * d.nb[1] + i11 * d.nb[2] + i12 * d.nb[3]);
#pragma unroll 4
        for (int64_t i = threadIdx.x; i < a.ne[0]; i += blockDim.x) y[i] = dequant_elem(a.type, row, i);
        return;
    }
    b -= h.n_rows;
    if (b < h.n_cpy) {
        for (int64_t e = (int64_t) b * blockDim.x + threadIdx.x; e < h.cp_n; e += (int64_t) h.n_cpy * blockDim.x) h.cp_dst[e] = f2h(h.cp_src[e]);
        return;
    }
    b -= h.n_cpy;
    if (b < h.n_tok && (int) threadIdx.x < h.half) {
        float cs, sn;
        rope_cos_sin((int) threadIdx.x, (float) h.pos[b], h.ff, h.rc, cs, sn);
        *(float2 *) (h.tab + ((size_t) b * h.half + threadIdx.x) * 2) = make_float2(cs, sn);
    }
}
void rope_host_consts(const rope_params & p, float & theta_scale, float & c0, float & c1);
// any of the three parts may be absent (idx == nullptr / cp_n == 0 / n_tok == 0)
void launch_step_head(hipStream_t s, const tdesc * a, const tdesc * idx, const tdesc * d, const float * cp_src, void * cp_dst, int64_t cp_n,
                      const int32_t * pos, const float * ff, const rope_params * p, int n_tok, float * tab) {
    step_head_args h{};
    if (a) {
        h.gr_a = *a;
        h.gr_idx = *idx;
        h.gr_d = *d;
        h.n_rows = (int) (idx->ne[0] * idx->ne[1] * idx->ne[2]);
    }
    h.cp_src = cp_src;
    h.cp_dst = (uint16_t *) cp_dst;
    h.cp_n = cp_n;
    h.n_cpy = cp_n > 0 ? (int) std::min<int64_t>(64, (cp_n + 4095) / 4096) : 0;
    if (n_tok > 0) {
        rope_host_consts(*p, h.rc.theta_scale, h.rc.corr0, h.rc.corr1);
        h.rc.freq_scale = p->freq_scale;
        h.rc.ext_factor = p->ext_factor;
        h.rc.attn_factor = p->attn_factor;
        h.half = p->n_dims / 2;
        h.n_tok = n_tok;
        h.pos = pos;
        h.ff = ff;
        h.tab = tab;
    }
    const int grid = h.n_rows + h.n_cpy + h.n_tok;
    if (grid <= 0) return;
    const int64_t width = std::max<int64_t>(a ? a->ne[0] : 0, std::max<int64_t>(cp_n > 0 ? 1024 : 0, h.half));
    hipLaunchKernelGGL(k_step_head, dim3((unsigned) grid), dim3((unsigned) std::min<int64_t>(1024, std::max<int64_t>(64, (width + 63) / 64 * 64))), 0, s, h);
}
void rope_host_consts(const rope_params & p, float & theta_scale, float & c0, float & c1) {
    theta_scale = powf(p.freq_base, -2.0f / (float) p.n_dims);
    // ggml_rope_yarn_corr_dims
    auto corr_dim = [&](float n_rot) { return (float) p.n_dims * logf((float) p.n_ctx_orig / (n_rot * 2.0f * (float) M_PI)) / (2.0f * logf(p.freq_base)); };
    c0 = fmaxf(0.0f, floorf(corr_dim(p.beta_fast)));
    c1 = fminf((float) (p.n_dims - 1), ceilf(corr_dim(p.beta_slow)));
}
void launch_rope_table(hipStream_t s, const int32_t * pos, const float * ff, const rope_params & p, int n_tok, float * tab) {
    rope_consts rc{};
    rope_host_consts(p, rc.theta_scale, rc.corr0, rc.corr1);
    rc.freq_scale = p.freq_scale;
    rc.ext_factor = p.ext_factor;
    rc.attn_factor = p.attn_factor;
    const int half = p.n_dims / 2;
    hipLaunchKernelGGL(k_rope_table, dim3((unsigned) n_tok), dim3((unsigned) ((half + 63) / 64 * 64)), 0, s, pos, ff, rc, half, tab);
}
// Batches: ROPE(q), ROPE(k) -> SET_ROWS(k cache), SET_ROWS(v cache) as ONE launch (four nodes of every layer; at a few
// dozen tokens each of them sits at the dependent-launch floor).  Query heads are rotated f32 -> f32 (in place when the allocator
// made the rope in-place), key heads are rotated and stored as f16 into the cache row the token's index names, value heads
// are converted.  Arithmetic per element is that of k_rope
// followed by k_set_rows (f32 result, then one f16 rounding), so fused and unfused graphs agree bit for bit.
__global__ void __launch_bounds__(256) k_rope_qk_store(const rope_store_args a) {
    // one workgroup per token; a thread owns rotation pair (tid & 63) [+ 64, ...] — its (cos, sin) is computed ONCE and serves every
    // head — and the four waves share out the head slots: [0, nh) query heads, [nh, nh + nkv) key heads, then value heads
    const int64_t t = blockIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float pos_f = (float) a.pos[t];
    const int n_pairs = a.p.n_dims / 2;
    const bool neox = (a.p.mode & GGML_ROPE_TYPE_NEOX) != 0;
    const rope_consts rc{a.theta_scale, a.p.freq_scale, a.p.ext_factor, a.p.attn_factor, a.corr0, a.corr1};
    const int64_t row = a.idx[t];
    // element i of head h of source s (0 q, 1 k, 2 v): from memory, or assembled from the split-K partial products of its projection
    auto ld = [&](const int sidx, const float * src, const int h, const int i) -> float {
        const float * part = a.sk[sidx].part;
        if (part == nullptr) return src[i];
        const int64_t e = t * a.sk[sidx].n + (int64_t) h * a.head_dim + i;
        float v = part[e];
        for (int k = 1; k < a.ks; ++k) v += part[(int64_t) k * a.sk[sidx].mn + e];
        if (a.sk[sidx].bias) v += a.sk[sidx].bias[(int64_t) h * a.head_dim + i];
        return v;
    };
    // blockIdx.y: this workgroup's share of the head slots (one share per token for big batches, where recomputing the angles
    // per head is the cost; many shares for a few dozen tokens, where filling the chip is)
    const int n_all = a.nh + 2 * a.nkv, per = (n_all + (int) gridDim.y - 1) / (int) gridDim.y;
    const int s_lo = (int) blockIdx.y * per, n_slots = min(n_all, s_lo + per);
    const int rot_hi = min(n_slots, a.nh + a.nkv);
    for (int ip0 = 0; ip0 < n_pairs && s_lo < rot_hi; ip0 += 64) {
        const int ip = ip0 + lane;
        const bool act = ip < n_pairs;
        float cs = 1.0f, sn = 0.0f;
        if (act) rope_cos_sin(ip, pos_f, a.ff, rc, cs, sn);
        const int ia = neox ? ip : 2 * ip, ib = neox ? ip + n_pairs : 2 * ip + 1;
        for (int slot = s_lo + w; slot < rot_hi; slot += 4) {
            if (!act) continue;
            if (slot < a.nh) {
                const float * src = (const float *) (a.q_src + slot * a.q_nb1 + t * a.q_nb2);
                float * dst = (float *) (a.q_dst + slot * a.qd_nb1 + t * a.qd_nb2);
                const float x0 = ld(0, src, slot, ia), x1 = ld(0, src, slot, ib);
                dst[ia] = x0 * cs - x1 * sn;
                dst[ib] = x0 * sn + x1 * cs;
            } else {
                const int h = slot - a.nh;
                const float * src = (const float *) (a.k_src + h * a.k_nb1 + t * a.k_nb2);
                uint16_t * dst = (uint16_t *) (a.k_cache + row * a.kc_nb1) + (int64_t) h * a.head_dim;
                const float x0 = ld(1, src, h, ia), x1 = ld(1, src, h, ib);
                dst[ia] = f2h(x0 * cs - x1 * sn);
                dst[ib] = f2h(x0 * sn + x1 * cs);
            }
        }
    }
    // pass-through tail of partially rotated heads, and the value heads
    for (int slot = s_lo + w; slot < n_slots; slot += 4) {
        if (slot < a.nh) {
            const float * src = (const float *) (a.q_src + slot * a.q_nb1 + t * a.q_nb2);
            float * dst = (float *) (a.q_dst + slot * a.qd_nb1 + t * a.qd_nb2);
            for (int i0 = a.p.n_dims + lane; i0 < a.head_dim; i0 += 64) dst[i0] = ld(0, src, slot, i0);
        } else if (slot < a.nh + a.nkv) {
            const int h = slot - a.nh;
            const float * src = (const float *) (a.k_src + h * a.k_nb1 + t * a.k_nb2);
            uint16_t * dst = (uint16_t *) (a.k_cache + row * a.kc_nb1) + (int64_t) h * a.head_dim;
            for (int i0 = a.p.n_dims + lane; i0 < a.head_dim; i0 += 64) dst[i0] = f2h(ld(1, src, h, i0));
        } else {
            const int h = slot - a.nh - a.nkv;
            const float * src = (const float *) (a.v_src + h * a.v_nb1 + t * a.v_nb2);
            if (a.v_idx) {
                const int64_t * ix = a.v_idx + ((int64_t) t * a.nkv + h) * a.head_dim;
                for (int i0 = lane; i0 < a.head_dim; i0 += 64) ((uint16_t *) a.v_cache)[ix[i0]] = f2h(ld(2, src, h, i0));
            } else {
                uint16_t * dst = (uint16_t *) (a.v_cache + row * a.vc_nb1) + (int64_t) h * a.head_dim;
                for (int i0 = lane; i0 < a.head_dim; i0 += 64) dst[i0] = f2h(ld(2, src, h, i0));
            }
        }
    }
}
// The same work for a prompt micro-batch, vectorised (round 4): a workgroup per token, a thread per CHUNK — 4 consecutive head dims ("normal"
// rotation: two adjacent pairs) or 4 + 4 dims half a head apart (NeoX) — of every query / key / value head; ALL of a thread's loads are
// requested first (the per-slot loop above waits one L2 round trip per head slot: 14 us for 512 tokens of Llama-3-8B against 23 MB of traffic),
// and (cos, sin) come from the per-graph-run table of k_rope_table (the chain of multiplies and the accurate cosf / sinf once per run, not once
// per layer).  Requires: whole heads rotated (n_dims == head_dim), head_dim % 8 == 0, projections in memory (no split-K partials); the V cache row-major, or transposed with llama.cpp's per-element indices (non-flash path).
// The arithmetic per element is that of k_rope_qk_store (x0 cs - x1 sn, x0 sn + x1 cs; f2h for the cache rows).
template <bool NEOX>
__global__ void __launch_bounds__(256) k_rope_qk_store_vec(const rope_store_args a, const float * __restrict__ tab) {
    const int64_t t = blockIdx.x;
    const int tid = threadIdx.x, hd = a.head_dim, half = hd >> 1;
    const int64_t row = a.idx[t];
    const int cph = NEOX ? hd / 8 : hd / 4;            // chunks per head
    const int n_rot = (a.nh + a.nkv) * cph, n_v = a.nkv * (hd / 4);
    constexpr int MAXC = 6;                            // chunks per thread and pass (Llama-3-8B: 1280 rotated chunks -> 5, + 1 value chunk)
    for (int c0 = 0; c0 < n_rot + n_v; c0 += 256 * MAXC) {
        float4 lo[MAXC], hi[MAXC], cs[MAXC], cs2[MAXC];
#pragma unroll
        for (int k = 0; k < MAXC; ++k) {
            const int c = c0 + tid + 256 * k;
            if (c < n_rot) {
                const int slot = c / cph, j = c - slot * cph;
                const float * src = slot < a.nh ? (const float *) (a.q_src + (int64_t) slot * a.q_nb1 + t * a.q_nb2) : (const float *) (a.k_src + (int64_t) (slot - a.nh) * a.k_nb1 + t * a.k_nb2);
                lo[k] = *(const float4 *) (src + 4 * j);
                if (NEOX) {
                    hi[k] = *(const float4 *) (src + half + 4 * j);
                    cs[k] = *(const float4 *) (tab + ((size_t) t * half + 4 * j) * 2);       // pairs 4j, 4j + 1
                    cs2[k] = *(const float4 *) (tab + ((size_t) t * half + 4 * j + 2) * 2);  // pairs 4j + 2, 4j + 3
                } else {
                    cs[k] = *(const float4 *) (tab + ((size_t) t * half + 2 * j) * 2);       // pairs 2j, 2j + 1
                }
            } else if (c < n_rot + n_v) {
                const int cv = c - n_rot, h = cv / (hd / 4), j = cv - h * (hd / 4);
                lo[k] = *(const float4 *) ((const float *) (a.v_src + (int64_t) h * a.v_nb1 + t * a.v_nb2) + 4 * j);
            }
        }
#pragma unroll
        for (int k = 0; k < MAXC; ++k) {
            const int c = c0 + tid + 256 * k;
            if (c < n_rot) {
                const int slot = c / cph, j = c - slot * cph;
                float4 o0, o1;
                if (NEOX) {  // element i pairs with i + half: (cos, sin) of pair i
                    o0 = make_float4(lo[k].x * cs[k].x - hi[k].x * cs[k].y, lo[k].y * cs[k].z - hi[k].y * cs[k].w, lo[k].z * cs2[k].x - hi[k].z * cs2[k].y, lo[k].w * cs2[k].z - hi[k].w * cs2[k].w);
                    o1 = make_float4(lo[k].x * cs[k].y + hi[k].x * cs[k].x, lo[k].y * cs[k].w + hi[k].y * cs[k].z, lo[k].z * cs2[k].y + hi[k].z * cs2[k].x, lo[k].w * cs2[k].w + hi[k].w * cs2[k].z);
                } else {     // elements (2i, 2i + 1)
                    o0 = make_float4(lo[k].x * cs[k].x - lo[k].y * cs[k].y, lo[k].x * cs[k].y + lo[k].y * cs[k].x, lo[k].z * cs[k].z - lo[k].w * cs[k].w, lo[k].z * cs[k].w + lo[k].w * cs[k].z);
                    o1 = o0;
                }
                if (slot < a.nh) {
                    float * dst = (float *) (a.q_dst + (int64_t) slot * a.qd_nb1 + t * a.qd_nb2);
                    *(float4 *) (dst + 4 * j) = o0;
                    if (NEOX) *(float4 *) (dst + half + 4 * j) = o1;
                } else {
                    uint16_t * dst = (uint16_t *) (a.k_cache + row * a.kc_nb1) + (int64_t) (slot - a.nh) * hd;
                    *(uint2 *) (dst + 4 * j) = make_uint2((uint32_t) f2h(o0.x) | ((uint32_t) f2h(o0.y) << 16), (uint32_t) f2h(o0.z) | ((uint32_t) f2h(o0.w) << 16));
                    if (NEOX) *(uint2 *) (dst + half + 4 * j) = make_uint2((uint32_t) f2h(o1.x) | ((uint32_t) f2h(o1.y) << 16), (uint32_t) f2h(o1.z) | ((uint32_t) f2h(o1.w) << 16));
                }
            } else if (c < n_rot + n_v) {
                const int cv = c - n_rot, h = cv / (hd / 4), j = cv - h * (hd / 4);
                if (a.v_idx) {  // the transposed V cache of the non-flash path: one element index per value (llama.cpp's v_idxs)
                    const int64_t * ix = a.v_idx + ((int64_t) t * a.nkv + h) * hd + 4 * j;
                    const int64_t i0 = ix[0], i1 = ix[1], i2 = ix[2], i3 = ix[3];
                    uint16_t * vc = (uint16_t *) a.v_cache;
                    vc[i0] = f2h(lo[k].x);
                    vc[i1] = f2h(lo[k].y);
                    vc[i2] = f2h(lo[k].z);
                    vc[i3] = f2h(lo[k].w);
                } else {
                    uint16_t * dst = (uint16_t *) (a.v_cache + row * a.vc_nb1) + (int64_t) h * hd;
                    *(uint2 *) (dst + 4 * j) = make_uint2((uint32_t) f2h(lo[k].x) | ((uint32_t) f2h(lo[k].y) << 16), (uint32_t) f2h(lo[k].z) | ((uint32_t) f2h(lo[k].w) << 16));
                }
            }
        }
    }
}
bool rope_qk_store_vec_ok(const rope_store_args & a, int n_tokens) {
    static const bool on = !getenv("GGML_MI355X_ROPE_VEC") || atoi(getenv("GGML_MI355X_ROPE_VEC")) != 0;
    if (!on || n_tokens < 33 || a.ks > 1 || a.sk[0].part || a.sk[1].part || a.sk[2].part) return false;
    if ((a.p.mode & ~GGML_ROPE_TYPE_NEOX) != 0 || a.p.n_dims != a.head_dim || (a.head_dim % 8) != 0) return false;
    auto al16 = [](const void * p) { return (((uintptr_t) p) & 15) == 0; };
    if (!al16(a.q_src) || !al16(a.q_dst) || !al16(a.k_src) || !al16(a.v_src) || !al16(a.k_cache) || !al16(a.v_cache)) return false;
    for (int64_t v : {a.q_nb1, a.q_nb2, a.qd_nb1, a.qd_nb2, a.k_nb1, a.k_nb2, a.v_nb1, a.v_nb2})
        if (v % 16) return false;
    return (a.kc_nb1 % 8) == 0 && (a.v_idx != nullptr || (a.vc_nb1 % 8) == 0);
}
void launch_rope_qk_store_vec(hipStream_t s, const rope_store_args & a, int n_tokens, const float * tab) {
    if (a.p.mode & GGML_ROPE_TYPE_NEOX) hipLaunchKernelGGL(k_rope_qk_store_vec<true>, dim3((unsigned) n_tokens), dim3(256), 0, s, a, tab);
    else hipLaunchKernelGGL(k_rope_qk_store_vec<false>, dim3((unsigned) n_tokens), dim3(256), 0, s, a, tab);
}
void launch_rope_qk_store(hipStream_t s, rope_store_args a, int n_tokens) {
    rope_host_consts(a.p, a.theta_scale, a.corr0, a.corr1);
    const int n_all = a.nh + 2 * a.nkv;
    const int shares = n_tokens >= 256 ? 1 : std::max(1, std::min((n_all + 3) / 4, 1024 / std::max(1, n_tokens)));
    hipLaunchKernelGGL(k_rope_qk_store, dim3((unsigned) n_tokens, (unsigned) shares), dim3(256), 0, s, a);
}
void launch_rope(hipStream_t s, const tdesc & a, const tdesc & pos, const float * ff, const tdesc & d, const rope_params & p) {
    float theta_scale, c0, c1;
    rope_host_consts(p, theta_scale, c0, c1);
    dim3 grid((unsigned) a.ne[1], (unsigned) a.ne[2], (unsigned) a.ne[3]);
    if (p.mode & GGML_ROPE_TYPE_MROPE) {
        if (a.type == GGML_TYPE_F16) hipLaunchKernelGGL(k_rope_multi<true>, grid, dim3(64), 0, s, a, (const int32_t *) pos.data, ff, d, p, theta_scale, c0, c1);
        else hipLaunchKernelGGL(k_rope_multi<false>, grid, dim3(64), 0, s, a, (const int32_t *) pos.data, ff, d, p, theta_scale, c0, c1);
        return;
    }
    if (a.type == GGML_TYPE_F16) hipLaunchKernelGGL(k_rope<true>, grid, dim3(64), 0, s, a, pos, ff, d, p, theta_scale, c0, c1);
    else hipLaunchKernelGGL(k_rope<false>, grid, dim3(64), 0, s, a, pos, ff, d, p, theta_scale, c0, c1);
}

// ------------------------------------------------------------------------------------------------ SOFT_MAX
// ggml_compute_forward_soft_max_f32 with llama-box's zero-sum guard (llama-box/patches/llama.cpp/ggml-cpu.patch:5-15):
// w = x*scale + slope*mask; max; p = expf(w - max); sum in double; if (isnan(sum) || sum == 0) sum = -inf; p *= 1/sum
__global__ void __launch_bounds__(256) k_soft_max(const tdesc a, const tdesc m, const int has_mask, const float * __restrict__ sinks, const tdesc d,
                                                  const float scale, const float max_bias, const float m0, const float m1, const uint32_t n_head_log2) {
    __shared__ double shd[4];
    __shared__ float shf[4];
    const int64_t row = blockIdx.x;
    const int64_t i1 = row % a.ne[1], i2 = (row / a.ne[1]) % a.ne[2], i3 = row / (a.ne[1] * a.ne[2]);
    const float * x = (const float *) (a.data + i1 * a.nb[1] + i2 * a.nb[2] + i3 * a.nb[3]);
    float * y = (float *) (d.data + i1 * d.nb[1] + i2 * d.nb[2] + i3 * d.nb[3]);
    const char * mp = has_mask ? m.data + i1 * m.nb[1] + (i2 % m.ne[2]) * m.nb[2] + (i3 % m.ne[3]) * m.nb[3] : nullptr;
    const uint32_t h = (uint32_t) i2;
    const float slope = max_bias > 0.0f ? (h < n_head_log2 ? powf(m0, (float) (h + 1)) : powf(m1, (float) (2 * (h - n_head_log2) + 1))) : 1.0f;
    const int64_t n = a.ne[0];
    float mx = -INFINITY;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        float w = x[i] * scale;
        if (mp) {
            const float mv = m.type == GGML_TYPE_F16 ? h2f(((const uint16_t *) mp)[i]) : ((const float *) mp)[i];
            w += slope * mv;
        }
        y[i] = w;
        mx = fmaxf(mx, w);
    }
    mx = block_max_f(mx, shf);
    if (sinks) mx = fmaxf(mx, sinks[i2]);
    double sum = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const float v = expf(y[i] - mx);
        y[i] = v;
        sum += (double) v;
    }
    sum = block_sum_d(sum, shd);
    if (sinks) sum += (double) expf(sinks[i2] - mx);
    if (isnan(sum) || sum == 0.0) sum = -INFINITY;
    const float inv = (float) (1.0 / sum);
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) y[i] *= inv;
}
// Rows of up to 256 NR values stay in registers between the three passes: one read, one write (the kernel above goes through the
// output row three times, each pass waiting for the one before — 13 us per call on the 32 rows x 2k cells of a decode step).  Same
// arithmetic in the same order (thread-strided partial sums in double, then the block tree), so the results are bit-identical.
template <int NR>
__global__ void __launch_bounds__(256) k_soft_max_reg(const tdesc a, const tdesc m, const int has_mask, const float * __restrict__ sinks, const tdesc d,
                                                      const float scale, const float max_bias, const float m0, const float m1, const uint32_t n_head_log2) {
    __shared__ double shd[4];
    __shared__ float shf[4];
    const int64_t row = blockIdx.x;
    const int64_t i1 = row % a.ne[1], i2 = (row / a.ne[1]) % a.ne[2], i3 = row / (a.ne[1] * a.ne[2]);
    const float * x = (const float *) (a.data + i1 * a.nb[1] + i2 * a.nb[2] + i3 * a.nb[3]);
    float * y = (float *) (d.data + i1 * d.nb[1] + i2 * d.nb[2] + i3 * d.nb[3]);
    const char * mp = has_mask ? m.data + i1 * m.nb[1] + (i2 % m.ne[2]) * m.nb[2] + (i3 % m.ne[3]) * m.nb[3] : nullptr;
    const uint32_t h = (uint32_t) i2;
    const float slope = max_bias > 0.0f ? (h < n_head_log2 ? powf(m0, (float) (h + 1)) : powf(m1, (float) (2 * (h - n_head_log2) + 1))) : 1.0f;
    const int n = (int) a.ne[0];
    const bool m16 = m.type == GGML_TYPE_F16;
    float w[NR];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int i = threadIdx.x + 256 * j;
        w[j] = -INFINITY;
        if (i < n) {
            float v = x[i] * scale;
            if (mp) v += slope * (m16 ? h2f(((const uint16_t *) mp)[i]) : ((const float *) mp)[i]);
            w[j] = v;
            mx = fmaxf(mx, v);
        }
    }
    mx = block_max_f(mx, shf);
    if (sinks) mx = fmaxf(mx, sinks[i2]);
    double sum = 0.0;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        if ((int) threadIdx.x + 256 * j < n) {
            w[j] = expf(w[j] - mx);
            sum += (double) w[j];
        }
    }
    sum = block_sum_d(sum, shd);
    if (sinks) sum += (double) expf(sinks[i2] - mx);
    if (isnan(sum) || sum == 0.0) sum = -INFINITY;
    const float inv = (float) (1.0 / sum);
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int i = threadIdx.x + 256 * j;
        if (i < n) y[i] = w[j] * inv;
    }
}
void launch_soft_max(hipStream_t s, const tdesc & a, const tdesc * mask, const float * sinks, const tdesc & d, float scale, float max_bias) {
    const uint32_t n_head = (uint32_t) a.ne[2];
    const uint32_t n_head_log2 = 1u << (uint32_t) floor(log2((double) n_head));
    const float m0 = powf(2.0f, -(max_bias) / (float) n_head_log2);
    const float m1 = powf(2.0f, -(max_bias / 2.0f) / (float) n_head_log2);
    const int64_t rows = a.ne[1] * a.ne[2] * a.ne[3];
    tdesc dummy = a;
#define SM_REG(NR) hipLaunchKernelGGL(k_soft_max_reg<NR>, dim3((unsigned) rows), dim3(256), 0, s, a, mask ? *mask : dummy, mask ? 1 : 0, sinks, d, scale, max_bias, m0, m1, n_head_log2)
    if (a.nb[0] == 4 && d.nb[0] == 4 && a.ne[0] <= 256 * 32) {
        if (a.ne[0] <= 256 * 4) SM_REG(4);
        else if (a.ne[0] <= 256 * 12) SM_REG(12);
        else SM_REG(32);
        return;
    }
#undef SM_REG
    hipLaunchKernelGGL(k_soft_max, dim3((unsigned) rows), dim3(256), 0, s, a, mask ? *mask : dummy, mask ? 1 : 0, sinks, d, scale, max_bias, m0, m1, n_head_log2);
}

// ---- SET_ROWS into a Q8_0 tensor (quantised KV cache): quantize_row_q8_0_ref per 32 values — d = amax / 127 stored as f16,
// q = roundf(x / d) with the unrounded d.  One wave per 256 source values (8 blocks, 8 lanes each); blocks are 34 bytes, so the
// quants go out as 16-bit stores.
__global__ void __launch_bounds__(64) k_set_rows_q8_0(const tdesc a, const tdesc idx, const tdesc d) {
    const int lane = threadIdx.x;
    const int64_t chunks = (a.ne[0] + 255) / 256;  // rows are whole blocks of 32; the last chunk of a row may be short
    const int64_t gid = blockIdx.x;
    const int64_t row = gid / chunks, ch = gid - row * chunks;
    const int64_t i01 = row % a.ne[1], i02 = (row / a.ne[1]) % a.ne[2], i03 = row / (a.ne[1] * a.ne[2]);
    const int64_t i12 = i03 % idx.ne[2], i11 = i02 % idx.ne[1];
    const int64_t r = *(const int64_t *) (idx.data + i01 * idx.nb[0] + i11 * idx.nb[1] + i12 * idx.nb[2]);
    const bool live = ch * 256 + lane * 4 < a.ne[0];
    const float4 x = live ? ((const float4 *) (a.data + i01 * a.nb[1] + i02 * a.nb[2] + i03 * a.nb[3]))[ch * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float v[4] = {x.x, x.y, x.z, x.w};
    float amax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
    amax = fmaxf(amax, dpp_f32<MI_DPP_QUAD_XOR1>(amax));
    amax = fmaxf(amax, dpp_f32<MI_DPP_QUAD_XOR2>(amax));
    amax = fmaxf(amax, dpp_f32<MI_DPP_HALF_MIRROR>(amax));
    const float dd = amax / 127.0f;
    const float id = dd != 0.0f ? 1.0f / dd : 0.0f;
    uint32_t packed = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) packed |= (uint32_t) ((int) roundf(v[k] * id) & 0xFF) << (8 * k);
    char * blk = d.data + r * d.nb[1] + i02 * d.nb[2] + i03 * d.nb[3] + (ch * 8 + (lane >> 3)) * 34;
    if (!live) return;
    uint16_t * qs = (uint16_t *) (blk + 2 + 4 * (lane & 7));
    qs[0] = (uint16_t) (packed & 0xFFFF);
    qs[1] = (uint16_t) (packed >> 16);
    if ((lane & 7) == 0) *(uint16_t *) blk = f2h(dd);
}
void launch_set_rows_q8_0(hipStream_t s, const tdesc & a, const tdesc & idx, const tdesc & d) {
    const int64_t blocks = ((a.ne[0] + 255) / 256) * a.ne[1] * a.ne[2] * a.ne[3];
    hipLaunchKernelGGL(k_set_rows_q8_0, dim3((unsigned) blocks), dim3(64), 0, s, a, idx, d);
}

// ---- CPY between a contiguous Q8_0 tensor and a contiguous F32 tensor: the K-shift of a quantised cache
// (llama.cpp build_rope_shift: cast K to f32, rope, cpy back).  Same 8-lanes-per-block layout as k_set_rows_q8_0.
__global__ void __launch_bounds__(256) k_cpy_q8_0_f32(const char * __restrict__ src, float * __restrict__ dst, const int64_t n4) {
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;  // 4 values each
    if (i >= n4) return;
    const char * blk = src + (i >> 3) * 34;
    const float d = h2f(ld16(blk));
    const uint32_t qs = ld32_a2(blk + 2 + 4 * (i & 7));
    ((float4 *) dst)[i] = make_float4(d * (float) (int8_t) qs, d * (float) (int8_t) (qs >> 8), d * (float) (int8_t) (qs >> 16), d * (float) (int8_t) (qs >> 24));
}
__global__ void __launch_bounds__(256) k_cpy_f32_q8_0(const float * __restrict__ src, char * __restrict__ dst, const int64_t n4) {
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    const bool live = i < n4;  // n4 is a multiple of 8 (whole blocks), so the 8 lanes of a block are live together
    const float4 x = live ? ((const float4 *) src)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float v[4] = {x.x, x.y, x.z, x.w};
    float amax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
    amax = fmaxf(amax, dpp_f32<MI_DPP_QUAD_XOR1>(amax));
    amax = fmaxf(amax, dpp_f32<MI_DPP_QUAD_XOR2>(amax));
    amax = fmaxf(amax, dpp_f32<MI_DPP_HALF_MIRROR>(amax));
    const float dd = amax / 127.0f;
    const float id = dd != 0.0f ? 1.0f / dd : 0.0f;
    uint32_t packed = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) packed |= (uint32_t) ((int) roundf(v[k] * id) & 0xFF) << (8 * k);
    if (!live) return;
    char * blk = dst + (i >> 3) * 34;
    uint16_t * qs = (uint16_t *) (blk + 2 + 4 * (i & 7));
    qs[0] = (uint16_t) (packed & 0xFFFF);
    qs[1] = (uint16_t) (packed >> 16);
    if ((i & 7) == 0) *(uint16_t *) blk = f2h(dd);
}
void launch_cpy_q8_0(hipStream_t s, const void * src, void * dst, int64_t n, bool to_q8) {
    const int64_t n4 = n / 4;
    const unsigned grid = (unsigned) ((n4 + 255) / 256);
    if (to_q8) hipLaunchKernelGGL(k_cpy_f32_q8_0, dim3(grid), dim3(256), 0, s, (const float *) src, (char *) dst, n4);
    else hipLaunchKernelGGL(k_cpy_q8_0_f32, dim3(grid), dim3(256), 0, s, (const char *) src, (float *) dst, n4);
}

// ---- small upload: copies `n` bytes from pinned host memory (device-visible) into device memory inside the stream
__global__ void __launch_bounds__(256) k_upload_small(char * __restrict__ dst, const char * __restrict__ src, const size_t n, const int vec) {
    const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (vec == 16) {
        if (i * 16 < n) ((uint4 *) dst)[i] = ((const uint4 *) src)[i];
    } else if (vec == 4) {
        if (i * 4 < n) ((uint32_t *) dst)[i] = ((const uint32_t *) src)[i];
    } else {
        if (i < n) dst[i] = src[i];
    }
}
void launch_upload_small(hipStream_t s, void * dst, const void * pinned_src, size_t n) {
    const uintptr_t al = (uintptr_t) dst | (uintptr_t) pinned_src | (uintptr_t) n;
    const int vec = (al & 15) == 0 ? 16 : ((al & 3) == 0 ? 4 : 1);
    const size_t items = (n + vec - 1) / vec;
    hipLaunchKernelGGL(k_upload_small, dim3((unsigned) ((items + 255) / 256)), dim3(256), 0, s, (char *) dst, (const char *) pinned_src, n, vec);
}

// several small uploads in ONE launch (a decode step brings token ids, positions, cache indices, a mask row and output ids:
// five dependent ~4.5 us kernels otherwise); blockIdx.y = segment
__global__ void __launch_bounds__(256) k_upload_multi(const upload_batch b) {
    const upload_seg sg = b.seg[blockIdx.y];
    const uintptr_t al = (uintptr_t) sg.dst | (uintptr_t) sg.src | (uintptr_t) sg.n;
    for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;; i += (size_t) gridDim.x * 256) {
        if ((al & 15) == 0) {
            if (i * 16 >= sg.n) break;
            ((uint4 *) sg.dst)[i] = ((const uint4 *) sg.src)[i];
        } else if ((al & 3) == 0) {
            if (i * 4 >= sg.n) break;
            ((uint32_t *) sg.dst)[i] = ((const uint32_t *) sg.src)[i];
        } else {
            if (i >= sg.n) break;
            sg.dst[i] = sg.src[i];
        }
    }
}
void launch_upload_multi(hipStream_t s, const upload_batch & b) {
    size_t most = 0;
    for (int i = 0; i < b.n; ++i) most = std::max(most, b.seg[i].n);
    const unsigned gx = (unsigned) std::min<size_t>(16, (most / 16 + 255) / 256 + 1);
    hipLaunchKernelGGL(k_upload_multi, dim3(gx, (unsigned) b.n), dim3(256), 0, s, b);
}

// rows of `width` bytes from one pitch to another; either side may be a peer device's memory (tp_inproc.cpp: the vocab shards of the logits into the
// main device's tensor, the KV shards into / out of the host's cache tensors)
__global__ void __launch_bounds__(256) k_copy2d(char * __restrict__ dst, const size_t dpitch, const char * __restrict__ src, const size_t spitch, const size_t width, const int vec) {
    const size_t row = blockIdx.y;
    char * d = dst + row * dpitch;
    const char * s = src + row * spitch;
    for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;; i += (size_t) gridDim.x * 256) {
        if (vec == 16) {
            if (i * 16 >= width) break;
            ((uint4 *) d)[i] = ((const uint4 *) s)[i];
        } else if (vec == 2) {
            if (i * 2 >= width) break;
            ((uint16_t *) d)[i] = ((const uint16_t *) s)[i];
        } else {
            if (i >= width) break;
            d[i] = s[i];
        }
    }
}
void launch_copy2d(hipStream_t s, void * dst, size_t dpitch, const void * src, size_t spitch, size_t width, size_t height) {
    if (width == 0 || height == 0) return;
    const uintptr_t al = (uintptr_t) dst | (uintptr_t) src | (uintptr_t) dpitch | (uintptr_t) spitch | (uintptr_t) width;
    const int vec = (al & 15) == 0 ? 16 : ((al & 1) == 0 ? 2 : 1);
    const size_t items = (width + vec - 1) / vec;
    const unsigned gx = (unsigned) std::min<size_t>(64, (items + 255) / 256);
    for (size_t r0 = 0; r0 < height; r0 += 65535) {
        const size_t h = std::min<size_t>(65535, height - r0);
        hipLaunchKernelGGL(k_copy2d, dim3(gx, (unsigned) h), dim3(256), 0, s, (char *) dst + r0 * dpitch, dpitch, (const char *) src + r0 * spitch, spitch, width, vec);
    }
}

// in-process tensor parallel, non-flash graphs: the host's per-element row indices of the transposed V cache (v_idxs[i * full + j] = j * n_ctx + slot_i)
// cut down to one device's rows [o0, o0 + ext) of every token and rebased onto its shard of the cache: dst[i * ext + j'] = src[i * full + o0 + j'] - o0 * n_ctx
__global__ void __launch_bounds__(256) k_rebase_row_index(int64_t * __restrict__ dst, const int64_t * __restrict__ src, const int64_t n_tok, const int64_t full, const int64_t ext,
                                                         const int64_t o0, const int64_t n_ctx) {
    const int64_t e = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (e >= n_tok * ext) return;
    const int64_t i = e / ext, j = e - i * ext;
    dst[e] = src[i * full + o0 + j] - o0 * n_ctx;
}
void launch_rebase_row_index(hipStream_t s, int64_t * dst, const int64_t * src, int64_t n_tok, int64_t full, int64_t ext, int64_t o0, int64_t n_ctx) {
    const int64_t n = n_tok * ext;
    if (n <= 0) return;
    hipLaunchKernelGGL(k_rebase_row_index, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, s, dst, src, n_tok, full, ext, o0, n_ctx);
}

MI_TU_TOUCH(ops)

}  // namespace mi355x
