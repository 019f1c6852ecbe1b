// shadow_probe.hip — stand-alone check + timing of the shadow-plane GEMM (scripts/ubench/mmq_shadow_dev.h; lab code, not part of libggml-mi355x.so).
//   build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I scripts/ubench -I llama_box_amd/csrc -I include -DGGML_MAX_NAME=128 [-DSH_KO=n] scripts/ubench/shadow_probe.hip -o scripts/ubench/shadow_probe.bin
//   run:    scripts/ubench/shadow_probe.bin            (prints one line per shape: max relative error of sampled outputs vs a CPU restatement, us, TFLOP/s)
// The CPU side restates the contract the kernel serves (ggml-cpu's vec_dot_q{4,6}_K_q8_K: integer block sums, one f32 scale-accumulate per
// super-block) in double precision on a sample of the outputs.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#ifndef SH_FORM
#define SH_FORM 2
#endif
#include "mmq_shadow_dev.h"

using namespace mi355x;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static uint16_t f2h_host(float f) { _Float16 h = (_Float16) f; uint16_t u; memcpy(&u, &h, 2); return u; }
static float h2f_host(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float) h; }

struct cpu_blk { int p[256]; float d, dmin; int m[8]; };  // s(k) q(k), scales
static cpu_blk decode(int qt, const uint8_t * b) {
    cpu_blk o{};
    if (qt == 4 || qt == 5) {
        o.d = h2f_host(*(const uint16_t *) b);
        o.dmin = h2f_host(*(const uint16_t *) (b + 2));
        const uint8_t * q = b + 4;
        for (int j = 0; j < 8; ++j) {
            int sc, m;
            if (j < 4) { sc = q[j] & 63; m = q[j + 4] & 63; }
            else { sc = (q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4); m = (q[j + 4] >> 4) | ((q[j] >> 6) << 4); }
            o.m[j] = m;
            const uint8_t * qs = b + (qt == 5 ? 48 : 16) + 32 * (j >> 1);
            const uint8_t * qh = b + 16;
            for (int l = 0; l < 32; ++l) {
                int v = (j & 1) ? (qs[l] >> 4) : (qs[l] & 0xF);
                if (qt == 5) v |= ((qh[l] >> j) & 1) << 4;
                o.p[32 * j + l] = sc * v;
            }
        }
    } else {
        uint16_t dh;
        memcpy(&dh, b + 208, 2);
        o.d = h2f_host(dh);
        const uint8_t * ql = b;
        const uint8_t * qh = b + 128;
        const int8_t * sc = (const int8_t *) (b + 192);
        for (int n = 0; n < 256; n += 128) {
            for (int l = 0; l < 32; ++l) {
                const int is = l / 16;
                const int q1 = (int) ((ql[l] & 0xF) | (((qh[l] >> 0) & 3) << 4)) - 32;
                const int q2 = (int) ((ql[l + 32] & 0xF) | (((qh[l] >> 2) & 3) << 4)) - 32;
                const int q3 = (int) ((ql[l] >> 4) | (((qh[l] >> 4) & 3) << 4)) - 32;
                const int q4 = (int) ((ql[l + 32] >> 4) | (((qh[l] >> 6) & 3) << 4)) - 32;
                o.p[n + l] = sc[is + 0] * q1;
                o.p[n + l + 32] = sc[is + 2] * q2;
                o.p[n + l + 64] = sc[is + 4] * q3;
                o.p[n + l + 96] = sc[is + 6] * q4;
            }
            ql += 64; qh += 32; sc += 8;
        }
    }
    return o;
}

int main(int argc, char ** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    struct shape { int qt, K, N, M, ks; };
    std::vector<shape> shapes = {{4, 1024, 256, 200, 1}, {6, 1024, 128, 300, 2}, {5, 512, 384, 512, 1},
                                 {4, 4096, 28672, 512, 1}, {4, 4096, 6144, 512, 1}, {4, 4096, 6144, 512, 2}, {4, 4096, 4096, 512, 1}, {4, 4096, 4096, 512, 2}, {4, 4096, 4096, 512, 4},
                                 {4, 14336, 4096, 512, 1}, {4, 14336, 4096, 512, 2}, {4, 14336, 4096, 512, 4}, {6, 14336, 4096, 512, 4}, {4, 4096, 28672, 2048, 1}, {4, 8192, 28672, 512, 1}};
    if (argc > 1 && atoi(argv[1]) == 1) shapes.resize(3);
    std::mt19937_64 rng(1234);
    CK(hipFuncSetAttribute((const void *) k_mmq_shadow<true>, hipFuncAttributeMaxDynamicSharedMemorySize, SH_LDS_BYTES));
    CK(hipFuncSetAttribute((const void *) k_mmq_shadow<false>, hipFuncAttributeMaxDynamicSharedMemorySize, SH_LDS_BYTES));
    CK(hipFuncSetAttribute((const void *) k_mmq_shadow3<true>, hipFuncAttributeMaxDynamicSharedMemorySize, SH3_LDS_BYTES));
    CK(hipFuncSetAttribute((const void *) k_mmq_shadow3<false>, hipFuncAttributeMaxDynamicSharedMemorySize, SH3_LDS_BYTES));
    CK(hipFuncSetAttribute((const void *) k_mmq_shadow_pp<true>, hipFuncAttributeMaxDynamicSharedMemorySize, SH2_LDS_BYTES));
    CK(hipFuncSetAttribute((const void *) k_mmq_shadow_pp<false>, hipFuncAttributeMaxDynamicSharedMemorySize, SH2_LDS_BYTES));
    for (const shape & sh : shapes) {
        const int bytes = sh.qt == 4 ? 144 : (sh.qt == 5 ? 176 : 210);
        const int nblk = sh.K / 256;
        const size_t w_nb1 = (size_t) nblk * bytes;
        std::vector<uint8_t> W((size_t) sh.N * w_nb1 + 256);
        for (size_t i = 0; i < W.size(); i += 8) { uint64_t r = rng(); memcpy(&W[i], &r, std::min<size_t>(8, W.size() - i)); }
        for (int n = 0; n < sh.N; ++n)
            for (int b = 0; b < nblk; ++b) {
                uint8_t * blk = &W[(size_t) n * w_nb1 + (size_t) b * bytes];
                const float d = 0.01f * (0.5f + (float) (rng() & 0xFFFF) / 65536.0f);
                const uint16_t dh = f2h_host(d), mh = f2h_host(d * 3.0f);
                if (sh.qt == 6) memcpy(blk + 208, &dh, 2);
                else { memcpy(blk, &dh, 2); memcpy(blk + 2, &mh, 2); }
            }
        std::vector<q8k_dev> A((size_t) sh.M * nblk);
        for (auto & q : A) {
            int bs[16] = {0};
            for (int i = 0; i < 256; ++i) {
                int v = (int) (rng() % 255) - 127;
                q.qs[i] = (int8_t) v;
                bs[i / 16] += v;
            }
            for (int g = 0; g < 16; ++g) q.bsums[g] = f2h_host((float) bs[g]);
            for (int g = 0; g < 8; ++g) q.bs32[g] = (int16_t) (bs[2 * g] + bs[2 * g + 1]);
            q.d = 0.02f * (0.5f + (float) (rng() & 0xFFFF) / 65536.0f);
            q.pad[0] = q.pad[1] = q.pad[2] = 0;
        }
        uint8_t * dW; q8k_dev * dA; char * planes, * meta; float * dst, * part = nullptr;
        const size_t pl_bytes = (size_t) (sh.N / 32) * nblk * SH_PANEL_SB, me_bytes = (size_t) (sh.N / 32) * nblk * SH_META;
        CK(hipMalloc(&dW, W.size())); CK(hipMalloc(&dA, A.size() * sizeof(q8k_dev))); CK(hipMalloc(&planes, pl_bytes)); CK(hipMalloc(&meta, me_bytes));
        CK(hipMalloc(&dst, (size_t) sh.M * sh.N * 4));
        if (sh.ks > 1) CK(hipMalloc(&part, (size_t) sh.ks * sh.M * sh.N * 4));
        CK(hipMemcpy(dW, W.data(), W.size(), hipMemcpyHostToDevice));
        CK(hipMemcpy(dA, A.data(), A.size() * sizeof(q8k_dev), hipMemcpyHostToDevice));
        CK(hipMemset(dst, 0xFF, (size_t) sh.M * sh.N * 4));
        const unsigned bgrid = (unsigned) ((sh.N / 32) * nblk);
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        if (sh.qt == 4) hipLaunchKernelGGL(k_shadow_build<4>, dim3(bgrid), dim3(256), 0, 0, dW, (int64_t) w_nb1, nblk, planes, meta);
        else if (sh.qt == 5) hipLaunchKernelGGL(k_shadow_build<5>, dim3(bgrid), dim3(256), 0, 0, dW, (int64_t) w_nb1, nblk, planes, meta);
        else hipLaunchKernelGGL(k_shadow_build<6>, dim3(bgrid), dim3(256), 0, 0, dW, (int64_t) w_nb1, nblk, planes, meta);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms_build = 0;
        CK(hipEventElapsedTime(&ms_build, e0, e1));
        shadow_args a{};
        a.n_mat = 1;
        a.K = sh.K; a.M = sh.M; a.act = dA;
        a.mat[0] = {planes, meta, sh.N, 0, dst, (int64_t) sh.N, nullptr, 0, part};
        a.n_groups = sh.N / 128;
        a.m_tiles = (sh.M + (SH_FORM == 3 ? SH3_BM : SH_BM) - 1) / (SH_FORM == 3 ? SH3_BM : SH_BM);
        a.ksplit = sh.ks;
        unsigned long long * d_st = nullptr;
#if SH_STAMP
        CK(hipMalloc(&d_st, 64 * 8));
        CK(hipMemset(d_st, 0, 64 * 8));
        a.stamps = d_st;
#endif
        const dim3 grid((unsigned) (((a.n_groups + 7) / 8) * 8 * a.m_tiles), (unsigned) sh.ks);
        auto launch = [&]() {
#if SH_FORM == 3
            if (sh.qt == 6) hipLaunchKernelGGL(k_mmq_shadow3<false>, grid, dim3(SH3_NW * 64), SH3_LDS_BYTES, 0, a);
            else hipLaunchKernelGGL(k_mmq_shadow3<true>, grid, dim3(SH3_NW * 64), SH3_LDS_BYTES, 0, a);
#elif SH_FORM == 2
            if (sh.qt == 6) hipLaunchKernelGGL(k_mmq_shadow_pp<false>, grid, dim3(SH_NW * 64), SH2_LDS_BYTES, 0, a);
            else hipLaunchKernelGGL(k_mmq_shadow_pp<true>, grid, dim3(SH_NW * 64), SH2_LDS_BYTES, 0, a);
#else
            if (sh.qt == 6) hipLaunchKernelGGL(k_mmq_shadow<false>, grid, dim3(SH_NW * 64), SH_LDS_BYTES, 0, a);
            else hipLaunchKernelGGL(k_mmq_shadow<true>, grid, dim3(SH_NW * 64), SH_LDS_BYTES, 0, a);
#endif
        };
        launch();
        CK(hipDeviceSynchronize());
        std::vector<float> out((size_t) sh.M * sh.N);
        if (sh.ks == 1) CK(hipMemcpy(out.data(), dst, out.size() * 4, hipMemcpyDeviceToHost));
        else {
            std::vector<float> pp((size_t) sh.ks * sh.M * sh.N);
            CK(hipMemcpy(pp.data(), part, pp.size() * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < out.size(); ++i) { float s = pp[i]; for (int k = 1; k < sh.ks; ++k) s += pp[(size_t) k * out.size() + i]; out[i] = s; }
        }
        // CPU check on a sample: every token of a few rows, a few tokens of every 97th row
        double max_rel = 0, ref_max = 0;
        long checked = 0;
        auto check = [&](int n, int m) {
            double acc = 0;
            for (int b = 0; b < nblk; ++b) {
                const cpu_blk wb = decode(sh.qt, &W[(size_t) n * w_nb1 + (size_t) b * bytes]);
                const q8k_dev & q = A[(size_t) m * nblk + b];
                long is = 0;
                for (int k = 0; k < 256; ++k) is += (long) wb.p[k] * q.qs[k];
                long ms = 0;
                if (sh.qt != 6) for (int j = 0; j < 8; ++j) ms += (long) wb.m[j] * q.bs32[j];
                acc += (double) q.d * ((double) wb.d * (double) is - (double) wb.dmin * (double) ms);
            }
            const double got = out[(size_t) m * sh.N + n];
            max_rel = std::max(max_rel, std::fabs(got - acc));
            ref_max = std::max(ref_max, std::fabs(acc));
            ++checked;
        };
        const int rows_full[] = {0, 31, 32, 63, 64, 127, sh.N - 1, sh.N / 2 + 5};
        for (int n : rows_full) if (n >= 0 && n < sh.N) for (int m = 0; m < sh.M; m += (sh.K > 4096 || sh.N > 8192 ? 7 : 1)) check(n, m);
        for (int n = 0; n < sh.N; n += 97) for (int m : {0, 1, 37, 63, 64, 200, 255, 256, sh.M - 1}) if (m < sh.M) check(n, m);
        // timing
        for (int i = 0; i < 3; ++i) launch();
        CK(hipDeviceSynchronize());
        const int reps = 20;
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        const double tflops = 2.0 * sh.K * (double) sh.N * sh.M / (us * 1e-6) / 1e12;
        printf("q%d_K K=%5d N=%5d M=%4d ks=%d grid=%4u x %d: max|err| %.3e (ref max %.3e, %ld outputs checked) %s | %8.2f us  %7.1f TFLOP/s (%.3f of 2.5 PF) | build %.1f us, shadow %.1f MB\n",
               sh.qt, sh.K, sh.N, sh.M, sh.ks, grid.x, sh.ks, max_rel, ref_max, checked, max_rel <= 2e-6 * ref_max + 1e-30 ? "OK" : "MISMATCH", us, tflops, tflops / 2500.0,
               ms_build * 1e3, (pl_bytes + me_bytes) / 1e6);
#if SH_STAMP
        {
            unsigned long long h[64];
            CK(hipMemcpy(h, d_st, sizeof(h), hipMemcpyDeviceToHost));
            const double steps = (double) nblk * 4 / sh.ks;
            printf("    stamps (cycles per step, workgroup 0; phases: 0 loop/compute tail, 1 vmcnt wait, 2 barrier, 3 request issue, 4 reads+mfma (all 4 steps -> per step), 5 fold per super-block / 4):\n");
            for (int w = 0; w < 8; ++w) {
                printf("      wave %d:", w);
                for (int i = 0; i < 6; ++i) printf(" %8.0f", (double) h[w * 8 + i] / steps);
                printf("\n");
            }
            CK(hipFree(d_st));
        }
#endif
        CK(hipFree(dW)); CK(hipFree(dA)); CK(hipFree(planes)); CK(hipFree(meta)); CK(hipFree(dst));
        if (part) CK(hipFree(part));
    }
    return 0;
}
