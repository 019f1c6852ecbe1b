// upload_probe.hip — how should set_tensor move a GGUF file's tensors (pageable / mmap'd host memory) to HBM?
//   A: hipMemcpy straight from pageable memory (what the runtime does internally: its own staging)
//   B: two pinned staging buffers, host memcpy (1..T threads) overlapped with hipMemcpyAsync
//   C: hipHostRegister the source range, one hipMemcpyAsync, unregister
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void par_memcpy(char * d, const char * s, size_t n, int T) {
    if (T <= 1) { memcpy(d, s, n); return; }
    std::vector<std::thread> th;
    const size_t per = (n / T + 4095) & ~(size_t) 4095;
    for (int t = 0; t < T; ++t) {
        const size_t o = (size_t) t * per;
        if (o >= n) break;
        th.emplace_back([=] { memcpy(d + o, s + o, std::min(per, n - o)); });
    }
    for (auto & x : th) x.join();
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const size_t N = (size_t) 2 << 30;
    char * src = (char *) aligned_alloc(4096, N);
    for (size_t i = 0; i < N; i += 4096) src[i] = (char) i;  // touch every page
    char * dev; CK(hipMalloc(&dev, N));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        CK(hipMemcpy(dev, src, N, hipMemcpyHostToDevice));
        printf("A  hipMemcpy pageable                : %.2f GB/s\n", N / (now() - t0) / 1e9);
    }
    for (size_t chunk : {(size_t) 8 << 20, (size_t) 32 << 20, (size_t) 128 << 20}) {
        for (int T : {1, 4, 8}) {
            char * pin[2]; hipEvent_t ev[2];
            for (int i = 0; i < 2; ++i) { CK(hipHostMalloc((void **) &pin[i], chunk, hipHostMallocDefault)); CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)); }
            double t0 = now();
            int k = 0;
            for (size_t o = 0; o < N; o += chunk, k ^= 1) {
                const size_t n = std::min(chunk, N - o);
                CK(hipEventSynchronize(ev[k]));
                par_memcpy(pin[k], src + o, n, T);
                CK(hipMemcpyAsync(dev + o, pin[k], n, hipMemcpyHostToDevice, s));
                CK(hipEventRecord(ev[k], s));
            }
            CK(hipStreamSynchronize(s));
            printf("B  pinned staging %4zu MiB x2, %d thr   : %.2f GB/s\n", chunk >> 20, T, N / (now() - t0) / 1e9);
            for (int i = 0; i < 2; ++i) { CK(hipHostFree(pin[i])); CK(hipEventDestroy(ev[i])); }
        }
    }
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        CK(hipHostRegister(src, N, hipHostRegisterDefault));
        double t1 = now();
        CK(hipMemcpyAsync(dev, src, N, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        double t2 = now();
        CK(hipHostUnregister(src));
        double t3 = now();
        printf("C  register %.3f s + copy %.2f GB/s + unregister %.3f s -> %.2f GB/s overall\n", t1 - t0, N / (t2 - t1) / 1e9, t3 - t2, N / (t3 - t0) / 1e9);
    }
    return 0;
}
