// upload_probe.cpp — what a host-buffer entry point pays for on this box (round 6):
// hipMalloc / hipFree of column-sized buffers, hipMemcpy from pageable memory (first and second call), hipHostMalloc of a
// staging ring, and a chunked pipeline (worker threads memcpy pageable -> pinned slot, hipMemcpyAsync slot -> device).
//   hipcc -O2 --offload-arch=gfx950 tools/probes/upload_probe.cpp -o /tmp/upload_probe -lpthread
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static void pipeline(const char *src, char *dst, size_t bytes, size_t chunk, int slots, int threads, char **pin, hipStream_t st, hipEvent_t *ev) {
    // slot s is free again when ev[s] has completed; `threads` workers fill each slot cooperatively
    size_t n_chunks = (bytes + chunk - 1) / chunk;
    for (size_t c = 0; c < n_chunks; ++c) {
        int s = (int)(c % slots);
        if (c >= (size_t)slots) CK(hipEventSynchronize(ev[s]));
        size_t off = c * chunk, len = std::min(chunk, bytes - off);
        if (threads <= 1) memcpy(pin[s], src + off, len);
        else {
            std::vector<std::thread> th;
            size_t per = (len / threads + 4095) & ~size_t(4095);
            for (int t = 0; t < threads; ++t) {
                size_t o = std::min(len, per * t), l = std::min(per, len - o);
                if (l) th.emplace_back([=] { memcpy(pin[s] + o, src + off + o, l); });
            }
            for (auto &t : th) t.join();
        }
        CK(hipMemcpyAsync(dst + off, pin[s], len, hipMemcpyHostToDevice, st));
        CK(hipEventRecord(ev[s], st));
    }
    CK(hipStreamSynchronize(st));
}

int main() {
    double t0 = now();
    CK(hipSetDevice(0)); CK(hipFree(nullptr));
    printf("hip init %.1f ms\n", 1e3 * (now() - t0));
    const size_t big = 400u << 20, small = 100u << 20;
    char *h = (char *)malloc(big); memset(h, 1, big);
    for (int rep = 0; rep < 3; ++rep) {
        void *d1, *d2; double a = now();
        CK(hipMalloc(&d1, big)); double b = now(); CK(hipMalloc(&d2, small)); double c = now();
        CK(hipMemcpy(d1, h, big, hipMemcpyHostToDevice)); double d = now();
        CK(hipMemcpy(d2, h, small, hipMemcpyHostToDevice)); double e = now();
        CK(hipFree(d1)); CK(hipFree(d2)); double f = now();
        printf("rep %d: hipMalloc 400M %.2f ms, 100M %.2f ms; hipMemcpy pageable 400M %.2f ms (%.1f GB/s), 100M %.2f ms (%.1f GB/s); 2 x hipFree %.2f ms\n", rep,
               1e3 * (b - a), 1e3 * (c - b), 1e3 * (d - c), big / (d - c) / 1e9, 1e3 * (e - d), small / (e - d) / 1e9, 1e3 * (f - e));
    }
    void *dev; CK(hipMalloc(&dev, big));
    { double a = now(); CK(hipHostRegister(h, big, hipHostRegisterDefault)); double b = now();
      CK(hipMemcpy(dev, h, big, hipMemcpyHostToDevice)); double c = now(); CK(hipHostUnregister(h)); double d = now();
      printf("hipHostRegister 400M %.2f ms, copy %.2f ms (%.1f GB/s), unregister %.2f ms\n", 1e3 * (b - a), 1e3 * (c - b), big / (c - b) / 1e9, 1e3 * (d - c)); }
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (size_t chunk : {size_t(4) << 20, size_t(16) << 20, size_t(32) << 20}) {
        const int slots = 4;
        char *pin[slots]; hipEvent_t ev[slots];
        double a = now();
        for (int s = 0; s < slots; ++s) { CK(hipHostMalloc((void **)&pin[s], chunk, hipHostMallocDefault)); CK(hipEventCreateWithFlags(&ev[s], hipEventDisableTiming)); }
        double b = now();
        printf("hipHostMalloc %d x %zu MiB: %.2f ms\n", slots, chunk >> 20, 1e3 * (b - a));
        for (int threads : {1, 2, 4, 8}) {
            double best = 1e9;
            for (int rep = 0; rep < 3; ++rep) { double c = now(); pipeline(h, (char *)dev, big, chunk, slots, threads, pin, st, ev); best = std::min(best, now() - c); }
            printf("  pipeline chunk %zu MiB, %d memcpy threads: 400M in %.2f ms (%.1f GB/s)\n", chunk >> 20, threads, 1e3 * best, big / best / 1e9);
        }
        { double c = now(); CK(hipMemcpyAsync(dev, pin[0], chunk, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); double d = now();
          printf("  pinned -> device %zu MiB: %.3f ms (%.1f GB/s)\n", chunk >> 20, 1e3 * (d - c), chunk / (d - c) / 1e9); }
        double c = now();
        for (int s = 0; s < slots; ++s) { CK(hipHostFree(pin[s])); CK(hipEventDestroy(ev[s])); }
        printf("  hipHostFree: %.2f ms\n", 1e3 * (now() - c));
    }
    // one big pinned allocation, for scale
    { double a = now(); void *p; CK(hipHostMalloc(&p, big, hipHostMallocDefault)); double b = now(); memcpy(p, h, big); double c = now();
      CK(hipMemcpy(dev, p, big, hipMemcpyHostToDevice)); double d = now(); CK(hipHostFree(p));
      printf("hipHostMalloc 400M %.2f ms, memcpy into it %.2f ms (%.1f GB/s), copy to device %.2f ms (%.1f GB/s)\n", 1e3 * (b - a), 1e3 * (c - b), big / (c - b) / 1e9, 1e3 * (d - c), big / (d - c) / 1e9); }
    return 0;
}
