// first_use_probe.cpp — what every FIRST use of a HIP facility costs in a fresh process on this box (round 6):
// the command-line hosts pay each of them exactly once, so their order and placement decide the tool's wall time.
//   hipcc -O2 --offload-arch=gfx950 tools/probes/first_use_probe.cpp -o tools/probes/first_use_probe ; ./first_use_probe [order]
// order: a string of letters, executed in sequence, each timed:
//   i hipSetDevice+hipFree(0)   m hipMalloc 64 MiB        s hipStreamCreate (non-blocking)   h hipHostMalloc 8 MiB   H hipHostMalloc 64 MiB
//   c pinned->device 8 MiB on the created stream (needs s,h,m)   n pinned->device 8 MiB on the NULL stream (needs h,m)
//   p pageable->device 64 KiB (hipMemcpy)   P pageable->device 64 MiB   d device->pageable 64 KiB   k first kernel launch   e hipEventCreate+record+sync
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void tiny(int *p) { if (p) p[threadIdx.x] = threadIdx.x; }
int main(int argc, char **argv) {
    const char *order = argc > 1 ? argv[1] : "imshcnpPdke";
    void *dev = nullptr, *pin = nullptr, *pin2 = nullptr;
    hipStream_t st = nullptr;
    char *page = (char *)malloc(64u << 20);
    memset(page, 1, 64u << 20);
    double total0 = now();
    for (const char *o = order; *o; ++o) {
        double t = now();
        const char *what = "?";
        switch (*o) {
            case 'i': what = "hipSetDevice + hipFree(0)"; CK(hipSetDevice(0)); CK(hipFree(nullptr)); break;
            case 'm': what = "hipMalloc 64 MiB"; CK(hipMalloc(&dev, 64u << 20)); break;
            case 's': what = "hipStreamCreateWithFlags"; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); break;
            case 'h': what = "hipHostMalloc 8 MiB"; CK(hipHostMalloc(&pin, 8u << 20, hipHostMallocDefault)); memset(pin, 2, 8u << 20); break;
            case 'H': what = "hipHostMalloc 64 MiB"; CK(hipHostMalloc(&pin2, 64u << 20, hipHostMallocDefault)); break;
            case 'c': what = "pinned -> device 8 MiB, created stream"; CK(hipMemcpyAsync(dev, pin, 8u << 20, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); break;
            case 'n': what = "pinned -> device 8 MiB, NULL stream"; CK(hipMemcpyAsync(dev, pin, 8u << 20, hipMemcpyHostToDevice, nullptr)); CK(hipStreamSynchronize(nullptr)); break;
            case 'p': what = "pageable -> device 64 KiB"; CK(hipMemcpy(dev, page, 65536, hipMemcpyHostToDevice)); break;
            case 'P': what = "pageable -> device 64 MiB"; CK(hipMemcpy(dev, page, 64u << 20, hipMemcpyHostToDevice)); break;
            case 'd': what = "device -> pageable 64 KiB"; CK(hipMemcpy(page, dev, 65536, hipMemcpyDeviceToHost)); break;
            case 'k': what = "kernel launch + sync"; hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, nullptr, (int *)dev); CK(hipStreamSynchronize(nullptr)); break;
            case 'e': { what = "hipEventCreate + record + sync"; hipEvent_t ev; CK(hipEventCreate(&ev)); CK(hipEventRecord(ev, nullptr)); CK(hipEventSynchronize(ev)); break; }
        }
        printf("  %c %-42s %8.2f ms\n", *o, what, 1e3 * (now() - t));
    }
    printf("order %s: %.1f ms in all\n", order, 1e3 * (now() - total0));
    return 0;
}
