// stream_probe.hip — timing-only probe: how fast can 64-lane waves stream two f64 columns from
// HBM on MI355X, as a function of the access pattern?  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_probe tools/stream_probe.hip && /tmp/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void fill_random(double *p, size_t n, unsigned long long seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = (i + 1) * seed;  // splitmix64 finaliser
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        p[i] = (double)(z >> 11) * (1.0 / 9007199254740992.0) * 0.3;
    }
}

__device__ __forceinline__ double2 ld(const double2 *p, bool nt) {
    if (nt) { double2 v; v.x = __builtin_nontemporal_load(&p->x); v.y = __builtin_nontemporal_load(&p->y); return v; }
    return *p;
}

// MODE 0: each wave owns contiguous 64-KiB chunks of each column (the product's mapping)
// MODE 1: consecutive waves read consecutive 1-KiB pieces: the chip-wide front is contiguous
// MODE 2: like 1 but a whole 256-thread workgroup reads 4 KiB contiguous per column per step and
//         steps by UNROLL pieces: block-contiguous chunks of UNROLL*4 KiB
template <int MODE, int UNROLL, bool NT, bool TWO>
__global__ __launch_bounds__(256) void probe(const double2 *__restrict__ a, const double2 *__restrict__ b,
                                             size_t n2 /* double2 per column */, double *sink) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    double acc = 0.0;
    if (MODE == 0) {
        const size_t chunk = 4096;  // double2 per 64 KiB
        for (size_t t = wave; t * chunk + chunk <= n2; t += n_waves) {
            const double2 *pa = a + t * chunk, *pb = b + t * chunk;
#pragma unroll 1
            for (int j = 0; j < 64; j += UNROLL) {
                double2 va[UNROLL], vb[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) { va[u] = ld(pa + (j + u) * 64 + lane, NT); if (TWO) vb[u] = ld(pb + (j + u) * 64 + lane, NT); }
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) { acc += va[u].x + va[u].y; if (TWO) acc += vb[u].x + vb[u].y; }
            }
        }
    } else if (MODE == 1) {
        const size_t pieces = n2 / 64;
        for (size_t p = wave * UNROLL; p + UNROLL <= pieces; p += n_waves * UNROLL) {
            double2 va[UNROLL], vb[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) { va[u] = ld(a + (p + u) * 64 + lane, NT); if (TWO) vb[u] = ld(b + (p + u) * 64 + lane, NT); }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) { acc += va[u].x + va[u].y; if (TWO) acc += vb[u].x + vb[u].y; }
        }
    } else if (MODE == 3) {
        // MODE 3: a workgroup owns a 128-KiB tile per column (8192 sites) and walks it in 4-KiB
        // block pieces, UNROLL pieces per step; workgroups stride over tiles
        const size_t tile = 8192;  // double2 per 128 KiB
        const int w = threadIdx.x;
        for (size_t t = blockIdx.x; t * tile + tile <= n2; t += gridDim.x) {
            const double2 *pa = a + t * tile, *pb = b + t * tile;
#pragma unroll 1
            for (int j = 0; j < 32; j += UNROLL) {
                double2 va[UNROLL], vb[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) { va[u] = ld(pa + (j + u) * 256 + w, NT); if (TWO) vb[u] = ld(pb + (j + u) * 256 + w, NT); }
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) { acc += va[u].x + va[u].y; if (TWO) acc += vb[u].x + vb[u].y; }
            }
        }
    } else {
        const size_t pieces = n2 / 256;  // 4-KiB block pieces
        const int w = threadIdx.x;       // 0..255
        for (size_t p = (size_t)blockIdx.x * UNROLL; p + UNROLL <= pieces; p += (size_t)gridDim.x * UNROLL) {
            double2 va[UNROLL], vb[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) { va[u] = ld(a + (p + u) * 256 + w, NT); if (TWO) vb[u] = ld(b + (p + u) * 256 + w, NT); }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) { acc += va[u].x + va[u].y; if (TWO) acc += vb[u].x + vb[u].y; }
        }
    }
    if (acc == 123.456) sink[0] = acc;  // keep the loads alive
}

// MODE L: LDS-DMA.  Each wave owns 64-KiB chunks like MODE 0 but brings every 1-KiB piece into
// its private LDS ring with global_load_lds_dwordx4 (no VGPR destination), then reads its own 16
// bytes back with ds_read_b128.  GROUP pieces per column are in flight per wave (2*GROUP KiB),
// two groups double-buffered.
template <int GROUP, bool NT>
__global__ __launch_bounds__(256) void probe_lds(const double2 *__restrict__ a, const double2 *__restrict__ b,
                                                 size_t n2, double *sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    char *ring = lds + (size_t)wib * (2 * 2 * GROUP * 1024);  // [buf][col][piece] x 1 KiB
    double acc = 0.0;
    const size_t chunk = 4096;
    constexpr int aux = NT ? 2 : 0;
    for (size_t t = wave; t * chunk + chunk <= n2; t += n_waves) {
        const double2 *pa = a + t * chunk, *pb = b + t * chunk;
        auto issue = [&](int buf, int j) {
#pragma unroll
            for (int u = 0; u < GROUP; ++u) {
                __builtin_amdgcn_global_load_lds(pa + (j + u) * 64 + lane, (__attribute__((address_space(3))) void *)(ring + ((buf * 2 + 0) * GROUP + u) * 1024), 16, 0, aux);
                __builtin_amdgcn_global_load_lds(pb + (j + u) * 64 + lane, (__attribute__((address_space(3))) void *)(ring + ((buf * 2 + 1) * GROUP + u) * 1024), 16, 0, aux);
            }
        };
        issue(0, 0);
        int buf = 0;
#pragma unroll 1
        for (int j = 0; j < 64; j += GROUP) {
            if (j + GROUP < 64) {
                issue(buf ^ 1, j + GROUP);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * GROUP) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
#pragma unroll
            for (int u = 0; u < GROUP; ++u) {
                const double2 va = *reinterpret_cast<const double2 *>(ring + ((buf * 2 + 0) * GROUP + u) * 1024 + lane * 16);
                const double2 vb = *reinterpret_cast<const double2 *>(ring + ((buf * 2 + 1) * GROUP + u) * 1024 + lane * 16);
                acc += va.x + va.y + vb.x + vb.y;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the ds_reads are done before the slot is refilled
            buf ^= 1;
        }
    }
    if (acc == 123.456) sink[0] = acc;
}

template <int GROUP, bool NT>
double run_lds(const double2 *a, const double2 *b, size_t n2, double *sink, int blocks, int reps) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t shmem = 4 * (2 * 2 * GROUP * 1024);
    CK(hipFuncSetAttribute((const void *)probe_lds<GROUP, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    std::vector<float> ms;
    for (int r = 0; r < reps + 2; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((probe_lds<GROUP, NT>), dim3(blocks), dim3(256), shmem, 0, a, b, n2, sink);
        CK(hipGetLastError());
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1)); if (r >= 2) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

template <int MODE, int UNROLL, bool NT, bool TWO>
double run(const double2 *a, const double2 *b, size_t n2, double *sink, int blocks, int reps) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int r = 0; r < reps + 2; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((probe<MODE, UNROLL, NT, TWO>), dim3(blocks), dim3(256), 0, 0, a, b, n2, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1)); if (r >= 2) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

int main() {
    const size_t n = 1000000000ull;     // sites
    const size_t n2 = n / 2;            // double2 per column
    double2 *a, *b; double *sink;
    CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&sink, 64));
    // pseudo-random contents: constant (memset) buffers stream measurably faster than real data on this
    // chip (less toggling), which would overstate the ceiling the product kernel can be held against
    hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, reinterpret_cast<double *>(a), n, 0x9E3779B97F4A7C15ull);
    hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, reinterpret_cast<double *>(b), n, 0xD1B54A32D192ED03ull);
    CK(hipDeviceSynchronize());
    if (getenv("PROBE_CONSTANT_DATA")) { CK(hipMemset(a, 1, n * 8)); CK(hipMemset(b, 2, n * 8)); printf("(constant data)\n"); }
    const double gb2 = 16.0 * n / 1e9, gb1 = 8.0 * n / 1e9;
    printf("| pattern | loads in flight/lane | nt | columns | grid | median ms | GB/s |\n|---|---|---|---|---|---|---|\n");
#define ROW(MODE, UN, NT, TWO, BL, NAME) { double t = run<MODE, UN, NT, TWO>(a, b, n2, sink, BL, 9); \
    printf("| %s | %d | %d | %d | %d | %.4f | %.0f |\n", NAME, UN * (TWO ? 2 : 1), (int)NT, TWO ? 2 : 1, BL, t, (TWO ? gb2 : gb1) / t * 1e3); }
    for (int rep = 0; rep < 2; ++rep) {
        ROW(0, 4, false, true, 2048, "wave-private 64 KiB chunks")
        ROW(0, 4, true, true, 2048, "wave-private 64 KiB chunks")
        ROW(1, 4, false, true, 2048, "wave-interleaved 1 KiB pieces")
        ROW(1, 4, true, true, 2048, "wave-interleaved 1 KiB pieces")
        ROW(1, 8, true, true, 2048, "wave-interleaved 1 KiB pieces")
        ROW(2, 4, true, true, 2048, "block-contiguous 4 KiB pieces")
        ROW(2, 8, true, true, 2048, "block-contiguous 4 KiB pieces")
        ROW(2, 4, true, true, 1024, "block-contiguous 4 KiB pieces")
        ROW(2, 4, true, true, 4096, "block-contiguous 4 KiB pieces")
        ROW(3, 4, true, true, 2048, "block-owned 128 KiB tiles, 4 KiB pieces")
        ROW(3, 4, true, true, 4096, "block-owned 128 KiB tiles, 4 KiB pieces")
        ROW(3, 8, true, true, 2048, "block-owned 128 KiB tiles, 4 KiB pieces")
        ROW(3, 4, true, true, 8192, "block-owned 128 KiB tiles, 4 KiB pieces")
#define ROWL(GROUP, NT, BL) { double t = run_lds<GROUP, NT>(a, b, n2, sink, BL, 9); \
    printf("| LDS-DMA ring, wave-private chunks | %d | %d | 2 | %d | %.4f | %.0f |\n", 2 * GROUP, (int)NT, BL, t, gb2 / t * 1e3); }
        ROWL(4, false, 512) ROWL(4, true, 512) ROWL(4, true, 1024) ROWL(4, true, 2048)
        ROWL(2, true, 1280) ROWL(2, true, 2048) ROWL(8, true, 512)
        ROW(0, 8, true, false, 2048, "wave-private, ONE column")
        ROW(1, 8, true, false, 2048, "wave-interleaved, ONE column")
        ROW(2, 8, true, false, 2048, "block-contiguous, ONE column")
    }
    return 0;
}
