// pgt_api.cpp — the C-ABI of libpgtwin (include/pgtwin.h): context, argument checks, the
// host-buffer entry points (copy in -> GPU -> copy out) and the device-resident entry points.
// There is deliberately no CPU code path here: without a HIP device every reduce fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "pgt_internal.h"

// Host I/O of the host-buffer entry points (pgt_fst_reduce & co., what INTEGRATION.md binds), kept per context:
//  * a PINNED STAGING RING for the column uploads: `workers` host threads copy 16-MiB pieces of the caller's pageable columns
//    into their own two pinned slots and queue the DMA of each piece themselves (hipMemcpyAsync on one copy stream), so the
//    host-side memcpy of piece k+1 overlaps the DMA of piece k and nothing of the caller's memory has to be page-locked.
//    A plain hipMemcpy from pageable memory lets the runtime pin the caller's pages first: 0.45 ms/MiB the first time a
//    range is seen (tools/probes/upload_probe.cpp on the gpurun box: 182 ms for 400 MiB, 7.5 ms for the same buffer again)
//    — and a command-line tool sees every range for the first time;
//  * a CACHED WORKSPACE (window table, rows, tree, genome-wide total): grown when a call needs more, freed by pgt_close —
//    tools that reduce in passes call these entry points once per pass with the same sizes.
struct HostIo {
    static constexpr int kMaxWorkers = 16, kSlots = 2 * kMaxWorkers;
    static constexpr size_t kChunk = (size_t)8 << 20;  // threshold for taking the ring at all: 4 of these
    // geometry: profiles/r06/host_api_ring_geometry_sweep.txt (2 GB of fst columns, steady state): 16-MiB pieces 38.8 ms = 51.5 GB/s
    // with 2 … 12 workers alike, 8 MiB 40.5, 4 MiB 44.5, 2 MiB 52; a plain hipMemcpy of the SAME buffer again 36.5 (the runtime
    // caches its pin of the caller's pages; a first call pays 85 ms more)
    // What the ring costs to set up (tools/probes/first_use_probe.cpp): hipHostMalloc 15 ms per 64 MiB; a stream of its own 18 ms
    // (it would only buy what nobody needs: the kernels that follow wait for the columns anyway) — so: the NULL stream, and the
    // smallest geometry that still runs at the link's rate.
    int workers = 2;                   // PGT_UPLOAD_WORKERS (1 … 16)
    size_t chunk = (size_t)16 << 20;   // PGT_UPLOAD_CHUNK_MIB (1 … 64); two slots of this size per worker: 64 MiB pinned
    bool ring_ready = false;
    char *pin[kSlots] = {};
    hipEvent_t slot_done[kSlots] = {};
    hipStream_t copy_stream = nullptr;  // the NULL stream (see above)
    enum { kWin = 0, kRows, kTree, kTot, kKinds };
    void *ws[kKinds] = {};
    size_t ws_bytes[kKinds] = {};
};

struct pgt_ctx {
    int device = 0;
    std::string error;
    bool profiling = false;
    bool have_timing = false;
    pgt::Hints hints;  // pgt_set_max_window / pgt_set_window_step (0 = unknown)
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};  // build start, build end, query end
    HostIo io;
};

namespace {

using namespace pgt;

int ctx_fail(pgt_ctx *ctx, int code, const std::string &msg) {
    if (ctx) ctx->error = msg;
    set_global_error(msg);
    return code;
}

int hip_check(pgt_ctx *ctx, hipError_t e, const char *what) {
    if (e == hipSuccess) return PGT_OK;
    return ctx_fail(ctx, PGT_EDEVICE, std::string(what) + ": " + hipGetErrorString(e));
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ABI 4: every row output carries its capacity; refuse before anything is launched
int room_check(pgt_ctx *ctx, const char *who, uint64_t rows, size_t row_bytes, size_t out_bytes) {
    if (rows != 0 && (rows > SIZE_MAX / row_bytes || rows * row_bytes > out_bytes))
        return ctx_fail(ctx, PGT_EARG, std::string(who) + ": the rows do not fit the output buffer (" + std::to_string(rows) + " rows of " +
                                           std::to_string(row_bytes) + " bytes, capacity " + std::to_string(out_bytes) + " bytes)");
    return PGT_OK;
}

// RAII device buffer for the host-buffer entry points (the columns: as large as the input, not cached)
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(pgt_ctx *ctx, size_t bytes, const char *what) {
        return hip_check(ctx, hipMalloc(&p, bytes ? bytes : 16), what);
    }
};

// PGT_TRACE_API=1: one stderr line per phase of a host-buffer entry point (where a call's milliseconds go)
struct ApiTrace {
    const char *who;
    bool on;
    std::chrono::steady_clock::time_point t;
    explicit ApiTrace(const char *w) : who(w), on(std::getenv("PGT_TRACE_API") != nullptr), t(std::chrono::steady_clock::now()) {}
    void lap(const char *what, size_t bytes = 0) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        const double ms = std::chrono::duration<double, std::milli>(now - t).count();
        if (bytes) std::fprintf(stderr, "[pgt-api] %s: %-18s %9.3f ms  %8.1f MB  %6.1f GB/s\n", who, what, ms, bytes / 1e6, bytes / ms / 1e6);
        else std::fprintf(stderr, "[pgt-api] %s: %-18s %9.3f ms\n", who, what, ms);
        t = now;
    }
};

// the cached workspace: -> a device pointer good for `bytes` until the next call that asks for more of the same kind
int workspace(pgt_ctx *ctx, int kind, size_t bytes, void **out) {
    HostIo &io = ctx->io;
    if (bytes == 0) bytes = 16;
    if (io.ws_bytes[kind] < bytes) {
        if (io.ws[kind]) (void)hipFree(io.ws[kind]);
        io.ws[kind] = nullptr;
        io.ws_bytes[kind] = 0;
        const size_t want = (bytes + (bytes >> 3) + 4095) & ~(size_t)4095;  // 12 % of slack: passes of slightly different sizes
        if (int rc = hip_check(ctx, hipMalloc(&io.ws[kind], want), "workspace: hipMalloc")) return rc;
        io.ws_bytes[kind] = want;
    }
    *out = io.ws[kind];
    return PGT_OK;
}

int ring_prepare(pgt_ctx *ctx) {
    HostIo &io = ctx->io;
    if (io.ring_ready) return PGT_OK;
    if (!io.pin[0]) {  // the geometry is fixed by the first attempt
        if (const char *e = std::getenv("PGT_UPLOAD_WORKERS")) io.workers = std::min(std::max(std::atoi(e), 1), (int)HostIo::kMaxWorkers);
        if (const char *e = std::getenv("PGT_UPLOAD_CHUNK_MIB")) io.chunk = (size_t)std::min(std::max(std::atoi(e), 1), 64) << 20;
    }
    for (int s = 0; s < 2 * io.workers; ++s) {  // (a slot that exists already: an earlier attempt failed further on)
        if (!io.pin[s])
            if (int rc = hip_check(ctx, hipHostMalloc(reinterpret_cast<void **>(&io.pin[s]), io.chunk, hipHostMallocDefault), "upload ring: hipHostMalloc")) {
                io.pin[s] = nullptr;
                return rc;
            }
        if (!io.slot_done[s])
            if (int rc = hip_check(ctx, hipEventCreateWithFlags(&io.slot_done[s], hipEventDisableTiming), "upload ring: hipEventCreate")) {
                io.slot_done[s] = nullptr;
                return rc;
            }
    }
    io.ring_ready = true;
    return PGT_OK;
}

void host_io_release(pgt_ctx *ctx) {
    HostIo &io = ctx->io;
    for (int s = 0; s < HostIo::kSlots; ++s) {
        if (io.pin[s]) (void)hipHostFree(io.pin[s]);
        if (io.slot_done[s]) (void)hipEventDestroy(io.slot_done[s]);
        io.pin[s] = nullptr;
        io.slot_done[s] = nullptr;
    }
    io.ring_ready = false;
    for (int k = 0; k < HostIo::kKinds; ++k) {
        if (io.ws[k]) (void)hipFree(io.ws[k]);
        io.ws[k] = nullptr;
        io.ws_bytes[k] = 0;
    }
}

// Host columns -> freshly allocated device columns, all of one call together.
struct UploadJob {
    DevBuf *dst;
    const void *src;
    size_t bytes;
    const char *what;
};

int upload_columns(pgt_ctx *ctx, UploadJob *jobs, int n_jobs, ApiTrace &trace) {
    size_t total = 0;
    for (int j = 0; j < n_jobs; ++j) {
        if (int rc = jobs[j].dst->alloc(ctx, jobs[j].bytes, jobs[j].what)) return rc;
        total += jobs[j].bytes;
    }
    trace.lap("alloc columns");
    const char *mode = std::getenv("PGT_UPLOAD");  // "plain": hipMemcpy from the caller's pageable memory, column after column
    if (total < 4 * HostIo::kChunk || (mode && std::strcmp(mode, "plain") == 0)) {
        for (int j = 0; j < n_jobs; ++j)
            if (jobs[j].bytes)
                if (int rc = hip_check(ctx, hipMemcpy(jobs[j].dst->p, jobs[j].src, jobs[j].bytes, hipMemcpyHostToDevice), jobs[j].what)) return rc;
        trace.lap("upload (hipMemcpy)", total);
        return PGT_OK;
    }
    if (int rc = ring_prepare(ctx)) return rc;
    trace.lap("staging ring");
    struct Piece {
        char *dst;
        const char *src;
        size_t bytes;
    };
    HostIo &io = ctx->io;
    std::vector<Piece> pieces;
    for (int j = 0; j < n_jobs; ++j)
        for (size_t off = 0; off < jobs[j].bytes; off += io.chunk)
            pieces.push_back({static_cast<char *>(jobs[j].dst->p) + off, static_cast<const char *>(jobs[j].src) + off,
                              std::min(io.chunk, jobs[j].bytes - off)});
    std::atomic<size_t> next{0};
    std::atomic<int> failed{0};  // a hipError_t, first one wins
    const int device = ctx->device;
    auto worker = [&](int w) {
        if (hipSetDevice(device) != hipSuccess) { int z = 0; failed.compare_exchange_strong(z, (int)hipErrorInvalidDevice); return; }
        bool used[2] = {false, false};
        for (int turn = 0;; turn ^= 1) {
            const size_t i = next.fetch_add(1);
            if (i >= pieces.size() || failed.load()) break;
            const int s = 2 * w + turn;
            hipError_t e = used[turn] ? hipEventSynchronize(io.slot_done[s]) : hipSuccess;  // the DMA that last read this slot
            if (e == hipSuccess) {
                std::memcpy(io.pin[s], pieces[i].src, pieces[i].bytes);
                e = hipMemcpyAsync(pieces[i].dst, io.pin[s], pieces[i].bytes, hipMemcpyHostToDevice, io.copy_stream);
            }
            if (e == hipSuccess) e = hipEventRecord(io.slot_done[s], io.copy_stream);
            used[turn] = true;
            if (e != hipSuccess) { int z = 0; failed.compare_exchange_strong(z, (int)e); break; }
        }
    };
    std::vector<std::thread> th;
    const int n_workers = (int)std::min<size_t>((size_t)io.workers, pieces.size());
    for (int w = 1; w < n_workers; ++w) {
        try {
            th.emplace_back(worker, w);
        } catch (...) {  // no thread to be had: the workers that exist (at least this one) take all pieces
            break;
        }
    }
    worker(0);
    for (auto &t : th) t.join();
    const hipError_t sync = hipStreamSynchronize(io.copy_stream);
    if (failed.load()) return hip_check(ctx, (hipError_t)failed.load(), "upload columns");
    if (int rc = hip_check(ctx, sync, "upload columns: synchronize")) return rc;
    trace.lap("upload (ring)", total);
    return PGT_OK;
}

// The three hints of a host table (pgt_table_hints): longest window; typical (median) length and typical step = median
// distance between consecutive window starts, both over a sample from the middle of the table (chromosome boundaries and
// the Q1 carry make a few distances irregular).  The typical length decides whether the group query fits, not the longest
// window: base-pair windows (dxyWindow) vary in their number of sites, and a strategy chosen for the longest would run the
// fallback of most.
void derive_hints(const pgt_win *win, uint64_t n_win, uint64_t *max_window, uint64_t *typical_window, uint64_t *window_step) {
    uint64_t m = 1;
    for (uint64_t i = 0; i < n_win; ++i)
        if (win[i].hi >= win[i].lo && win[i].hi - win[i].lo > m) m = win[i].hi - win[i].lo;
    std::vector<uint64_t> d, len;
    const uint64_t from = n_win > 4097 ? n_win / 2 - 2048 : 0, to = n_win > 4097 ? from + 4096 : (n_win ? n_win - 1 : 0);
    for (uint64_t i = from; i < to; ++i) {
        if (win[i + 1].lo >= win[i].lo) d.push_back(win[i + 1].lo - win[i].lo);
        if (win[i].hi >= win[i].lo) len.push_back(win[i].hi - win[i].lo);
    }
    uint64_t typical = 0, step = 0;
    if (!len.empty()) {
        std::nth_element(len.begin(), len.begin() + len.size() / 2, len.end());
        typical = len[len.size() / 2];
    }
    if (!d.empty()) {
        std::nth_element(d.begin(), d.begin() + d.size() / 2, d.end());
        step = d[d.size() / 2];
    }
    *max_window = m;
    *typical_window = typical;
    *window_step = step;
}

// RAII: the host-buffer entry points know the table, so they set the hints that are unset themselves
struct HintScope {
    pgt_ctx *ctx;
    pgt::Hints saved;
    HintScope(pgt_ctx *c, const pgt_win *win, uint64_t n_win) : ctx(c), saved(c->hints) {
        uint64_t m = 1, typical = 0, step = 0;
        derive_hints(win, n_win, &m, &typical, &step);
        // an explicit hint (the caller knows the whole table: a shard of it arrives here) stays, and stands for the typical
        // length too — unless THIS table contradicts it (a window longer than the hint: a stale pgt_set_max_window of an
        // earlier, unrelated table; the query would walk hundreds of nodes of too low a level), then the table speaks
        if (saved.max_window == 0 || m > saved.max_window) {
            c->hints.max_window = m;
            c->hints.typical_window = typical;
            if (saved.max_window != 0) c->hints.window_step = step;  // a stale longest window means a stale step as well
        }
        if (saved.window_step == 0) c->hints.window_step = step;
    }
    ~HintScope() { ctx->hints = saved; }
};

int check_windows_host(pgt_ctx *ctx, const pgt_win *win, uint64_t n_win, uint64_t n, bool need_coords_or_sites) {
    for (uint64_t i = 0; i < n_win; ++i) {
        if (win[i].lo > win[i].hi || win[i].hi > n)
            return ctx_fail(ctx, PGT_EARG, "window range outside [0, n_sites]");
        if (need_coords_or_sites && win[i].lo == win[i].hi && !(win[i].flags & PGT_WIN_COORDS))
            return ctx_fail(ctx, PGT_EARG, "empty window without explicit coordinates");
    }
    return PGT_OK;
}

struct EvSet {
    void *b0 = nullptr, *b1 = nullptr, *q1 = nullptr;
};
EvSet events_for(pgt_ctx *ctx) {
    EvSet e;
    if (ctx->profiling) {
        e.b0 = ctx->ev[0];
        e.b1 = ctx->ev[1];
        e.q1 = ctx->ev[2];
        ctx->have_timing = true;
    }
    return e;
}

// Every entry point runs on ctx's device and leaves the CALLER's current device as it found it (a
// torch caller whose current device differs must not have it switched under its feet).
struct DeviceScope {
    int saved = -1;
    bool switched = false;
    int enter(pgt_ctx *ctx) {
        if (!ctx) return ctx_fail(nullptr, PGT_EARG, "NULL context");
        if (hipGetDevice(&saved) != hipSuccess) saved = -1;
        if (saved == ctx->device) return PGT_OK;
        if (int rc = hip_check(ctx, hipSetDevice(ctx->device), "hipSetDevice")) return rc;
        switched = true;
        return PGT_OK;
    }
    ~DeviceScope() {
        if (switched && saved >= 0) (void)hipSetDevice(saved);
    }
};
#define PGT_USE_DEVICE(ctx)  \
    DeviceScope device_scope; \
    if (int rc_ = device_scope.enter(ctx)) return rc_

}  // namespace

extern "C" {

int pgt_abi_version(void) { return PGT_ABI_VERSION; }

pgt_ctx *pgt_open(int device) {
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev == 0) {
        set_global_error(std::string("pgt_open: no HIP device available (") +
                         (e != hipSuccess ? hipGetErrorString(e) : "device count is 0") +
                         "); libpgtwin has no CPU fallback");
        return nullptr;
    }
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) device = 0;
    }
    if (device >= n_dev) {
        set_global_error("pgt_open: device ordinal out of range");
        return nullptr;
    }
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) {
        set_global_error(std::string("pgt_open: hipGetDeviceProperties: ") + hipGetErrorString(e));
        return nullptr;
    }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_global_error(std::string("pgt_open: device is ") + prop.gcnArchName +
                         ", but libpgtwin carries gfx950 (MI355X) code objects only");
        return nullptr;
    }
    int caller_device = -1;
    if (hipGetDevice(&caller_device) != hipSuccess) caller_device = -1;
    if ((e = hipSetDevice(device)) != hipSuccess) {
        set_global_error(std::string("pgt_open: hipSetDevice: ") + hipGetErrorString(e));
        return nullptr;
    }
    struct Restore {  // the caller's current device is left as it was found
        int dev, ctx_dev;
        ~Restore() { if (dev >= 0 && dev != ctx_dev) (void)hipSetDevice(dev); }
    } restore{caller_device, device};
    std::string init_err;
    if (init_kernels(&init_err) != PGT_OK || init_af_kernels(&init_err) != PGT_OK) {
        set_global_error("pgt_open: " + init_err);
        return nullptr;
    }
    pgt_ctx *ctx = new pgt_ctx;
    ctx->device = device;
    for (auto &ev : ctx->ev)
        if ((e = hipEventCreate(&ev)) != hipSuccess) {
            set_global_error(std::string("pgt_open: hipEventCreate: ") + hipGetErrorString(e));
            pgt_close(ctx);
            return nullptr;
        }
    return ctx;
}

void pgt_close(pgt_ctx *ctx) {
    if (!ctx) return;
    int saved = -1;
    const bool sw = hipGetDevice(&saved) == hipSuccess && saved != ctx->device && hipSetDevice(ctx->device) == hipSuccess;
    host_io_release(ctx);
    for (auto &ev : ctx->ev)
        if (ev) (void)hipEventDestroy(ev);
    if (sw) (void)hipSetDevice(saved);
    delete ctx;
}

int pgt_prepare_host_io(pgt_ctx *ctx, uint64_t expected_column_bytes) {
    PGT_USE_DEVICE(ctx);
    // the ring only when uploads will take it (0 = unknown: yes); 15 ms of hipHostMalloc a small input need not wait for
    const bool ring = expected_column_bytes == 0 || expected_column_bytes >= 4 * HostIo::kChunk;
    if (ring)
        if (int rc = ring_prepare(ctx)) return rc;
    // The FIRST copy of a process in each direction and flavour pays for the runtime's own set-up (its DMA queues and staging
    // buffers: 20-30 ms for the first GPU operation of any kind, 8-9 ms more for the first host-to-device DMA, ~5 ms for the first
    // small pageable copy; tools/probes/first_use_probe.cpp, host_api_probe.py) — pay it here, beside the caller's parse, with
    // 64 KiB instead of inside the first reduce.
    HostIo &io = ctx->io;
    void *scratch = nullptr;
    if (int rc = workspace(ctx, HostIo::kWin, 65536, &scratch)) return rc;
    std::vector<char> pageable(65536, 0);
    void *pinned = ring ? static_cast<void *>(io.pin[0]) : nullptr;
    if (!pinned)
        if (int rc = hip_check(ctx, hipHostMalloc(&pinned, 65536, hipHostMallocDefault), "host io warm-up: hipHostMalloc")) return rc;
    int rc = hip_check(ctx, hipMemcpyAsync(scratch, pinned, 65536, hipMemcpyHostToDevice, io.copy_stream), "host io warm-up");
    if (!rc) rc = hip_check(ctx, hipStreamSynchronize(io.copy_stream), "host io warm-up");
    if (!ring) (void)hipHostFree(pinned);
    if (rc) return rc;
    if (int rc2 = hip_check(ctx, hipMemcpy(scratch, pageable.data(), pageable.size(), hipMemcpyHostToDevice), "host io warm-up")) return rc2;
    return hip_check(ctx, hipMemcpy(pageable.data(), scratch, pageable.size(), hipMemcpyDeviceToHost), "host io warm-up");
}

const char *pgt_last_error(const pgt_ctx *ctx) {
    if (ctx) return ctx->error.c_str();
    return global_error().c_str();
}

size_t pgt_tree_bytes(int stat, uint64_t n_sites) {
    if (stat < PGT_STAT_FST || stat > PGT_STAT_EXT) return 0;
    return tree_layout(stat, n_sites).bytes;
}

int pgt_set_profiling(pgt_ctx *ctx, int enabled) {
    if (!ctx) return ctx_fail(nullptr, PGT_EARG, "NULL context");
    ctx->profiling = enabled != 0;
    ctx->have_timing = false;
    return PGT_OK;
}

int pgt_set_max_window(pgt_ctx *ctx, uint64_t max_window_sites) {
    if (!ctx) return ctx_fail(nullptr, PGT_EARG, "NULL context");
    ctx->hints.max_window = max_window_sites;
    ctx->hints.typical_window = 0;  // the explicit value stands for the typical length too
    return PGT_OK;
}

int pgt_set_typical_window(pgt_ctx *ctx, uint64_t typical_sites) {
    if (!ctx) return ctx_fail(nullptr, PGT_EARG, "NULL context");
    ctx->hints.typical_window = typical_sites;
    return PGT_OK;
}

int pgt_table_hints(const pgt_win *win, uint64_t n_win, uint64_t *max_window, uint64_t *typical_window, uint64_t *window_step) {
    if ((n_win && !win) || !max_window || !typical_window || !window_step)
        return ctx_fail(nullptr, PGT_EARG, "pgt_table_hints: NULL argument");
    derive_hints(win, n_win, max_window, typical_window, window_step);
    if (*window_step == 0) *window_step = UINT64_MAX;  // "one wave per window", said explicitly: 0 would let a slice estimate its own
    return PGT_OK;
}

int pgt_set_window_step(pgt_ctx *ctx, uint64_t step_sites) {
    if (!ctx) return ctx_fail(nullptr, PGT_EARG, "NULL context");
    ctx->hints.window_step = step_sites;
    return PGT_OK;
}

int pgt_last_kernel_ms(pgt_ctx *ctx, float *build_ms, float *query_ms) {
    if (!ctx || !build_ms || !query_ms) return ctx_fail(ctx, PGT_EARG, "pgt_last_kernel_ms: NULL argument");
    if (!ctx->have_timing) return ctx_fail(ctx, PGT_EARG, "pgt_last_kernel_ms: no profiled call yet");
    if (int rc = hip_check(ctx, hipEventSynchronize(ctx->ev[2]), "hipEventSynchronize")) return rc;
    if (int rc = hip_check(ctx, hipEventElapsedTime(build_ms, ctx->ev[0], ctx->ev[1]), "hipEventElapsedTime")) return rc;
    return hip_check(ctx, hipEventElapsedTime(query_ms, ctx->ev[1], ctx->ev[2]), "hipEventElapsedTime");
}

/* ---------------- device-resident entry points ---------------- */

int pgt_fst_reduce_pairs_dev(pgt_ctx *ctx, const uint32_t *pos, const double *const *a, const double *const *b,
                             uint32_t n_pairs, uint64_t n, const pgt_win *win, uint64_t n_win, pgt_fst_row *out,
                             size_t out_bytes, void *tree, size_t tree_bytes, void *stream) {
    PGT_USE_DEVICE(ctx);
    if (n == 0 && n_win == 0) return PGT_OK;  // a shard without windows (more ranks than windows)
    if (!a || !b || n_pairs == 0 || !tree || (n_win && (!win || !out || !pos)))
        return ctx_fail(ctx, PGT_EARG, "pgt_fst_reduce: NULL argument");
    for (uint32_t p = 0; p < n_pairs; ++p)
        if (!a[p] || !b[p] || !aligned16(a[p]) || !aligned16(b[p]))
            return ctx_fail(ctx, PGT_EARG, "pgt_fst_reduce: f64 columns must be non-NULL and 16-byte aligned");
    if (!aligned16(tree) || tree_bytes < (size_t)n_pairs * pgt_tree_bytes(PGT_STAT_FST, n))
        return ctx_fail(ctx, PGT_EARG, "pgt_fst_reduce: tree workspace too small or misaligned");
    if (n_win > UINT64_MAX / n_pairs) return ctx_fail(ctx, PGT_EARG, "pgt_fst_reduce: n_pairs * n_win overflows");
    if (int rc = room_check(ctx, "pgt_fst_reduce", (uint64_t)n_pairs * n_win, sizeof(pgt_fst_row), out_bytes)) return rc;
    const EvSet e = events_for(ctx);
    return launch_fst(pos, a, b, n_pairs, n, win, n_win, out, tree, stream, e.b0, e.b1, e.q1, &ctx->error, ctx->hints);
}

int pgt_fst_reduce_dev(pgt_ctx *ctx, const uint32_t *pos, const double *a, const double *b, uint64_t n,
                       const pgt_win *win, uint64_t n_win, pgt_fst_row *out, size_t out_bytes, void *tree,
                       size_t tree_bytes, void *stream) {
    const double *pa[1] = {a}, *pb[1] = {b};
    return pgt_fst_reduce_pairs_dev(ctx, pos, pa, pb, 1, n, win, n_win, out, out_bytes, tree, tree_bytes, stream);
}

int pgt_het_reduce_dev(pgt_ctx *ctx, const uint32_t *pos, const int8_t *g, uint64_t n, const pgt_win *win,
                       uint64_t n_win, pgt_het_row *out, size_t out_bytes, void *tree, size_t tree_bytes, void *stream) {
    PGT_USE_DEVICE(ctx);
    if (n == 0 && n_win == 0) return PGT_OK;
    if (!g || !tree || (n_win && (!win || !out || !pos))) return ctx_fail(ctx, PGT_EARG, "pgt_het_reduce: NULL argument");
    if (!aligned16(g)) return ctx_fail(ctx, PGT_EARG, "pgt_het_reduce: genotype column must be 16-byte aligned");
    if (n >= (1ull << 32)) return ctx_fail(ctx, PGT_EARG, "pgt_het_reduce: at most 2^32-1 sites per call");
    if (!aligned16(tree) || tree_bytes < pgt_tree_bytes(PGT_STAT_HET, n))
        return ctx_fail(ctx, PGT_EARG, "pgt_het_reduce: tree workspace too small or misaligned");
    if (int rc = room_check(ctx, "pgt_het_reduce", n_win, sizeof(pgt_het_row), out_bytes)) return rc;
    const EvSet e = events_for(ctx);
    return launch_het(pos, g, n, win, n_win, out, tree, stream, e.b0, e.b1, e.q1, &ctx->error, ctx->hints);
}

int pgt_dxy_reduce_dev(pgt_ctx *ctx, const uint32_t *pos, const double *p1, const double *p2, const int32_t *n1,
                       const int32_t *n2, uint64_t n, int minind, const pgt_win *win, uint64_t n_win,
                       pgt_dxy_row *out, size_t out_bytes, pgt_dxy_total *tot, void *tree, size_t tree_bytes, void *stream) {
    PGT_USE_DEVICE(ctx);
    if (n == 0 && n_win == 0 && !tot) return PGT_OK;
    if (!p1 || !p2 || !n1 || !n2 || !tree || (n_win && (!win || !out)))
        return ctx_fail(ctx, PGT_EARG, "pgt_dxy_reduce: NULL argument");
    if (!aligned16(p1) || !aligned16(p2) || !aligned16(n1) || !aligned16(n2))
        return ctx_fail(ctx, PGT_EARG, "pgt_dxy_reduce: the f64 and i32 columns must be 16-byte aligned");
    if (n >= (1ull << 32)) return ctx_fail(ctx, PGT_EARG, "pgt_dxy_reduce: at most 2^32-1 sites per call");
    if (!aligned16(tree) || tree_bytes < pgt_tree_bytes(PGT_STAT_DXY, n))
        return ctx_fail(ctx, PGT_EARG, "pgt_dxy_reduce: tree workspace too small or misaligned");
    if (int rc = room_check(ctx, "pgt_dxy_reduce", n_win, sizeof(pgt_dxy_row), out_bytes)) return rc;
    const EvSet e = events_for(ctx);
    return launch_dxy(pos, p1, p2, n1, n2, n, minind, win, n_win, out, tot, tree, stream, e.b0, e.b1, e.q1,
                      &ctx->error, ctx->hints);
}

int pgt_dxy_het_reduce_dev(pgt_ctx *ctx, const uint32_t *pos, const double *p1, const double *p2, const int32_t *n1,
                           const int32_t *n2, const int8_t *g1, const int8_t *g2, uint64_t n, int minind,
                           const pgt_win *win, uint64_t n_win, pgt_dxy_row *dxy_out, size_t dxy_out_bytes, pgt_dxy_total *tot,
                           pgt_het_row *het_out1, pgt_het_row *het_out2, size_t het_out_bytes, void *tree, size_t tree_bytes,
                           void *stream) {
    PGT_USE_DEVICE(ctx);
    if (n == 0 && n_win == 0 && !tot) return PGT_OK;
    if (!p1 || !p2 || !n1 || !n2 || !g1 || !g2 || !tree || (n_win && (!win || !dxy_out || !het_out1 || !het_out2 || !pos)))
        return ctx_fail(ctx, PGT_EARG, "pgt_dxy_het_reduce: NULL argument");
    if (!aligned16(p1) || !aligned16(p2) || !aligned16(g1) || !aligned16(g2) ||
        !aligned16(n1) || !aligned16(n2))
        return ctx_fail(ctx, PGT_EARG, "pgt_dxy_het_reduce: the f64, i32 and i8 columns must be 16-byte aligned");
    if (n >= (1ull << 32)) return ctx_fail(ctx, PGT_EARG, "pgt_dxy_het_reduce: at most 2^32-1 sites per call");
    if (!aligned16(tree) || tree_bytes < pgt_tree_bytes(PGT_STAT_DXY, n) + 2 * pgt_tree_bytes(PGT_STAT_HET, n))
        return ctx_fail(ctx, PGT_EARG, "pgt_dxy_het_reduce: tree workspace too small or misaligned");
    if (int rc = room_check(ctx, "pgt_dxy_het_reduce (dxy rows)", n_win, sizeof(pgt_dxy_row), dxy_out_bytes)) return rc;
    if (int rc = room_check(ctx, "pgt_dxy_het_reduce (het rows)", n_win, sizeof(pgt_het_row), het_out_bytes)) return rc;
    const EvSet e = events_for(ctx);
    return launch_dxy_het(pos, p1, p2, n1, n2, g1, g2, n, minind, win, n_win, dxy_out, tot, het_out1, het_out2, tree,
                          stream, e.b0, e.b1, e.q1, &ctx->error, ctx->hints);
}

size_t pgt_af_tree_bytes(uint32_t n_pops, uint64_t n_sites) {
    if (n_pops < 2 || n_pops > (uint32_t)kAfMaxPops) return 0;
    return af_tree_bytes(tree_layout(PGT_STAT_FST, n_sites), (int)(n_pops + n_pops * (n_pops - 1) / 2));
}

int pgt_fst_af_reduce_dev(pgt_ctx *ctx, const uint32_t *pos, const double *const *freq, const double *nsamp,
                          uint32_t n_pops, uint64_t n, const pgt_win *win, uint64_t n_win, pgt_fst_row *out, size_t out_bytes,
                          void *tree, size_t tree_bytes, void *stream) {
    PGT_USE_DEVICE(ctx);
    if (n == 0 && n_win == 0) return PGT_OK;
    if (!freq || !nsamp || !tree || (n_win && (!win || !out || !pos)))
        return ctx_fail(ctx, PGT_EARG, "pgt_fst_af_reduce: NULL argument");
    if (n_pops < 2 || n_pops > (uint32_t)kAfMaxPops) return ctx_fail(ctx, PGT_EARG, "pgt_fst_af_reduce: 2 <= n_pops <= 8");
    for (uint32_t k = 0; k < n_pops; ++k) {
        if (!freq[k] || !aligned16(freq[k]))
            return ctx_fail(ctx, PGT_EARG, "pgt_fst_af_reduce: frequency columns must be non-NULL and 16-byte aligned");
        if (!(nsamp[k] > 0.0)) return ctx_fail(ctx, PGT_EARG, "pgt_fst_af_reduce: sample sizes must be positive");
    }
    if (!aligned16(tree) || tree_bytes < pgt_af_tree_bytes(n_pops, n))
        return ctx_fail(ctx, PGT_EARG, "pgt_fst_af_reduce: tree workspace too small or misaligned");
    if (n_win > UINT64_MAX / 28) return ctx_fail(ctx, PGT_EARG, "pgt_fst_af_reduce: n_pairs * n_win overflows");
    if (int rc = room_check(ctx, "pgt_fst_af_reduce", (uint64_t)(n_pops * (n_pops - 1) / 2) * n_win, sizeof(pgt_fst_row), out_bytes)) return rc;
    const EvSet e = events_for(ctx);
    return launch_fst_af(pos, freq, nsamp, n_pops, n, win, n_win, out, tree, stream, e.b0, e.b1, e.q1, &ctx->error,
                         ctx->hints);
}

int pgt_extreme_reduce_dev(pgt_ctx *ctx, const uint32_t *pos, const double *score, uint64_t n, int mode, double cutoff,
                           const pgt_win *win, uint64_t n_win, pgt_ext_row *out, size_t out_bytes, void *tree,
                           size_t tree_bytes, void *stream) {
    PGT_USE_DEVICE(ctx);
    if (n == 0 && n_win == 0) return PGT_OK;
    if (!score || !tree || (n_win && (!win || !out || !pos))) return ctx_fail(ctx, PGT_EARG, "pgt_extreme_reduce: NULL argument");
    if (mode < PGT_EXT_IHS || mode > PGT_EXT_XP_MIN) return ctx_fail(ctx, PGT_EARG, "pgt_extreme_reduce: unknown mode");
    if (!aligned16(score)) return ctx_fail(ctx, PGT_EARG, "pgt_extreme_reduce: score column must be 16-byte aligned");
    if (n >= 0xFFFFFFFFull) return ctx_fail(ctx, PGT_EARG, "pgt_extreme_reduce: at most 2^32-2 sites per call");
    if (!aligned16(tree) || tree_bytes < pgt_tree_bytes(PGT_STAT_EXT, n))
        return ctx_fail(ctx, PGT_EARG, "pgt_extreme_reduce: tree workspace too small or misaligned");
    if (int rc = room_check(ctx, "pgt_extreme_reduce", n_win, sizeof(pgt_ext_row), out_bytes)) return rc;
    const EvSet e = events_for(ctx);
    return launch_ext(pos, score, n, mode, cutoff, win, n_win, out, tree, stream, e.b0, e.b1, e.q1, &ctx->error,
                      ctx->hints);
}

/* ---------------- device-side text ingest ---------------- */

int pgt_ingest_text(pgt_ctx *ctx, const char *text, size_t len, const uint8_t *tokens, int n_tokens, pgt_ingest **out) {
    PGT_USE_DEVICE(ctx);
    return ingest_text(ctx->device, text, len, tokens, n_tokens, 0, out, &ctx->error);
}

int pgt_ingest_text_behind(pgt_ctx *ctx, const char *text, size_t len, const uint8_t *tokens, int n_tokens, uint64_t rows_in_front,
                           pgt_ingest **out) {
    PGT_USE_DEVICE(ctx);
    return ingest_text(ctx->device, text, len, tokens, n_tokens, rows_in_front, out, &ctx->error);
}

/* ---------------- multi-GPU row exchange ---------------- */

static_assert(sizeof(hipIpcMemHandle_t) == PGT_IPC_HANDLE_BYTES, "pgt_ipc_handle must hold a hipIpcMemHandle_t");

int pgt_peer_access(pgt_ctx *ctx, int peer_device) {
    PGT_USE_DEVICE(ctx);
    if (peer_device == ctx->device) return PGT_OK;
    int can = 0;
    if (int rc = hip_check(ctx, hipDeviceCanAccessPeer(&can, ctx->device, peer_device), "hipDeviceCanAccessPeer")) return rc;
    if (!can) return ctx_fail(ctx, PGT_EDEVICE, "pgt_peer_access: no peer access between the two devices");
    const hipError_t e = hipDeviceEnablePeerAccess(peer_device, 0);
    if (e == hipErrorPeerAccessAlreadyEnabled) {
        (void)hipGetLastError();
        return PGT_OK;
    }
    return hip_check(ctx, e, "hipDeviceEnablePeerAccess");
}

int pgt_rowbuf_create(pgt_ctx *ctx, size_t bytes, void **dev_ptr, pgt_ipc_handle *handle) {
    PGT_USE_DEVICE(ctx);
    if (!dev_ptr || !handle) return ctx_fail(ctx, PGT_EARG, "pgt_rowbuf_create: NULL argument");
    void *p = nullptr;
    if (int rc = hip_check(ctx, hipMalloc(&p, bytes ? bytes : 256), "pgt_rowbuf_create: hipMalloc")) return rc;
    hipIpcMemHandle_t h;
    if (int rc = hip_check(ctx, hipIpcGetMemHandle(&h, p), "pgt_rowbuf_create: hipIpcGetMemHandle")) {
        (void)hipFree(p);
        return rc;
    }
    std::memcpy(handle->opaque, &h, sizeof h);
    *dev_ptr = p;
    return PGT_OK;
}

int pgt_rowbuf_open(pgt_ctx *ctx, const pgt_ipc_handle *handle, void **dev_ptr) {
    PGT_USE_DEVICE(ctx);
    if (!dev_ptr || !handle) return ctx_fail(ctx, PGT_EARG, "pgt_rowbuf_open: NULL argument");
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle->opaque, sizeof h);
    void *p = nullptr;
    if (int rc = hip_check(ctx, hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess), "pgt_rowbuf_open: hipIpcOpenMemHandle"))
        return rc;
    *dev_ptr = p;
    return PGT_OK;
}

int pgt_rowbuf_close(pgt_ctx *ctx, void *dev_ptr, int owner) {
    PGT_USE_DEVICE(ctx);
    if (!dev_ptr) return PGT_OK;
    return owner ? hip_check(ctx, hipFree(dev_ptr), "pgt_rowbuf_close: hipFree")
                 : hip_check(ctx, hipIpcCloseMemHandle(dev_ptr), "pgt_rowbuf_close: hipIpcCloseMemHandle");
}

int pgt_rowbuf_read(pgt_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes, void *stream) {
    PGT_USE_DEVICE(ctx);
    if (bytes == 0) return PGT_OK;
    if (!host_dst || !dev_src) return ctx_fail(ctx, PGT_EARG, "pgt_rowbuf_read: NULL argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc = hip_check(ctx, hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, s), "pgt_rowbuf_read: copy")) return rc;
    return hip_check(ctx, hipStreamSynchronize(s), "pgt_rowbuf_read: synchronize");
}

int pgt_rowbuf_fill(pgt_ctx *ctx, void *dev_ptr, size_t bytes, uint64_t seed, void *stream) {
    PGT_USE_DEVICE(ctx);
    if (bytes == 0) return PGT_OK;
    if (!dev_ptr || (bytes & 7u) || (reinterpret_cast<uintptr_t>(dev_ptr) & 7u))
        return ctx_fail(ctx, PGT_EARG, "pgt_rowbuf_fill: NULL or misaligned argument");
    return launch_fill_pattern(static_cast<uint64_t *>(dev_ptr), bytes / 8, seed, stream, &ctx->error);
}

/* ---------------- plain device buffers ---------------- */

int pgt_dev_alloc(pgt_ctx *ctx, size_t bytes, void **dev_ptr) {
    PGT_USE_DEVICE(ctx);
    if (!dev_ptr) return ctx_fail(ctx, PGT_EARG, "pgt_dev_alloc: NULL argument");
    return hip_check(ctx, hipMalloc(dev_ptr, bytes ? bytes : 16), "pgt_dev_alloc: hipMalloc");
}

int pgt_dev_memory(pgt_ctx *ctx, size_t *free_bytes, size_t *total_bytes) {
    PGT_USE_DEVICE(ctx);
    size_t f = 0, t = 0;
    if (int rc = hip_check(ctx, hipMemGetInfo(&f, &t), "pgt_dev_memory: hipMemGetInfo")) return rc;
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return PGT_OK;
}

int pgt_dev_free(pgt_ctx *ctx, void *dev_ptr) {
    PGT_USE_DEVICE(ctx);
    if (!dev_ptr) return PGT_OK;
    return hip_check(ctx, hipFree(dev_ptr), "pgt_dev_free: hipFree");
}

int pgt_dev_upload(pgt_ctx *ctx, void *dev_dst, const void *host_src, size_t bytes) {
    PGT_USE_DEVICE(ctx);
    if (bytes == 0) return PGT_OK;
    if (!dev_dst || !host_src) return ctx_fail(ctx, PGT_EARG, "pgt_dev_upload: NULL argument");
    return hip_check(ctx, hipMemcpy(dev_dst, host_src, bytes, hipMemcpyHostToDevice), "pgt_dev_upload: hipMemcpy");
}

int pgt_dev_copy(pgt_ctx *dst_ctx, void *dst, pgt_ctx *src_ctx, const void *src, size_t bytes) {
    PGT_USE_DEVICE(dst_ctx);
    if (!src_ctx) return ctx_fail(dst_ctx, PGT_EARG, "pgt_dev_copy: NULL source context");
    if (bytes == 0) return PGT_OK;
    if (!dst || !src) return ctx_fail(dst_ctx, PGT_EARG, "pgt_dev_copy: NULL argument");
    if (src_ctx->device == dst_ctx->device)
        return hip_check(dst_ctx, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice), "pgt_dev_copy: hipMemcpy");
    return hip_check(dst_ctx, hipMemcpyPeer(dst, dst_ctx->device, src, src_ctx->device, bytes), "pgt_dev_copy: hipMemcpyPeer");
}

/* ---------------- host-buffer entry points ---------------- */

extern "C++" {
namespace {

// What every *_cols entry point does around its device call: window table up (cached workspace), rows and tree from the
// cached workspace, kernels, rows down.  `run` gets (device windows, device rows, tree, tree bytes, device total or NULL).
template <class Row, class Run>
int reduce_with_workspace(pgt_ctx *ctx, const char *who, int stat, uint64_t n, const pgt_win *win, uint64_t n_win, Row *out, size_t out_bytes,
                          pgt_dxy_total *tot, Run run) {
    ApiTrace trace(who);
    if (int rc = room_check(ctx, who, n_win, sizeof(Row), out_bytes)) return rc;
    if (int rc = check_windows_host(ctx, win, n_win, n, true)) return rc;
    const HintScope hint(ctx, win, n_win);
    void *dwin = nullptr, *dout = nullptr, *dtree = nullptr, *dtot = nullptr;
    const size_t tb = pgt_tree_bytes(stat, n);
    if (int rc = workspace(ctx, HostIo::kWin, n_win * sizeof(pgt_win), &dwin)) return rc;
    if (int rc = workspace(ctx, HostIo::kRows, n_win * sizeof(Row), &dout)) return rc;
    if (int rc = workspace(ctx, HostIo::kTree, tb, &dtree)) return rc;
    if (tot)
        if (int rc = workspace(ctx, HostIo::kTot, sizeof(pgt_dxy_total), &dtot)) return rc;
    trace.lap("workspace");
    if (n_win)
        if (int rc = hip_check(ctx, hipMemcpy(dwin, win, n_win * sizeof(pgt_win), hipMemcpyHostToDevice), "upload windows")) return rc;
    trace.lap("upload windows", n_win * sizeof(pgt_win));
    if (int rc = run(static_cast<const pgt_win *>(dwin), static_cast<Row *>(dout), dtree, tb, static_cast<pgt_dxy_total *>(dtot))) return rc;
    if (int rc = hip_check(ctx, hipStreamSynchronize(nullptr), "kernels")) return rc;
    trace.lap("kernels");
    if (n_win)
        if (int rc = hip_check(ctx, hipMemcpy(out, dout, n_win * sizeof(Row), hipMemcpyDeviceToHost), "download rows")) return rc;
    if (tot)
        if (int rc = hip_check(ctx, hipMemcpy(tot, dtot, sizeof(pgt_dxy_total), hipMemcpyDeviceToHost), "download total")) return rc;
    trace.lap("download rows", n_win * sizeof(Row));
    return PGT_OK;
}

}  // namespace
}  // extern "C++"

int pgt_extreme_reduce_cols(pgt_ctx *ctx, const uint32_t *d_pos, const double *d_score, uint64_t n, int mode, double cutoff,
                            const pgt_win *win, uint64_t n_win, pgt_ext_row *out, size_t out_bytes) {
    PGT_USE_DEVICE(ctx);
    if ((n && (!d_pos || !d_score)) || (n_win && (!win || !out))) return ctx_fail(ctx, PGT_EARG, "pgt_extreme_reduce_cols: NULL argument");
    return reduce_with_workspace<pgt_ext_row>(ctx, "pgt_extreme_reduce_cols", PGT_STAT_EXT, n, win, n_win, out, out_bytes, nullptr,
        [&](const pgt_win *dw, pgt_ext_row *dr, void *tree, size_t tb, pgt_dxy_total *) {
            return pgt_extreme_reduce_dev(ctx, d_pos, d_score, n, mode, cutoff, dw, n_win, dr, n_win * sizeof(pgt_ext_row), tree, tb, nullptr);
        });
}

int pgt_extreme_reduce(pgt_ctx *ctx, const uint32_t *pos, const double *score, uint64_t n, int mode, double cutoff,
                       const pgt_win *win, uint64_t n_win, pgt_ext_row *out) {
    PGT_USE_DEVICE(ctx);
    if ((n && (!pos || !score)) || (n_win && (!win || !out))) return ctx_fail(ctx, PGT_EARG, "pgt_extreme_reduce: NULL argument");
    ApiTrace trace("pgt_extreme_reduce");
    DevBuf dpos, ds;
    UploadJob jobs[] = {{&dpos, pos, n * sizeof(uint32_t), "upload pos"}, {&ds, score, n * sizeof(double), "upload scores"}};
    if (int rc = upload_columns(ctx, jobs, 2, trace)) return rc;
    return pgt_extreme_reduce_cols(ctx, static_cast<uint32_t *>(dpos.p), static_cast<double *>(ds.p), n, mode, cutoff, win, n_win, out,
                                   (size_t)n_win * sizeof(pgt_ext_row));
}


/* device columns (e.g. from pgt_ingest_text), window table and rows in HOST memory */
int pgt_fst_reduce_cols(pgt_ctx *ctx, const uint32_t *d_pos, const double *d_a, const double *d_b, uint64_t n,
                        const pgt_win *win, uint64_t n_win, pgt_fst_row *out, size_t out_bytes) {
    PGT_USE_DEVICE(ctx);
    if ((n && (!d_pos || !d_a || !d_b)) || (n_win && (!win || !out))) return ctx_fail(ctx, PGT_EARG, "pgt_fst_reduce: NULL argument");
    return reduce_with_workspace<pgt_fst_row>(ctx, "pgt_fst_reduce_cols", PGT_STAT_FST, n, win, n_win, out, out_bytes, nullptr,
        [&](const pgt_win *dw, pgt_fst_row *dr, void *tree, size_t tb, pgt_dxy_total *) {
            return pgt_fst_reduce_dev(ctx, d_pos, d_a, d_b, n, dw, n_win, dr, n_win * sizeof(pgt_fst_row), tree, tb, nullptr);
        });
}

int pgt_fst_reduce(pgt_ctx *ctx, const uint32_t *pos, const double *a, const double *b, uint64_t n,
                   const pgt_win *win, uint64_t n_win, pgt_fst_row *out) {
    PGT_USE_DEVICE(ctx);
    if ((n && (!pos || !a || !b)) || (n_win && (!win || !out))) return ctx_fail(ctx, PGT_EARG, "pgt_fst_reduce: NULL argument");
    ApiTrace trace("pgt_fst_reduce");
    DevBuf dpos, da, db;
    UploadJob jobs[] = {{&dpos, pos, n * sizeof(uint32_t), "upload pos"}, {&da, a, n * sizeof(double), "upload a"},
                        {&db, b, n * sizeof(double), "upload b"}};
    if (int rc = upload_columns(ctx, jobs, 3, trace)) return rc;
    return pgt_fst_reduce_cols(ctx, static_cast<uint32_t *>(dpos.p), static_cast<double *>(da.p), static_cast<double *>(db.p), n,
                               win, n_win, out, n_win * sizeof(pgt_fst_row));
}

int pgt_het_reduce_cols(pgt_ctx *ctx, const uint32_t *d_pos, const int8_t *d_g, uint64_t n, const pgt_win *win,
                        uint64_t n_win, pgt_het_row *out, size_t out_bytes) {
    PGT_USE_DEVICE(ctx);
    if ((n && (!d_pos || !d_g)) || (n_win && (!win || !out))) return ctx_fail(ctx, PGT_EARG, "pgt_het_reduce: NULL argument");
    return reduce_with_workspace<pgt_het_row>(ctx, "pgt_het_reduce_cols", PGT_STAT_HET, n, win, n_win, out, out_bytes, nullptr,
        [&](const pgt_win *dw, pgt_het_row *dr, void *tree, size_t tb, pgt_dxy_total *) {
            return pgt_het_reduce_dev(ctx, d_pos, d_g, n, dw, n_win, dr, n_win * sizeof(pgt_het_row), tree, tb, nullptr);
        });
}

int pgt_het_reduce(pgt_ctx *ctx, const uint32_t *pos, const int8_t *g, uint64_t n, const pgt_win *win,
                   uint64_t n_win, pgt_het_row *out) {
    PGT_USE_DEVICE(ctx);
    if ((n && (!pos || !g)) || (n_win && (!win || !out))) return ctx_fail(ctx, PGT_EARG, "pgt_het_reduce: NULL argument");
    ApiTrace trace("pgt_het_reduce");
    DevBuf dpos, dg;
    UploadJob jobs[] = {{&dpos, pos, n * sizeof(uint32_t), "upload pos"}, {&dg, g, n * sizeof(int8_t), "upload genotypes"}};
    if (int rc = upload_columns(ctx, jobs, 2, trace)) return rc;
    return pgt_het_reduce_cols(ctx, static_cast<uint32_t *>(dpos.p), static_cast<int8_t *>(dg.p), n, win, n_win, out,
                               n_win * sizeof(pgt_het_row));
}

int pgt_dxy_reduce_cols(pgt_ctx *ctx, const uint32_t *d_pos, const double *d_p1, const double *d_p2, const int32_t *d_n1,
                        const int32_t *d_n2, uint64_t n, int minind, const pgt_win *win, uint64_t n_win,
                        pgt_dxy_row *out, size_t out_bytes, pgt_dxy_total *tot) {
    PGT_USE_DEVICE(ctx);
    if ((n && (!d_pos || !d_p1 || !d_p2 || !d_n1 || !d_n2)) || (n_win && (!win || !out)))
        return ctx_fail(ctx, PGT_EARG, "pgt_dxy_reduce: NULL argument");
    return reduce_with_workspace<pgt_dxy_row>(ctx, "pgt_dxy_reduce_cols", PGT_STAT_DXY, n, win, n_win, out, out_bytes, tot,
        [&](const pgt_win *dw, pgt_dxy_row *dr, void *tree, size_t tb, pgt_dxy_total *dtot) {
            return pgt_dxy_reduce_dev(ctx, d_pos, d_p1, d_p2, d_n1, d_n2, n, minind, dw, n_win, dr, n_win * sizeof(pgt_dxy_row), dtot, tree, tb,
                                      nullptr);
        });
}

int pgt_dxy_reduce(pgt_ctx *ctx, const uint32_t *pos, const double *p1, const double *p2, const int32_t *n1,
                   const int32_t *n2, uint64_t n, int minind, const pgt_win *win, uint64_t n_win,
                   pgt_dxy_row *out, pgt_dxy_total *tot) {
    PGT_USE_DEVICE(ctx);
    if ((n && (!pos || !p1 || !p2 || !n1 || !n2)) || (n_win && (!win || !out)))
        return ctx_fail(ctx, PGT_EARG, "pgt_dxy_reduce: NULL argument");
    ApiTrace trace("pgt_dxy_reduce");
    DevBuf dpos, d1, d2, dn1, dn2;
    UploadJob jobs[] = {{&dpos, pos, n * sizeof(uint32_t), "upload pos"}, {&d1, p1, n * sizeof(double), "upload p1"},
                        {&d2, p2, n * sizeof(double), "upload p2"}, {&dn1, n1, n * sizeof(int32_t), "upload n1"},
                        {&dn2, n2, n * sizeof(int32_t), "upload n2"}};
    if (int rc = upload_columns(ctx, jobs, 5, trace)) return rc;
    return pgt_dxy_reduce_cols(ctx, static_cast<uint32_t *>(dpos.p), static_cast<double *>(d1.p), static_cast<double *>(d2.p),
                               static_cast<int32_t *>(dn1.p), static_cast<int32_t *>(dn2.p), n, minind, win, n_win, out,
                               n_win * sizeof(pgt_dxy_row), tot);
}

/* ---------------- window tables built on the device ---------------- */

struct pgt_wintab {
    int device = 0;
    uint64_t n_win = 0;
    uint32_t W = 0, S = 0;
    std::vector<uint64_t> first;  // n_runs + 1
    pgt_win *d_win = nullptr;
};

int pgt_wintab_sites(pgt_ctx *ctx, const uint64_t *run_len, size_t n_runs, uint32_t W, uint32_t S, pgt_wintab **out) {
    PGT_USE_DEVICE(ctx);
    if (!out) return ctx_fail(ctx, PGT_EARG, "pgt_wintab_sites: NULL argument");
    std::vector<RunPlan> plan;
    uint64_t count = 0;
    if (int rc = plan_site_windows(run_len, n_runs, W, S, plan, &count)) return ctx_fail(ctx, rc, global_error());
    std::unique_ptr<pgt_wintab> tab(new pgt_wintab);
    tab->device = ctx->device;
    tab->n_win = count;
    tab->W = W;
    tab->S = S;
    tab->first.resize(n_runs + 1);
    for (size_t r = 0; r < n_runs; ++r) tab->first[r] = plan[r].out0;
    tab->first[n_runs] = count;
    if (count) {
        DevBuf dplan;
        if (int rc = dplan.alloc(ctx, plan.size() * sizeof(RunPlan), "alloc window plan")) return rc;
        if (int rc = hip_check(ctx, hipMemcpy(dplan.p, plan.data(), plan.size() * sizeof(RunPlan), hipMemcpyHostToDevice), "upload window plan")) return rc;
        void *p = nullptr;
        if (int rc = hip_check(ctx, hipMalloc(&p, count * sizeof(pgt_win)), "alloc window table")) return rc;
        tab->d_win = static_cast<pgt_win *>(p);
        int rc = launch_windows_from_plan(static_cast<const RunPlan *>(dplan.p), n_runs, count, W, S, tab->d_win, nullptr, &ctx->error);
        if (!rc) rc = hip_check(ctx, hipStreamSynchronize(nullptr), "window table kernel");  // dplan is freed on return
        if (rc) {
            (void)hipFree(tab->d_win);
            return rc;
        }
    }
    *out = tab.release();
    return PGT_OK;
}

uint64_t pgt_wintab_size(const pgt_wintab *tab) { return tab ? tab->n_win : 0; }
const uint64_t *pgt_wintab_first(const pgt_wintab *tab) { return tab ? tab->first.data() : nullptr; }
const pgt_win *pgt_wintab_device(const pgt_wintab *tab) { return tab ? tab->d_win : nullptr; }
void pgt_wintab_free(pgt_wintab *tab) {
    if (!tab) return;
    int saved = -1;
    const bool sw = hipGetDevice(&saved) == hipSuccess && saved != tab->device && hipSetDevice(tab->device) == hipSuccess;
    if (tab->d_win) (void)hipFree(tab->d_win);
    if (sw) (void)hipSetDevice(saved);
    delete tab;
}

namespace {
// the hints a table built from (W, S) implies, for the duration of one reduce
struct TabHints {
    pgt_ctx *ctx;
    pgt::Hints saved;
    TabHints(pgt_ctx *c, const pgt_wintab *t) : ctx(c), saved(c->hints) {
        c->hints.max_window = t->W;
        c->hints.typical_window = 0;  // W stands for the typical length
        c->hints.window_step = t->S;
    }
    ~TabHints() { ctx->hints = saved; }
};
int tab_check(pgt_ctx *ctx, const pgt_wintab *tab, const void *out, const char *who) {
    if (!tab || (tab->n_win && !out)) return ctx_fail(ctx, PGT_EARG, std::string(who) + ": NULL argument");
    if (tab->device != ctx->device) return ctx_fail(ctx, PGT_EARG, std::string(who) + ": the window table lives on another device");
    return PGT_OK;
}
}  // namespace

extern "C++" {
namespace {
// the *_tab entry points: table already on the device; rows and tree from the cached workspace
template <class Row, class Run>
int reduce_tab(pgt_ctx *ctx, const char *who, int stat, uint64_t n, const pgt_wintab *tab, Row *out, pgt_dxy_total *tot, ApiTrace &trace, Run run) {
    const TabHints hint(ctx, tab);
    void *dout = nullptr, *dtree = nullptr, *dtot = nullptr;
    const size_t tb = pgt_tree_bytes(stat, n);
    if (int rc = workspace(ctx, HostIo::kRows, tab->n_win * sizeof(Row), &dout)) return rc;
    if (int rc = workspace(ctx, HostIo::kTree, tb, &dtree)) return rc;
    if (tot)
        if (int rc = workspace(ctx, HostIo::kTot, sizeof(pgt_dxy_total), &dtot)) return rc;
    trace.lap("workspace");
    if (int rc = run(static_cast<Row *>(dout), dtree, tb, static_cast<pgt_dxy_total *>(dtot))) return rc;
    if (int rc = hip_check(ctx, hipStreamSynchronize(nullptr), who)) return rc;
    trace.lap("kernels");
    if (tab->n_win)
        if (int rc = hip_check(ctx, hipMemcpy(out, dout, tab->n_win * sizeof(Row), hipMemcpyDeviceToHost), "download rows")) return rc;
    if (tot)
        if (int rc = hip_check(ctx, hipMemcpy(tot, dtot, sizeof(pgt_dxy_total), hipMemcpyDeviceToHost), "download total")) return rc;
    trace.lap("download rows", tab->n_win * sizeof(Row));
    return PGT_OK;
}
}  // namespace
}  // extern "C++"

int pgt_fst_reduce_tab(pgt_ctx *ctx, const uint32_t *pos, const double *a, const double *b, uint64_t n, int cols_on_device,
                       const pgt_wintab *tab, pgt_fst_row *out, size_t out_bytes) {
    PGT_USE_DEVICE(ctx);
    if (int rc = tab_check(ctx, tab, out, "pgt_fst_reduce_tab")) return rc;
    if (int rc = room_check(ctx, "pgt_fst_reduce_tab", tab->n_win, sizeof(pgt_fst_row), out_bytes)) return rc;
    if (n && (!pos || !a || !b)) return ctx_fail(ctx, PGT_EARG, "pgt_fst_reduce_tab: NULL argument");
    if (tab->n_win == 0) return PGT_OK;
    ApiTrace trace("pgt_fst_reduce_tab");
    DevBuf dpos, da, db;
    if (!cols_on_device) {
        UploadJob jobs[] = {{&dpos, pos, n * sizeof(uint32_t), "upload pos"}, {&da, a, n * sizeof(double), "upload a"},
                            {&db, b, n * sizeof(double), "upload b"}};
        if (int rc = upload_columns(ctx, jobs, 3, trace)) return rc;
        pos = static_cast<uint32_t *>(dpos.p); a = static_cast<double *>(da.p); b = static_cast<double *>(db.p);
    }
    return reduce_tab<pgt_fst_row>(ctx, "fst kernels", PGT_STAT_FST, n, tab, out, nullptr, trace,
        [&](pgt_fst_row *dr, void *tree, size_t tb, pgt_dxy_total *) {
            return pgt_fst_reduce_dev(ctx, pos, a, b, n, tab->d_win, tab->n_win, dr, tab->n_win * sizeof(pgt_fst_row), tree, tb, nullptr);
        });
}

int pgt_het_reduce_tab(pgt_ctx *ctx, const uint32_t *pos, const int8_t *g, uint64_t n, int cols_on_device, const pgt_wintab *tab,
                       pgt_het_row *out, size_t out_bytes) {
    PGT_USE_DEVICE(ctx);
    if (int rc = tab_check(ctx, tab, out, "pgt_het_reduce_tab")) return rc;
    if (int rc = room_check(ctx, "pgt_het_reduce_tab", tab->n_win, sizeof(pgt_het_row), out_bytes)) return rc;
    if (n && (!pos || !g)) return ctx_fail(ctx, PGT_EARG, "pgt_het_reduce_tab: NULL argument");
    if (tab->n_win == 0) return PGT_OK;
    ApiTrace trace("pgt_het_reduce_tab");
    DevBuf dpos, dg;
    if (!cols_on_device) {
        UploadJob jobs[] = {{&dpos, pos, n * sizeof(uint32_t), "upload pos"}, {&dg, g, n * sizeof(int8_t), "upload genotypes"}};
        if (int rc = upload_columns(ctx, jobs, 2, trace)) return rc;
        pos = static_cast<uint32_t *>(dpos.p); g = static_cast<int8_t *>(dg.p);
    }
    return reduce_tab<pgt_het_row>(ctx, "het kernels", PGT_STAT_HET, n, tab, out, nullptr, trace,
        [&](pgt_het_row *dr, void *tree, size_t tb, pgt_dxy_total *) {
            return pgt_het_reduce_dev(ctx, pos, g, n, tab->d_win, tab->n_win, dr, tab->n_win * sizeof(pgt_het_row), tree, tb, nullptr);
        });
}

int pgt_dxy_reduce_tab(pgt_ctx *ctx, const uint32_t *pos, const double *p1, const double *p2, const int32_t *n1, const int32_t *n2,
                       uint64_t n, int minind, int cols_on_device, const pgt_wintab *tab, pgt_dxy_row *out, size_t out_bytes,
                       pgt_dxy_total *tot) {
    PGT_USE_DEVICE(ctx);
    if (int rc = tab_check(ctx, tab, out, "pgt_dxy_reduce_tab")) return rc;
    if (int rc = room_check(ctx, "pgt_dxy_reduce_tab", tab->n_win, sizeof(pgt_dxy_row), out_bytes)) return rc;
    if (n && (!pos || !p1 || !p2 || !n1 || !n2)) return ctx_fail(ctx, PGT_EARG, "pgt_dxy_reduce_tab: NULL argument");
    ApiTrace trace("pgt_dxy_reduce_tab");
    DevBuf dpos, d1, d2, dn1, dn2;
    if (!cols_on_device) {
        UploadJob jobs[] = {{&dpos, pos, n * sizeof(uint32_t), "upload pos"}, {&d1, p1, n * sizeof(double), "upload p1"},
                            {&d2, p2, n * sizeof(double), "upload p2"}, {&dn1, n1, n * sizeof(int32_t), "upload n1"},
                            {&dn2, n2, n * sizeof(int32_t), "upload n2"}};
        if (int rc = upload_columns(ctx, jobs, 5, trace)) return rc;
        pos = static_cast<uint32_t *>(dpos.p); p1 = static_cast<double *>(d1.p); p2 = static_cast<double *>(d2.p);
        n1 = static_cast<int32_t *>(dn1.p); n2 = static_cast<int32_t *>(dn2.p);
    }
    return reduce_tab<pgt_dxy_row>(ctx, "dxy kernels", PGT_STAT_DXY, n, tab, out, tot, trace,
        [&](pgt_dxy_row *dr, void *tree, size_t tb, pgt_dxy_total *dtot) {
            return pgt_dxy_reduce_dev(ctx, pos, p1, p2, n1, n2, n, minind, tab->d_win, tab->n_win, dr, tab->n_win * sizeof(pgt_dxy_row), dtot, tree,
                                      tb, nullptr);
        });
}

/* copy of a device column of an ingest object to the host (dxyWindow's site synchronisation and bp-window
 * table work on host positions) */
int pgt_ingest_download(pgt_ctx *ctx, const pgt_ingest *ing, int token, void *host_dst, size_t bytes) {
    PGT_USE_DEVICE(ctx);
    void *src = pgt_ingest_column(ing, token);
    if (bytes == 0) return PGT_OK;
    if (!src || !host_dst) return ctx_fail(ctx, PGT_EARG, "pgt_ingest_download: no such column");
    if (bytes > ingest_column_bytes(ing, token)) return ctx_fail(ctx, PGT_EARG, "pgt_ingest_download: more bytes than the column holds");
    return hip_check(ctx, hipMemcpy(host_dst, src, bytes, hipMemcpyDeviceToHost), "pgt_ingest_download");
}

}  // extern "C"
