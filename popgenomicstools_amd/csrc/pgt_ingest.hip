// pgt_ingest.hip — device-side ingest of the tools' text inputs.
//
// What is replaced: the per-line text parse of the reference's streaming loops
//   fstWindow.cpp:123-146 (`chr pos a b`)   hetWindow.cpp:121-144 (`chr pos genotype`)
//   dxyWindow.cpp:141-153,282-292,399-403 (`chr pos major minor ref freq nInd`)
// which is ~94 % of the reference's wall time and, after round 1's 32-thread host parser, still 0.3 s of
// this host's 0.4 s at 10^8 lines.  Here the raw text crosses PCIe once (it is about as large as the
// columns it becomes) and is parsed where the columns are needed:
//   1. count_lines_kernel   line-starting newlines per 4-KiB block of text (16 bytes per lane);
//   2. scan_blocks_kernel   exclusive prefix over the block counts -> first row of every block;
//   3. parse_lines_kernel   the lane that owns a newline parses the line behind it: chromosome token
//                           (compared with the previous line's: run starts), then the format's tokens.
// Exactness: a decimal with at most 15 significant digits and a power of ten within 10^+-22 is ONE
// correctly rounded f64 multiplication or division of two exactly representable numbers (Clinger's fast
// path) — the same bits as the strtod behind the reference's `ss >> double`.  The kernel converts only
// such tokens (and plain digit strings for the integers).  Every line that holds anything else — longer
// mantissas, big exponents, inf/nan, signs in odd places, missing tokens — is put on a "slow line" list and
// parsed on the host by the same std::from_chars calls the host parser uses; the host patches the
// columns and decides whether the line is an error.  Semantics kept from the host parser: blank-only
// line = end of data (fstWindow.cpp:125), first unparsable line before that = error with its line
// number, a last line without newline is accepted, extra columns are ignored, \r counts as blank.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <charconv>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <climits>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "pgt_internal.h"

namespace pgt {
namespace {

constexpr int kLaneBytes = 16;
constexpr int kBlockThreads = 256;
constexpr uint64_t kBlockBytes = (uint64_t)kLaneBytes * kBlockThreads;  // 4 KiB of text per workgroup
constexpr int kMaxTokens = 12;  // xpehhWindow: id pos + 7 numeric fields, the score in the last
constexpr uint32_t kListCap = 1u << 20;  // slow lines / run starts the device may report before the host takes over
constexpr uint64_t kMaxLine = 1u << 16;  // a lane never walks further than this through one line: longer lines (not
                                         // the tools' tables) make the ingest refuse the input (PGT_EDOMAIN), the
                                         // host parser takes it

struct Spec {
    uint8_t tok[kMaxTokens];
    int n;
    int chr_prefix;  // tokens[0] == PGT_TOK_CHR_PREFIX: the chromosome is the first token up to its first '_'
};
struct ListEntry {  // a run start (name token) or a slow line (line start)
    uint64_t row, off;
    uint32_t len, pad_;
};
struct Columns {
    void *col[kMaxTokens];  // per token: device column, or NULL for CHR / SKIP
};
struct Counters {
    unsigned long long first_empty;  // smallest row of a blank-only line
    uint32_t n_runs, n_slow;         // entries appended (may exceed kListCap: overflow)
    uint32_t refuse, pad_;           // set when a line is too long for a lane to walk
};

__device__ __forceinline__ bool is_blank(char c) { return c == ' ' || c == '\t' || c == '\r'; }
__device__ __forceinline__ bool is_sep(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n'; }

// bit k set: byte k of the lane's 16 is a newline that starts a line (i.e. is not the last byte of the text)
__device__ __forceinline__ uint32_t newline_mask(const char *txt, uint64_t base, uint64_t len) {
    if (base >= len) return 0;
    uint32_t m = 0;
    if (base + kLaneBytes <= len) {
        const uint4 w = *reinterpret_cast<const uint4 *>(txt + base);  // the buffer is 16-byte aligned and padded
        const uint32_t v[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t x = v[q] ^ 0x0A0A0A0Au;  // newline bytes become 0
            const uint32_t z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);  // 0x80 per zero byte
            m |= (((z >> 7) & 1u) | ((z >> 14) & 2u) | ((z >> 21) & 4u) | ((z >> 28) & 8u)) << (4 * q);
        }
    } else {
        for (int k = 0; base + k < len; ++k) m |= (uint32_t)(txt[base + k] == '\n') << k;
    }
    // a newline that is the very last byte starts no line
    if (len - 1 >= base && len - 1 < base + kLaneBytes) m &= ~(1u << (uint32_t)(len - 1 - base));
    return m;
}

__global__ __launch_bounds__(kBlockThreads) void count_lines_kernel(const char *txt, uint64_t len, uint32_t *block_count) {
    __shared__ uint32_t part[kBlockThreads / kWave];
    const uint64_t base = ((uint64_t)blockIdx.x * kBlockThreads + threadIdx.x) * kLaneBytes;
    uint32_t c = (uint32_t)__popc(newline_mask(txt, base, len));
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) c += (uint32_t)__shfl_xor((int)c, d, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_count[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// one workgroup: exclusive prefix of n counts (n = text bytes / 4096: < 10^7 even for 32 GB of text)
__global__ __launch_bounds__(1024) void scan_blocks_kernel(const uint32_t *count, uint64_t *first_row, uint64_t n, uint64_t *total) {
    __shared__ uint64_t part[1024];
    const uint64_t per = (n + 1023) / 1024, lo = per * threadIdx.x, hi = lo + per < n ? lo + per : n;
    uint64_t s = 0;
    for (uint64_t i = lo; i < hi; ++i) s += count[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t run = 0;
        for (int t = 0; t < 1024; ++t) { const uint64_t v = part[t]; part[t] = run; run += v; }
        *total = run;
    }
    __syncthreads();
    uint64_t run = part[threadIdx.x];
    for (uint64_t i = lo; i < hi; ++i) { first_row[i] = run; run += count[i]; }
}

struct Token { uint64_t b, e; };
// `limit`: the walk stops there even without a separator (callers check p against it: line too long)
__device__ __forceinline__ Token next_token(const char *txt, uint64_t &p, uint64_t len, uint64_t limit) {
    if (limit > len) limit = len;
    while (p < limit && is_blank(txt[p])) ++p;
    Token t{p, p};
    while (p < limit && !is_sep(txt[p])) ++p;
    t.e = p;
    return t;
}

// digits only (optional leading '+'), value in u64 without overflow (at most 19 digits): else not plain
__device__ __forceinline__ bool plain_u64(const char *txt, Token t, uint64_t &v) {
    uint64_t p = t.b;
    if (p < t.e && txt[p] == '+') ++p;
    if (p == t.e || t.e - p > 19) return false;
    uint64_t x = 0;
    for (; p < t.e; ++p) {
        const unsigned d = (unsigned)(txt[p] - '0');
        if (d > 9u) return false;
        x = x * 10 + d;
    }
    v = x;
    return true;
}
__device__ __forceinline__ bool plain_i64(const char *txt, Token t, long long &v) {
    uint64_t p = t.b;
    bool neg = false;
    if (p < t.e && txt[p] == '-') { neg = true; ++p; }
    else if (p < t.e && txt[p] == '+') ++p;  // "+-1" falls out below: '-' is no digit -> slow line
    if (p == t.e || t.e - p > 18) return false;
    long long x = 0;
    for (; p < t.e; ++p) {
        const unsigned d = (unsigned)(txt[p] - '0');
        if (d > 9u) return false;
        x = x * 10 + (long long)d;
    }
    v = neg ? -x : x;
    return true;
}
__constant__ double kPow10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                  1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
// [+-]digits[.digits][(e|E)[+-]digits] with at most 15 significant digits and a net power of ten in [-22, 22]
__device__ __forceinline__ bool plain_f64(const char *txt, Token t, double &v) {
    uint64_t p = t.b;
    bool neg = false;
    if (p < t.e && txt[p] == '-') { neg = true; ++p; }
    else if (p < t.e && txt[p] == '+') { ++p; if (p < t.e && (txt[p] == '-' || txt[p] == '+')) return false; }
    uint64_t m = 0;
    int sig = 0, frac = 0;
    bool any = false;
    for (; p < t.e && (unsigned)(txt[p] - '0') <= 9u; ++p) {
        any = true;
        if (sig >= 15) return false;
        m = m * 10 + (unsigned)(txt[p] - '0');
        sig += m != 0;
    }
    if (p < t.e && txt[p] == '.') {
        for (++p; p < t.e && (unsigned)(txt[p] - '0') <= 9u; ++p) {
            any = true;
            if (sig >= 15) return false;
            m = m * 10 + (unsigned)(txt[p] - '0');
            sig += m != 0;
            ++frac;
        }
    }
    if (!any) return false;
    int e10 = 0;
    if (p < t.e && (txt[p] == 'e' || txt[p] == 'E')) {
        ++p;
        bool eneg = false;
        if (p < t.e && (txt[p] == '-' || txt[p] == '+')) eneg = txt[p++] == '-';
        if (p == t.e || (unsigned)(txt[p] - '0') > 9u) return false;
        for (; p < t.e && (unsigned)(txt[p] - '0') <= 9u; ++p) {
            e10 = e10 * 10 + (txt[p] - '0');
            if (e10 > 9999) return false;
        }
        if (eneg) e10 = -e10;
    }
    if (p != t.e) return false;
    e10 -= frac;
    if (e10 < -22 || e10 > 22) return false;
    const double x = e10 < 0 ? __ddiv_rn((double)m, kPow10[-e10]) : __dmul_rn((double)m, kPow10[e10]);
    v = neg ? -x : x;
    return true;
}

__device__ __forceinline__ void append(ListEntry *list, uint32_t *n, uint64_t row, uint64_t off, uint32_t len) {
    const uint32_t k = atomicAdd(n, 1u);
    if (k < kListCap) list[k] = ListEntry{row, off, len, 0u};
}

// selscan locus ids `chr_position`: the chromosome is the id up to its first '_' (extractChr, ihsWindow.cpp:80-92)
__device__ __forceinline__ Token chr_prefix(const char *txt, Token t) {
    for (uint64_t q = t.b; q < t.e; ++q)
        if (txt[q] == '_') return Token{t.b, q};
    return t;
}

__device__ void parse_line(const char *txt, uint64_t len, uint64_t s, uint64_t row, const Spec &spec, const Columns &cols,
                           Counters *cnt, ListEntry *runs, ListEntry *slow) {
    uint64_t p = s;
    const uint64_t limit = s + kMaxLine;
    Token chr = next_token(txt, p, len, limit);
    if (p >= limit && limit < len) {  // no end of line (or of token) within kMaxLine bytes: not for the device path
        atomicExch(&cnt->refuse, 1u);
        return;
    }
    if (chr.b == chr.e) {  // blank-only line: end of data (fstWindow.cpp:125)
        atomicMin(&cnt->first_empty, (unsigned long long)row);
        return;
    }
    if (spec.chr_prefix) chr = chr_prefix(txt, chr);  // may be empty ("_123"): an empty name is a name like any other
    // run start?  compare the chromosome token with the previous line's
    bool new_run = row == 0;
    if (!new_run) {
        uint64_t q = s - 1;  // txt[s-1] is the newline that ends the previous line
        const uint64_t stop = s > kMaxLine ? s - kMaxLine : 0;
        while (q > stop && txt[q - 1] != '\n') --q;
        if (q == stop && stop > 0) {  // the previous line is longer than kMaxLine
            atomicExch(&cnt->refuse, 1u);
            return;
        }
        uint64_t pp = q;
        Token prev = next_token(txt, pp, s - 1, s - 1);
        if (spec.chr_prefix) prev = chr_prefix(txt, prev);
        new_run = prev.e - prev.b != chr.e - chr.b;
        for (uint64_t k = 0; !new_run && k < chr.e - chr.b; ++k) new_run = txt[prev.b + k] != txt[chr.b + k];
    }
    if (new_run) append(runs, &cnt->n_runs, row, chr.b, (uint32_t)(chr.e - chr.b));
    bool ok = true;
    for (int k = 1; k < spec.n && ok; ++k) {  // token 0 is the chromosome
        const Token t = next_token(txt, p, len, limit);
        if (p >= limit && limit < len) {
            atomicExch(&cnt->refuse, 1u);
            return;
        }
        switch (spec.tok[k]) {
            case PGT_TOK_SKIP: ok = t.e > t.b; break;
            case PGT_TOK_U32: {
                uint64_t v;
                ok = plain_u64(txt, t, v) && v <= 0xFFFFFFFFull;
                if (ok) static_cast<uint32_t *>(cols.col[k])[row] = (uint32_t)v;
                break;
            }
            case PGT_TOK_F64:
            case PGT_TOK_FREQ: {
                double v;
                ok = plain_f64(txt, t, v) && (spec.tok[k] == PGT_TOK_F64 || (v >= 0.0 && v <= 1.0));
                if (ok) static_cast<double *>(cols.col[k])[row] = v;
                break;
            }
            case PGT_TOK_I8: {
                long long v;
                ok = plain_i64(txt, t, v);
                if (ok) static_cast<int8_t *>(cols.col[k])[row] = (int8_t)(v < -128 ? -128 : (v > 127 ? 127 : v));
                break;
            }
            case PGT_TOK_I32: {
                long long v;
                ok = plain_i64(txt, t, v);
                if (ok) static_cast<int32_t *>(cols.col[k])[row] = (int32_t)(v < INT_MIN ? INT_MIN : (v > INT_MAX ? INT_MAX : v));
                break;
            }
            default: ok = false;
        }
    }
    if (!ok) append(slow, &cnt->n_slow, row, s, 0u);  // the host parses this line with from_chars and decides
}

__global__ __launch_bounds__(kBlockThreads) void parse_lines_kernel(const char *txt, uint64_t len, const uint64_t *first_row, Spec spec,
                                                                     Columns cols, Counters *cnt, ListEntry *runs, ListEntry *slow) {
    __shared__ uint32_t part[kBlockThreads / kWave];
    const int lane = threadIdx.x & (kWave - 1), wib = threadIdx.x >> 6;
    const uint64_t base = ((uint64_t)blockIdx.x * kBlockThreads + threadIdx.x) * kLaneBytes;
    const uint32_t mask = newline_mask(txt, base, len);
    const uint32_t c = (uint32_t)__popc(mask);
    uint32_t incl = c;  // inclusive scan over the wave, then over the 4 waves
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)incl, d, kWave);
        if (lane >= d) incl += t;
    }
    if (lane == kWave - 1) part[wib] = incl;
    __syncthreads();
    uint32_t before = incl - c;
    for (int w = 0; w < wib; ++w) before += part[w];
    // rows: the line starting at byte 0 is row 0; the line behind the k-th line-starting newline is row k + 1
    uint64_t row = first_row[blockIdx.x] + before + 1;
    if (blockIdx.x == 0 && threadIdx.x == 0 && len > 0) parse_line(txt, len, 0, 0, spec, cols, cnt, runs, slow);
    for (uint32_t m = mask; m != 0; m &= m - 1, ++row) {
        const uint64_t s = base + (uint64_t)(__ffs((int)m) - 1) + 1;
        parse_line(txt, len, s, row, spec, cols, cnt, runs, slow);
    }
}

template <class T>
__global__ void scatter_kernel(T *col, const uint64_t *rows, const T *vals, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) col[rows[i]] = vals[i];
}

size_t elem_bytes(uint8_t tok) {
    switch (tok) {
        case PGT_TOK_U32: case PGT_TOK_I32: return 4;
        case PGT_TOK_F64: case PGT_TOK_FREQ: return 8;
        case PGT_TOK_I8: return 1;
        default: return 0;
    }
}

// ---- host side of a slow line: exactly the conversions of the host parser (host/host_common.h) ----
struct HostCursor {
    const char *p, *end;
    std::pair<const char *, const char *> token() {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\r')) ++p;
        const char *b = p;
        while (p < end && *p != ' ' && *p != '\t' && *p != '\r' && *p != '\n') ++p;
        return {b, p};
    }
};
bool host_u32(std::pair<const char *, const char *> t, uint32_t &v) {
    if (t.first == t.second) return false;
    const char *b = t.first;
    if (*b == '+') ++b;
    unsigned long long x = 0;
    auto r = std::from_chars(b, t.second, x);
    if (r.ec != std::errc() || r.ptr != t.second || x > 0xFFFFFFFFull) return false;
    v = (uint32_t)x;
    return true;
}
bool host_i64(std::pair<const char *, const char *> t, long long &v) {
    if (t.first == t.second) return false;
    const char *b = t.first;
    if (*b == '+') ++b;
    auto r = std::from_chars(b, t.second, v);
    return r.ec == std::errc() && r.ptr == t.second;
}
bool host_f64(std::pair<const char *, const char *> t, double &v) {
    if (t.first == t.second) return false;
    const char *b = t.first;
    if (*b == '+') ++b;
    auto r = std::from_chars(b, t.second, v);
    return r.ec == std::errc() && r.ptr == t.second;
}

}  // namespace
}  // namespace pgt

using namespace pgt;

struct pgt_ingest {
    int device = 0;
    uint64_t rows = 0;
    int64_t bad_line = -1;
    bool blank_before_end = false;  // the data ended at a blank line (possibly the last line of the text)
    int n_tokens = 0;
    uint8_t tok[kMaxTokens] = {};
    void *col[kMaxTokens] = {};   // first PARSED row of every column
    void *base[kMaxTokens] = {};  // the allocation: `front` rows of room before col[k] (pgt_ingest_text_behind), else == col[k]
    uint64_t front = 0;
    std::vector<uint64_t> run_len, name_off;
    std::vector<uint32_t> name_len;
};

namespace {
struct DevMem {
    void *p = nullptr;
    ~DevMem() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
    void *release() { void *q = p; p = nullptr; return q; }
};
int ingest_fail(std::string *err, int code, const std::string &msg) {
    if (err) *err = msg;
    set_global_error(msg);
    return code;
}
}  // namespace

namespace pgt {

int ingest_text(int device, const char *text, size_t len, const uint8_t *tokens, int n_tokens, uint64_t front, pgt_ingest **out, std::string *err) {
    if (!out || (len && !text) || !tokens || n_tokens < 2 || n_tokens > kMaxTokens || (tokens[0] != PGT_TOK_CHR && tokens[0] != PGT_TOK_CHR_PREFIX))
        return ingest_fail(err, PGT_EARG, "pgt_ingest_text: bad argument (the first token must be PGT_TOK_CHR or PGT_TOK_CHR_PREFIX, 2..12 tokens)");
    for (int k = 1; k < n_tokens; ++k)
        if (tokens[k] == PGT_TOK_CHR || tokens[k] > PGT_TOK_FREQ)
            return ingest_fail(err, PGT_EARG, "pgt_ingest_text: unknown token kind");
    auto hip = [&](hipError_t e, const char *what) {
        return e == hipSuccess ? PGT_OK : ingest_fail(err, PGT_EDEVICE, std::string("pgt_ingest_text: ") + what + ": " + hipGetErrorString(e));
    };
    // PGT_HOST_TIMING=1 (the hosts' phase timer): where the ingest spends its time, on stderr
    const bool timing = std::getenv("PGT_HOST_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        (void)hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[pgt-host]   ingest: %-20s %9.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    std::unique_ptr<pgt_ingest> ing(new pgt_ingest);
    ing->device = device;
    ing->front = front;
    ing->n_tokens = n_tokens;
    std::memcpy(ing->tok, tokens, (size_t)n_tokens);
    struct FreeCols {  // columns are handed over only on success
        pgt_ingest *g;
        ~FreeCols() { if (g) for (int k = 0; k < kMaxTokens; ++k) if (g->base[k]) { (void)hipFree(g->base[k]); g->base[k] = g->col[k] = nullptr; } }
    } free_cols{ing.get()};
    if (len == 0) {
        free_cols.g = nullptr;
        *out = ing.release();
        return PGT_OK;
    }

    // 1. the text, padded so that every lane may load its 16 bytes
    const uint64_t n_blocks = (len + kBlockBytes - 1) / kBlockBytes;
    DevMem dtxt, dcount, dfirst, dtotal, dcnt, druns, dslow;
    if (int rc = hip(dtxt.alloc(n_blocks * kBlockBytes), "alloc text")) return rc;
    lap("alloc text");
    // one hipMemcpy of the pageable mapping: 25 GB/s (3.2 GB in 129 ms).  Tried: 4 threads staging interleaved
    // 8-MiB chunks through their own pinned buffers and streams — 154 ms (the pinned allocations and the extra
    // memcpy cost more than the overlap buys); hipHostRegister of the mapping + copy: 31 ms/GB to register,
    // then 57.6 GB/s — 156 ms for the same 3.2 GB; four threads each registering, copying and unregistering
    // their own 128-MiB slices: 113-126 ms against 88 ms for the plain copy in the same session (registration
    // does not scale with threads); 2 / 4 / 8 threads each copying its slice with hipMemcpyAsync on its own stream,
    // no explicit pinning: 100 / 137-163 / 179 ms against 83-95 ms for the one plain copy.  Kept simple.
    if (int rc = hip(hipMemcpy(dtxt.p, text, len, hipMemcpyHostToDevice), "upload text")) return rc;
    lap("upload text");
    if (int rc = hip(hipMemsetAsync(static_cast<char *>(dtxt.p) + len, 0, n_blocks * kBlockBytes - len, nullptr), "pad text")) return rc;
    // 2. lines
    if (int rc = hip(dcount.alloc(n_blocks * sizeof(uint32_t)), "alloc counts")) return rc;
    if (int rc = hip(dfirst.alloc(n_blocks * sizeof(uint64_t)), "alloc offsets")) return rc;
    if (int rc = hip(dtotal.alloc(sizeof(uint64_t)), "alloc total")) return rc;
    const char *txt = static_cast<const char *>(dtxt.p);
    hipLaunchKernelGGL(count_lines_kernel, dim3((unsigned)n_blocks), dim3(kBlockThreads), 0, nullptr, txt, (uint64_t)len,
                       static_cast<uint32_t *>(dcount.p));
    hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(1024), 0, nullptr, static_cast<const uint32_t *>(dcount.p),
                       static_cast<uint64_t *>(dfirst.p), n_blocks, static_cast<uint64_t *>(dtotal.p));
    uint64_t newlines = 0;
    if (int rc = hip(hipMemcpy(&newlines, dtotal.p, sizeof newlines, hipMemcpyDeviceToHost), "line count")) return rc;
    const uint64_t n_lines = newlines + 1;
    lap("count lines");
    // 3. columns + parse
    Spec spec{};
    spec.n = n_tokens;
    spec.chr_prefix = tokens[0] == PGT_TOK_CHR_PREFIX;
    Columns cols{};
    for (int k = 0; k < n_tokens; ++k) {
        spec.tok[k] = tokens[k];
        if (const size_t eb = elem_bytes(tokens[k])) {
            if (int rc = hip(hipMalloc(&ing->base[k], (front + n_lines) * eb + 16), "alloc column")) return rc;
            ing->col[k] = static_cast<char *>(ing->base[k]) + front * eb;
            cols.col[k] = ing->col[k];
        }
    }
    if (int rc = hip(dcnt.alloc(sizeof(Counters)), "alloc counters")) return rc;
    if (int rc = hip(druns.alloc((size_t)kListCap * sizeof(ListEntry)), "alloc run list")) return rc;
    if (int rc = hip(dslow.alloc((size_t)kListCap * sizeof(ListEntry)), "alloc slow list")) return rc;
    lap("alloc columns");
    Counters c0{~0ull, 0u, 0u, 0u, 0u};
    if (int rc = hip(hipMemcpy(dcnt.p, &c0, sizeof c0, hipMemcpyHostToDevice), "init counters")) return rc;
    hipLaunchKernelGGL(parse_lines_kernel, dim3((unsigned)n_blocks), dim3(kBlockThreads), 0, nullptr, txt, (uint64_t)len,
                       static_cast<const uint64_t *>(dfirst.p), spec, cols, static_cast<Counters *>(dcnt.p),
                       static_cast<ListEntry *>(druns.p), static_cast<ListEntry *>(dslow.p));
    if (int rc = hip(hipGetLastError(), "parse kernel")) return rc;
    Counters cnt{};
    if (int rc = hip(hipMemcpy(&cnt, dcnt.p, sizeof cnt, hipMemcpyDeviceToHost), "counters")) return rc;
    lap("parse kernel");
    if (cnt.refuse || cnt.n_runs > kListCap || cnt.n_slow > kListCap)
        return ingest_fail(err, PGT_EDOMAIN, "pgt_ingest_text: a line longer than 64 KiB, or more than 2^20 chromosome runs or irregular "
                                             "lines: parse this input on the host");
    const uint64_t n_rows = std::min<uint64_t>(n_lines, cnt.first_empty);
    ing->blank_before_end = cnt.first_empty != ~0ull;  // also when it is the text's last line: a caller's next piece lies behind it
    std::vector<ListEntry> runs(cnt.n_runs), slow(cnt.n_slow);
    if (cnt.n_runs)
        if (int rc = hip(hipMemcpy(runs.data(), druns.p, runs.size() * sizeof(ListEntry), hipMemcpyDeviceToHost), "run list")) return rc;
    if (cnt.n_slow)
        if (int rc = hip(hipMemcpy(slow.data(), dslow.p, slow.size() * sizeof(ListEntry), hipMemcpyDeviceToHost), "slow list")) return rc;
    auto by_row = [](const ListEntry &a, const ListEntry &b) { return a.row < b.row; };
    std::sort(runs.begin(), runs.end(), by_row);
    std::sort(slow.begin(), slow.end(), by_row);

    // 4. slow lines: the host parser's conversions, in row order; the first failure before the end of data is the error
    std::vector<std::vector<uint64_t>> prow(n_tokens);
    std::vector<std::vector<unsigned char>> pval(n_tokens);
    for (const ListEntry &e : slow) {
        if (e.row >= n_rows) break;
        HostCursor c{text + e.off, text + len};
        c.token();  // chromosome
        bool ok = true;
        unsigned char tmp[kMaxTokens][8];
        for (int k = 1; k < n_tokens && ok; ++k) {
            const auto t = c.token();
            switch (tokens[k]) {
                case PGT_TOK_SKIP: break;  // the host parser does not look at these tokens at all
                case PGT_TOK_U32: { uint32_t v; ok = host_u32(t, v); std::memcpy(tmp[k], &v, 4); break; }
                case PGT_TOK_F64: { double v; ok = host_f64(t, v); std::memcpy(tmp[k], &v, 8); break; }
                case PGT_TOK_FREQ: { double v; ok = host_f64(t, v) && v >= 0.0 && v <= 1.0; std::memcpy(tmp[k], &v, 8); break; }
                case PGT_TOK_I8: { long long v; ok = host_i64(t, v); const int8_t w = (int8_t)std::clamp<long long>(v, -128, 127); std::memcpy(tmp[k], &w, 1); break; }
                case PGT_TOK_I32: { long long v; ok = host_i64(t, v); const int32_t w = (int32_t)std::clamp<long long>(v, INT_MIN, INT_MAX); std::memcpy(tmp[k], &w, 4); break; }
                default: ok = false;
            }
        }
        if (!ok) { ing->bad_line = (int64_t)e.row; break; }
        for (int k = 1; k < n_tokens; ++k)
            if (const size_t eb = elem_bytes(tokens[k])) {
                prow[k].push_back(e.row);
                pval[k].insert(pval[k].end(), tmp[k], tmp[k] + eb);
            }
    }
    const uint64_t keep = ing->bad_line >= 0 ? (uint64_t)ing->bad_line : n_rows;  // rows that are good
    for (int k = 1; k < n_tokens; ++k) {
        const size_t eb = elem_bytes(tokens[k]), np = prow[k].size();
        if (!eb || !np) continue;
        DevMem drow, dval;
        if (int rc = hip(drow.alloc(np * 8), "alloc patches")) return rc;
        if (int rc = hip(dval.alloc(np * eb), "alloc patches")) return rc;
        if (int rc = hip(hipMemcpy(drow.p, prow[k].data(), np * 8, hipMemcpyHostToDevice), "upload patches")) return rc;
        if (int rc = hip(hipMemcpy(dval.p, pval[k].data(), np * eb, hipMemcpyHostToDevice), "upload patches")) return rc;
        const dim3 grid((unsigned)((np + 255) / 256));
        const uint64_t *r = static_cast<const uint64_t *>(drow.p);
        if (eb == 8) hipLaunchKernelGGL(scatter_kernel<uint64_t>, grid, dim3(256), 0, nullptr, static_cast<uint64_t *>(ing->col[k]), r, static_cast<const uint64_t *>(dval.p), (uint64_t)np);
        else if (eb == 4) hipLaunchKernelGGL(scatter_kernel<uint32_t>, grid, dim3(256), 0, nullptr, static_cast<uint32_t *>(ing->col[k]), r, static_cast<const uint32_t *>(dval.p), (uint64_t)np);
        else hipLaunchKernelGGL(scatter_kernel<uint8_t>, grid, dim3(256), 0, nullptr, static_cast<uint8_t *>(ing->col[k]), r, static_cast<const uint8_t *>(dval.p), (uint64_t)np);
        if (int rc = hip(hipDeviceSynchronize(), "patch kernel")) return rc;
    }
    if (int rc = hip(hipDeviceSynchronize(), "ingest kernels")) return rc;

    lap("slow lines + patches");
    // 5. chromosome runs of the good rows
    ing->rows = keep;
    for (size_t i = 0; i < runs.size() && runs[i].row < keep; ++i) {
        const uint64_t next = i + 1 < runs.size() && runs[i + 1].row < keep ? runs[i + 1].row : keep;
        ing->run_len.push_back(next - runs[i].row);
        ing->name_off.push_back(runs[i].off);
        ing->name_len.push_back(runs[i].len);
    }
    free_cols.g = nullptr;
    *out = ing.release();
    return PGT_OK;
}

size_t ingest_column_bytes(const pgt_ingest *g, int token) {
    if (!g || token < 0 || token >= g->n_tokens || !g->col[token]) return 0;
    return (size_t)g->rows * elem_bytes(g->tok[token]);
}

}  // namespace pgt

extern "C" {

uint64_t pgt_ingest_rows(const pgt_ingest *g) { return g ? g->rows : 0; }
int64_t pgt_ingest_bad_line(const pgt_ingest *g) { return g ? g->bad_line : -1; }
int pgt_ingest_blank_before_end(const pgt_ingest *g) { return g && g->blank_before_end ? 1 : 0; }
void *pgt_ingest_column(const pgt_ingest *g, int token) {
    return g && token >= 0 && token < g->n_tokens ? g->col[token] : nullptr;
}
void *pgt_ingest_column_base(const pgt_ingest *g, int token) {
    return g && token >= 0 && token < g->n_tokens ? g->base[token] : nullptr;
}
size_t pgt_ingest_runs(const pgt_ingest *g, const uint64_t **run_len, const uint64_t **name_off, const uint32_t **name_len) {
    if (!g) return 0;
    if (run_len) *run_len = g->run_len.data();
    if (name_off) *name_off = g->name_off.data();
    if (name_len) *name_len = g->name_len.data();
    return g->run_len.size();
}
void pgt_ingest_free(pgt_ingest *g) {
    if (!g) return;
    int saved = -1;
    const bool sw = hipGetDevice(&saved) == hipSuccess && saved != g->device && hipSetDevice(g->device) == hipSuccess;
    for (auto &c : g->base)
        if (c) (void)hipFree(c);
    if (sw) (void)hipSetDevice(saved);
    delete g;
}

}  // extern "C"
