// host_common.h — what the three retained C++ hosts share: reading whitespace-separated text
// columns into structure-of-arrays buffers (in parallel), chromosome run bookkeeping, error exit.
//
// The hosts keep the reference tools' command lines and TSV (fstWindow.cpp:37-67,88;
// hetWindow.cpp:34-64,87; dxyWindow.cpp:63-139,190,429-433).  What changes is the middle: the
// reference streams line by line through a W-entry buffer and calls calcWindow per window; the
// hosts parse the whole input into SoA columns, build the window table once
// (pgt_build_windows_*) and hand both to the GPU through include/pgtwin.h.
//
// Ingest: ~94 % of the reference's wall time is libstdc++ text parsing
// (getline + stringstream + operator>>).  Here the file is mapped (or inflated, for .gz), cut into
// one chunk per thread at line boundaries; a first pass counts lines so that every column is
// allocated once, a second pass parses each chunk with std::from_chars straight into its slice —
// same values bit for bit (from_chars and the strtod behind operator>> are both correctly rounded).
// Measured (8 threads, 10^7 fst lines, 320 MB): 0.19 s, vs 4.4 s for the whole reference run.
#pragma once

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <charconv>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <memory>
#include <new>
#include <string>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "pgtwin.h"

namespace pgthost {

// The reference tools `return -1` from main on error, i.e. exit status 255.
// _exit, not exit: a background thread may still be inside HIP start-up (DeviceOpener), and running
// static destructors under it is not safe; nothing of value is buffered at this point.
[[noreturn]] inline void die(const std::string &msg) {
    std::fflush(stdout);
    std::fprintf(stderr, "%s\n", msg.c_str());
    std::fflush(stderr);
    _exit(255);
}

inline void check(int rc, const pgt_ctx *ctx) {
    if (rc != PGT_OK) die(std::string("libpgtwin: ") + pgt_last_error(ctx));
}

// ---- phase timing on stderr when PGT_HOST_TIMING is set ------------------------------------
// Milliseconds since the kernel created this process (exec, dynamic linking of the HIP runtime and static
// initialisers included): /proc/self/stat field 22 (start time in clock ticks since boot) against the
// monotonic clock, which Linux counts from boot as well (minus suspend; irrelevant here).
inline double process_age_ms() {
    FILE *f = std::fopen("/proc/self/stat", "r");
    if (!f) return -1.0;
    char buf[1024];
    const size_t got = std::fread(buf, 1, sizeof buf - 1, f);
    std::fclose(f);
    buf[got] = 0;
    const char *p = std::strrchr(buf, ')');  // the command name may contain spaces
    if (!p) return -1.0;
    unsigned long long start = 0;
    int field = 2;
    for (++p; *p && field < 22; ++p)
        if (*p == ' ') {
            ++field;
            if (field == 22) start = std::strtoull(p + 1, nullptr, 10);
        }
    if (!start) return -1.0;
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (ts.tv_sec + ts.tv_nsec * 1e-9 - (double)start / (double)sysconf(_SC_CLK_TCK)) * 1e3;
}

struct PhaseTimer {
    bool on = std::getenv("PGT_HOST_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    PhaseTimer() {
        if (on) std::fprintf(stderr, "[pgt-host] %-14s %9.3f ms\n", "process start", process_age_ms());
    }
    void lap(const char *what) {
        if (!on) return;
        auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[pgt-host] %-14s %9.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

// End of a successful run: everything is printed, so skip the teardown (unmapping gigabytes of input
// text and columns, destroying the HIP runtime: 0.1-0.3 s at 10^8 lines that no user is waiting for).
[[noreturn]] inline void finish(PhaseTimer &timer) {
    // a failed write (ENOSPC, EPIPE, closed pipe) must not look like success: the TSV would be truncated with exit status 0
    if (std::fflush(stdout) != 0 || std::ferror(stdout)) {
        std::fprintf(stderr, "Error writing the output: %s\n", std::strerror(errno));
        std::fflush(stderr);
        _exit(255);
    }
    timer.lap("print");
    if (timer.on) std::fprintf(stderr, "[pgt-host] %-14s %9.3f ms\n", "total", process_age_ms());
    std::fflush(stderr);
    _exit(0);
}

// Uninitialised host memory for columns, row arrays and inflated text.  From 8 MiB on: an anonymous mapping aligned to
// 2 MiB with MADV_HUGEPAGE (this pool runs transparent huge pages in `madvise` mode) — a 10^8-line table is 2 GB of
// columns, i.e. 500 000 page faults while the parser threads fill them and as many pages to give back when the process
// ends, or 1 000 of each with huge pages: fstWindow on 10^8 lines with the host parser 0.81 -> 0.52 s of wall time, the
// parse phase 329 -> 242 ms (profiles/r03/host_huge_pages_ab.txt, alternating runs on one box).
struct HostBuf {
    void *p = nullptr;
    size_t mapped = 0;  // bytes of the mapping (0: malloc)
    HostBuf() = default;
    HostBuf(const HostBuf &) = delete;
    HostBuf &operator=(const HostBuf &) = delete;
    ~HostBuf() { release(); }
    void release() {
        if (!p) return;
        if (mapped) munmap(p, mapped); else std::free(p);
        p = nullptr;
        mapped = 0;
    }
    void alloc(size_t bytes) {
        release();
        const size_t huge = (size_t)2 << 20;
        if (bytes >= ((size_t)8 << 20)) {
            const size_t len = (bytes + huge - 1) / huge * huge;
            void *q = mmap(nullptr, len + huge, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (q != MAP_FAILED) {
                char *lo = static_cast<char *>(q), *at = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(lo) + huge - 1) / huge * huge);
                if (at > lo) munmap(lo, (size_t)(at - lo));
                if (at + len < lo + len + huge) munmap(at + len, (size_t)(lo + len + huge - (at + len)));
#ifdef MADV_HUGEPAGE
                madvise(at, len, MADV_HUGEPAGE);
#endif
                p = at;
                mapped = len;
                return;
            }
        }
        p = std::malloc(bytes ? bytes : 1);
        if (!p) throw std::bad_alloc();
    }
};

// ---- input text: mmap for plain files, zlib for gzip (dxyWindow.cpp:82-83,256-278) ----------
class Text {
  public:
    Text() = default;
    Text(const Text &) = delete;
    Text &operator=(const Text &) = delete;
    ~Text() {
        if (map_) munmap(map_, map_len_);
    }
    bool open(const char *path) {
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {  // pipes etc.: read through zlib
            ::close(fd);
            return slurp(path);
        }
        unsigned char magic[2] = {0, 0};
        const ssize_t got = pread(fd, magic, 2, 0);
        if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
            const int r = st.st_size >= 28 ? inflate_bgzf(fd, (size_t)st.st_size) : 0;
            unsigned char tail[4] = {0, 0, 0, 0};  // ISIZE of the (last) member: the text's size when there is one member < 4 GiB
            if (!r && st.st_size >= 18 && pread(fd, tail, 4, st.st_size - 4) != 4) std::memset(tail, 0, 4);
            ::close(fd);
            const size_t isize = (size_t)tail[0] | (size_t)tail[1] << 8 | (size_t)tail[2] << 16 | (size_t)tail[3] << 24;
            return r ? r > 0 : slurp(path, std::max(isize, (size_t)st.st_size));
        }
        if (st.st_size == 0) {
            ::close(fd);
            b_ = e_ = "";
            return true;
        }
        map_len_ = (size_t)st.st_size;
        map_ = mmap(nullptr, map_len_, PROT_READ, MAP_PRIVATE, fd, 0);
        ::close(fd);
        if (map_ == MAP_FAILED) {
            map_ = nullptr;
            return slurp(path);
        }
        madvise(map_, map_len_, MADV_SEQUENTIAL | MADV_WILLNEED);
        b_ = static_cast<const char *>(map_);
        e_ = b_ + map_len_;
        prefault();
        return true;
    }
    const char *begin() const { return b_; }
    const char *end() const { return e_; }
    size_t size() const { return (size_t)(e_ - b_); }

  private:
    // Large inputs: map the page-cache pages into this process now, on all threads (MADV_POPULATE_READ, or
    // one read per page where the kernel lacks it), instead of one fault at a time inside whoever reads the
    // text first — the parser threads, or the single hipMemcpy that feeds the device parser.
    void prefault() {
        if (map_len_ < (64u << 20)) return;
        unsigned T = std::thread::hardware_concurrency();
        T = T ? (T > 32 ? 32 : T) : 1;
        if (const char *e = std::getenv("PGT_HOST_THREADS")) T = (unsigned)std::max(1, std::atoi(e));
        const size_t page = 4096, per = ((map_len_ / T) + page - 1) / page * page;
        std::vector<std::thread> th;
        char *base = static_cast<char *>(map_);
        for (unsigned t = 0; t < T; ++t) {
            const size_t lo = per * t, hi = std::min(map_len_, lo + per);
            if (lo >= hi) break;
            th.emplace_back([=] {
#ifdef MADV_POPULATE_READ
                if (madvise(base + lo, hi - lo, MADV_POPULATE_READ) == 0) return;
#endif
                volatile char sink = 0;
                for (size_t i = lo; i < hi; i += page) sink = sink + base[i];
            });
        }
        for (auto &x : th) x.join();
    }
    // ANGSD writes its .mafs.gz through bgzf: a series of independent gzip members of at most 64 KiB of text,
    // each announcing its own compressed size in a "BC" extra field.  Such a file is inflated block-parallel
    // (zlib's raw inflate per member, CRC-32 and length checked as gzread would); the text is the same bytes
    // the reference's gzip_decompressor (dxyWindow.cpp:256-278) or gzread produce from the same file.
    // -> 1 done, -1 a member is damaged (the input cannot be read), 0 not bgzf from the first byte to the
    // last: nothing was done, the caller reads the file through gzread (any gzip stream, trailing garbage
    // ignored as zlib does).
    int inflate_bgzf(int fd, size_t zlen) {
        struct Member { size_t data, data_len, out; uint32_t crc, isize; };
        void *zmap = mmap(nullptr, zlen, PROT_READ, MAP_PRIVATE, fd, 0);
        if (zmap == MAP_FAILED) return 0;
        const unsigned char *z = static_cast<const unsigned char *>(zmap);
        auto u16 = [&](size_t o) { return (size_t)z[o] | (size_t)z[o + 1] << 8; };
        auto u32 = [&](size_t o) { return (uint32_t)z[o] | (uint32_t)z[o + 1] << 8 | (uint32_t)z[o + 2] << 16 | (uint32_t)z[o + 3] << 24; };
        std::vector<Member> mem;
        size_t off = 0, total = 0;
        bool bgzf = true;
        while (bgzf && off < zlen) {
            bgzf = false;
            if (zlen - off < 26 || z[off] != 0x1f || z[off + 1] != 0x8b || z[off + 2] != 8 || z[off + 3] != 4) break;  // FLG = FEXTRA only
            const size_t xlen = u16(off + 10), hdr = 12 + xlen;
            if (hdr + 8 > zlen - off) break;
            size_t bsize = 0;
            for (size_t x = off + 12; x + 4 <= off + hdr;) {  // subfields: SI1 SI2 SLEN data
                const size_t slen = u16(x + 2);
                if (z[x] == 'B' && z[x + 1] == 'C' && slen == 2 && x + 6 <= off + hdr) bsize = u16(x + 4) + 1;
                x += 4 + slen;
            }
            if (bsize < hdr + 8 || bsize > zlen - off) break;
            const uint32_t isize = u32(off + bsize - 4);
            if (isize > (1u << 16)) break;
            mem.push_back(Member{off + hdr, bsize - hdr - 8, total, u32(off + bsize - 8), isize});
            total += isize;
            off += bsize;
            bgzf = true;
        }
        if (!bgzf || mem.empty()) {
            munmap(zmap, zlen);
            return 0;
        }
        try {
            big_.alloc(total ? total : 1);  // not value-initialised: the inflating threads touch the pages first
        } catch (const std::bad_alloc &) {  // members announcing more text than this machine can hold
            munmap(zmap, zlen);
            return -1;
        }
        unsigned T = std::thread::hardware_concurrency();
        T = T ? (T > 32 ? 32 : T) : 1;
        if (const char *e = std::getenv("PGT_HOST_THREADS")) T = (unsigned)std::max(1, std::atoi(e));
        T = (unsigned)std::min<size_t>(T, (mem.size() + 63) / 64);
        std::vector<char> bad(T, 0);
        std::vector<std::thread> th;
        char *out = static_cast<char *>(big_.p);
        for (unsigned t = 0; t < T; ++t)
            th.emplace_back([&, t] {
                z_stream zs{};
                if (inflateInit2(&zs, -15) != Z_OK) { bad[t] = 1; return; }
                for (size_t i = mem.size() * t / T, e = mem.size() * (t + 1) / T; i < e && !bad[t]; ++i) {
                    const Member &m = mem[i];
                    zs.next_in = const_cast<unsigned char *>(z + m.data);
                    zs.avail_in = (uInt)m.data_len;
                    zs.next_out = reinterpret_cast<unsigned char *>(out + m.out);
                    zs.avail_out = m.isize;
                    const int rc = inflate(&zs, Z_FINISH);
                    if (rc != Z_STREAM_END || zs.avail_out != 0 || zs.avail_in != 0 ||
                        (uint32_t)crc32(crc32(0L, Z_NULL, 0), reinterpret_cast<const unsigned char *>(out + m.out), m.isize) != m.crc)
                        bad[t] = 1;
                    inflateReset(&zs);
                }
                inflateEnd(&zs);
            });
        for (auto &x : th) x.join();
        munmap(zmap, zlen);
        if (std::find(bad.begin(), bad.end(), 1) != bad.end()) return -1;
        b_ = static_cast<const char *>(big_.p);
        e_ = b_ + total;
        return 1;
    }
    bool slurp(const char *path, size_t expect = 0) {
        gzFile f = gzopen(path, "rb");
        if (!f) return false;
        if (expect) own_.reserve(expect + 1);  // a hint: no regrowth copies when it is right
        gzbuffer(f, 1 << 20);
        std::vector<char> buf(1 << 22);
        int n;
        while ((n = gzread(f, buf.data(), (unsigned)buf.size())) > 0) own_.append(buf.data(), (size_t)n);
        int zerr = Z_OK;
        (void)gzerror(f, &zerr);  // a file cut inside a member reads as a short stream: gzread returns 0, the error says so
        const bool ok = n == 0 && (zerr == Z_OK || zerr == Z_STREAM_END);
        gzclose(f);
        b_ = own_.data();
        e_ = b_ + own_.size();
        return ok;
    }
    void *map_ = nullptr;
    size_t map_len_ = 0;
    std::string own_;
    HostBuf big_;  // text inflated from a bgzf file
    const char *b_ = "", *e_ = b_;
};

// ---- tokenising ------------------------------------------------------------------------------
using Tok = std::pair<const char *, const char *>;

struct Cursor {
    const char *p, *end;
    bool at_eol() const { return p >= end || *p == '\n'; }
    void skip_blank() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\r')) ++p; }
    Tok token() {  // next whitespace-delimited token of the current line ("" at end of line)
        skip_blank();
        const char *b = p;
        while (p < end && *p != ' ' && *p != '\t' && *p != '\r' && *p != '\n') ++p;
        return {b, p};
    }
    void next_line() {
        const void *nl = p < end ? std::memchr(p, '\n', (size_t)(end - p)) : nullptr;
        p = nl ? static_cast<const char *>(nl) + 1 : end;
    }
};

inline bool to_u32(Tok t, uint32_t &v) {
    if (t.first == t.second) return false;
    const char *b = t.first;
    if (*b == '+') ++b;
    unsigned long long x = 0;
    auto r = std::from_chars(b, t.second, x);
    if (r.ec != std::errc() || r.ptr != t.second || x > 0xFFFFFFFFull) return false;
    v = (uint32_t)x;
    return true;
}

inline bool to_i64(Tok t, long long &v) {
    if (t.first == t.second) return false;
    const char *b = t.first;
    if (*b == '+') ++b;
    auto r = std::from_chars(b, t.second, v);
    return r.ec == std::errc() && r.ptr == t.second;
}

// Correctly rounded, like the strtod behind the reference's `ss >> double`.  (A hand-written Clinger fast
// path for short decimals was measured at 13.5 ns per token against 18.9 ns for from_chars: not worth a
// second conversion routine in a parity-critical spot; tests/host_parse_check.cpp checks this one
// against strtod.)
inline bool to_f64(Tok t, double &v) {
    if (t.first == t.second) return false;
    const char *b = t.first;
    if (*b == '+') ++b;
    auto r = std::from_chars(b, t.second, v);
    return r.ec == std::errc() && r.ptr == t.second;
}

// Chromosome runs: a new run starts whenever the name differs from the previous line's
// (the reference compares adjacent names only, fstWindow.cpp:132).
struct Runs {
    std::vector<std::string> name;
    std::vector<uint64_t> len;
    void add(const char *b, const char *e, uint64_t count = 1) {
        const size_t n = (size_t)(e - b);
        if (name.empty() || name.back().size() != n || std::memcmp(name.back().data(), b, n) != 0) {
            name.emplace_back(b, e);
            len.push_back(0);
        }
        len.back() += count;
    }
    void append(const Runs &o) {  // runs of the next chunk: its first run may continue our last
        for (size_t r = 0; r < o.name.size(); ++r) add(o.name[r].data(), o.name[r].data() + o.name[r].size(), o.len[r]);
    }
};

inline int host_threads() {
    if (const char *e = std::getenv("PGT_HOST_THREADS")) return std::max(1, std::atoi(e));
    const unsigned hw = std::thread::hardware_concurrency();
    return (int)std::min<unsigned>(hw ? hw : 1, 32);
}

// ---- parallel line parser -----------------------------------------------------------------------
// Uninitialised column: pages are first touched by the thread that parses into them.
template <class T>
struct Column {
    HostBuf p;
    T *view = nullptr;  // set when the column lives in a mapped cache file instead of `p`
    size_t n = 0;
    void alloc(size_t cap) { p.alloc((cap ? cap : 1) * sizeof(T)); view = nullptr; }
    void borrow(T *mapped) { p.release(); view = mapped; }
    T *data() { return view ? view : static_cast<T *>(p.p); }
    const T *data() const { return view ? view : static_cast<const T *>(p.p); }
    size_t size() const { return n; }
    T &operator[](size_t i) { return data()[i]; }
    const T &operator[](size_t i) const { return data()[i]; }
};

// Two passes over the text, both parallel over chunks cut at line boundaries:
//   1. count the lines of every chunk -> row offset of every chunk, one allocation per column;
//   2. parse every chunk straight into its slice of the final columns.
// Table must provide  void alloc(size_t rows)  and  bool parse_line(Cursor&, size_t row, Runs&)
// (consume the tokens of one non-empty line into row `row`; false if it cannot).
// Parsing stops at the first empty line, as the reference loops do (fstWindow.cpp:125,
// hetWindow.cpp:123, dxyWindow.cpp:313); a final line without '\n' is accepted (the reference
// hangs on it).  Returns the number of rows; `runs` receives the chromosome runs.
template <class Table>
size_t parse_table(const char *b, const char *e, Table &tab, Runs &runs, const char *what, const char *path,
                   size_t first_line_no, std::string *error = nullptr) {  // error: receives the message instead of exiting
    const size_t len = (size_t)(e - b);
    int T = host_threads();
    if (len < (1u << 20)) T = 1;
    std::vector<const char *> cut(T + 1, e);
    cut[0] = b;
    for (int t = 1; t < T; ++t) {
        const char *p = b + len / T * t;
        if (p < cut[t - 1]) p = cut[t - 1];
        const void *nl = p < e ? std::memchr(p, '\n', (size_t)(e - p)) : nullptr;
        cut[t] = nl ? static_cast<const char *>(nl) + 1 : e;
    }
    auto run_all = [&](auto &&fn) {
        if (T == 1) { fn(0); return; }
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back(fn, t);
        for (auto &x : th) x.join();
    };
    std::vector<size_t> lines(T, 0), off(T + 1, 0);
    run_all([&](int t) {
        size_t k = (size_t)std::count(cut[t], cut[t + 1], '\n');
        if (cut[t + 1] > cut[t] && cut[t + 1][-1] != '\n') ++k;  // last line without newline
        lines[t] = k;
    });
    for (int t = 0; t < T; ++t) off[t + 1] = off[t] + lines[t];
    tab.alloc(off[T]);

    struct Result { size_t rows = 0; bool stopped = false; const char *bad = nullptr; Runs runs; };
    std::vector<Result> res(T);
    run_all([&](int t) {
        Result &r = res[t];
        Cursor c{cut[t], cut[t + 1]};
        size_t row = off[t];
        while (c.p < c.end) {
            const char *line = c.p;
            c.skip_blank();
            if (c.at_eol()) { r.stopped = true; break; }
            if (!tab.parse_line(c, row, r.runs)) { r.bad = line; break; }
            ++row;
            c.next_line();
        }
        r.rows = row - off[t];
    });
    // rows are contiguous up to the first chunk that stopped early (then everything after is dropped)
    size_t n = 0;
    for (int t = 0; t < T; ++t) {
        if (res[t].bad) {
            const size_t line_no = first_line_no + (size_t)std::count(b, res[t].bad, '\n');
            const std::string msg = std::string(what) + " on line " + std::to_string(line_no) + " of " + path;
            if (!error) die(msg);
            *error = msg;
            return 0;
        }
        runs.append(res[t].runs);
        n = off[t] + res[t].rows;
        if (res[t].stopped) break;
    }
    return n;
}

// ---- binary column cache ------------------------------------------------------
// The reference re-parses its text input on every run (fstWindow.cpp:123-146); at 10^8 lines the parse is
// 0.3 s of this host's 0.4 s.  With PGT_COLUMN_CACHE=<directory> in the environment (the command line stays
// the reference's) the parsed structure-of-arrays columns and chromosome runs of an input file are written
// to <directory>/<key>.pgtcols and mapped straight back on later runs; key = hash of (tool tag, resolved
// path, size, mtime in ns), so an edited or replaced input is simply parsed again.  Plain regular files
// only (a pipe has no identity to key on).  Layout: "PGTCOLS1", u64 rows, u64 n_runs, u64 n_cols, per run
// {u64 len, u64 name bytes, name padded to 8}, then per column {u64 bytes, data padded to 64}.
class ColumnCache {
  public:
    struct Col { void *data; size_t elem; };  // in: the parsed column; out: pointer into the mapping
    ColumnCache(const char *tag, const char *input) {
        const char *dir = std::getenv("PGT_COLUMN_CACHE");
        if (!dir || !*dir) return;
        struct stat st;
        if (stat(input, &st) != 0 || !S_ISREG(st.st_mode)) return;
        char real[4096];
        if (!realpath(input, real)) return;
        uint64_t h = 1469598103934665603ull;  // FNV-1a
        auto mix = [&](const void *p, size_t n) {
            for (size_t i = 0; i < n; ++i) h = (h ^ static_cast<const unsigned char *>(p)[i]) * 1099511628211ull;
        };
        const uint64_t size = (uint64_t)st.st_size, mt = (uint64_t)st.st_mtim.tv_sec * 1000000000ull + (uint64_t)st.st_mtim.tv_nsec;
        mix(tag, std::strlen(tag) + 1); mix(real, std::strlen(real) + 1); mix(&size, 8); mix(&mt, 8);
        char name[32];
        std::snprintf(name, sizeof name, "%016llx", (unsigned long long)h);
        path_ = std::string(dir) + "/" + name + ".pgtcols";
    }
    ~ColumnCache() {
        if (map_) munmap(map_, map_len_);
    }
    bool enabled() const { return !path_.empty(); }
    // maps the cache file; fills rows, runs and the column pointers (cols[i].elem must match)
    bool load(size_t &rows, Runs &runs, std::vector<Col> &cols) {
        if (!enabled()) return false;
        const int fd = ::open(path_.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0 || st.st_size < 32) { ::close(fd); return false; }
        map_len_ = (size_t)st.st_size;
        map_ = mmap(nullptr, map_len_, PROT_READ, MAP_PRIVATE, fd, 0);
        ::close(fd);
        if (map_ == MAP_FAILED) { map_ = nullptr; return false; }
        const char *b = static_cast<const char *>(map_), *e = b + map_len_, *p = b;
        auto room = [&](uint64_t bytes) { return bytes <= (uint64_t)(e - p); };  // sizes are compared, never added to pointers
        auto u64 = [&](uint64_t &v) { if (!room(8)) return false; std::memcpy(&v, p, 8); p += 8; return true; };
        uint64_t n = 0, n_runs = 0, n_cols = 0;
        if (std::memcmp(p, "PGTCOLS1", 8) != 0) return false;
        p += 8;
        if (!u64(n) || !u64(n_runs) || !u64(n_cols) || n_cols != cols.size()) return false;
        Runs r;
        for (uint64_t k = 0; k < n_runs; ++k) {
            uint64_t len = 0, nb = 0;
            if (!u64(len) || !u64(nb) || nb > map_len_ || !room((nb + 7) & ~7ull)) return false;
            r.name.emplace_back(p, p + nb);
            r.len.push_back(len);
            p += (nb + 7) & ~7ull;
        }
        for (auto &c : cols) {
            uint64_t bytes = 0;
            if (!u64(bytes) || n > map_len_ || bytes != n * c.elem) return false;
            const size_t at = ((size_t)(p - b) + 63) & ~(size_t)63;
            if (at > map_len_ || bytes > map_len_ - at) return false;
            p = b + at;
            c.data = const_cast<char *>(p);
            p += bytes;
        }
        madvise(map_, map_len_, MADV_SEQUENTIAL | MADV_WILLNEED);
        rows = (size_t)n;
        runs = std::move(r);
        return true;
    }
    // written to a temporary name and renamed, so that a concurrent reader never sees half a file;
    // failures (read-only directory, disk full) are silent: the cache is an optimisation
    void store(size_t rows, const Runs &runs, const std::vector<Col> &cols) const {
        if (!enabled()) return;
        const std::string tmp = path_ + "." + std::to_string((long)getpid()) + ".tmp";
        FILE *f = std::fopen(tmp.c_str(), "wb");
        if (!f) return;
        bool ok = true;
        size_t at = 0;
        auto put = [&](const void *p, size_t n) { ok = ok && std::fwrite(p, 1, n, f) == n; at += n; };
        auto u64 = [&](uint64_t v) { put(&v, 8); };
        static const char zeros[64] = {0};
        put("PGTCOLS1", 8);
        u64(rows); u64(runs.name.size()); u64(cols.size());
        for (size_t k = 0; k < runs.name.size(); ++k) {
            u64(runs.len[k]); u64(runs.name[k].size());
            put(runs.name[k].data(), runs.name[k].size());
            put(zeros, (8 - runs.name[k].size() % 8) % 8);
        }
        for (const auto &c : cols) {
            u64((uint64_t)rows * c.elem);
            put(zeros, (64 - at % 64) % 64);
            put(c.data, rows * c.elem);
        }
        ok = std::fclose(f) == 0 && ok;
        if (!ok || std::rename(tmp.c_str(), path_.c_str()) != 0) std::remove(tmp.c_str());
    }

  private:
    std::string path_;
    void *map_ = nullptr;
    size_t map_len_ = 0;
};

// ---- number formatting of the TSV rows --------------------------------------------------------------
// The tools print with `std::cout << double` = printf("%g") (fstWindow.cpp:88).  With one row per site
// (-stepsize 1) that conversion is the slowest thing left: ~0.4 us per sprintf.  fmt_g6
// produces the same bytes: the value is scaled to six significant digits by ONE exact power of ten (error
// below 1.2e-10 of a unit in the last digit); unless the scaled value lies within 1e-6 of a rounding boundary
// — then, and for nan / inf / magnitudes outside 1e-17..1e27, it simply calls snprintf — the rounded digits
// are the ones printf finds from the exact binary value.  tests/host_parse_check.cpp compares the two on
// tens of millions of values.
inline char *put_u32(char *p, uint32_t v) {
    char tmp[10];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}
inline size_t fmt_g6(double v, char *o) {
    static const double kP10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                    1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    char *p = o;
    if (v == 0.0) {
        if (std::signbit(v)) *p++ = '-';
        *p++ = '0';
        return (size_t)(p - o);
    }
    const double a = std::fabs(v);
    if (!(a >= 1e-17 && a < 1e27)) return (size_t)std::snprintf(o, 32, "%g", v);  // also nan and inf
    int X = (int)std::floor(std::log10(a));  // decimal exponent, possibly one off next to a power of ten
    auto scale = [&](int x) { return 5 - x >= 0 ? a * kP10[5 - x] : a / kP10[x - 5]; };
    double sc = scale(X);
    if (sc < 1e5) sc = scale(--X);
    else if (sc >= 1e6) sc = scale(++X);
    if (!(sc >= 1e5 && sc < 1e6)) return (size_t)std::snprintf(o, 32, "%g", v);
    const double fl = std::floor(sc), frac = sc - fl;
    if (std::fabs(frac - 0.5) < 1e-6) return (size_t)std::snprintf(o, 32, "%g", v);  // too close to call: let printf decide
    uint32_t n = (uint32_t)fl + (frac > 0.5 ? 1u : 0u);
    if (n == 1000000u) { n = 100000u; ++X; }
    char d[6];
    for (int k = 5; k >= 0; --k) { d[k] = (char)('0' + n % 10); n /= 10; }
    int nd = 6;
    while (nd > 1 && d[nd - 1] == '0') --nd;  // %g drops trailing zeros
    if (std::signbit(v)) *p++ = '-';
    if (X < -4 || X >= 6) {  // d.ddddde+XX
        *p++ = d[0];
        if (nd > 1) { *p++ = '.'; for (int k = 1; k < nd; ++k) *p++ = d[k]; }
        *p++ = 'e';
        int ex = X;
        if (ex < 0) { *p++ = '-'; ex = -ex; } else *p++ = '+';
        if (ex < 10) *p++ = '0';
        p = put_u32(p, (uint32_t)ex);
    } else if (X >= 0) {  // ddd.ddd
        for (int k = 0; k <= X; ++k) *p++ = k < nd ? d[k] : '0';
        if (nd > X + 1) { *p++ = '.'; for (int k = X + 1; k < nd; ++k) *p++ = d[k]; }
    } else {  // 0.000ddd
        *p++ = '0'; *p++ = '.';
        for (int k = 0; k < -X - 1; ++k) *p++ = '0';
        for (int k = 0; k < nd; ++k) *p++ = d[k];
    }
    return (size_t)(p - o);
}
// one TSV row: chromosome name, then unsigned columns and one %g column at position `g_at` (0-based among the numbers)
inline size_t put_row(char *o, const std::string &chr, std::initializer_list<uint32_t> before, double g, std::initializer_list<uint32_t> after) {
    char *p = o;
    std::memcpy(p, chr.data(), chr.size());
    p += chr.size();
    for (uint32_t u : before) { *p++ = '\t'; p = put_u32(p, u); }
    *p++ = '\t';
    p += fmt_g6(g, p);
    for (uint32_t u : after) { *p++ = '\t'; p = put_u32(p, u); }
    *p++ = '\n';
    return (size_t)(p - o);
}

// ---- TSV writer -----------------------------------------------------------------------------
// With -winsize 1 -stepsize 1 style runs (dxyWindow.cpp:47) the number of rows approaches the
// number of sites and formatting becomes the bottleneck.  Rows are formatted by
// all threads into per-thread buffers, one block of rows at a time, and the buffers are written in
// row order.  fmt(i, out) must write row i (the same printf conversions the tools always used:
// %g == default std::ostream precision, fstWindow.cpp:88) and return its length, at most
// max_row_bytes.
template <class Fmt>
void write_rows(size_t n, size_t max_row_bytes, Fmt fmt, FILE *out = stdout) {
    const int T = n < 200000 ? 1 : host_threads();
    const size_t block = 1u << 20;  // rows per round
    std::vector<std::vector<char>> buf(T);
    std::vector<size_t> used(T);
    for (size_t b0 = 0; b0 < n; b0 += block) {
        const size_t b1 = std::min(n, b0 + block), per = (b1 - b0 + T - 1) / T;
        auto work = [&](int t) {
            const size_t r0 = std::min(b1, b0 + per * t), r1 = std::min(b1, r0 + per);
            if (buf[t].size() < (r1 - r0) * max_row_bytes) buf[t].resize((r1 - r0) * max_row_bytes);
            char *p = buf[t].data();
            for (size_t i = r0; i < r1; ++i) p += fmt(i, p);
            used[t] = (size_t)(p - buf[t].data());
        };
        if (T == 1) work(0);
        else {
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t) th.emplace_back(work, t);
            for (auto &x : th) x.join();
        }
        for (int t = 0; t < T; ++t)
            if (used[t] && std::fwrite(buf[t].data(), 1, used[t], out) != used[t]) die("write error on output");
    }
    std::fflush(out);
}

inline size_t longest_name(const Runs &r) {
    size_t m = 0;
    for (const auto &s : r.name) m = std::max(m, s.size());
    return m;
}

inline int device_from_env() {
    const char *d = std::getenv("PGT_DEVICE");
    return d ? std::atoi(d) : 0;
}
// PGT_DEVICES=0,1,2,...: the GPUs a run may use (one context and one host thread each); default: the one of PGT_DEVICE.
// The same ordinal may be listed twice (two contexts on one GPU: how the multi-GPU path is tested on a one-GPU box).
inline std::vector<int> devices_from_env() {
    std::vector<int> ids;
    if (const char *l = std::getenv("PGT_DEVICES")) {
        for (const char *p = l; *p;) {
            char *q = nullptr;
            const long v = std::strtol(p, &q, 10);
            if (q == p) break;
            ids.push_back((int)v);
            p = *q == ',' ? q + 1 : q;
        }
        if (ids.size() > 64) ids.resize(64);
    }
    if (ids.empty()) ids.push_back(device_from_env());
    return ids;
}

inline pgt_ctx *open_or_die() {
    pgt_ctx *ctx = pgt_open(device_from_env());
    if (!ctx) die(std::string("libpgtwin: ") + pgt_last_error(nullptr));
    return ctx;
}

// HIP start-up (50-160 ms) overlapped with parsing: the devices are opened on background threads (one per
// context) as soon as the arguments are known to be valid; get() joins.  Inputs that produce no window never
// call get() and therefore still run without a GPU, as before.
class DeviceOpener {
  public:
    explicit DeviceOpener(std::vector<int> ids = devices_from_env()) : ids_(std::move(ids)), ctx_(ids_.size(), nullptr), err_(ids_.size()) {
        for (size_t k = 0; k < ids_.size(); ++k)
            th_.emplace_back([this, k] {
                ctx_[k] = pgt_open(ids_[k]);
                if (!ctx_[k]) err_[k] = pgt_last_error(nullptr);  // thread-local in the library: copy it here
                // Will this run upload HOST columns (pgt_*_reduce with host pointers: the host parser, the column cache)?  Then
                // the pinned staging ring and the runtime's one-time set-up of its first copies (50 ... 80 ms in all,
                // profiles/r06/first_use_probe.txt) are paid HERE, beside the parse, instead of inside the first
                // reduce.  The main thread says so with plan_host_io() as soon as it knows; runs that parse on the GPU skip it.
                {
                    std::unique_lock<std::mutex> lock(m_);
                    cv_.wait(lock, [this] { return host_io_ >= 0; });
                }
                if (host_io_ == 1 && ctx_[k]) (void)pgt_prepare_host_io(ctx_[k], host_io_bytes_);  // a failure is not fatal: the reduce reports it if it needs the ring
            });
    }
    ~DeviceOpener() {
        join();
        for (pgt_ctx *c : ctx_)
            if (c) pgt_close(c);
    }
    size_t count() const { return ids_.size(); }
    // the first call decides; get() decides "no" if nobody has said anything by then.  text_bytes: the size of the input text
    // (an upper bound of the columns parsed from it: below the ring's threshold only the set-up is paid, not the ring); 0 = unknown
    void plan_host_io(bool wanted, uint64_t text_bytes = 0) {
        if (const char *e = std::getenv("PGT_PREPARE_HOST_IO"); e && std::atoi(e) == 0) wanted = false;  // A/B knob: set-up inside the first reduce, as until round 5
        {
            std::lock_guard<std::mutex> lock(m_);
            if (host_io_ < 0) {
                host_io_ = wanted ? 1 : 0;
                host_io_bytes_ = text_bytes;
            }
        }
        cv_.notify_all();
    }
    pgt_ctx *get(size_t k = 0) {
        join();
        if (!ctx_[k]) die("libpgtwin: " + err_[k]);
        return ctx_[k];
    }

  private:
    void join() {
        plan_host_io(false);
        for (auto &t : th_)
            if (t.joinable()) t.join();
    }
    std::mutex m_;
    std::condition_variable cv_;
    int host_io_ = -1;  // -1 undecided, 0 no, 1 yes
    uint64_t host_io_bytes_ = 0;
    std::vector<int> ids_;
    std::vector<pgt_ctx *> ctx_;
    std::vector<std::string> err_;
    std::vector<std::thread> th_;
};

// ---- device-side ingest (pgt_ingest_text) ---------------------------------------------------------
// From 2 GiB of text on, the GPU parses (the tail of) the table: its text crosses PCIe once, its columns never visit the
// host.  Below that the host parser wins or ties: it runs beside HIP start-up (70-260 ms on this pool), which the device path
// has to wait for, it writes its columns into huge pages (round 3), and small inputs keep working without a GPU.  Measured,
// fstWindow end to end, alternating runs, quartile of 12 (profiles/r03/hybrid_ingest_ab.txt): 1.6 GB of text 0.17 s with the
// host parser against 0.24 with the device parser; 3.3 GB 0.37 against 0.32 (0.31 with the head on the host, ingest_hybrid
// below); 6.7 GB 1.00 against 0.52 (0.54).  (Until round 3 the switch was at 512 MiB: the host parser's columns then lived in
// 4-KiB pages and cost 0.2 s more at 3.3 GB.)  PGT_GPU_INGEST=0 forces the host parser, =1 the device one.  With
// PGT_COLUMN_CACHE the host path is taken (the cache holds host columns).
inline bool gpu_ingest_wanted(size_t text_bytes) {
    if (const char *c = std::getenv("PGT_COLUMN_CACHE"); c && *c) return false;
    if (const char *e = std::getenv("PGT_GPU_INGEST")) return std::atoi(e) != 0;
    return text_bytes >= ((size_t)2 << 30);
}
struct DeviceTable {
    pgt_ingest *ing = nullptr;
    size_t n = 0;
    bool from_base = false;  // the table starts in the room in front of the parsed rows (ingest_hybrid: the head parsed on the host)
    DeviceTable() = default;
    DeviceTable(const DeviceTable &) = delete;
    DeviceTable &operator=(const DeviceTable &) = delete;
    ~DeviceTable() { if (ing) pgt_ingest_free(ing); }
    template <class T> T *col(int token) const { return static_cast<T *>(from_base ? pgt_ingest_column_base(ing, token) : pgt_ingest_column(ing, token)); }
};
// true: `tab` and `runs` hold the parsed table (errors in the text die here with the host parser's message);
// false: the input has too many irregular lines for the device path — parse it on the host
// error: receives the message about a bad line instead of the exit (two files parsed side by side: the caller reports
// the first file's problem first, whichever thread met its problem first)
// rows_in_front: room for that many rows before the parsed ones in every column (pgt_ingest_text_behind)
inline bool ingest_on_device(pgt_ctx *ctx, const char *b, const char *e, const uint8_t *spec, int n_tokens, const char *what,
                             const char *path, size_t first_line_no, DeviceTable &tab, Runs &runs, std::string *error = nullptr,
                             uint64_t rows_in_front = 0) {
    const int rc = rows_in_front ? pgt_ingest_text_behind(ctx, b, (size_t)(e - b), spec, n_tokens, rows_in_front, &tab.ing)
                                 : pgt_ingest_text(ctx, b, (size_t)(e - b), spec, n_tokens, &tab.ing);
    if (rc == PGT_EDOMAIN) return false;
    check(rc, ctx);
    const int64_t bad = pgt_ingest_bad_line(tab.ing);
    if (bad >= 0) {
        const std::string msg = std::string(what) + " on line " + std::to_string(first_line_no + (size_t)bad) + " of " + path;
        if (!error) die(msg);
        *error = msg;
        return true;
    }
    tab.n = (size_t)pgt_ingest_rows(tab.ing);
    const uint64_t *len = nullptr, *off = nullptr;
    const uint32_t *nlen = nullptr;
    const size_t n_runs = pgt_ingest_runs(tab.ing, &len, &off, &nlen);
    for (size_t r = 0; r < n_runs; ++r) runs.add(b + off[r], b + off[r] + nlen[r], len[r]);
    return true;
}

// ---- the rows coming back from the GPU ----------------------------------------------------------------------
// With one row per site (-stepsize 1) the row array is hundreds of megabytes: value-initialising it (std::vector)
// and taking its page faults one at a time inside the device-to-host copy costs more than the kernels.  Here it is
// left uninitialised — every row is written by the copy — and its pages are touched by all threads first.
template <class T>
struct RowArray {
    HostBuf p;
    size_t n;
    explicit RowArray(size_t rows) : n(rows) {
        p.alloc((rows ? rows : 1) * sizeof(T));
        const size_t bytes = rows * sizeof(T), page = 4096;
        if (bytes < ((size_t)32 << 20)) return;
        const int T_ = host_threads();
        char *base = static_cast<char *>(p.p);
        const size_t per = (bytes / (size_t)T_ + page - 1) / page * page;
        std::vector<std::thread> th;
        for (int t = 0; t < T_; ++t) {
            const size_t lo = per * (size_t)t, hi = std::min(bytes, lo + per);
            if (lo >= hi) break;
            th.emplace_back([=] { for (size_t i = lo; i < hi; i += page) base[i] = 0; });
        }
        for (auto &x : th) x.join();
    }
    T *data() { return static_cast<T *>(p.p); }
    size_t size() const { return n; }
    T &operator[](size_t i) { return data()[i]; }
    const T &operator[](size_t i) const { return static_cast<const T *>(p.p)[i]; }
};

// ---- the site-window table of a run: on the host, or — from 2^20 windows on — on the device -------------------
// With `-stepsize 1` there is one window per site: 32 bytes of table per window that the host would fill and
// upload only for the GPU to read once.  pgt_wintab_sites writes the same table in GPU memory from the run
// lengths; the host keeps the index of every run's first window and finds a row's chromosome by bisection.
struct SiteWindows {
    std::vector<pgt_win> win;    // the table, when it lives on the host
    pgt_wintab *tab = nullptr;   // ... or the device table
    const uint64_t *first = nullptr;
    size_t n_runs = 0, n = 0;
    SiteWindows() = default;
    SiteWindows(const SiteWindows &) = delete;
    SiteWindows &operator=(const SiteWindows &) = delete;
    ~SiteWindows() { if (tab) pgt_wintab_free(tab); }
    uint32_t label(size_t i) const {
        if (!tab) return win[i].label_run;
        return (uint32_t)(std::upper_bound(first, first + n_runs + 1, (uint64_t)i) - first - 1);
    }
    // get_ctx: called only when the table goes to the device (it waits for HIP start-up)
    template <class GetCtx>
    void build(const Runs &runs, uint32_t W, uint32_t S, GetCtx &&get_ctx, PhaseTimer *timer = nullptr, bool host_only = false) {
        check(pgt_build_windows_sites(runs.len.data(), runs.len.size(), W, S, nullptr, 0, &n), nullptr);
        const char *force = std::getenv("PGT_DEVICE_WINTAB");  // 0 / 1: never / always (tests); default by size
        const bool on_device = !host_only && (force ? std::atoi(force) != 0 : n >= ((size_t)1 << 20));  // host_only: the multi-GPU path shards a host table
        if (n == 0) return;
        if (on_device) {
            pgt_ctx *ctx = get_ctx();
            if (timer) timer->lap("wait for HIP");
            check(pgt_wintab_sites(ctx, runs.len.data(), runs.len.size(), W, S, &tab), ctx);
            first = pgt_wintab_first(tab);
            n_runs = runs.len.size();
        } else {
            win.resize(n);
            check(pgt_build_windows_sites(runs.len.data(), runs.len.size(), W, S, win.data(), win.size(), &n), nullptr);
        }
    }
};

// Both speed hints from the tool's own arguments: every GPU that reduces a slice of the table — and the single-GPU run —
// then takes the same query strategy and tree levels, whatever its slice looks like (rows are bitwise independent of the
// number of GPUs only under identical hints).
inline void set_site_hints(pgt_ctx *ctx, uint32_t W, uint32_t S) {
    check(pgt_set_max_window(ctx, W), ctx);
    check(pgt_set_window_step(ctx, S), ctx);
}

// ---- several GPUs in one host process (PGT_DEVICES=0,1,...) ---------------------------------------------------
// The reference is one thread over one stream (fstWindow.cpp:109-155); here the window table is cut into one
// contiguous block per GPU (pgt_plan_shards: balanced by sites, block starts tree-node aligned, halo <= one window),
// every GPU gets its own host thread and context, reduces its block from the columns of ITS site range and writes its
// rows into its slice of the one row array the main thread prints — the CLI-level form of bench.py's N > 1 run.
// Where the text is parsed on the GPUs (large inputs), the file is cut at line starts into one piece per GPU: each
// piece crosses its own PCIe link and is parsed where it lands; a GPU then gathers the columns of its shard from the
// pieces that hold them (mostly its own: device-to-device on one GPU; the halo: a peer copy of <= W + 65535 sites).
struct DevicePiece {   // one GPU's piece of the text, parsed there
    pgt_ctx *ctx = nullptr;
    pgt_ingest *ing = nullptr;
    uint64_t row0 = 0, rows = 0;  // global site range [row0, row0 + rows)
};
struct GatherColumn {  // token of the ingest object, element size
    int token;
    size_t elem;
};

// Cut [b,e) at line starts into `parts` pieces of about equal bytes
inline std::vector<const char *> cut_at_lines(const char *b, const char *e, size_t parts) {
    std::vector<const char *> cut(parts + 1, e);
    cut[0] = b;
    const size_t len = (size_t)(e - b);
    for (size_t t = 1; t < parts; ++t) {
        const char *p = b + len / parts * t;
        if (p < cut[t - 1]) p = cut[t - 1];
        const void *nl = p < e ? std::memchr(p, '\n', (size_t)(e - p)) : nullptr;
        cut[t] = nl ? static_cast<const char *>(nl) + 1 : e;
    }
    return cut;
}

// The last line of [b,e) holds nothing but blanks (or nothing at all): for the parser of that piece the data simply ends
// there, but it ends there for the pieces behind it as well
inline bool ends_with_blank_line(const char *b, const char *e) {
    if (e == b) return false;
    const char *p = e;
    if (p[-1] == '\n') --p;
    for (; p > b && p[-1] != '\n'; --p)
        if (p[-1] != ' ' && p[-1] != '\t' && p[-1] != '\r') return false;
    return true;
}

// Parse one file on all devices: piece k on device k.  true: `pieces` (one per device that got data, in file order) and
// `runs` (stitched at the seams) describe the table, *n_rows its length; false: some piece was refused by the device
// parser (too many irregular lines) — parse on the host.  Text errors die with the host parser's message.
inline bool ingest_on_devices(DeviceOpener &device, const char *b, const char *e, const uint8_t *spec, int n_tokens, const char *what,
                              const char *path, std::vector<DevicePiece> &pieces, Runs &runs, size_t *n_rows, size_t first_line_no = 1) {
    const size_t N = device.count();
    const std::vector<const char *> cut = cut_at_lines(b, e, N);
    std::vector<pgt_ingest *> ing(N, nullptr);
    std::vector<int> rc(N, PGT_OK);
    std::vector<std::string> msg(N);
    for (size_t k = 0; k < N; ++k) (void)device.get(k);  // all contexts are open (or the run has died) before the threads start
    {
        std::vector<std::thread> th;
        for (size_t k = 0; k < N; ++k)
            th.emplace_back([&, k] {
                if (cut[k + 1] == cut[k]) return;
                pgt_ctx *c = device.get(k);
                rc[k] = pgt_ingest_text(c, cut[k], (size_t)(cut[k + 1] - cut[k]), spec, n_tokens, &ing[k]);
                if (rc[k] != PGT_OK) msg[k] = pgt_last_error(c);
            });
        for (auto &t : th) t.join();
    }
    auto free_all = [&] { for (pgt_ingest *g : ing) if (g) pgt_ingest_free(g); };
    for (size_t k = 0; k < N; ++k)
        if (rc[k] == PGT_EDOMAIN) { free_all(); return false; }
    for (size_t k = 0; k < N; ++k)
        if (rc[k] != PGT_OK) die("libpgtwin: " + msg[k]);
    uint64_t row = 0;
    for (size_t k = 0; k < N; ++k) {
        if (!ing[k]) continue;
        const int64_t bad = pgt_ingest_bad_line(ing[k]);
        if (bad >= 0) die(std::string(what) + " on line " + std::to_string(first_line_no + row + (uint64_t)bad) + " of " + path);
        const uint64_t rows = pgt_ingest_rows(ing[k]);
        const uint64_t *len = nullptr, *off = nullptr;
        const uint32_t *nlen = nullptr;
        const size_t n_runs = pgt_ingest_runs(ing[k], &len, &off, &nlen);
        for (size_t r = 0; r < n_runs; ++r) runs.add(cut[k] + off[r], cut[k] + off[r] + nlen[r], len[r]);  // add() merges a run that continues across the seam
        pieces.push_back(DevicePiece{device.get(k), ing[k], row, rows});
        row += rows;
        const bool stopped = pgt_ingest_blank_before_end(ing[k]) != 0 || ends_with_blank_line(cut[k], cut[k + 1]);
        ing[k] = nullptr;
        if (stopped) break;  // a blank line ends the data (fstWindow.cpp:125): the pieces behind it are not part of the table
    }
    free_all();
    *n_rows = (size_t)row;
    return true;
}
inline void free_pieces(std::vector<DevicePiece> &pieces) {
    for (DevicePiece &p : pieces)
        if (p.ing) pgt_ingest_free(p.ing);
    pieces.clear();
}

// The sharded reduce.  host_reduce(ctx, site_lo, n_sites, win, n_win, out, out_bytes): columns on the host (the entry point
// uploads the slice); cols_reduce(ctx, dcols, n_sites, win, n_win, out, out_bytes): dcols[c] = this GPU's gathered copy of
// gather[c].  Exactly one of the two paths is taken: pieces empty -> host columns.
template <class Row, class HostReduce, class ColsReduce>
void reduce_on_devices(DeviceOpener &device, const std::vector<pgt_win> &win, uint32_t W, uint32_t S, std::vector<DevicePiece> &pieces,
                       const std::vector<GatherColumn> &gather, Row *rows, HostReduce host_reduce, ColsReduce cols_reduce) {
    const size_t N = device.count();
    std::vector<pgt_shard> shard(N);
    check(pgt_plan_shards(win.data(), win.size(), (uint32_t)N, shard.data()), nullptr);
    for (size_t k = 0; k < N; ++k) (void)device.get(k);
    std::vector<std::thread> th;
    for (size_t k = 0; k < N; ++k)
        th.emplace_back([&, k] {
            const pgt_shard sh = shard[k];
            const size_t n_local = (size_t)(sh.win_end - sh.win_begin);
            if (n_local == 0) return;
            pgt_ctx *ctx = device.get(k);
            set_site_hints(ctx, W, S);
            std::vector<pgt_win> local(win.begin() + (ptrdiff_t)sh.win_begin, win.begin() + (ptrdiff_t)sh.win_end);
            for (pgt_win &w : local) { w.lo -= sh.site_lo; w.hi -= sh.site_lo; }
            const uint64_t n_sites = sh.site_hi - sh.site_lo;
            Row *out = rows + sh.win_begin;
            if (pieces.empty()) {
                check(host_reduce(ctx, sh.site_lo, n_sites, local.data(), n_local, out, n_local * sizeof(Row)), ctx);
                return;
            }
            std::vector<void *> dcols(gather.size(), nullptr);
            for (size_t c = 0; c < gather.size(); ++c) {
                check(pgt_dev_alloc(ctx, (size_t)n_sites * gather[c].elem + 16, &dcols[c]), ctx);
                for (const DevicePiece &p : pieces) {
                    const uint64_t g0 = std::max<uint64_t>(sh.site_lo, p.row0), g1 = std::min<uint64_t>(sh.site_hi, p.row0 + p.rows);
                    if (g0 >= g1) continue;
                    const char *src = static_cast<const char *>(pgt_ingest_column(p.ing, gather[c].token)) + (g0 - p.row0) * gather[c].elem;
                    check(pgt_dev_copy(ctx, static_cast<char *>(dcols[c]) + (g0 - sh.site_lo) * gather[c].elem, p.ctx, src,
                                       (size_t)(g1 - g0) * gather[c].elem), ctx);
                }
            }
            check(cols_reduce(ctx, dcols.data(), n_sites, local.data(), n_local, out, n_local * sizeof(Row)), ctx);
            for (void *d : dcols) check(pgt_dev_free(ctx, d), ctx);
        });
    for (auto &t : th) t.join();
}

// ---- large inputs, one GPU: the host parses the head of the text while HIP starts, the GPU the rest --------------------------
// HIP start-up (70-260 ms) is dead time for the device parser, and what the host parser needs for 1-2 GB of text (14 GB/s
// on this box's share of cores).  So wherever the device parser is taken (2 GiB of text and more) the text is cut at a line
// start near 1 GiB: the host threads parse
// the head into huge-page columns beside HIP start-up and the GPU parses the tail behind room for the head's rows
// (pgt_ingest_text_behind), into which the head's columns are then uploaded (20 B per line instead of 33 B of text): one
// contiguous column per field, no copy.  Measured, fstWindow end to end, alternating runs on two boxes
// (profiles/r03/hybrid_ingest_ab.txt; medians): 3.3 GB 0.44 -> 0.30 s and 0.34 -> 0.32 s, 6.7 GB 0.55 -> 0.59, 10 GB 0.69 -> 0.70:
// a gain where HIP start-up is a large part of the run, nothing beyond; a head of 2-3 GiB is worse than 1 GiB (its parse then
// runs beside the upload of the tail and slows it).  PGT_HYBRID_HOST_BYTES=<n> moves the cut (0: off).
//   0  not taken (small input, switched off, or the device refused the tail): nothing was changed
//   1  `tab` and `runs` hold the table on the GPU
//   2  the data ended at a blank line inside the head: `host_tab` (n rows) and `runs` hold it on the host
struct HybridColumn { int token; size_t elem; const void *host; };  // host: the head's column, valid after the parse
template <class Table, class HostColumns>
int ingest_hybrid(DeviceOpener &device, const char *b, const char *e, const uint8_t *spec, int n_tokens, const char *what, const char *path,
                  Table &host_tab, HostColumns host_columns, DeviceTable &tab, Runs &runs, size_t *n_rows, PhaseTimer &timer) {
    size_t host_bytes = (size_t)1 << 30, least = (size_t)5 << 28;  // cut near 1 GiB, a tail of 256 MiB at least (the caller asks from 2 GiB on)
    if (const char *v = std::getenv("PGT_HYBRID_HOST_BYTES")) {
        host_bytes = (size_t)std::max<long long>(std::atoll(v), 0);
        least = host_bytes + 1;
    }
    if (host_bytes == 0 || (size_t)(e - b) < least) return 0;
    const void *nl = std::memchr(b + host_bytes - 1, '\n', (size_t)(e - (b + host_bytes - 1)));
    const char *cut = nl ? static_cast<const char *>(nl) + 1 : e;
    if (cut >= e) return 0;
    // lines of the head: the tail's messages carry global line numbers
    size_t head_lines = 0;
    {
        const int T = host_threads();
        const std::vector<const char *> part = cut_at_lines(b, cut, (size_t)T);
        std::vector<size_t> k(T, 0);
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back([&, t] { k[t] = (size_t)std::count(part[t], part[t + 1], '\n'); });
        for (auto &x : th) x.join();
        for (size_t v : k) head_lines += v;
    }
    std::string err_h, err_d;
    Runs runs_h, runs_d;
    size_t n_h = 0;
    std::thread head([&] { n_h = parse_table(b, cut, host_tab, runs_h, what, path, 1, &err_h); });
    pgt_ctx *ctx = device.get();
    const bool ok_d = ingest_on_device(ctx, cut, e, spec, n_tokens, what, path, head_lines + 1, tab, runs_d, &err_d, head_lines);
    head.join();
    timer.lap("head on the host, tail on the GPU");
    if (!err_h.empty()) die(err_h);
    auto drop_tail = [&] {
        if (tab.ing) pgt_ingest_free(tab.ing);
        tab.ing = nullptr;
        tab.n = 0;
    };
    if (n_h < head_lines) {  // a blank line in the head: the data ended there
        drop_tail();
        runs = runs_h;
        *n_rows = n_h;
        return 2;
    }
    if (!ok_d) return 0;  // too many irregular lines in the tail: the caller parses everything on the host
    if (!err_d.empty()) die(err_d);
    const size_t n = n_h + tab.n;  // n_h == head_lines: the head's rows fill the room in front exactly
    for (const HybridColumn &c : host_columns(host_tab))
        check(pgt_dev_upload(ctx, pgt_ingest_column_base(tab.ing, c.token), c.host, n_h * c.elem), ctx);
    tab.from_base = true;
    tab.n = n;
    runs = runs_h;
    runs.append(runs_d);
    *n_rows = n;
    timer.lap("columns joined");
    return 1;
}

// ---- a table larger than the GPU: reduced in PASSES (PGT_MAX_RESIDENT_SITES, or by itself when the text would not fit) -----
// The reference streams with O(W) memory (fstWindow.cpp:111-113); the paths above hold one whole input on the GPU.  Here
// a first, cheap pass over the text on the host threads finds the chromosome runs and the byte position of every
// 65536th row (no number is converted); the window table is built from the runs, cut by pgt_plan_shards into blocks of
// at most the resident limit, and every block is one pass: the text of ITS rows [site_lo, site_hi) (+ the halo of one window
// that pgt_plan_shards adds; block starts are multiples of 65536 rows, so they are marked) is parsed on the GPU, the block's
// windows are reduced and its rows are printed before the next pass starts — the output streams as the reference's does.
// GPU memory per pass: the text and the columns of one block.  A bad line is reported when its pass reaches it (rows of
// earlier passes are out by then — the reference would have printed them too).  With PGT_DEVICES the blocks go round the
// listed GPUs (every GPU uploads over its own PCIe link) and are printed in order.
constexpr uint64_t kMarkEvery = 65536;  // rows between two byte marks = the smallest shard alignment of pgt_plan_shards

// splitmix64's finaliser: the per-row mixer of the position digests below
inline uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// runs of chromosome names, the number of rows and the row marks of [b,e); stops at the first blank line like the parsers.
// pos_digest (optional): one 64-bit digest of the SECOND token (the position) per block of 65536 rows — the sum over the
// block's rows of mix64(value of a plain digit string, or a hash of the token's bytes otherwise; + the row's index), so two
// files list the same positions in the same rows exactly when their digests agree (up to 2^-64): how dxyWindow's passes
// learn BEFORE the first row is printed whether the two MAF files list the same sites, without holding a position column.
inline size_t scan_runs_and_marks(const char *b, const char *e, Runs &runs, std::vector<const char *> &mark, const char **data_end,
                                  std::vector<uint64_t> *pos_digest = nullptr) {
    int T = host_threads();
    if ((size_t)(e - b) < (1u << 20)) T = 1;
    const std::vector<const char *> cut = cut_at_lines(b, e, (size_t)T);
    std::vector<size_t> lines(T, 0), off(T + 1, 0);
    auto run_all = [&](auto &&fn) {
        if (T == 1) { fn(0); return; }
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back(fn, t);
        for (auto &x : th) x.join();
    };
    run_all([&](int t) {
        size_t k = (size_t)std::count(cut[t], cut[t + 1], '\n');
        if (cut[t + 1] > cut[t] && cut[t + 1][-1] != '\n') ++k;  // last line without newline
        lines[t] = k;
    });
    for (int t = 0; t < T; ++t) off[t + 1] = off[t] + lines[t];
    mark.assign(off[T] / kMarkEvery + 2, nullptr);
    struct Result { size_t rows = 0; bool stopped = false; const char *end = nullptr; Runs runs; std::vector<uint64_t> digest; };
    std::vector<Result> res(T);
    run_all([&](int t) {
        Result &r = res[t];
        Cursor c{cut[t], cut[t + 1]};
        size_t row = off[t];
        const size_t block0 = row / kMarkEvery;  // r.digest[k]: this piece's share of block block0 + k
        while (c.p < c.end) {
            const char *line = c.p;
            c.skip_blank();
            if (c.at_eol()) { r.stopped = true; r.end = line; break; }
            if (row % kMarkEvery == 0) mark[row / kMarkEvery] = line;
            const Tok chr = c.token();
            r.runs.add(chr.first, chr.second);
            if (pos_digest) {
                c.skip_blank();
                uint64_t v = 0;
                bool digits = !c.at_eol();
                const char *q = c.p;
                if (q < c.end && *q == '+') ++q;  // as to_u32
                for (; q < c.end && *q != ' ' && *q != '\t' && *q != '\n' && *q != '\r'; ++q) {
                    if (*q < '0' || *q > '9' || q - c.p > 18) digits = false;
                    v = digits ? v * 10 + (uint64_t)(*q - '0') : mix64(v ^ (uint64_t)(unsigned char)*q);
                }
                if (!digits) v = mix64(v ^ 0xD1B54A32D192ED03ull);  // not a position: some digest that no digit string has by construction
                const size_t k = row / kMarkEvery - block0;
                if (k >= r.digest.size()) r.digest.resize(k + 1, 0);
                r.digest[k] += mix64(v + 0x9E3779B97F4A7C15ull * (uint64_t)(row % kMarkEvery));
            }
            ++row;
            c.next_line();
        }
        if (!r.stopped) r.end = cut[t + 1];
        r.rows = row - off[t];
    });
    size_t n = 0;
    *data_end = b;
    if (pos_digest) pos_digest->assign(off[T] / kMarkEvery + 1, 0);
    for (int t = 0; t < T; ++t) {
        runs.append(res[t].runs);
        n = off[t] + res[t].rows;
        *data_end = res[t].end;
        if (pos_digest)
            for (size_t k = 0; k < res[t].digest.size(); ++k) (*pos_digest)[off[t] / kMarkEvery + k] += res[t].digest[k];
        if (res[t].stopped) break;
    }
    if (pos_digest) pos_digest->resize((n + kMarkEvery - 1) / kMarkEvery);
    mark.resize(n / kMarkEvery + 2);
    mark[(n + kMarkEvery - 1) / kMarkEvery] = *data_end;  // the mark behind the last row
    return n;
}

// 0: the whole input fits (the usual paths); else the largest number of sites one pass may hold.
// PGT_MAX_RESIDENT_SITES=<n> forces passes (tests, or a GPU shared with other work).
// the decision itself: the device parser holds the text and the columns of what it parses at once (the reduce after it:
// columns, a tree of < 2 % of them and the rows); resident if that needs less than 80 % of the free memory, else passes of
// what fits 60 % of it
inline uint64_t resident_limit_for(size_t text_bytes, double bytes_per_line, size_t bytes_per_site_on_gpu, size_t free_bytes, int texts) {
    const double per_site = (double)texts * bytes_per_line + 1.1 * (double)bytes_per_site_on_gpu;
    if ((double)text_bytes / bytes_per_line * per_site < 0.8 * (double)free_bytes) return 0;
    return std::max<uint64_t>((uint64_t)(0.6 * (double)free_bytes / per_site), 1);
}
template <class GetCtx>
inline uint64_t resident_limit(const char *b, const char *e, size_t bytes_per_site_on_gpu, GetCtx &&get_ctx, int texts = 1) {  // texts: files like this one parsed side by side
    if (const char *v = std::getenv("PGT_MAX_RESIDENT_SITES")) {
        const long long lim = std::atoll(v);
        return lim > 0 ? (uint64_t)lim : 0;
    }
    const size_t text_bytes = (size_t)(e - b);
    if (text_bytes * (size_t)texts < ((size_t)8 << 30)) return 0;  // far below any MI355X: do not even ask
    size_t free_b = 0, total_b = 0;
    pgt_ctx *ctx = get_ctx();
    check(pgt_dev_memory(ctx, &free_b, &total_b), ctx);
    const size_t sample = std::min<size_t>(text_bytes, (size_t)8 << 20);  // bytes per line from the head of the file
    const double per_line = (double)sample / (double)std::max<size_t>((size_t)std::count(b, b + sample, '\n'), 1);
    return resident_limit_for(text_bytes, per_line, bytes_per_site_on_gpu, free_b, texts);
}

// The passes.  parse_and_reduce(ctx, piece_begin, piece_end, first_row /*global*/, rows_in_piece, win /*rebased*/, n_win, out, error)
// (error: NULL = exit on a bad line, else the message goes there) parses the piece (device parser; host parser where the device refuses) and reduces the block's windows (n_win = 0: parse only);
// print(rows, n_win, win /*the block's entries of the global table: label_run*/) writes them.  Host memory: the mapped
// text and the window table (24 bytes per window).
template <class Row, class ParseReduce, class Print>
void reduce_in_passes(DeviceOpener &device, const char *b, const char *e, uint32_t W, uint32_t S, uint64_t max_resident, Runs &runs,
                      PhaseTimer &timer, ParseReduce parse_and_reduce, Print print) {
    pgt_ctx *ctx = device.get();
    std::vector<const char *> mark;
    const char *data_end = b;
    const size_t n = scan_runs_and_marks(b, e, runs, mark, &data_end);
    timer.lap("scan runs");
    size_t n_win = 0;
    check(pgt_build_windows_sites(runs.len.data(), runs.len.size(), W, S, nullptr, 0, &n_win), nullptr);
    const uint64_t per_pass = std::max<uint64_t>(max_resident, 2 * (uint64_t)W + 2 * kMarkEvery);
    // rows [from, to) parsed only, from a multiple of 65536: rows no window covers still have to be well-formed (the resident
    // paths parse the whole text before they look at the windows)
    auto parse_only = [&](uint64_t from, uint64_t to) {
        const uint64_t step = per_pass / kMarkEvery * kMarkEvery;
        for (uint64_t row0 = from; row0 < to; row0 += step) {
            const uint64_t row1 = std::min<uint64_t>(row0 + step, to), m1 = std::min<uint64_t>((row1 + kMarkEvery - 1) / kMarkEvery, (n + kMarkEvery - 1) / kMarkEvery);
            parse_and_reduce(ctx, mark[row0 / kMarkEvery], mark[m1], row0, std::min<uint64_t>(m1 * kMarkEvery, n) - row0, nullptr, 0, nullptr, nullptr);
        }
    };
    if (n_win == 0) {
        parse_only(0, n);
        timer.lap("passes");
        return;
    }
    std::vector<pgt_win> win(n_win);
    check(pgt_build_windows_sites(runs.len.data(), runs.len.size(), W, S, win.data(), win.size(), &n_win), nullptr);
    const uint32_t passes = (uint32_t)std::min<uint64_t>((n + per_pass - 1) / per_pass + 1, 1u << 20);
    std::vector<pgt_shard> shard(passes);
    check(pgt_plan_shards(win.data(), n_win, passes, shard.data()), nullptr);
    timer.lap("window table");
    for (uint32_t p = 0; p < passes; ++p)
        if (shard[p].win_end > shard[p].win_begin) {
            parse_only(0, shard[p].site_lo);  // a first run without windows
            break;
        }
    // one pass: block p on context c -> its rows
    auto run_pass = [&](uint32_t p, pgt_ctx *c, std::vector<Row> &rows, std::string *error) {
        const pgt_shard sh = shard[p];
        const size_t n_local = (size_t)(sh.win_end - sh.win_begin);
        if (sh.site_lo % kMarkEvery != 0) die("pgt_plan_shards returned a block start that is not a multiple of 65536");
        // every row is parsed by some pass, also rows no window covers (dropped tails of a run): a bad line there is an error
        // in the resident paths, so it is one here — a piece reaches to the next block's first row, the last one to the end
        uint64_t cover_hi = n;
        for (uint32_t q = p + 1; q < passes; ++q)
            if (shard[q].win_end > shard[q].win_begin) { cover_hi = std::max<uint64_t>(sh.site_hi, shard[q].site_lo); break; }
        const uint64_t hi_mark = std::min<uint64_t>((cover_hi + kMarkEvery - 1) / kMarkEvery, (n + kMarkEvery - 1) / kMarkEvery);
        const char *pb = mark[sh.site_lo / kMarkEvery], *pe = mark[hi_mark];
        const uint64_t rows_in_piece = std::min<uint64_t>(hi_mark * kMarkEvery, n) - sh.site_lo;
        std::vector<pgt_win> local(win.begin() + (ptrdiff_t)sh.win_begin, win.begin() + (ptrdiff_t)sh.win_end);
        for (pgt_win &w : local) { w.lo -= sh.site_lo; w.hi -= sh.site_lo; }
        rows.resize(n_local);
        parse_and_reduce(c, pb, pe, sh.site_lo, rows_in_piece, local.data(), n_local, rows.data(), error);
    };
    const size_t N = device.count();
    for (size_t k = 0; k < N; ++k) set_site_hints(device.get(k), W, S);
    if (N == 1) {
        std::vector<Row> rows;
        for (uint32_t p = 0; p < passes; ++p) {
            if (shard[p].win_end == shard[p].win_begin) continue;
            run_pass(p, ctx, rows, nullptr);
            print(rows.data(), rows.size(), win.data() + shard[p].win_begin);
        }
        timer.lap("passes");
        return;
    }
    // PGT_DEVICES: block p goes to GPU p mod N (a pass is bound by its text upload, and every GPU has its own PCIe link);
    // the main thread prints the blocks in order as they finish — and reports a bad line when its block's turn has come, so
    // that the message and the rows before it are those of the one-GPU run; a GPU runs at most two rounds ahead of the printer
    struct Done { std::vector<Row> rows; std::string error; bool ready = false; };
    std::vector<Done> done(passes);
    std::mutex mu;
    std::condition_variable cv;
    uint32_t printed = 0;  // blocks below this are out
    std::vector<std::thread> th;
    for (size_t k = 0; k < N; ++k)
        th.emplace_back([&, k] {
            for (uint32_t p = (uint32_t)k; p < passes; p += (uint32_t)N) {
                if (shard[p].win_end == shard[p].win_begin) continue;
                {
                    std::unique_lock<std::mutex> lock(mu);
                    cv.wait(lock, [&] { return p < printed + 2 * N; });
                }
                std::vector<Row> rows;
                std::string error;
                run_pass(p, device.get(k), rows, &error);
                const bool failed = !error.empty();
                {
                    std::lock_guard<std::mutex> lock(mu);
                    done[p].rows = std::move(rows);
                    done[p].error = std::move(error);
                    done[p].ready = true;
                }
                cv.notify_all();
                if (failed) return;  // the main thread ends the run when this block's turn comes
            }
        });
    for (uint32_t p = 0; p < passes; ++p) {
        if (shard[p].win_end != shard[p].win_begin) {
            std::vector<Row> rows;
            {
                std::unique_lock<std::mutex> lock(mu);
                cv.wait(lock, [&] { return done[p].ready; });
                if (!done[p].error.empty()) die(done[p].error);
                rows = std::move(done[p].rows);
            }
            print(rows.data(), rows.size(), win.data() + shard[p].win_begin);
        }
        {
            std::lock_guard<std::mutex> lock(mu);
            printed = p + 1;
        }
        cv.notify_all();
    }
    for (auto &t : th) t.join();
    timer.lap("passes");
}

// Window size / step size as fstWindow.cpp:51-64 reads them (atoi); zero, negative or
// non-numeric values are refused.  The reference only warns for a bad step and then crashes
// (SURVEY.md §4 Q9); a step larger than the window crashes it too.  Here all of these exit 255.
inline void parse_window_args(int argc, char **argv, uint32_t &W, uint32_t &S) {
    if (argc > 2) {
        const int w = std::atoi(argv[2]);
        if (w <= 0) die("Window size must be a positive integer");
        W = (uint32_t)w;
    }
    if (argc > 3) {
        const int s = std::atoi(argv[3]);
        if (s <= 0) die("Step size must be a positive integer");
        S = (uint32_t)s;
    }
    if (S > W) die("Step size must not exceed the window size");
}

}  // namespace pgthost
