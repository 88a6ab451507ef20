// host_common.h — what the three retained C++ hosts share: reading whitespace-separated text
// columns into structure-of-arrays buffers, chromosome run bookkeeping, error exit.
//
// The hosts keep the reference tools' command lines and TSV (fstWindow.cpp:37-67,88;
// hetWindow.cpp:34-64,87; dxyWindow.cpp:63-139,190,429-433).  What changes is the middle: the
// reference streams line by line through a W-entry buffer and calls calcWindow per window; the
// hosts parse the whole input into SoA columns, build the window table once
// (pgt_build_windows_*) and hand both to the GPU through include/pgtwin.h.
#pragma once

#include <zlib.h>

#include <charconv>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "pgtwin.h"

namespace pgthost {

// The reference tools `return -1` from main on error, i.e. exit status 255.
[[noreturn]] inline void die(const std::string &msg) {
    std::fprintf(stderr, "%s\n", msg.c_str());
    std::exit(255);
}

inline void check(int rc, const pgt_ctx *ctx) {
    if (rc != PGT_OK) die(std::string("libpgtwin: ") + pgt_last_error(ctx));
}

// Whole file into memory, transparently gunzipped (dxyWindow sniffs the gzip magic 0x1f8b and
// wraps the stream in a gzip filter, dxyWindow.cpp:82-83,256-278; zlib's gzread does both).
inline bool slurp(const char *path, std::string &out) {
    gzFile f = gzopen(path, "rb");
    if (!f) return false;
    gzbuffer(f, 1 << 20);
    out.clear();
    std::vector<char> buf(1 << 22);
    int n;
    while ((n = gzread(f, buf.data(), (unsigned)buf.size())) > 0) out.append(buf.data(), (size_t)n);
    const bool ok = n == 0;
    gzclose(f);
    return ok;
}

struct Cursor {
    const char *p, *end;
    bool at_eol() const { return p >= end || *p == '\n'; }
    void skip_blank() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\r')) ++p; }
    // next whitespace-delimited token of the current line ("" at end of line)
    std::pair<const char *, const char *> token() {
        skip_blank();
        const char *b = p;
        while (p < end && *p != ' ' && *p != '\t' && *p != '\r' && *p != '\n') ++p;
        return {b, p};
    }
    void next_line() {
        while (p < end && *p != '\n') ++p;
        if (p < end) ++p;
    }
};

inline bool to_u32(std::pair<const char *, const char *> t, uint32_t &v) {
    if (t.first == t.second) return false;
    const char *b = t.first;
    if (*b == '+') ++b;
    unsigned long long x = 0;
    auto r = std::from_chars(b, t.second, x);
    if (r.ec != std::errc() || r.ptr != t.second || x > 0xFFFFFFFFull) return false;
    v = (uint32_t)x;
    return true;
}

inline bool to_i64(std::pair<const char *, const char *> t, long long &v) {
    if (t.first == t.second) return false;
    const char *b = t.first;
    if (*b == '+') ++b;
    auto r = std::from_chars(b, t.second, v);
    return r.ec == std::errc() && r.ptr == t.second;
}

// Correctly rounded, like the strtod behind the reference's `ss >> double`.
inline bool to_f64(std::pair<const char *, const char *> t, double &v) {
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
    void add(const char *b, const char *e) {
        const size_t n = (size_t)(e - b);
        if (name.empty() || name.back().size() != n || std::memcmp(name.back().data(), b, n) != 0) {
            name.emplace_back(b, e);
            len.push_back(0);
        }
        ++len.back();
    }
};

inline int device_from_env() {
    const char *d = std::getenv("PGT_DEVICE");
    return d ? std::atoi(d) : 0;
}

inline pgt_ctx *open_or_die() {
    pgt_ctx *ctx = pgt_open(device_from_env());
    if (!ctx) die(std::string("libpgtwin: ") + pgt_last_error(nullptr));
    return ctx;
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
