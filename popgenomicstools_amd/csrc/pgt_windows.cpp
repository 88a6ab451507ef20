// pgt_windows.cpp — host side of the boundary: window tables and the multi-GPU shard plan.
//
// The reference decides window by window, while it streams, when calcWindow is called
// (fstWindow.cpp:132-138,150-152; hetWindow.cpp:130-136,148-150; dxyWindow.cpp:334-378,407-426).
// Every such call reduces the last `fill` entries of its buffer, i.e. one contiguous range of
// the entry stream, so the same decisions can be taken per chromosome RUN in closed form:
//
//   a run contributes L entries; it starts with a carry of n0 (< W) entries left in the buffer;
//   the buffer is full before entry j_k = W - n0 + k*S of the run (k = 0,1,...), and a window
//   [.. , g + j_k) of W entries is emitted there if that entry exists (j_k <= L-1);
//   at the end of the run the buffer holds n_end = n0 + L - K*S entries (K windows emitted);
//   then the tool-specific end-of-run rule decides flush / carry / leak.
//
// Cost O(#windows + #runs); nothing here touches the GPU.
#include <algorithm>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "pgt_internal.h"

namespace pgt {
namespace {

thread_local std::string g_error;

// Entry-stream windows shared by both modes.  `entries[r]` is the number of buffer entries run r
// pushes (sites, or bp slots).  bp_rules selects dxyWindow's slot-mode end-of-run rule
// (dxyWindow.cpp:353-355: flush only if fill > W-S, and a run that does not flush leaks its
// entries into the next one) instead of the site-mode rule (fstWindow.cpp:132-134: flush
// whenever fill > 0).  Ranges are in ENTRY coordinates.
//
// Two phases, so that neither counting nor filling walks the windows one after the other:
//   plan_entry_windows   O(#runs): per run the index of its first window, the number K of full
//                        windows (an arithmetic progression: hi_k = hi0 + k*S) and its end-of-run window;
//   for_each_window      every window by index, in parallel chunks once there are many (the
//                        `-stepsize 1` regime: as many windows as sites).
uint64_t plan_entry_windows(const uint64_t *entries, size_t n_runs, uint32_t W, uint32_t S, bool bp_rules,
                            std::vector<RunPlan> &plan) {
    plan.assign(n_runs, RunPlan{});
    uint64_t g = 0;    // entries pushed before this run
    uint64_t n0 = 0;   // buffer fill carried into this run
    uint64_t out = 0;  // windows emitted so far
    for (size_t r = 0; r < n_runs; ++r) {
        RunPlan &p = plan[r];
        p.out0 = out;
        const uint64_t L = entries[r];
        if (L >= 1 && L - 1 >= (uint64_t)W - n0) {
            p.K = (L - 1 - ((uint64_t)W - n0)) / S + 1;
            p.hi0 = g + ((uint64_t)W - n0);
        }
        const uint64_t n_end = n0 + L - p.K * S;
        g += L;
        const bool last = r + 1 == n_runs;
        auto tail = [&] { p.tail = 1; p.tail_lo = g - n_end; p.tail_hi = g; };
        if (last) {  // fstWindow.cpp:150-152 / dxyWindow.cpp:424-426
            if (n_end > (uint64_t)(W - S) && n_end <= W) tail();
            n0 = 0;
        } else if (bp_rules) {  // dxyWindow.cpp:353-355
            if (n_end > (uint64_t)(W - S)) {
                tail();
                n0 = n_end == W ? W - S : 0;
            } else {
                n0 = n_end;  // Q3: neither printed nor reset
            }
        } else {  // fstWindow.cpp:132-134 with calcWindow's carry rule :92-103
            if (n_end > 0) {
                tail();
                n0 = n_end == W ? W - S : 0;
            } else {
                n0 = 0;
            }
        }
        out += p.K + (p.tail ? 1 : 0);
    }
    return out;
}

// fn(index, lo, hi, label_run) for the windows [0, limit) of the plan; fn must be safe to call concurrently
// for different indices.
template <class Fn>
void for_each_window(const std::vector<RunPlan> &plan, uint64_t limit, uint32_t W, uint32_t S, Fn &&fn) {
    auto range = [&](uint64_t i0, uint64_t i1) {
        // first run whose windows reach index i0
        size_t r = (size_t)(std::upper_bound(plan.begin(), plan.end(), i0,
                                             [](uint64_t i, const RunPlan &p) { return i < p.out0; }) - plan.begin());
        r = r ? r - 1 : 0;
        for (; r < plan.size() && plan[r].out0 < i1; ++r) {
            const RunPlan &p = plan[r];
            const uint64_t k0 = i0 > p.out0 ? i0 - p.out0 : 0;
            const uint64_t k1 = std::min<uint64_t>(p.K, i1 - p.out0);
            for (uint64_t k = k0; k < k1; ++k) fn(p.out0 + k, p.hi0 + k * S - W, p.hi0 + k * S, (uint32_t)r);
            if (p.tail && p.out0 + p.K >= i0 && p.out0 + p.K < i1) fn(p.out0 + p.K, p.tail_lo, p.tail_hi, (uint32_t)r);
        }
    };
    unsigned T = std::min(std::thread::hardware_concurrency(), 32u);
    if (limit < (1u << 20) || T < 2) {
        range(0, limit);
        return;
    }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; ++t)
        th.emplace_back(range, limit / T * t, t + 1 == T ? limit : limit / T * (t + 1));
    for (auto &x : th) x.join();
}

int fail(int code, const std::string &msg) {
    g_error = msg;
    return code;
}

}  // namespace

int plan_site_windows(const uint64_t *run_len, size_t n_runs, uint32_t W, uint32_t S, std::vector<RunPlan> &plan, uint64_t *count) {
    if (!count || (n_runs && !run_len)) return fail(PGT_EARG, "site windows: NULL argument");
    // the reference segfaults or exits outside 1 <= S <= W; refuse instead
    if (W < 1 || S < 1 || S > W) return fail(PGT_EARG, "window size and step must satisfy 1 <= step <= window");
    for (size_t r = 0; r < n_runs; ++r)
        if (run_len[r] == 0) return fail(PGT_EARG, "site windows: empty chromosome run");
    *count = plan_entry_windows(run_len, n_runs, W, S, false, plan);
    return PGT_OK;
}

void set_global_error(const std::string &msg) { g_error = msg; }
const std::string &global_error() { return g_error; }

}  // namespace pgt

using namespace pgt;

extern "C" int pgt_build_windows_sites(const uint64_t *run_len, size_t n_runs, uint32_t W, uint32_t S,
                                       pgt_win *out, size_t cap, size_t *n_out) {
    if (!n_out) return fail(PGT_EARG, "pgt_build_windows_sites: NULL argument");
    std::vector<RunPlan> plan;
    uint64_t count = 0;
    if (int rc = plan_site_windows(run_len, n_runs, W, S, plan, &count)) return rc;
    *n_out = (size_t)count;
    if (out)
        for_each_window(plan, std::min<uint64_t>(count, cap), W, S, [&](uint64_t i, uint64_t lo, uint64_t hi, uint32_t label) {
            pgt_win w;
            w.lo = lo;
            w.hi = hi;
            w.label_run = label;
            w.flags = 0;
            w.start = w.end = 0;
            out[i] = w;
        });
    if (out && count > cap) return fail(PGT_ECAP, "pgt_build_windows_sites: output capacity too small");
    return PGT_OK;
}

extern "C" int pgt_build_windows_bp(const uint32_t *pos, const uint64_t *run_len, const uint32_t *chr_len,
                                    size_t n_runs, uint32_t W, uint32_t S, pgt_win *out, size_t cap,
                                    size_t *n_out) {
    if (!n_out || (n_runs && (!run_len || !chr_len || !pos)))
        return fail(PGT_EARG, "pgt_build_windows_bp: NULL argument");
    if (W < 1 || S < 1 || S > W) return fail(PGT_EARG, "window size and step must satisfy 1 <= step <= window");

    // slots of run r: bp 1..max(chr_len, last data position) — the reference pads up to the data
    // site first (dxyWindow.cpp:365) and only then up to the chromosome length (:345,:415)
    std::vector<uint64_t> slots(n_runs), slot_base(n_runs + 1, 0), site_base(n_runs + 1, 0);
    for (size_t r = 0; r < n_runs; ++r) {
        if (run_len[r] == 0) return fail(PGT_EARG, "pgt_build_windows_bp: empty chromosome run");
        if (chr_len[r] == 0) return fail(PGT_EDOMAIN, "pgt_build_windows_bp: chromosome length must be > 0");
        site_base[r + 1] = site_base[r] + run_len[r];
        const uint32_t *p = pos + site_base[r];
        if (p[0] < 1) return fail(PGT_EDOMAIN, "pgt_build_windows_bp: positions are 1-based");
        for (uint64_t i = 1; i < run_len[r]; ++i)
            if (p[i] <= p[i - 1])
                return fail(PGT_EDOMAIN, "pgt_build_windows_bp: positions must increase strictly inside a chromosome");
        slots[r] = std::max<uint64_t>(chr_len[r], p[run_len[r] - 1]);
        slot_base[r + 1] = slot_base[r] + slots[r];
    }

    // global slot G -> (run, bp) and -> first data site whose slot is >= G
    auto run_of = [&](uint64_t G) {
        return (size_t)(std::upper_bound(slot_base.begin(), slot_base.end(), G) - slot_base.begin()) - 1;
    };
    auto first_site_at_or_after = [&](uint64_t G) -> uint64_t {
        if (G >= slot_base[n_runs]) return site_base[n_runs];
        const size_t r = run_of(G);
        const uint32_t bp = (uint32_t)(G - slot_base[r]) + 1;
        const uint32_t *p = pos + site_base[r];
        return site_base[r] + (uint64_t)(std::lower_bound(p, p + run_len[r], bp) - p);
    };

    std::vector<RunPlan> plan;
    const uint64_t count = plan_entry_windows(slots.data(), n_runs, W, S, true, plan);
    *n_out = (size_t)count;
    if (out)
        for_each_window(plan, std::min<uint64_t>(count, cap), W, S, [&](uint64_t i, uint64_t Glo, uint64_t Ghi, uint32_t label) {
            pgt_win w;
            w.lo = first_site_at_or_after(Glo);
            w.hi = first_site_at_or_after(Ghi);
            w.label_run = label;
            w.flags = PGT_WIN_COORDS;
            const size_t r0 = run_of(Glo), r1 = run_of(Ghi - 1);
            w.start = (uint32_t)(Glo - slot_base[r0]) + 1;       // dxywin[0].first
            w.end = (uint32_t)(Ghi - 1 - slot_base[r1]) + 1;     // dxywin[nsites-1].first
            out[i] = w;
        });
    if (out && count > cap) return fail(PGT_ECAP, "pgt_build_windows_bp: output capacity too small");
    return PGT_OK;
}

// ihsWindow.cpp:147-218 / xpehhWindow.cpp:158-229: unlike the sliding tools, window membership here
// depends on the order of events (a site with pos >= winend opens the next window and then sits in
// it even if pos == its end), so the table is produced by running the tool's own loop over the
// positions — integer compares only, O(n).
extern "C" int pgt_build_windows_extreme(const uint32_t *pos, const uint64_t *run_len, const uint32_t *chr_len,
                                         size_t n_runs, uint32_t W, pgt_win *out, size_t cap, size_t *n_out) {
    if (!n_out || (n_runs && (!run_len || !pos))) return fail(PGT_EARG, "pgt_build_windows_extreme: NULL argument");
    if (W < 1) return fail(PGT_EARG, "Window size must be a positive integer");
    if (n_runs == 0) return fail(PGT_EDOMAIN, "pgt_build_windows_extreme: no sites");
    size_t count = 0;
    auto emit = [&](uint64_t lo, uint64_t hi, uint32_t label, uint32_t ws, uint32_t we) {
        if (out && count < cap) {
            pgt_win w;
            w.lo = lo; w.hi = hi; w.label_run = label; w.flags = PGT_WIN_COORDS; w.start = ws; w.end = we;
            out[count] = w;
        }
        ++count;
    };
    uint64_t i = 0;
    uint32_t ws = 1, we = ws + (W - 1);  // the first window of the first chromosome is not clamped (:133-134,154-157)
    auto advance = [&](uint32_t len) {
        ws = we + 1;
        we = ws + (W - 1);
        if (len && we > len) we = len;
    };
    for (size_t r = 0; r < n_runs; ++r) {
        if (run_len[r] == 0) return fail(PGT_EARG, "pgt_build_windows_extreme: empty chromosome run");
        const uint32_t len = chr_len ? chr_len[r] : 0;
        if (r > 0) {  // chromosome change: back to [1,W], clamped (:169-175)
            ws = 1;
            we = ws + (W - 1);
            if (len && we > len) we = len;
        }
        const uint64_t first = i, run_end = i + run_len[r];
        uint64_t lo = i;
        for (; i < run_end; ++i) {
            const uint32_t p = pos[i];
            if (len && p > len) return fail(PGT_EDOMAIN, "position beyond the chromosome length given with -chrlen");
            // the site that triggers a chromosome change takes the `if` branch of :160 and is never tested
            // against the window end; every other site (the very first of the file included) is (:176)
            if ((r == 0 || i != first) && p >= we) {
                emit(lo, i, (uint32_t)r, ws, we);
                advance(len);
                lo = i;
                while (p > we) {  // empty windows in the gap (:184-189)
                    emit(i, i, (uint32_t)r, ws, we);
                    advance(len);
                }
            }
        }
        emit(lo, run_end, (uint32_t)r, ws, we);  // chromosome change (:161) or end of input (:212)
        while (we < len) {                        // trailing windows up to the chromosome length (:162-167, :213-218)
            advance(len);
            emit(run_end, run_end, (uint32_t)r, ws, we);
        }
    }
    *n_out = count;
    if (out && count > cap) return fail(PGT_ECAP, "pgt_build_windows_extreme: output capacity too small");
    return PGT_OK;
}

extern "C" int pgt_plan_shards(const pgt_win *win, uint64_t n_win, uint32_t n_ranks, pgt_shard *out) {
    if (!out || n_ranks == 0 || (n_win && !win)) return fail(PGT_EARG, "pgt_plan_shards: bad argument");
    uint64_t total = 0, max_len = 0;
    for (uint64_t i = 0; i < n_win; ++i) {
        if (win[i].hi < win[i].lo) return fail(PGT_EARG, "pgt_plan_shards: window with hi < lo");
        if (i && win[i].lo < win[i - 1].lo) return fail(PGT_EARG, "pgt_plan_shards: windows must be ordered by lo");
        total = std::max(total, win[i].hi);
        max_len = std::max(max_len, win[i].hi - win[i].lo);
    }
    // Align shard starts to the largest power of two not above the longest window, never below 2^16.
    // Every tree node is a power of two of sites (2^7,2^13,2^19,.. for f64 column pairs, 2^10,2^16,2^22,..
    // for int8, 2^8,2^14,2^20,.. for the extreme-score column), and a query only touches nodes that fit
    // inside its window, so each of them divides this alignment whatever the statistic: every node a
    // query touches covers the same sites as in the single-GPU tree.
    uint64_t align = 1ull << 16;
    for (int sh = 17; sh <= 62; ++sh)
        if ((1ull << sh) <= max_len) align = 1ull << sh;
    uint64_t w = 0;
    for (uint32_t r = 0; r < n_ranks; ++r) {
        // rank r takes the windows whose lo falls in its 1/n_ranks slice of the site axis
        const uint64_t cut = r + 1 == n_ranks ? UINT64_MAX : (uint64_t)((__uint128_t)total * (r + 1) / n_ranks);
        pgt_shard s;
        s.win_begin = w;
        uint64_t hi = 0;
        while (w < n_win && (win[w].lo < cut || r + 1 == n_ranks)) { hi = std::max(hi, win[w].hi); ++w; }
        s.win_end = w;
        if (s.win_end > s.win_begin) {
            s.site_lo = win[s.win_begin].lo / align * align;
            s.site_hi = hi;
        } else {
            s.site_lo = s.site_hi = 0;
        }
        out[r] = s;
    }
    return PGT_OK;
}
