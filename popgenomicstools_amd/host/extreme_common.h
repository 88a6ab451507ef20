// extreme_common.h — shared by the ihsWindow and xpehhWindow hosts: both tools are one loop
// (ihsWindow.cpp:123-221, xpehhWindow.cpp:126-232) over selscan *.norm lines
//     <chr>_<anything>  pos  f0 f1 f2 ...
// with the chromosome taken from the locus id up to its first '_' (ihsWindow.cpp:82-93) and the
// score in numeric field 4 (iHS) or 6 (XP-EHH) after the position.
#pragma once

#include <map>

#include "host_common.h"

namespace pgthost {

struct ScoreTable {
    Column<uint32_t> pos;
    Column<double> score;
    int score_field = 4;
    void alloc(size_t rows) { pos.alloc(rows); score.alloc(rows); }
    bool parse_line(Cursor &c, size_t i, Runs &runs) {
        const Tok id = c.token();
        const char *us = static_cast<const char *>(std::memchr(id.first, '_', (size_t)(id.second - id.first)));
        if (!to_u32(c.token(), pos[i])) return false;
        Tok t{};
        for (int k = 0; k <= score_field; ++k) t = c.token();
        if (!to_f64(t, score[i])) return false;  // the reference would silently reuse the previous line's value
        runs.add(id.first, us ? us : id.second);
        return true;
    }
};

// -chrlen FILE: chr <TAB> length; std::map::insert keeps the first entry of a name (ihsWindow.cpp:112-121)
inline std::map<std::string, uint32_t> read_chrlen(const char *path) {
    std::map<std::string, uint32_t> m;
    Text text;
    if (!text.open(path)) die(std::string("Unable to open chromosome length file ") + path);
    Cursor c{text.begin(), text.end()};
    while (c.p < c.end) {
        const Tok name = c.token();
        uint32_t len = 0;
        if (name.first != name.second && to_u32(c.token(), len)) m.insert({std::string(name.first, name.second), len});
        c.next_line();
    }
    return m;
}

// Everything after argument parsing: parse, window table, GPU reduce, TSV (ihsWindow.cpp:101-110).
inline int run_extreme(const char *path, bool skip_header, int score_field, uint32_t W, int mode, double cutoff,
                       const char *chrlen_path, const char *open_error) {
    PhaseTimer timer;
    Text text;
    if (!text.open(path)) die(std::string(open_error) + path);
    DeviceOpener device;
    std::map<std::string, uint32_t> lenmap;
    if (chrlen_path) lenmap = read_chrlen(chrlen_path);
    const char *b = text.begin();
    size_t first_line = 1;
    if (skip_header) {  // xpehhWindow.cpp:149-154
        if (text.size() == 0) {
            std::fprintf(stderr, "Input XP-EHH file had zero sites\n");
            return 0;
        }
        Cursor h{b, text.end()};
        h.next_line();
        b = h.p;
        first_line = 2;
    }
    ScoreTable tab;
    tab.score_field = score_field;
    Runs runs;
    const size_t n = parse_table(b, text.end(), tab, runs, "cannot parse '<chr>_<id> pos ... score ...'", path, first_line);
    timer.lap("parse");
    if (n == 0) {  // the reference prints its initial window with an empty chromosome name (:212)
        std::printf("\t1\t%u\tNA\tNA\tNA\t0\n", 1u + (W - 1));
        return 0;
    }
    std::vector<uint32_t> chr_len(runs.name.size(), 0);
    for (size_t r = 0; r < runs.name.size(); ++r) {
        auto it = lenmap.find(runs.name[r]);
        if (it != lenmap.end()) chr_len[r] = it->second;
    }
    size_t n_win = 0;
    check(pgt_build_windows_extreme(tab.pos.data(), runs.len.data(), chr_len.data(), runs.len.size(), W, nullptr, 0, &n_win), nullptr);
    std::vector<pgt_win> win(n_win);
    check(pgt_build_windows_extreme(tab.pos.data(), runs.len.data(), chr_len.data(), runs.len.size(), W, win.data(), win.size(), &n_win), nullptr);
    timer.lap("window table");
    pgt_ctx *ctx = device.get();
    timer.lap("wait for HIP");
    std::vector<pgt_ext_row> rows(n_win);
    check(pgt_extreme_reduce(ctx, tab.pos.data(), tab.score.data(), n, mode, cutoff, win.data(), n_win, rows.data()), ctx);
    timer.lap("gpu reduce");
    write_rows(n_win, longest_name(runs) + 96, [&](size_t i, char *o) {
        const pgt_ext_row &r = rows[i];
        const char *chr = runs.name[win[i].label_run].c_str();
        if (r.nsites > 0)  // score, its position, proportion beyond the cutoff, SNPs in the window
            return (size_t)std::sprintf(o, "%s\t%u\t%u\t%g\t%u\t%g\t%u\n", chr, r.start, r.end, r.value, r.position,
                                        (double)(int)r.nbig / r.nsites, r.nsites);
        return (size_t)std::sprintf(o, "%s\t%u\t%u\tNA\tNA\tNA\t0\n", chr, r.start, r.end);
    });
    finish(timer);
}

}  // namespace pgthost
