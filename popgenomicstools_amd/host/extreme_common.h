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
// The same capabilities as fstWindow / hetWindow / dxyWindow (round 4): large tables (2 GiB of text and more, or
// PGT_GPU_INGEST=1) are parsed on the GPU — the locus id's `chr_` prefix is the device parser's PGT_TOK_CHR_PREFIX — with
// only the position column (4 B per site) coming back for the window rules, which are history dependent
// (pgt_build_windows_extreme); PGT_DEVICES=0,1,..: the text is cut at line starts into one piece per GPU, every GPU reduces
// its block of the window table (pgt_plan_shards) from its gathered slice of the two columns; the rows are integers and a
// maximum, so they are the single-GPU run's bit for bit.  An input that one pass cannot hold (resident_limit: text + columns
// against the free device memory) is REFUSED before anything is printed — these tools have no passes mode: a selscan *.norm
// table has one line per SNP, 288 GB hold more than 4 * 10^9 of them.
inline int run_extreme(const char *path, bool skip_header, int score_field, uint32_t W, int mode, double cutoff,
                       const char *chrlen_path, const char *open_error) {
    PhaseTimer timer;
    Text text;
    if (!text.open(path)) die(std::string(open_error) + path);
    DeviceOpener device;
    const bool multi = device.count() > 1;
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
    const char *what = "cannot parse '<chr>_<id> pos ... score ...'";
    // id pos f0 .. f<score_field>: the score is the (score_field + 1)-th token behind the position
    uint8_t spec[12] = {PGT_TOK_CHR_PREFIX, PGT_TOK_U32};
    const int score_tok = 2 + score_field, n_tokens = score_tok + 1;
    for (int k = 2; k < score_tok; ++k) spec[k] = PGT_TOK_SKIP;
    spec[score_tok] = PGT_TOK_F64;
    // (several GPUs: each holds its piece of the text and its slice of the columns)
    // The limit is PER GPU (PGT_MAX_RESIDENT_SITES, or what the free memory holds of a per-GPU share of this text) and is
    // compared with the table's REAL line count: a limit the table stays under refuses nothing (round 5: any positive
    // PGT_MAX_RESIDENT_SITES used to refuse every table).
    if (const uint64_t per_gpu = resident_limit(b, b + (size_t)(text.end() - b) / device.count(), 4 + 8, [&] { return device.get(); })) {
        uint64_t lines = (uint64_t)std::count(b, text.end(), '\n');
        if (text.end() > b && text.end()[-1] != '\n') ++lines;
        const uint64_t gpus = device.count();
        const uint64_t fits = per_gpu > UINT64_MAX / gpus ? UINT64_MAX : per_gpu * gpus;
        if (lines > fits) {
            const char *env = std::getenv("PGT_MAX_RESIDENT_SITES");
            die("libpgtwin: the table has " + std::to_string(lines) + " lines, more than the " + std::to_string(fits) + " SNPs that fit (" +
                (env ? "PGT_MAX_RESIDENT_SITES=" + std::string(env) : std::string("the free memory")) + " x " + std::to_string(gpus) +
                " GPU" + (gpus > 1 ? "s" : "") + "), and the extreme-score tools have no passes mode (give more GPUs with PGT_DEVICES=0,1,..)");
        }
    }

    ScoreTable tab;
    tab.score_field = score_field;
    Runs runs;
    size_t n = 0;
    DeviceTable dtab;
    std::vector<DevicePiece> pieces;
    bool on_device = false;
    if (gpu_ingest_wanted((size_t)(text.end() - b))) {
        pgt_ctx *c = device.get();
        timer.lap("wait for HIP");
        if (multi) {
            on_device = ingest_on_devices(device, b, text.end(), spec, n_tokens, what, path, pieces, runs, &n, first_line);
        } else {
            on_device = ingest_on_device(c, b, text.end(), spec, n_tokens, what, path, first_line, dtab, runs);
            n = dtab.n;
        }
        timer.lap(on_device ? "gpu parse" : "gpu parse (refused)");
        if (on_device) {  // the window rules walk the positions on the host: 4 B per site come back
            tab.pos.alloc(n);
            if (multi) {
                for (const DevicePiece &p : pieces)
                    check(pgt_ingest_download(p.ctx, p.ing, 1, tab.pos.data() + p.row0, (size_t)p.rows * sizeof(uint32_t)), p.ctx);
            } else if (n) {
                check(pgt_ingest_download(c, dtab.ing, 1, tab.pos.data(), n * sizeof(uint32_t)), c);
            }
            timer.lap("positions");
        }
    }
    if (!on_device) {
        device.plan_host_io(true, (uint64_t)(text.end() - b));  // the host parser's columns will be uploaded: staging ring (inputs from 32 MiB) + first-copy set-up beside the parse
        runs = Runs{};
        n = parse_table(b, text.end(), tab, runs, what, path, first_line);
        timer.lap("parse");
    }
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
    if (multi) {
        uint64_t longest = 1;  // every context gets the hint of the WHOLE table (levels built), not of its block
        for (const pgt_win &w : win) longest = std::max<uint64_t>(longest, w.hi - w.lo);
        reduce_on_devices<pgt_ext_row>(
            device, win, (uint32_t)std::min<uint64_t>(longest, 0xFFFFFFFFu), 0, pieces, {{1, sizeof(uint32_t)}, {score_tok, sizeof(double)}}, rows.data(),
            [&](pgt_ctx *c, uint64_t lo, uint64_t n_k, const pgt_win *w, size_t nw, pgt_ext_row *out, size_t) {
                return pgt_extreme_reduce(c, tab.pos.data() + lo, tab.score.data() + lo, n_k, mode, cutoff, w, nw, out);
            },
            [&](pgt_ctx *c, void *const *d, uint64_t n_k, const pgt_win *w, size_t nw, pgt_ext_row *out, size_t bytes) {
                return pgt_extreme_reduce_cols(c, static_cast<const uint32_t *>(d[0]), static_cast<const double *>(d[1]), n_k, mode, cutoff, w, nw,
                                               out, bytes);
            });
        free_pieces(pieces);
    } else if (on_device) {
        check(pgt_extreme_reduce_cols(ctx, dtab.col<uint32_t>(1), dtab.col<double>(score_tok), n, mode, cutoff, win.data(), n_win, rows.data(),
                                      rows.size() * sizeof(rows[0])), ctx);
    } else {
        check(pgt_extreme_reduce(ctx, tab.pos.data(), tab.score.data(), n, mode, cutoff, win.data(), n_win, rows.data()), ctx);
    }
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
