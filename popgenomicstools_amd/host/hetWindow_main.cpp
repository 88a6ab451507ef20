// hetWindow (MI355X host) — sliding-window individual heterozygosity = #het / #non-missing.
// Same command line and TSV as the reference tool (hetWindow.cpp:20-32 usage, :34-64 arguments,
// :87 row format: column 6 is the NON-MISSING count); reduction on the GPU via include/pgtwin.h.
//
//   hetWindow <genotypes file> [window size (sites)] [step size (sites)]
//   input lines:  chr  pos  g   (g: 0/1/2, negative = missing)
#include <algorithm>

#include "host_common.h"

using namespace pgthost;

static void usage(unsigned W, unsigned S) {
    std::printf("\nUsage:\n"
                "hetWindow [genotypes file] [window size (number sites)] [step size (number sites)]\n"
                "default window size: %u\ndefault step size: %u\n\n"
                "Output:\n(1) chromosome\n(2) window start\n(3) window end\n(4) window midpoint position\n"
                "(5) heterozygosity\n(6) Number sites in window\n\n", W, S);
}

int main(int argc, char **argv) {
    uint32_t W = 1, S = 1;  // hetWindow.cpp:159-160
    if (argc < 2) {
        usage(W, S);
        return 0;
    }
    PhaseTimer timer;
    {   // the reference opens the file before it looks at the other arguments (hetWindow.cpp:42-46)
        FILE *probe = std::fopen(argv[1], "rb");
        if (!probe) die(std::string("Unable to open genotypes file ") + argv[1]);
        std::fclose(probe);
    }
    parse_window_args(argc, argv, W, S);
    DeviceOpener device;  // HIP start-up runs beside the parse; PGT_DEVICES=0,1,..: one context and host thread per GPU
    const bool multi = device.count() > 1;
    std::vector<DevicePiece> pieces;  // multi-GPU device ingest: one parsed piece of the text per GPU

    // chr pos genotype  (hetWindow.cpp:128,139), parsed in parallel chunks straight into the columns
    struct Table {
        Column<uint32_t> pos;
        Column<int8_t> g;
        void alloc(size_t rows) { pos.alloc(rows); g.alloc(rows); }
        bool parse_line(Cursor &c, size_t i, Runs &runs) {
            const Tok chr = c.token();
            long long v;
            if (!to_u32(c.token(), pos[i]) || !to_i64(c.token(), v)) return false;
            // only `>= 0` and `== 1` are ever tested (hetWindow.cpp:78-80): clamping to int8 keeps both
            g[i] = (int8_t)std::clamp<long long>(v, -128, 127);
            runs.add(chr.first, chr.second);
            return true;
        }
    } tab;
    Runs runs;
    size_t n = 0;
    DeviceTable dtab;  // the table when it was parsed on the GPU
    Text text;         // the input text (not opened when the column cache answers)
    ColumnCache cache("hetWindow", argv[1]);  // only with PGT_COLUMN_CACHE=<dir>
    std::vector<ColumnCache::Col> cols = {{nullptr, sizeof(uint32_t)}, {nullptr, sizeof(int8_t)}};
    bool on_device = false;
    if (cache.load(n, runs, cols)) {
        device.plan_host_io(true);  // host columns will be uploaded: stage and warm up beside what is left to do
        tab.pos.borrow(static_cast<uint32_t *>(cols[0].data));
        tab.g.borrow(static_cast<int8_t *>(cols[1].data));
        timer.lap("cache map");
    } else {
        if (!text.open(argv[1])) die(std::string("Unable to open genotypes file ") + argv[1]);
        const char *what = "hetWindow: cannot parse 'chr pos genotype'";
        static const uint8_t spec[] = {PGT_TOK_CHR, PGT_TOK_U32, PGT_TOK_I8};
        if (const uint64_t resident = resident_limit(text.begin(), text.end(), 4 + 1, [&] { return device.get(); })) {
            // larger than the GPU (or PGT_MAX_RESIDENT_SITES): block by block, rows printed as the blocks finish
            reduce_in_passes<pgt_het_row>(
                device, text.begin(), text.end(), W, S, resident, runs, timer,
                [&](pgt_ctx *c, const char *pb, const char *pe, uint64_t first_row, uint64_t n_rows, const pgt_win *w, size_t nw, pgt_het_row *out, std::string *error) {
                    DeviceTable piece;
                    Runs piece_runs;
                    if (ingest_on_device(c, pb, pe, spec, 3, what, argv[1], first_row + 1, piece, piece_runs, error)) {
                        if (error && !error->empty()) return;
                        if (piece.n != n_rows) die("hetWindow: a pass parsed another number of rows than the first scan counted");
                        if (nw) check(pgt_het_reduce_cols(c, piece.col<uint32_t>(1), piece.col<int8_t>(2), piece.n, w, nw, out, nw * sizeof(*out)), c);
                    } else {
                        decltype(tab) t;
                        const size_t k = parse_table(pb, pe, t, piece_runs, what, argv[1], first_row + 1, error);
                        if (error && !error->empty()) return;
                        if (k != n_rows) die("hetWindow: a pass parsed another number of rows than the first scan counted");
                        if (nw) check(pgt_het_reduce(c, t.pos.data(), t.g.data(), k, w, nw, out), c);
                    }
                },
                [&](const pgt_het_row *r, size_t nw, const pgt_win *w) {
                    write_rows(nw, longest_name(runs) + 80, [&](size_t i, char *o) {
                        return put_row(o, runs.name[w[i].label_run], {r[i].start, r[i].end, r[i].mid}, r[i].h, {r[i].nonmissing});
                    });
                });
            finish(timer);
        }
        bool parsed = false;  // by the hybrid path, into the host table (the data ended inside its head)
        if (gpu_ingest_wanted(text.size()) && !multi) {  // large inputs: head on the host beside HIP start-up, tail on the GPU
            const int h = ingest_hybrid(device, text.begin(), text.end(), spec, 3, what, argv[1], tab,
                                        [](decltype(tab) &t) {
                                            return std::vector<HybridColumn>{{1, sizeof(uint32_t), t.pos.data()}, {2, sizeof(int8_t), t.g.data()}};
                                        }, dtab, runs, &n, timer);
            on_device = h == 1;
            parsed = h == 2;
        }
        if (gpu_ingest_wanted(text.size()) && !on_device && !parsed) {
            pgt_ctx *c = device.get();
            timer.lap("wait for HIP");
            if (multi) {
                on_device = ingest_on_devices(device, text.begin(), text.end(), spec, 3, what, argv[1], pieces, runs, &n);
            } else {
                on_device = ingest_on_device(c, text.begin(), text.end(), spec, 3, what, argv[1], 1, dtab, runs);
                n = dtab.n;
            }
            timer.lap(on_device ? "gpu parse" : "gpu parse (refused)");
        }
        if (!on_device && !parsed) {
            device.plan_host_io(true, text.size());  // the host parser's columns will be uploaded: staging ring (inputs from 32 MiB) + first-copy set-up beside the parse
            n = parse_table(text.begin(), text.end(), tab, runs, what, argv[1], 1);
            timer.lap("parse");
            if (cache.enabled()) {
                cols[0].data = tab.pos.data(); cols[1].data = tab.g.data();
                cache.store(n, runs, cols);
                timer.lap("cache write");
            }
        }
    }

    SiteWindows sw;
    sw.build(runs, W, S, [&] { return device.get(); }, &timer, multi);
    const size_t n_win = sw.n;
    if (n_win == 0) return 0;

    timer.lap("window table");
    pgt_ctx *ctx = device.get();
    RowArray<pgt_het_row> rows(n_win);
    timer.lap("wait for HIP");
    set_site_hints(ctx, W, S);  // the strategy follows the tool's arguments, on one GPU as on several
    const uint32_t *pos = on_device && !multi ? dtab.col<uint32_t>(1) : tab.pos.data();
    const int8_t *g = on_device && !multi ? dtab.col<int8_t>(2) : tab.g.data();
    if (multi) {
        reduce_on_devices<pgt_het_row>(
            device, sw.win, W, S, pieces, {{1, sizeof(uint32_t)}, {2, sizeof(int8_t)}}, rows.data(),
            [&](pgt_ctx *c, uint64_t lo, uint64_t n_k, const pgt_win *w, size_t nw, pgt_het_row *out, size_t) {
                return pgt_het_reduce(c, pos + lo, g + lo, n_k, w, nw, out);
            },
            [&](pgt_ctx *c, void *const *d, uint64_t n_k, const pgt_win *w, size_t nw, pgt_het_row *out, size_t bytes) {
                return pgt_het_reduce_cols(c, static_cast<const uint32_t *>(d[0]), static_cast<const int8_t *>(d[1]), n_k, w, nw, out, bytes);
            });
        free_pieces(pieces);
    } else if (sw.tab)
        check(pgt_het_reduce_tab(ctx, pos, g, n, on_device, sw.tab, rows.data(), rows.size() * sizeof(rows[0])), ctx);
    else if (on_device)
        check(pgt_het_reduce_cols(ctx, pos, g, n, sw.win.data(), n_win, rows.data(), rows.size() * sizeof(rows[0])), ctx);
    else
        check(pgt_het_reduce(ctx, pos, g, n, sw.win.data(), n_win, rows.data()), ctx);
    timer.lap("gpu reduce");

    // chr start end mid h nonmissing (hetWindow.cpp:87)
    write_rows(n_win, longest_name(runs) + 80, [&](size_t i, char *o) {
        return put_row(o, runs.name[sw.label(i)], {rows[i].start, rows[i].end, rows[i].mid}, rows[i].h, {rows[i].nonmissing});
    });
    finish(timer);
}
