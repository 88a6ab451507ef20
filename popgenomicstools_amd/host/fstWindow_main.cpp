// fstWindow (MI355X host) — sliding-window FST = Σa/Σb from ANGSD variance components.
// Same command line and TSV as the reference tool (fstWindow.cpp:23-35 usage, :37-67 arguments,
// :88 row format); the per-window reduction runs on the GPU through include/pgtwin.h.
//
//   fstWindow <variance component file> [window size (sites)] [step size (sites)]
//   input lines:  chr  pos  a  b          output: chr start end mid fst nsites
#include "host_common.h"

using namespace pgthost;

static void usage(unsigned W, unsigned S) {
    std::printf("\nUsage:\n"
                "fstWindow [ANGSD fst variance component file] [window size (number sites)] [step size (number sites)]\n"
                "default window size: %u\ndefault step size: %u\n\n"
                "Output:\n(1) chromosome\n(2) window start\n(3) window end\n(4) window midpoint position\n"
                "(5) Fst\n(6) Number sites in window\n\n", W, S);
}

int main(int argc, char **argv) {
    uint32_t W = 1, S = 1;  // fstWindow.cpp:161-162
    if (argc < 2) {
        usage(W, S);
        return 0;
    }
    PhaseTimer timer;
    {   // the reference opens the file before it looks at the other arguments (fstWindow.cpp:45-49)
        FILE *probe = std::fopen(argv[1], "rb");
        if (!probe) die(std::string("Unable to open Fst variance components file ") + argv[1]);
        std::fclose(probe);
    }
    parse_window_args(argc, argv, W, S);
    DeviceOpener device;  // HIP start-up runs beside the parse; PGT_DEVICES=0,1,..: one context and host thread per GPU
    const bool multi = device.count() > 1;
    std::vector<DevicePiece> pieces;  // multi-GPU device ingest: one parsed piece of the text per GPU

    // chr pos a b  (fstWindow.cpp:130,141), parsed in parallel chunks straight into the columns
    struct Table {
        Column<uint32_t> pos;
        Column<double> a, b;
        void alloc(size_t rows) { pos.alloc(rows); a.alloc(rows); b.alloc(rows); }
        bool parse_line(Cursor &c, size_t i, Runs &runs) {
            const Tok chr = c.token();
            if (!to_u32(c.token(), pos[i]) || !to_f64(c.token(), a[i]) || !to_f64(c.token(), b[i])) return false;
            runs.add(chr.first, chr.second);
            return true;
        }
    } tab;
    Runs runs;
    size_t n = 0;
    DeviceTable dtab;  // the table when it was parsed on the GPU
    Text text;         // the input text (not opened when the column cache answers)
    ColumnCache cache("fstWindow", argv[1]);  // only with PGT_COLUMN_CACHE=<dir>
    std::vector<ColumnCache::Col> cols = {{nullptr, sizeof(uint32_t)}, {nullptr, sizeof(double)}, {nullptr, sizeof(double)}};
    bool on_device = false;
    if (cache.load(n, runs, cols)) {
        device.plan_host_io(true);  // host columns will be uploaded: stage and warm up beside what is left to do
        tab.pos.borrow(static_cast<uint32_t *>(cols[0].data));
        tab.a.borrow(static_cast<double *>(cols[1].data));
        tab.b.borrow(static_cast<double *>(cols[2].data));
        timer.lap("cache map");
    } else {
        if (!text.open(argv[1])) die(std::string("Unable to open Fst variance components file ") + argv[1]);
        const char *what = "fstWindow: cannot parse 'chr pos a b'";
        static const uint8_t spec[] = {PGT_TOK_CHR, PGT_TOK_U32, PGT_TOK_F64, PGT_TOK_F64};
        if (const uint64_t resident = resident_limit(text.begin(), text.end(), 4 + 8 + 8, [&] { return device.get(); })) {
            // larger than the GPU (or PGT_MAX_RESIDENT_SITES): block by block, rows printed as the blocks finish
            const size_t row_bytes_max = 80;
            reduce_in_passes<pgt_fst_row>(
                device, text.begin(), text.end(), W, S, resident, runs, timer,
                [&](pgt_ctx *c, const char *pb, const char *pe, uint64_t first_row, uint64_t n_rows, const pgt_win *w, size_t nw, pgt_fst_row *out, std::string *error) {
                    DeviceTable piece;
                    Runs piece_runs;
                    if (ingest_on_device(c, pb, pe, spec, 4, what, argv[1], first_row + 1, piece, piece_runs, error)) {
                        if (error && !error->empty()) return;
                        if (piece.n != n_rows) die("fstWindow: a pass parsed another number of rows than the first scan counted");
                        if (nw) check(pgt_fst_reduce_cols(c, piece.col<uint32_t>(1), piece.col<double>(2), piece.col<double>(3), piece.n, w, nw, out,
                                                  nw * sizeof(*out)), c);
                    } else {
                        decltype(tab) t;
                        const size_t k = parse_table(pb, pe, t, piece_runs, what, argv[1], first_row + 1, error);
                        if (error && !error->empty()) return;
                        if (k != n_rows) die("fstWindow: a pass parsed another number of rows than the first scan counted");
                        if (nw) check(pgt_fst_reduce(c, t.pos.data(), t.a.data(), t.b.data(), k, w, nw, out), c);
                    }
                },
                [&](const pgt_fst_row *r, size_t nw, const pgt_win *w) {
                    write_rows(nw, longest_name(runs) + row_bytes_max, [&](size_t i, char *o) {
                        return put_row(o, runs.name[w[i].label_run], {r[i].start, r[i].end, r[i].mid}, r[i].fst, {r[i].n});
                    });
                });
            finish(timer);
        }
        bool parsed = false;  // by the hybrid path, into the host table (the data ended inside its head)
        if (gpu_ingest_wanted(text.size()) && !multi) {  // large inputs: head on the host beside HIP start-up, tail on the GPU
            const int h = ingest_hybrid(device, text.begin(), text.end(), spec, 4, what, argv[1], tab,
                                        [](decltype(tab) &t) {
                                            return std::vector<HybridColumn>{{1, sizeof(uint32_t), t.pos.data()}, {2, sizeof(double), t.a.data()},
                                                                             {3, sizeof(double), t.b.data()}};
                                        }, dtab, runs, &n, timer);
            on_device = h == 1;
            parsed = h == 2;
        }
        if (gpu_ingest_wanted(text.size()) && !on_device && !parsed) {
            pgt_ctx *c = device.get();
            timer.lap("wait for HIP");
            if (multi) {
                on_device = ingest_on_devices(device, text.begin(), text.end(), spec, 4, what, argv[1], pieces, runs, &n);
            } else {
                on_device = ingest_on_device(c, text.begin(), text.end(), spec, 4, what, argv[1], 1, dtab, runs);
                n = dtab.n;
            }
            timer.lap(on_device ? "gpu parse" : "gpu parse (refused)");
        }
        if (!on_device && !parsed) {
            device.plan_host_io(true, text.size());  // the host parser's columns will be uploaded: staging ring (inputs from 32 MiB) + first-copy set-up beside the parse
            n = parse_table(text.begin(), text.end(), tab, runs, what, argv[1], 1);
            timer.lap("parse");
            if (cache.enabled()) {
                cols[0].data = tab.pos.data(); cols[1].data = tab.a.data(); cols[2].data = tab.b.data();
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
    RowArray<pgt_fst_row> rows(n_win);
    timer.lap("wait for HIP");
    set_site_hints(ctx, W, S);  // the strategy follows the tool's arguments, on one GPU as on several
    const uint32_t *pos = on_device && !multi ? dtab.col<uint32_t>(1) : tab.pos.data();
    const double *a = on_device && !multi ? dtab.col<double>(2) : tab.a.data(), *b = on_device && !multi ? dtab.col<double>(3) : tab.b.data();
    if (multi) {
        reduce_on_devices<pgt_fst_row>(
            device, sw.win, W, S, pieces, {{1, sizeof(uint32_t)}, {2, sizeof(double)}, {3, sizeof(double)}}, rows.data(),
            [&](pgt_ctx *c, uint64_t lo, uint64_t n_k, const pgt_win *w, size_t nw, pgt_fst_row *out, size_t) {
                return pgt_fst_reduce(c, pos + lo, a + lo, b + lo, n_k, w, nw, out);
            },
            [&](pgt_ctx *c, void *const *d, uint64_t n_k, const pgt_win *w, size_t nw, pgt_fst_row *out, size_t bytes) {
                return pgt_fst_reduce_cols(c, static_cast<const uint32_t *>(d[0]), static_cast<const double *>(d[1]),
                                           static_cast<const double *>(d[2]), n_k, w, nw, out, bytes);
            });
        free_pieces(pieces);
    } else if (sw.tab)
        check(pgt_fst_reduce_tab(ctx, pos, a, b, n, on_device, sw.tab, rows.data(), rows.size() * sizeof(rows[0])), ctx);
    else if (on_device)
        check(pgt_fst_reduce_cols(ctx, pos, a, b, n, sw.win.data(), n_win, rows.data(), rows.size() * sizeof(rows[0])), ctx);
    else
        check(pgt_fst_reduce(ctx, pos, a, b, n, sw.win.data(), n_win, rows.data()), ctx);
    timer.lap("gpu reduce");

    // chr start end mid fst nsites; %g == std::ostream default formatting (fstWindow.cpp:88)
    write_rows(n_win, longest_name(runs) + 80, [&](size_t i, char *o) {
        return put_row(o, runs.name[sw.label(i)], {rows[i].start, rows[i].end, rows[i].mid}, rows[i].fst, {rows[i].n});
    });
    finish(timer);
}
