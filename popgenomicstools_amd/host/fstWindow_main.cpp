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
    std::string text;
    if (!slurp(argv[1], text)) die(std::string("Unable to open Fst variance components file ") + argv[1]);
    parse_window_args(argc, argv, W, S);

    Runs runs;
    std::vector<uint32_t> pos;
    std::vector<double> a, b;
    const size_t guess = text.size() / 24 + 16;
    pos.reserve(guess); a.reserve(guess); b.reserve(guess);
    Cursor c{text.data(), text.data() + text.size()};
    size_t line = 0;
    while (c.p < c.end) {
        ++line;
        c.skip_blank();
        if (c.at_eol()) break;  // the reference loop ends at the first empty line (fstWindow.cpp:125)
        auto chr = c.token();
        uint32_t p;
        double x, y;
        if (!to_u32(c.token(), p) || !to_f64(c.token(), x) || !to_f64(c.token(), y))
            die("fstWindow: cannot parse 'chr pos a b' on line " + std::to_string(line) + " of " + argv[1]);
        runs.add(chr.first, chr.second);
        pos.push_back(p); a.push_back(x); b.push_back(y);
        c.next_line();
    }
    std::string().swap(text);

    size_t n_win = 0;
    check(pgt_build_windows_sites(runs.len.data(), runs.len.size(), W, S, nullptr, 0, &n_win), nullptr);
    if (n_win == 0) return 0;
    std::vector<pgt_win> win(n_win);
    check(pgt_build_windows_sites(runs.len.data(), runs.len.size(), W, S, win.data(), win.size(), &n_win), nullptr);

    pgt_ctx *ctx = open_or_die();
    std::vector<pgt_fst_row> rows(n_win);
    check(pgt_fst_reduce(ctx, pos.data(), a.data(), b.data(), pos.size(), win.data(), n_win, rows.data()), ctx);
    pgt_close(ctx);

    static char obuf[1 << 20];
    std::setvbuf(stdout, obuf, _IOFBF, sizeof obuf);
    for (size_t i = 0; i < n_win; ++i)  // %g == std::ostream default formatting (fstWindow.cpp:88)
        std::printf("%s\t%u\t%u\t%u\t%g\t%u\n", runs.name[win[i].label_run].c_str(), rows[i].start, rows[i].end,
                    rows[i].mid, rows[i].fst, rows[i].n);
    return 0;
}
