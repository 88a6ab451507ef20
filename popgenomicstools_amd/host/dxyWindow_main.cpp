// dxyWindow (MI355X host) — sliding-window dxy from two ANGSD .mafs files (plain or gzip).
// Same options, defaults, messages and TSV as the reference tool (dxyWindow.cpp:34-61 help,
// :63-139 arguments, :190 row format, :429-433 genome-wide line); per-site dxy and all window
// sums run on the GPU through include/pgtwin.h.
//
//   dxyWindow [options] <pop1 maf file> <pop2 maf file>      (options come BEFORE the two files)
//
// Site synchronisation: the reference's catch-up loops (dxyWindow.cpp:315-331) are only well
// defined when both files list identical or nested site sets.  This host
// computes the intersection of the two files by (chromosome run, position), which is what the
// reference produces on that domain, and is defined outside it as well.
// PGT_DXY_SYNC=reference (an environment switch, the CLI surface is unchanged) replays those loops
// instead — pair_as_the_reference() below — so that inputs on which the reference terminates
// normally but oddly (a position coincidence across chromosomes, an extra Pop2 site at a
// chromosome end, no shared site) print the reference's bytes: INTEGRATION.md §3a.
#include <map>

#include "host_common.h"

using namespace pgthost;

static void help(unsigned W, unsigned S, int minind, int fixedsite, int skip_missing) {
    std::printf("\ndxyWindow [options] <pop1 maf file> <pop2 maf file>\n\nOptions:\n"
                "%-14s%-8sWindow size in base pairs (0 for global calculation) [%u]\n"
                "%-14s%-8sNumber of base pairs to progress window [%u]\n"
                "%-14s%-8sMinimum number of individuals in each population with data [%d]\n"
                "%-14s%-8s(1) Use fixed number of sites from MAF input for each window (window sizes may vary) or (0) constant window size [%d]\n"
                "%-14s%-8sTwo-column TSV file with each row having (1) chromsome name (2) chromosome size in base pairs\n"
                "%-14s%-8sDo not print windows with zero effective sites if INT=1 [%d]\n"
                "\nNotes:\n"
                "* -winsize 1 -stepsize 1 calculates per site dxy\n"
                "* -sizefile is REQUIRED(!) with -fixedsite 0 (the default)\n"
                "* Both input MAF files need to have the same chromosomes in the same order\n"
                "* Assumes SNPs are biallelic across populations\n"
                "* For global Dxy calculations only columns 4, 5, and 6 below are printed\n"
                "* Input MAF files can contain all sites (including monomorphic sites) or just variable sites\n"
                "* -fixedsite 1 -winsize 500 would for example ensure that all windows contain 500 SNPs\n"
                "\nOutput:\n(1) chromosome\n(2) Window start\n(3) Window end\n(4) dxy\n"
                "(5) number sites in MAF input that were analyzed\n"
                "(6) number of sites in MAF input that were skipped due to too few individuals\n\n",
                "-winsize", "INT", W, "-stepsize", "INT", S, "-minind", "INT", minind, "-fixedsite", "INT", fixedsite,
                "-sizefile", "FILE", "-skip_missing", "INT", skip_missing);
}

struct Maf {
    Runs runs;
    DeviceTable dev;  // set when the file was parsed on the GPU: pos / freq / nind are then tokens 1 / 5 / 6 there
    pgt_ctx *ctx = nullptr;  // ... the context of that GPU
    bool on_device = false;
    Column<uint32_t> pos;
    Column<double> freq;
    Column<int32_t> nind;
    size_t n = 0;
    void reset() {  // back to empty: the host parser fills it from scratch
        if (dev.ing) pgt_ingest_free(dev.ing);
        dev.ing = nullptr;
        dev.n = 0;
        on_device = false;
        runs = Runs{};
        n = 0;
    }
    void alloc(size_t rows) { pos.alloc(rows); freq.alloc(rows); nind.alloc(rows); }
    // chr pos major minor ref freq nind — only chr, pos, freq, nind are used (dxyWindow.cpp:141-153)
    bool parse_line(Cursor &c, size_t i, Runs &r) {
        const Tok chr = c.token();
        long long k;
        bool ok = to_u32(c.token(), pos[i]);
        c.token(); c.token(); c.token();  // major minor ref
        ok = ok && to_f64(c.token(), freq[i]) && to_i64(c.token(), k);
        // a frequency outside [0,1] would make dxy negative, which the reference neither counts nor
        // skips (dxyWindow.cpp:180-185): refuse it
        if (!ok || !(freq[i] >= 0.0 && freq[i] <= 1.0)) return false;
        nind[i] = (int32_t)std::max<long long>(std::min<long long>(k, INT32_MAX), INT32_MIN);
        r.add(chr.first, chr.second);
        return true;
    }
};

static const char *const kMafWhat = "dxyWindow: cannot parse MAF line (chr pos major minor ref freq nind, freq in [0,1])";
static const uint8_t kMafSpec[] = {PGT_TOK_CHR, PGT_TOK_U32, PGT_TOK_SKIP, PGT_TOK_SKIP, PGT_TOK_SKIP, PGT_TOK_FREQ, PGT_TOK_I32};

// the device path of read_maf (plain or gzipped text of at least 8 MiB, no column cache): only the position
// column comes back to the host (site synchronisation and the bp-window table work on it)
static bool read_maf_on_device(pgt_ctx *ctx, const Text &text, const char *path, Maf &m, std::string *error = nullptr) {
    Cursor hdr{text.begin(), text.end()};
    hdr.next_line();  // header (dxyWindow.cpp:284)
    if (!ingest_on_device(ctx, hdr.p, text.end(), kMafSpec, 7, kMafWhat, path, 2, m.dev, m.runs, error)) return false;
    if (error && !error->empty()) return true;
    m.n = m.dev.n;
    m.ctx = ctx;
    m.on_device = true;
    m.pos.alloc(m.n);
    check(pgt_ingest_download(ctx, m.dev.ing, 1, m.pos.data(), m.n * sizeof(uint32_t)), ctx);
    return true;
}
// the remaining columns of a file parsed on the GPU, for the host-side merge of differing site sets
static void fetch_columns(Maf &m) {
    if (!m.on_device) return;
    pgt_ctx *ctx = m.ctx;
    m.freq.alloc(m.n);
    m.nind.alloc(m.n);
    check(pgt_ingest_download(ctx, m.dev.ing, 5, m.freq.data(), m.n * sizeof(double)), ctx);
    check(pgt_ingest_download(ctx, m.dev.ing, 6, m.nind.data(), m.n * sizeof(int32_t)), ctx);
    m.on_device = false;
}

// error: a message instead of an exit, so that the caller reports Pop1's problems before Pop2's whichever thread
// met its problem first
// opened: the file's text when the caller has it already (NULL: open it here)
static void read_maf(const char *path, const char *which, Maf &m, ColumnCache &cache, std::string &error, const Text *opened = nullptr) {
    std::vector<ColumnCache::Col> cols = {{nullptr, sizeof(uint32_t)}, {nullptr, sizeof(double)}, {nullptr, sizeof(int32_t)}};
    if (cache.load(m.n, m.runs, cols)) {  // only with PGT_COLUMN_CACHE=<dir>; plain (not gzipped-by-name-only) regular files
        m.pos.borrow(static_cast<uint32_t *>(cols[0].data));
        m.freq.borrow(static_cast<double *>(cols[1].data));
        m.nind.borrow(static_cast<int32_t *>(cols[2].data));
        return;
    }
    Text own;
    if (!opened && !own.open(path)) {
        error = std::string("Unable to open ") + which + " MAF file: " + path;
        return;
    }
    const Text &text = opened ? *opened : own;
    Cursor hdr{text.begin(), text.end()};
    hdr.next_line();  // header (dxyWindow.cpp:284)
    m.n = parse_table(hdr.p, text.end(), m, m.runs, kMafWhat, path, 2, &error);
    if (!error.empty()) return;
    if (cache.enabled()) {
        cols[0].data = m.pos.data(); cols[1].data = m.freq.data(); cols[2].data = m.nind.data();
        cache.store(m.n, m.runs, cols);
    }
}

// ---- PGT_DXY_SYNC=reference: dxyWindow.cpp:315-331 replayed over the two parsed site lists ----------------------------
// The reference keeps one current line per file and, when chromosome or position differ, advances ONE of them:
//   Pop1 (`:317-323`) when the names agree and Pop1's position is smaller, or the names differ and Pop2's name is not the
//        chromosome of the last processed site — until the POSITIONS are equal (names are not looked at), or Pop1 ends;
//   Pop2 (`:324-330`) otherwise — while its position is SMALLER — and gives the whole run up unless the positions then agree.
// What it then processes is Pop1's line with Pop2's frequency and count beside it, under Pop1's chromosome name (`:332`).  A
// give-up ends the main loop exactly as the end of a file does (`:323,329` break to `:406`), so the reference's output is that
// of its window machine on the pairs processed so far: the list this function returns.  "getline fails" is "no further
// parsed line" here (both parsers stop at the first empty line as `while (!maf1line.empty())` does, `:313`).
// -> the pairs (index in file 1, index in file 2); `last_chr`: the chromosome the closing code (`:407-426`) works on —
// the last pair's, or the first line's when nothing was paired.
static std::vector<std::pair<size_t, size_t>> pair_as_the_reference(const Maf &m1, const Maf &m2, std::string &last_chr) {
    auto run_of = [](const Runs &r) {  // site index -> run index, by a cursor that only moves forward
        return [&r, run = (size_t)0, end = (size_t)(r.len.empty() ? 0 : r.len[0])](size_t i) mutable {
            while (i >= end && run + 1 < r.len.size()) end += r.len[++run];
            return run;
        };
    };
    auto r1 = run_of(m1.runs), r2 = run_of(m2.runs);
    std::vector<std::pair<size_t, size_t>> pairs;
    size_t i = 0, j = 0;
    std::string chr = m1.runs.name[0];
    for (;;) {
        const std::string &c1 = m1.runs.name[r1(i)], &c2 = m2.runs.name[r2(j)];
        if (m1.pos[i] != m2.pos[j] || c1 != c2) {  // :316
            if ((c1 == c2 && m1.pos[i] < m2.pos[j]) || (c1 != c2 && c2 != chr)) {  // :317
                while (m1.pos[i] != m2.pos[j] && i + 1 < m1.n) ++i;  // :319-322
                if (m1.pos[i] != m2.pos[j]) break;                     // :323
            } else {
                while (m2.pos[j] < m1.pos[i] && j + 1 < m2.n) ++j;    // :326-329
                if (m1.pos[i] != m2.pos[j]) break;                     // :330
            }
        }
        chr = m1.runs.name[r1(i)];  // :332
        pairs.emplace_back(i, j);
        if (i + 1 >= m1.n) break;   // :399
        ++i;
        if (j + 1 >= m2.n) break;   // :402
        ++j;
    }
    last_chr = chr;
    return pairs;
}

// The reference's closing code on a chromosome of `len` base pairs of which NO site was processed (`:407-426` with nsites = 0,
// positer = 1): every slot is a placeholder, every window `chr start end 0 0 0` (case H10 of tests/golden/dxy_hand_walked.json).
static void print_placeholder_chromosome(const std::string &chr, uint64_t len, uint64_t W, uint64_t S, int skip_missing) {
    if (skip_missing) return;  // neffective == 0: the row is dropped (`:189`)
    uint64_t first = 1, n = 0, p = 1;
    while (p <= len) {
        if (n == W) {  // `:413`: the buffer is full before the next slot goes in
            std::printf("%s\t%llu\t%llu\t0\t0\t0\n", chr.c_str(), (unsigned long long)first, (unsigned long long)(first + W - 1));
            first += S;
            n = W - S;
        }
        const uint64_t take = std::min<uint64_t>(W - n, len - p + 1);
        n += take;
        p += take;
    }
    if (n > W - S && n <= W)  // `:424`
        std::printf("%s\t%llu\t%llu\t0\t0\t0\n", chr.c_str(), (unsigned long long)first, (unsigned long long)(first + n - 1));
}

// ---- several GPUs (PGT_DEVICES=0,1,...) ------------------------------------------------------------------------
// The window table is cut into one contiguous block per GPU (pgt_plan_shards) as for fstWindow; the genome-wide line
// (dxyWindow.cpp:382-385,429-433) has to see every site once, also sites no window covers, so the site axis is cut at the
// block starts into one OWNED range per GPU, and a GPU adds its owned range to its windows as extra windows of 65536 sites
// (block starts are multiples of 65536 on every GPU: a block's sum is taken over the same tree nodes wherever it is
// reduced).  The main thread adds the block rows in block order.  neff and nskip of the line are integers and exact; the sum
// is the single-GPU total to rounding (another association of the same additions; six digits are printed).
// Columns: on the host (every GPU uploads its slice over its own PCIe link), or where the two files were parsed — file 1 on
// the first GPU, file 2 on the second — from where a GPU copies its slice device to device.
struct DxyColumns {
    const uint32_t *pos = nullptr;
    const double *p1 = nullptr, *p2 = nullptr;
    const int32_t *n1 = nullptr, *n2 = nullptr;
    pgt_ctx *ctx1 = nullptr, *ctx2 = nullptr;  // set: device pointers of these contexts (pos, p1, n1 on ctx1; p2, n2 on ctx2)
};
constexpr uint64_t kTotalBlock = 65536;

// The plan of N blocks: window blocks (pgt_plan_shards), owned site ranges [cut[k], cut[k+1]) and the hints of the whole table
struct DxyBlockPlan {
    std::vector<pgt_shard> shard;
    std::vector<uint64_t> cut;
    uint64_t h_max = 0, h_typical = 0, h_step = 0;
};
static DxyBlockPlan plan_dxy_blocks(const std::vector<pgt_win> &win, uint64_t n_sites, size_t N) {
    DxyBlockPlan pl;
    pl.shard.resize(N);
    check(pgt_plan_shards(win.data(), win.size(), (uint32_t)N, pl.shard.data()), nullptr);
    check(pgt_table_hints(win.data(), win.size(), &pl.h_max, &pl.h_typical, &pl.h_step), nullptr);  // every block is reduced as the whole table would be
    pl.cut.assign(N + 1, 0);
    pl.cut[N] = n_sites;
    for (size_t k = N - 1; k >= 1; --k) {
        const bool has = pl.shard[k].win_end > pl.shard[k].win_begin;
        if (win.empty()) pl.cut[k] = std::min<uint64_t>(n_sites / N * k / kTotalBlock * kTotalBlock, pl.cut[k + 1]);  // global dxy only
        else pl.cut[k] = has ? std::min<uint64_t>(pl.shard[k].site_lo, pl.cut[k + 1]) : pl.cut[k + 1];
        if (pl.cut[k] % kTotalBlock != 0 && pl.cut[k] != n_sites) die("pgt_plan_shards returned a block start that is not a multiple of 65536");
    }
    return pl;
}

// Block k on ctx: its windows + the 65536-site blocks of its owned range, re-based to its first site, through
// reduce(first site, sites, windows, n, rows out).  false: the block is empty.  out = the window rows, then the block rows.
template <class Reduce>
static bool run_dxy_block(const DxyBlockPlan &pl, size_t k, const std::vector<pgt_win> &win, pgt_ctx *ctx, Reduce reduce,
                          std::vector<pgt_dxy_row> &out, size_t &n_own) {
    const pgt_shard sh = pl.shard[k];
    n_own = (size_t)(sh.win_end - sh.win_begin);
    const uint64_t b0 = pl.cut[k] / kTotalBlock, b1 = pl.cut[k + 1] > pl.cut[k] ? (pl.cut[k + 1] + kTotalBlock - 1) / kTotalBlock : b0;
    const size_t n_blocks = (size_t)(b1 - b0);
    if (n_own + n_blocks == 0) return false;
    const uint64_t lo = pl.cut[k], hi = std::max<uint64_t>(n_own ? sh.site_hi : lo, pl.cut[k + 1]);
    std::vector<pgt_win> local(n_own + n_blocks);
    std::copy(win.begin() + (ptrdiff_t)sh.win_begin, win.begin() + (ptrdiff_t)sh.win_end, local.begin());
    for (size_t i = 0; i < n_blocks; ++i) {
        pgt_win &w = local[n_own + i];
        w.lo = (b0 + i) * kTotalBlock;
        w.hi = std::min<uint64_t>(w.lo + kTotalBlock, pl.cut[k + 1]);
        w.label_run = 0;
        w.flags = PGT_WIN_COORDS;  // no coordinates to look up
        w.start = w.end = 0;
    }
    for (pgt_win &w : local) { w.lo -= lo; w.hi -= lo; }
    check(pgt_set_max_window(ctx, pl.h_max), ctx);
    check(pgt_set_typical_window(ctx, pl.h_typical), ctx);
    check(pgt_set_window_step(ctx, pl.h_step), ctx);
    out.resize(local.size());
    reduce(lo, hi - lo, local.data(), local.size(), out.data());
    return true;
}
static void add_block_rows(pgt_dxy_total &tot, const pgt_dxy_row *b, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        tot.sum += b[i].sum;
        tot.neff += b[i].neff;
        tot.nskip += b[i].nskip;
    }
}

static void reduce_dxy_on_devices(DeviceOpener &device, const std::vector<pgt_win> &win, const DxyColumns &col, uint64_t n_sites,
                                  int minind, pgt_dxy_row *rows, pgt_dxy_total *tot) {
    const size_t N = device.count();
    const DxyBlockPlan pl = plan_dxy_blocks(win, n_sites, N);
    for (size_t k = 0; k < N; ++k) (void)device.get(k);
    std::vector<std::vector<pgt_dxy_row>> blocks(N);
    std::vector<std::thread> th;
    for (size_t k = 0; k < N; ++k)
        th.emplace_back([&, k] {
            pgt_ctx *ctx = device.get(k);
            std::vector<pgt_dxy_row> out;
            size_t n_own = 0;
            const bool any = run_dxy_block(pl, k, win, ctx, [&](uint64_t lo, uint64_t n_k, const pgt_win *w, size_t nw, pgt_dxy_row *o) {
                if (!col.ctx1) {
                    check(pgt_dxy_reduce(ctx, col.pos + lo, col.p1 + lo, col.p2 + lo, col.n1 + lo, col.n2 + lo, n_k, minind, w, nw, o, nullptr), ctx);
                    return;
                }
                const struct { const void *src; pgt_ctx *from; size_t elem; } part[5] = {
                    {col.pos, col.ctx1, 4}, {col.p1, col.ctx1, 8}, {col.p2, col.ctx2, 8}, {col.n1, col.ctx1, 4}, {col.n2, col.ctx2, 4}};
                void *d[5] = {};
                for (int c = 0; c < 5; ++c) {
                    check(pgt_dev_alloc(ctx, (size_t)n_k * part[c].elem + 16, &d[c]), ctx);
                    check(pgt_dev_copy(ctx, d[c], part[c].from, static_cast<const char *>(part[c].src) + lo * part[c].elem,
                                       (size_t)n_k * part[c].elem), ctx);
                }
                check(pgt_dxy_reduce_cols(ctx, static_cast<uint32_t *>(d[0]), static_cast<double *>(d[1]), static_cast<double *>(d[2]),
                                          static_cast<int32_t *>(d[3]), static_cast<int32_t *>(d[4]), n_k, minind, w, nw, o, nw * sizeof(*o), nullptr), ctx);
                for (void *p : d) check(pgt_dev_free(ctx, p), ctx);
            }, out, n_own);
            if (!any) return;
            std::copy(out.begin(), out.begin() + (ptrdiff_t)n_own, rows + pl.shard[k].win_begin);
            blocks[k].assign(out.begin() + (ptrdiff_t)n_own, out.end());
        });
    for (auto &t : th) t.join();
    *tot = pgt_dxy_total{};
    for (size_t k = 0; k < N; ++k) add_block_rows(*tot, blocks[k].data(), blocks[k].size());  // GPUs in order = blocks in order
}

// ---- two MAF files larger than the GPU: in passes (PGT_MAX_RESIDENT_SITES, or decided from the free memory) ----------------
// As for fstWindow (host_common.h: reduce_in_passes), with the blocks of the several-GPU path above run one after the other
// on the first GPU: a first scan of both texts for runs, row marks and a digest of the position column per 65536 rows; the
// passes need both files to list the SAME sites (run for run and position for position — both known from that scan, before
// anything is printed; files whose site lists differ go through the host merge, which holds everything: -> false, the
// resident path runs).  Base-pair windows need every position
// before the first window is known: one extra pass over file 1 that keeps only its position column (4 B per site on the
// host).  Then per block: the text of its rows of both files -> device parser -> reduce -> its rows printed; the genome-wide
// line from the blocks' 65536-site rows, in order, as on several GPUs.
static bool dxy_in_passes(DeviceOpener &device, const Text &t1, const Text &t2, const char *path1, const char *path2, uint32_t W, uint32_t S,
                          int minind, int fixedsite, int skip_missing, const std::map<std::string, uint32_t> &chrsize,
                          uint64_t max_resident, PhaseTimer &timer) {
    struct File { const char *b, *e, *end; Runs runs; std::vector<const char *> mark; std::vector<uint64_t> pos_digest; size_t n; const char *path; } f[2];
    const Text *texts[2] = {&t1, &t2};
    const char *paths[2] = {path1, path2};
    for (int i = 0; i < 2; ++i) {
        Cursor hdr{texts[i]->begin(), texts[i]->end()};
        hdr.next_line();  // header (dxyWindow.cpp:284)
        f[i].b = hdr.p;
        f[i].e = texts[i]->end();
        f[i].path = paths[i];
        f[i].n = scan_runs_and_marks(f[i].b, f[i].e, f[i].runs, f[i].mark, &f[i].end, &f[i].pos_digest);
    }
    timer.lap("scan runs");
    // anything but two files with the same runs goes through the resident path — it parses both files completely before it
    // looks at their chromosomes, so a bad line is reported before "Chromosomes in MAF files differ", as the fuzzer insists
    if (f[0].n == 0 || f[1].n == 0 || f[0].n != f[1].n || f[0].runs.name != f[1].runs.name || f[0].runs.len != f[1].runs.len) return false;
    // ... and so does a pair with the same runs but other POSITIONS (the reference's sync loop, dxyWindow.cpp:315-331, and the
    // resident host merge accept such input): decided here, from the scan's per-block digests of the position column, before
    // a single row is on stdout — not block by block in the middle of the run
    if (f[0].pos_digest != f[1].pos_digest) return false;
    const uint64_t n = f[0].n;
    const Runs &runs = f[0].runs;
    pgt_ctx *ctx = device.get();
    timer.lap("wait for HIP");
    const uint64_t last_mark = (n + kMarkEvery - 1) / kMarkEvery;
    // rows [row0, row1) of file i (row0 a multiple of 65536) parsed: on the GPU, or by the host parser where the device refuses
    struct Piece { Maf m; uint64_t rows = 0; };
    auto parse_piece = [&](int i, uint64_t row0, uint64_t row1, Piece &pc) {
        const uint64_t m1 = std::min<uint64_t>((row1 + kMarkEvery - 1) / kMarkEvery, last_mark);
        const char *pb = f[i].mark[row0 / kMarkEvery], *pe = f[i].mark[m1];
        pc.rows = std::min<uint64_t>(m1 * kMarkEvery, n) - row0;
        if (ingest_on_device(ctx, pb, pe, kMafSpec, 7, kMafWhat, f[i].path, row0 + 2, pc.m.dev, pc.m.runs)) {
            pc.m.n = pc.m.dev.n;
            pc.m.ctx = ctx;
            pc.m.on_device = true;
        } else {
            pc.m.n = parse_table(pb, pe, pc.m, pc.m.runs, kMafWhat, f[i].path, row0 + 2);
        }
        if (pc.m.n != pc.rows) die("dxyWindow: a pass parsed another number of rows than the first scan counted");
    };
    const uint64_t per_pass = std::max<uint64_t>(max_resident, 4 * kMarkEvery) / kMarkEvery * kMarkEvery;
    std::vector<uint32_t> pos_all;  // base-pair windows only
    std::vector<pgt_win> win;
    if (W > 0) {
        size_t n_win = 0;
        if (fixedsite) {
            check(pgt_build_windows_sites(runs.len.data(), runs.len.size(), W, S, nullptr, 0, &n_win), nullptr);
            win.resize(n_win);
            if (n_win) check(pgt_build_windows_sites(runs.len.data(), runs.len.size(), W, S, win.data(), win.size(), &n_win), nullptr);
        } else {
            pos_all.resize(n);
            for (uint64_t row0 = 0; row0 < n; row0 += per_pass) {
                Piece pc;
                parse_piece(0, row0, std::min<uint64_t>(row0 + per_pass, n), pc);
                if (pc.m.on_device) check(pgt_ingest_download(ctx, pc.m.dev.ing, 1, pos_all.data() + row0, pc.rows * sizeof(uint32_t)), ctx);
                else std::memcpy(pos_all.data() + row0, pc.m.pos.data(), pc.rows * sizeof(uint32_t));
            }
            timer.lap("positions");
            std::vector<uint32_t> chr_len(runs.name.size());
            for (size_t r = 0; r < runs.name.size(); ++r) {
                auto it = chrsize.find(runs.name[r]);
                if (it == chrsize.end()) die("Unable to determine size for " + runs.name[r]);  // dxyWindow.cpp:340-343
                chr_len[r] = it->second;
            }
            check(pgt_build_windows_bp(pos_all.data(), runs.len.data(), chr_len.data(), runs.len.size(), W, S, nullptr, 0, &n_win), nullptr);
            win.resize(n_win);
            if (n_win) check(pgt_build_windows_bp(pos_all.data(), runs.len.data(), chr_len.data(), runs.len.size(), W, S, win.data(), win.size(), &n_win), nullptr);
        }
    }
    const size_t passes = (size_t)std::min<uint64_t>((n + per_pass - 1) / per_pass + (win.empty() ? 0 : 1), 1u << 20);
    const DxyBlockPlan pl = plan_dxy_blocks(win, n, std::max<size_t>(passes, 1));
    timer.lap("window table");
    pgt_dxy_total tot{};
    std::vector<pgt_dxy_row> out;
    std::vector<uint32_t> pa, pb2;
    for (size_t k = 0; k < pl.shard.size(); ++k) {
        size_t n_own = 0;
        const bool any = run_dxy_block(pl, k, win, ctx, [&](uint64_t lo, uint64_t n_k, const pgt_win *w, size_t nw, pgt_dxy_row *o) {
            Piece a, b;
            parse_piece(0, lo, lo + n_k, a);
            parse_piece(1, lo, lo + n_k, b);
            // the same sites, position for position?
            const uint32_t *q[2] = {nullptr, nullptr};
            Piece *pcs[2] = {&a, &b};
            std::vector<uint32_t> *buf[2] = {&pa, &pb2};
            for (int i = 0; i < 2; ++i) {
                if (pcs[i]->m.on_device) {
                    buf[i]->resize(pcs[i]->rows);
                    check(pgt_ingest_download(ctx, pcs[i]->m.dev.ing, 1, buf[i]->data(), pcs[i]->rows * sizeof(uint32_t)), ctx);
                    q[i] = buf[i]->data();
                } else q[i] = pcs[i]->m.pos.data();
            }
            if (std::memcmp(q[0], q[1], a.rows * sizeof(uint32_t)) != 0)  // cannot happen after the digest comparison above
                die("dxyWindow: internal error: the position digests of the two MAF files agree but a block's parsed positions differ");
            if (a.m.on_device && b.m.on_device) {
                check(pgt_dxy_reduce_cols(ctx, a.m.dev.col<uint32_t>(1), a.m.dev.col<double>(5), b.m.dev.col<double>(5), a.m.dev.col<int32_t>(6),
                                          b.m.dev.col<int32_t>(6), a.rows, minind, w, nw, o, nw * sizeof(*o), nullptr), ctx);
            } else {
                fetch_columns(a.m);
                fetch_columns(b.m);
                check(pgt_dxy_reduce(ctx, q[0], a.m.freq.data(), b.m.freq.data(), a.m.nind.data(), b.m.nind.data(), a.rows, minind, w, nw, o, nullptr), ctx);
            }
        }, out, n_own);
        if (!any) continue;
        const pgt_win *gw = win.data() + pl.shard[k].win_begin;
        // chr start end dxy neffective nskip, unless -skip_missing drops the row (dxyWindow.cpp:189-191)
        write_rows(n_own, longest_name(runs) + 80, [&](size_t i, char *o) -> size_t {
            if (!(out[i].neff > 0 || !skip_missing)) return 0;
            return put_row(o, runs.name[gw[i].label_run], {out[i].start, out[i].end}, out[i].sum, {out[i].neff, out[i].nskip});
        });
        add_block_rows(tot, out.data() + n_own, out.size() - n_own);
    }
    timer.lap("passes");
    std::fprintf(W == 0 ? stdout : stderr, "%g\t%llu\t%llu\n", tot.sum, (unsigned long long)tot.neff, (unsigned long long)tot.nskip);
    return true;
}

int main(int argc, char **argv) {
    uint32_t W = 0, S = 0;  // dxyWindow.cpp:529-534
    int minind = 1, fixedsite = 0, skip_missing = 0;
    const char *sizefile = nullptr;
    if (argc < 3) {
        help(W, S, minind, fixedsite, skip_missing);
        return 0;
    }
    // the last two arguments are the MAF files; option/value pairs precede them (dxyWindow.cpp:97-126)
    for (int i = 1; i < argc - 2; i += 2) {
        const char *opt = argv[i], *val = argv[i + 1];
        if (!std::strcmp(opt, "-winsize")) W = (uint32_t)std::atoi(val);
        else if (!std::strcmp(opt, "-stepsize")) S = (uint32_t)std::atoi(val);
        else if (!std::strcmp(opt, "-minind")) {
            minind = std::atoi(val);
            if (minind <= 0) die("-minind must be at least 1");
        } else if (!std::strcmp(opt, "-sizefile")) sizefile = val;
        else if (!std::strcmp(opt, "-fixedsite")) fixedsite = std::atoi(val);
        else if (!std::strcmp(opt, "-skip_missing")) skip_missing = std::atoi(val);
        else die(std::string("Unknown command: ") + opt);
    }
    if (W > 0 && S < 1) die("Must specify a -stepsize > 0 when -winsize is > 0");
    if (!fixedsite && !sizefile) die("Must supply size file unless -fixedsite 1");
    if (W > 0 && S > W) die("-stepsize must not exceed -winsize");                      // reference: crash (Q9)
    if (W == 0 && !fixedsite) die("-winsize 0 (global dxy) requires -fixedsite 1");      // reference: crash (Q10)

    std::map<std::string, uint32_t> chrsize;  // dxyWindow.cpp:155-170
    if (!fixedsite) {
        Text text;
        if (!text.open(sizefile)) die(std::string("Unable to open sizefile: ") + sizefile);
        Cursor c{text.begin(), text.end()};
        while (c.p < c.end) {
            auto name = c.token();
            uint32_t len = 0;
            if (name.first == name.second || !to_u32(c.token(), len) || len == 0)
                die("Unable to correctly parse chromosome size file");
            chrsize.insert({std::string(name.first, name.second), len});
            c.next_line();
        }
    }

    PhaseTimer timer;
    DeviceOpener device;  // HIP start-up runs beside the parse; PGT_DEVICES=0,1,..: one context and host thread per GPU
    const bool multi = device.count() > 1;
    Maf m1, m2;
    ColumnCache c1("dxyWindow maf", argv[argc - 2]), c2("dxyWindow maf", argv[argc - 1]);  // own the mappings the columns may borrow
    bool parsed = false;
    // Without the column cache both texts are opened here, side by side (a one-member .gz inflates on one thread,
    // a bgzf file on all of them), and serve whichever parser runs.  They are never unmapped or freed: giving
    // back gigabytes of text costs more than everything downstream of the parse, and the process ends by _exit.
    Text &t1 = *new Text, &t2 = *new Text;
    bool open1 = false, open2 = false;
    if (!c1.enabled()) {
        std::thread other([&] { open1 = t1.open(argv[argc - 2]); });
        open2 = t2.open(argv[argc - 1]);
        other.join();
        timer.lap("open");
        // inputs larger than the GPU (or PGT_MAX_RESIDENT_SITES): block by block — unless the site lists differ
        if (open1 && open2)
            if (const uint64_t resident = resident_limit(t1.begin(), t1.end(), 2 * (4 + 8 + 4), [&] { return device.get(); }, 2))
                if (dxy_in_passes(device, t1, t2, argv[argc - 2], argv[argc - 1], W, S, minind, fixedsite, skip_missing, chrsize, resident, timer))
                    finish(timer);
        // large inputs: parse both files on the GPU, one after the other (one context, one thread); a file that
        // cannot be opened is left to the host path below, which reports Pop1's problems first
        if (open1 && open2 && gpu_ingest_wanted(std::min(t1.size(), t2.size()))) {
            pgt_ctx *c = device.get();
            timer.lap("wait for HIP");
            if (multi) {  // one file per GPU, side by side: each text crosses its own PCIe link
                std::string e1, e2;
                bool ok1 = false, ok2 = false;
                pgt_ctx *c2 = device.get(1);
                std::thread other([&] { ok1 = read_maf_on_device(c, t1, argv[argc - 2], m1, &e1); });
                ok2 = read_maf_on_device(c2, t2, argv[argc - 1], m2, &e2);
                other.join();
                if (ok1 && !e1.empty()) die(e1);  // Pop1's problems first, as the reference meets them
                if (ok1 && ok2 && !e2.empty()) die(e2);
                parsed = ok1 && ok2;
            } else
                parsed = read_maf_on_device(c, t1, argv[argc - 2], m1) && read_maf_on_device(c, t2, argv[argc - 1], m2);
            if (!parsed) { m1.reset(); m2.reset(); }
            timer.lap(parsed ? "gpu parse" : "gpu parse (refused)");
        }
    }
    if (!parsed) {  // the two files are independent: parse them side by side on the host
        device.plan_host_io(true, open1 && open2 ? (uint64_t)t1.size() + t2.size() : 0);  // their columns will be uploaded: staging ring (inputs from 32 MiB) + first-copy set-up beside the parse
        std::string e1, e2;
        std::thread th([&] { read_maf(argv[argc - 2], "Pop1", m1, c1, e1, open1 ? &t1 : nullptr); });
        read_maf(argv[argc - 1], "Pop2", m2, c2, e2, open2 ? &t2 : nullptr);
        th.join();
        if (!e1.empty()) die(e1);
        if (!e2.empty()) die(e2);
        timer.lap("parse");
    }
    if (m1.n == 0 || m2.n == 0) die("dxyWindow: a MAF file holds no sites");
    if (m1.runs.name[0] != m2.runs.name[0]) die("Chromosomes in MAF files differ");  // dxyWindow.cpp:295-298

    // The sites common to both files, by (run, position).  Usual case first: both files list exactly the
    // same sites (ANGSD run on one site list) -> the parsed columns are used as they are, nothing is copied.
    Runs runs;
    std::vector<uint32_t> pos_v;
    std::vector<double> p1_v, p2_v;
    std::vector<int32_t> n1_v, n2_v;
    const uint32_t *pos = nullptr;
    const double *p1 = nullptr, *p2 = nullptr;
    const int32_t *n1 = nullptr, *n2 = nullptr;
    size_t n_sites = 0;
    const char *sync_env = std::getenv("PGT_DXY_SYNC");
    if (sync_env && std::strcmp(sync_env, "reference") && std::strcmp(sync_env, "intersection"))
        die("PGT_DXY_SYNC must be 'intersection' (the default) or 'reference'");
    const bool sync_as_reference = sync_env && !std::strcmp(sync_env, "reference");
    const bool same_sites = m1.n == m2.n && m1.runs.name == m2.runs.name && m1.runs.len == m2.runs.len &&
                            std::memcmp(m1.pos.data(), m2.pos.data(), m1.n * sizeof(uint32_t)) == 0;
    const bool on_device = same_sites && m1.on_device && m2.on_device;  // frequencies and counts stay on the GPU
    if (same_sites) {
        if (!on_device) { fetch_columns(m1); fetch_columns(m2); }
        runs = m1.runs;
        pos = m1.pos.data(); p1 = m1.freq.data(); p2 = m2.freq.data(); n1 = m1.nind.data(); n2 = m2.nind.data();
        n_sites = m1.n;
    } else if (sync_as_reference) {
        fetch_columns(m1);
        fetch_columns(m2);
        std::string last_chr;
        const auto pairs = pair_as_the_reference(m1, m2, last_chr);
        if (pairs.empty()) {  // no site processed at all: the closing code alone (`:407-433`)
            if (W > 0 && !fixedsite) {
                auto it = chrsize.find(last_chr);
                if (it == chrsize.end()) die("Unable to determine size for " + last_chr);  // dxyWindow.cpp:410-413
                print_placeholder_chromosome(last_chr, it->second, W, S, skip_missing);
            }
            std::fprintf(W == 0 ? stdout : stderr, "0\t0\t0\n");
            finish(timer);
        }
        size_t run1 = 0, end1 = m1.runs.len[0];
        for (const auto &pr : pairs) {  // Pop1's line, Pop2's frequency and count beside it, under Pop1's chromosome name
            while (pr.first >= end1) end1 += m1.runs.len[++run1];
            const std::string &chr1 = m1.runs.name[run1];
            pos_v.push_back(m1.pos[pr.first]);
            p1_v.push_back(m1.freq[pr.first]); p2_v.push_back(m2.freq[pr.second]);
            n1_v.push_back(m1.nind[pr.first]); n2_v.push_back(m2.nind[pr.second]);
            runs.add(chr1.data(), chr1.data() + chr1.size());
        }
        pos = pos_v.data(); p1 = p1_v.data(); p2 = p2_v.data(); n1 = n1_v.data(); n2 = n2_v.data();
        n_sites = pos_v.size();
    } else {
        fetch_columns(m1);
        fetch_columns(m2);
        size_t r1 = 0, r2 = 0, o1 = 0, o2 = 0;
        while (r1 < m1.runs.name.size() && r2 < m2.runs.name.size()) {
            const std::string &chr1 = m1.runs.name[r1], &chr2 = m2.runs.name[r2];
            if (chr1 != chr2) {  // skip the run that the other file does not have next
                bool later_in_1 = false;
                for (size_t k = r1 + 1; k < m1.runs.name.size() && !later_in_1; ++k) later_in_1 = m1.runs.name[k] == chr2;
                if (later_in_1) { o1 += m1.runs.len[r1]; ++r1; } else { o2 += m2.runs.len[r2]; ++r2; }
                continue;
            }
            size_t i = o1, j = o2;
            const size_t e1 = o1 + m1.runs.len[r1], e2 = o2 + m2.runs.len[r2];
            const size_t before = pos_v.size();
            while (i < e1 && j < e2) {
                if (m1.pos[i] < m2.pos[j]) ++i;
                else if (m2.pos[j] < m1.pos[i]) ++j;
                else {
                    pos_v.push_back(m1.pos[i]);
                    p1_v.push_back(m1.freq[i]); p2_v.push_back(m2.freq[j]);
                    n1_v.push_back(m1.nind[i]); n2_v.push_back(m2.nind[j]);
                    ++i; ++j;
                }
            }
            if (pos_v.size() > before) runs.add(chr1.data(), chr1.data() + chr1.size(), pos_v.size() - before);
            o1 = e1; o2 = e2; ++r1; ++r2;
        }
        pos = pos_v.data(); p1 = p1_v.data(); p2 = p2_v.data(); n1 = n1_v.data(); n2 = n2_v.data();
        n_sites = pos_v.size();
    }
    if (n_sites == 0) die("dxyWindow: the two MAF files share no site");

    SiteWindows sw;  // fixed-site windows: on the host, or on the device when there are very many (-stepsize 1)
    std::vector<pgt_win> &win = sw.win;
    if (W > 0) {
        size_t n_win = 0;
        if (fixedsite) {
            sw.build(runs, W, S, [&] { return device.get(); }, &timer, multi);  // several GPUs shard a host table
        } else {
            std::vector<uint32_t> chr_len(runs.name.size());
            for (size_t r = 0; r < runs.name.size(); ++r) {
                auto it = chrsize.find(runs.name[r]);
                if (it == chrsize.end()) die("Unable to determine size for " + runs.name[r]);  // dxyWindow.cpp:340-343
                chr_len[r] = it->second;
            }
            check(pgt_build_windows_bp(pos, runs.len.data(), chr_len.data(), runs.len.size(), W, S, nullptr, 0, &n_win), nullptr);
            win.resize(n_win);
            check(pgt_build_windows_bp(pos, runs.len.data(), chr_len.data(), runs.len.size(), W, S, win.data(), win.size(), &n_win), nullptr);
        }
    }

    timer.lap("sync + table");
    pgt_ctx *ctx = device.get();
    timer.lap("wait for HIP");
    const size_t n_rows = sw.tab ? sw.n : win.size();
    RowArray<pgt_dxy_row> rows(n_rows);
    pgt_dxy_total tot{};
    if (on_device) {  // frequencies and counts are on the GPU
        pos = m1.dev.col<uint32_t>(1); p1 = m1.dev.col<double>(5); p2 = m2.dev.col<double>(5);
        n1 = m1.dev.col<int32_t>(6); n2 = m2.dev.col<int32_t>(6);
    }
    if (multi) {
        DxyColumns col{pos, p1, p2, n1, n2, on_device ? m1.ctx : nullptr, on_device ? m2.ctx : nullptr};
        reduce_dxy_on_devices(device, win, col, n_sites, minind, rows.data(), &tot);
    } else if (sw.tab)
        check(pgt_dxy_reduce_tab(ctx, pos, p1, p2, n1, n2, n_sites, minind, on_device, sw.tab, rows.data(), rows.size() * sizeof(rows[0]), &tot), ctx);
    else if (on_device)
        check(pgt_dxy_reduce_cols(ctx, pos, p1, p2, n1, n2, n_sites, minind, win.data(), win.size(), rows.data(), rows.size() * sizeof(rows[0]), &tot), ctx);
    else
        check(pgt_dxy_reduce(ctx, pos, p1, p2, n1, n2, n_sites, minind, win.data(), win.size(), rows.data(), &tot), ctx);
    timer.lap("gpu reduce");

    // chr start end dxy neffective nskip, unless -skip_missing drops the row (dxyWindow.cpp:189-191)
    write_rows(n_rows, longest_name(runs) + 80, [&](size_t i, char *o) -> size_t {
        if (!(rows[i].neff > 0 || !skip_missing)) return 0;
        return put_row(o, runs.name[sw.tab ? sw.label(i) : win[i].label_run], {rows[i].start, rows[i].end}, rows[i].sum,
                       {rows[i].neff, rows[i].nskip});
    });
    // genome-wide line: stdout for the global run, stderr beside windows (dxyWindow.cpp:429-433)
    std::fprintf(W == 0 ? stdout : stderr, "%g\t%llu\t%llu\n", tot.sum, (unsigned long long)tot.neff,
                 (unsigned long long)tot.nskip);
    finish(timer);
}
