// CPU check of the hosts' own ingest code (popgenomicstools_amd/host/host_common.h), built with ASan + UBSan
// by tests/test_sanitizers.py:
//   1. to_f64 against strtod (the conversion behind the reference's operator>>) on random and hand-picked
//      tokens: whatever it accepts has strtod's bits;
//   1b. fmt_g6 (the hosts' TSV number formatting) against snprintf("%g") on random bit patterns, short decimals,
//       values at and next to rounding boundaries and powers of ten: the same bytes;
//   2. parse_table on a generated table, single- and multi-chunk;
//   3. ColumnCache: store -> load gives the same bytes; truncated, foreign and size-mismatched files are refused;
//   4. Text on gzip input: a bgzf file (block-parallel inflate), a plain gzip file and a bgzf file followed by a
//      plain member (both through gzread) give the text back; a damaged block or a truncated file is refused;
//   5. cut_at_lines (the multi-GPU text cut): the pieces tile the text and start at line starts; chromosome runs parsed
//      per piece and stitched with Runs::add are the runs of the whole text;
//   6. scan_runs_and_marks (first scan of the passes mode): rows, runs, end of the data, the byte marks, the position digests;
//   6b. ScoreTable / read_chrlen (ihsWindow / xpehhWindow: locus-id prefix, score field, errors with line numbers);
//   7. HostBuf (huge-page mappings for columns, rows and inflated text): alignment, size, every byte writable;
//   8. resident_limit_for: which inputs are reduced in passes, and of how many sites.
#include <dirent.h>

#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "host_common.h"
#include "extreme_common.h"

using namespace pgthost;

static int fails = 0;
static const double kPow10Check[20] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19};
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "FAIL %s:%d %s\n", __FILE__, __LINE__, #c); ++fails; } } while (0)

static void check_token(const std::string &s) {
    double a = 123.0;
    const Tok t{s.data(), s.data() + s.size()};
    const bool fa = to_f64(t, a);
    if (fa) {
        const char *start = s.c_str() + (s[0] == '+' ? 1 : 0);
        char *end = nullptr;
        const double c = std::strtod(start, &end);
        if (*end == 0 && std::memcmp(&a, &c, 8) != 0) { std::fprintf(stderr, "strtod differs on '%s'\n", s.c_str()); ++fails; }
    }
}

// one bgzf member (SAM spec 4.1): gzip header with the "BC" extra field, raw deflate data, CRC-32, length
static std::string bgzf_member(const char *p, size_t n) {
    std::vector<unsigned char> def(compressBound(n) + 64);
    z_stream zs{};
    deflateInit2(&zs, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
    zs.next_in = reinterpret_cast<unsigned char *>(const_cast<char *>(p));
    zs.avail_in = (uInt)n;
    zs.next_out = def.data();
    zs.avail_out = (uInt)def.size();
    deflate(&zs, Z_FINISH);
    const size_t dn = def.size() - zs.avail_out;
    deflateEnd(&zs);
    const size_t bsize = 18 + dn + 8;
    std::string m("\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0", 16);
    m += (char)((bsize - 1) & 0xff);
    m += (char)((bsize - 1) >> 8);
    m.append(reinterpret_cast<char *>(def.data()), dn);
    const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), reinterpret_cast<const unsigned char *>(p), (uInt)n), len = (uint32_t)n;
    for (int k = 0; k < 4; ++k) m += (char)(crc >> (8 * k));
    for (int k = 0; k < 4; ++k) m += (char)(len >> (8 * k));
    return m;
}
static void put_file(const std::string &path, const std::string &bytes) {
    FILE *f = std::fopen(path.c_str(), "wb");
    std::fwrite(bytes.data(), 1, bytes.size(), f);
    std::fclose(f);
}
static bool text_is(const std::string &path, const std::string &want) {
    Text t;
    return t.open(path.c_str()) && t.size() == want.size() && std::memcmp(t.begin(), want.data(), want.size()) == 0;
}

int main(int argc, char **argv) {
    const long n = argc > 1 ? std::atol(argv[1]) : 300000;
    std::mt19937_64 rng(argc > 2 ? std::atoll(argv[2]) : 7);
    for (const char *s : {"0", "-0", "+0", "0.0", "-0.000000", "1", "1.", ".5", "-.5", "+.5", "1e5", "1E5", "1e+5", "1e-5", "1e", "1e+",
                          "e5", ".", "-", "+", "+-1", "-+1", "++1", "1.5e-05", "0.123456", "-0.000012", "123456789", "0.300000",
                          "999999999999999", "9999999999999999", "0.1234567890123456789", "1e22", "1e23", "1e-22", "1e-23", "123e20",
                          "1e400", "1e-400", "nan", "inf", "-inf", "NaN", "0x10", "1,5", "1.5x", " 1", "1 ", "00012.5000", "1e0005",
                          "4.9e-324", "2.2250738585072014e-308", "17976931348623157e292", "0.000000000000000000001"})
        check_token(s);
    for (long i = 0; i < n; ++i) {
        std::string s;
        const int kind = (int)(rng() % 6);
        if (rng() % 7 == 0) s += rng() % 2 ? "-" : "+";
        const int nint = (int)(rng() % (kind == 0 ? 3 : 12)), nfrac = (int)(rng() % (kind == 1 ? 20 : 8));
        for (int k = 0; k < nint; ++k) s += (char)('0' + rng() % 10);
        if (nfrac || rng() % 5 == 0) {
            s += '.';
            for (int k = 0; k < nfrac; ++k) s += (char)('0' + rng() % 10);
        }
        if (kind >= 4) {
            s += rng() % 2 ? 'e' : 'E';
            if (rng() % 2) s += rng() % 2 ? '-' : '+';
            s += std::to_string(rng() % (kind == 4 ? 30 : 330));
        }
        if (rng() % 50 == 0) s += (char)("xe.-+ "[rng() % 6]);
        check_token(s);
    }

    // ---- fmt_g6 == printf("%g") ------------------------------------------------------------------
    {
        auto same = [&](double v) {
            char a[40], b[40];
            const size_t la = fmt_g6(v, a);
            const int lb = std::snprintf(b, sizeof b, "%g", v);
            if ((int)la != lb || std::memcmp(a, b, la) != 0) {
                a[la] = 0;
                std::fprintf(stderr, "fmt_g6 differs on %.17g: '%s' vs '%s'\n", v, a, b);
                ++fails;
            }
        };
        for (double v : {0.0, -0.0, 1.0, -1.0, 0.5, 0.1, 100000.0, 999999.0, 999999.5, 999999.4999999999, 1e6, 1e-4, 0.0001, 0.00009999995,
                         0.000099999949999, 1e-5, 123456.5, 123456.49999999999, 123457.5, 0.0900737, -4e-05, 1.25e+06, 8.1e-15, 1e22, 1e23,
                         1e-17, 9.9e-18, 1e27, 9.99999e26, 5e-324, 1.7976931348623157e308, 2.5, 3.5, 0.000123456, 1234.56789, 99999.95,
                         9.999995, 9.9999949999999, 0.3333333333333333, 2.0 / 3.0, 1e5 + 0.5, 12345.65, 1.000005, 1.0000049999999999})
            same(v), same(-v);
        same(std::nan("")); same(HUGE_VAL); same(-HUGE_VAL);
        for (long i = 0; i < 4 * n; ++i) {
            const int kind = (int)(rng() % 6);
            double v;
            if (kind == 0) {  // random bit pattern in a sane exponent range
                uint64_t bits = rng();
                bits = (bits & ~(0x7FFull << 52)) | ((uint64_t)(1023 - 70 + rng() % 140) << 52);
                std::memcpy(&v, &bits, 8);
            } else if (kind == 1) {  // short decimals
                v = (double)(int64_t)(rng() % 20000000 - 10000000) / kPow10Check[rng() % 10];
            } else if (kind == 2) {  // next to a rounding boundary of the sixth digit
                const double base = (double)(100000 + rng() % 900000) + 0.5;
                v = std::nextafter(base, (rng() % 2) ? 0.0 : 1e9);
                for (int k = (int)(rng() % 3); k > 0; --k) v = std::nextafter(v, (rng() % 2) ? 0.0 : 1e9);
                v *= kPow10Check[rng() % 10] / kPow10Check[rng() % 20];
            } else if (kind == 3) {  // ratios, as the window statistics are
                v = ((double)(rng() % 1000000) - 300000.0) / (double)(1 + rng() % 3000000);
            } else if (kind == 4) {  // around powers of ten
                v = kPow10Check[rng() % 20] / kPow10Check[rng() % 20];
                for (int k = (int)(rng() % 4); k > 0; --k) v = std::nextafter(v, (rng() % 2) ? 0.0 : 1e300);
            } else {
                v = (double)(rng() % 4000000000ull);
            }
            same(v);
        }
    }

    // ---- parse_table + ColumnCache round trip --------------------------------------------------
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
    };
    const size_t rows = 200000;
    std::string text;
    std::vector<double> va(rows), vb(rows);
    for (size_t i = 0; i < rows; ++i) {
        char line[96];
        va[i] = (double)(int64_t)(rng() % 2000000 - 1000000) / 1e6;
        vb[i] = (double)(rng() % 300001) / 1e6;
        text.append(line, (size_t)std::snprintf(line, sizeof line, "chr%zu\t%zu\t%.6f\t%.6f\n", i * 7 / rows + 1, i % 50000 + 1, va[i], vb[i]));
    }
    for (int threads : {1, 5}) {
        setenv("PGT_HOST_THREADS", std::to_string(threads).c_str(), 1);
        std::string padded = text;
        if (threads > 1) padded.append(std::string((1u << 20) + 7, ' '));  // > 1 MiB so that the parallel path is taken
        Table tab;
        Runs runs;
        const size_t got = parse_table(padded.data(), padded.data() + text.size(), tab, runs, "check", "mem", 1);
        CHECK(got == rows && runs.name.size() == 7);
        for (size_t i = 0; i < rows && got == rows; i += 997) CHECK(tab.a[i] == std::strtod(std::to_string(va[i]).c_str(), nullptr) && tab.b[i] == vb[i]);
        if (threads == 1) {
            // cache round trip on this table
            const std::string dir = argc > 3 ? argv[3] : "/tmp";
            setenv("PGT_COLUMN_CACHE", dir.c_str(), 1);
            const std::string input = dir + "/host_parse_check_input.txt";
            FILE *f = std::fopen(input.c_str(), "wb");
            std::fwrite(text.data(), 1, text.size(), f);
            std::fclose(f);
            std::vector<ColumnCache::Col> cols = {{tab.pos.data(), 4}, {tab.a.data(), 8}, {tab.b.data(), 8}};
            {
                ColumnCache c("check", input.c_str());
                CHECK(c.enabled());
                size_t n0 = 0;
                Runs r0;
                std::vector<ColumnCache::Col> probe = {{nullptr, 4}, {nullptr, 8}, {nullptr, 8}};
                CHECK(!c.load(n0, r0, probe));  // nothing there yet
                c.store(rows, runs, cols);
            }
            {
                ColumnCache c("check", input.c_str());
                size_t n1 = 0;
                Runs r1;
                std::vector<ColumnCache::Col> back = {{nullptr, 4}, {nullptr, 8}, {nullptr, 8}};
                CHECK(c.load(n1, r1, back) && n1 == rows && r1.name == runs.name && r1.len == runs.len);
                if (n1 == rows) {
                    CHECK(std::memcmp(back[0].data, tab.pos.data(), rows * 4) == 0);
                    CHECK(std::memcmp(back[1].data, tab.a.data(), rows * 8) == 0);
                    CHECK(std::memcmp(back[2].data, tab.b.data(), rows * 8) == 0);
                    CHECK((reinterpret_cast<uintptr_t>(back[1].data) & 63) == 0);
                }
                std::vector<ColumnCache::Col> wrong = {{nullptr, 4}, {nullptr, 8}};  // another tool's column set
                ColumnCache c2("check", input.c_str());
                CHECK(!c2.load(n1, r1, wrong));
                std::vector<ColumnCache::Col> wrong2 = {{nullptr, 8}, {nullptr, 8}, {nullptr, 8}};  // element size mismatch
                ColumnCache c3("check", input.c_str());
                CHECK(!c3.load(n1, r1, wrong2));
            }
            {   // damaged cache files: header words replaced by extreme values, the file cut at random lengths — load
                // must refuse or return something consistent, and never read outside the mapping (ASan watches)
                std::string cache_file;
                if (DIR *dh = opendir(dir.c_str())) {
                    while (dirent *de = readdir(dh)) {
                        const std::string nm = de->d_name;
                        if (nm.size() > 8 && nm.substr(nm.size() - 8) == ".pgtcols") cache_file = dir + "/" + nm;
                    }
                    closedir(dh);
                }
                CHECK(!cache_file.empty());
                FILE *cf = std::fopen(cache_file.c_str(), "rb");
                std::string good((size_t)rows * 20 + 4096, 0);
                good.resize(std::fread(&good[0], 1, good.size(), cf));
                std::fclose(cf);
                const uint64_t evil[] = {~0ull, 1ull << 63, (1ull << 61) + 1, good.size(), good.size() - 7, 0ull, rows + 1, 1ull << 32};
                for (int trial = 0; trial < 200; ++trial) {
                    std::string bad = good;
                    if (trial % 4 == 3) bad.resize(rng() % good.size());
                    else {
                        const size_t word = 8 + 8 * (rng() % 40);  // header, run table and the first column header
                        const uint64_t v = evil[rng() % 8];
                        if (word + 8 <= bad.size()) std::memcpy(&bad[word], &v, 8);
                    }
                    put_file(cache_file, bad);
                    ColumnCache c("check", input.c_str());
                    size_t n3 = 0;
                    Runs r3;
                    std::vector<ColumnCache::Col> b3 = {{nullptr, 4}, {nullptr, 8}, {nullptr, 8}};
                    if (c.load(n3, r3, b3)) {  // accepted: every column must lie inside the file
                        volatile unsigned char sink = 0;
                        for (auto &col : b3)
                            if (n3) sink = sink + static_cast<unsigned char *>(col.data)[0] + static_cast<unsigned char *>(col.data)[n3 * col.elem - 1];
                    }
                }
                put_file(cache_file, good);
            }
            ColumnCache other("another tool", input.c_str());  // the tag is part of the key
            size_t n2 = 0;
            Runs r2;
            std::vector<ColumnCache::Col> b2 = {{nullptr, 4}, {nullptr, 8}, {nullptr, 8}};
            CHECK(!other.load(n2, r2, b2));
            std::remove(input.c_str());
        }
    }
    {   // 4. gzip input
        const std::string dir = argc > 3 ? argv[3] : "/tmp";
        std::string text;
        for (long i = 0; i < 60000; ++i) text += "chr" + std::to_string(i / 9000) + "\t" + std::to_string(i * 7 + 1) + "\t0." + std::to_string(100000 + rng() % 900000) + "\n";
        std::string bg;
        std::vector<size_t> starts;
        for (size_t o = 0; o < text.size();) {
            const size_t n = std::min<size_t>(text.size() - o, 500 + rng() % (o % 3 ? 4000 : 64000));  // ragged blocks, all <= 64 KiB
            starts.push_back(bg.size());
            bg += bgzf_member(text.data() + o, n);
            o += n;
        }
        const std::string eof = bgzf_member("", 0);  // the empty member bgzf files end with
        for (int threads : {1, 5}) {
            setenv("PGT_HOST_THREADS", std::to_string(threads).c_str(), 1);
            put_file(dir + "/t.bgzf.gz", bg + eof);
            CHECK(text_is(dir + "/t.bgzf.gz", text));
            put_file(dir + "/t.bgzf.gz", bg);  // no end marker: still complete members
            CHECK(text_is(dir + "/t.bgzf.gz", text));
        }
        CHECK(starts.size() > 64);
        {
            gzFile g = gzopen((dir + "/t.plain.gz").c_str(), "wb");
            gzwrite(g, text.data(), (unsigned)text.size());
            gzclose(g);
            CHECK(text_is(dir + "/t.plain.gz", text));
            // bgzf members followed by an ordinary gzip member: not bgzf to the end -> gzread, which reads all members
            FILE *f = std::fopen((dir + "/t.plain.gz").c_str(), "rb");
            std::string plain(1 << 22, 0);
            plain.resize(std::fread(&plain[0], 1, plain.size(), f));
            std::fclose(f);
            put_file(dir + "/t.mixed.gz", bg + plain);
            CHECK(text_is(dir + "/t.mixed.gz", text + text));
        }
        {
            std::string hurt = bg + eof;
            hurt[starts[starts.size() / 2] + 40] ^= 0x55;  // inside the deflate data of a middle block
            put_file(dir + "/t.hurt.gz", hurt);
            Text t;
            CHECK(!t.open((dir + "/t.hurt.gz").c_str()));
            std::string crc = bg + eof;
            crc[starts[3] - 6] ^= 1;  // the CRC-32 of block 2
            put_file(dir + "/t.hurt.gz", crc);
            Text t2;
            CHECK(!t2.open((dir + "/t.hurt.gz").c_str()));
            put_file(dir + "/t.hurt.gz", bg.substr(0, starts[5] + 100));  // cut inside block 5
            Text t3;
            CHECK(!t3.open((dir + "/t.hurt.gz").c_str()));
        }
        for (const char *n : {"/t.bgzf.gz", "/t.plain.gz", "/t.mixed.gz", "/t.hurt.gz"}) std::remove((dir + n).c_str());
    }
    {   // 5. the multi-GPU text cut (cut_at_lines) and the stitching of chromosome runs across the seams (Runs::add)
        for (int trial = 0; trial < 300; ++trial) {
            std::string text;
            Runs whole;
            const int n_lines = (int)(rng() % 400);
            int chr = 0;
            for (int i = 0; i < n_lines; ++i) {
                if (rng() % 17 == 0) ++chr;
                const std::string name = "c" + std::to_string(chr);
                whole.add(name.data(), name.data() + name.size());
                text += name + "\t" + std::to_string(i + 1) + "\t0." + std::to_string(rng() % 1000) + "\t0.5\n";
            }
            if (n_lines && rng() % 4 == 0) text.pop_back();  // last line without newline
            const size_t parts = 1 + rng() % 9;
            const char *b = text.data(), *e = b + text.size();
            const std::vector<const char *> cut = cut_at_lines(b, e, parts);
            CHECK(cut.size() == parts + 1 && cut.front() == b && cut.back() == e);
            Runs stitched;
            for (size_t k = 0; k < parts; ++k) {
                CHECK(cut[k] <= cut[k + 1]);
                CHECK(cut[k] == b || cut[k] == e || cut[k][-1] == '\n');  // every piece starts at a line start
                Runs piece;  // what the parser of piece k reports: runs with names pointing into the text
                for (const char *p = cut[k]; p < cut[k + 1];) {
                    const char *tab = static_cast<const char *>(std::memchr(p, '\t', (size_t)(cut[k + 1] - p)));
                    piece.add(p, tab);
                    const char *nl = static_cast<const char *>(std::memchr(p, '\n', (size_t)(cut[k + 1] - p)));
                    p = nl ? nl + 1 : cut[k + 1];
                }
                for (size_t r = 0; r < piece.name.size(); ++r)
                    stitched.add(piece.name[r].data(), piece.name[r].data() + piece.name[r].size(), piece.len[r]);
            }
            CHECK(stitched.name == whole.name && stitched.len == whole.len);
        }
    }
    {   // 5b. a piece whose LAST line is blank ends the data for the pieces behind it (found by tests/ingest_fuzz.py: the device
        //     parser of such a piece sees nothing unusual)
        CHECK(!ends_with_blank_line("", ""));
        const char *yes[] = {"\n", " \n", "c 1 2\n\n", "c 1 2\n \t\r\n", "c 1 2\n  ", "c 1 2\n\r\n"};
        const char *no[] = {"c 1 2\n", "c 1 2", "\nc 1 2\n", " \nc 1 2", "c 1 2\nx \n"};
        for (const char *t : yes) CHECK(ends_with_blank_line(t, t + std::strlen(t)));
        for (const char *t : no) CHECK(!ends_with_blank_line(t, t + std::strlen(t)));
    }
    {   // 6. the first scan of the passes mode (scan_runs_and_marks): rows, runs, the end of the data and a mark at every 65536th row
        for (int trial = 0; trial < 6; ++trial) {
            std::string text;
            Runs whole;
            const size_t n_lines = trial == 0 ? 0 : trial == 1 ? 65536 : trial == 2 ? 131072 : 100000 + rng() % 200000;
            const size_t blank_at = trial >= 4 ? n_lines / 2 + rng() % 1000 : ~(size_t)0;  // the parsers stop at the first blank line
            std::vector<size_t> line_start;
            size_t rows = 0, chr = 0;
            for (size_t i = 0; i < n_lines; ++i) {
                if (i == blank_at) { text += " \t\n"; continue; }
                if (rng() % 30011 == 0) ++chr;
                const std::string name = "scaffold_" + std::to_string(chr);
                if (i < blank_at) { whole.add(name.data(), name.data() + name.size()); line_start.push_back(text.size()); ++rows; }
                text += name + "\t" + std::to_string(i + 1) + "\t0." + std::to_string(rng() % 1000) + "\t0.5\n";
            }
            if (n_lines && trial == 3) text.pop_back();  // last line without newline
            const char *b = text.data(), *e = b + text.size(), *data_end = nullptr;
            Runs runs;
            std::vector<const char *> mark;
            const size_t n = scan_runs_and_marks(b, e, runs, mark, &data_end);
            CHECK(n == rows);
            CHECK(runs.name == whole.name && runs.len == whole.len);
            CHECK(data_end == (blank_at < n_lines ? b + line_start.back() + (std::strchr(b + line_start.back(), '\n') - (b + line_start.back())) + 1 : e));
            CHECK(mark.size() >= (n + kMarkEvery - 1) / kMarkEvery + 1);
            for (size_t k = 0; k * kMarkEvery < n; ++k) CHECK(mark[k] == b + line_start[k * kMarkEvery]);
            CHECK(mark[(n + kMarkEvery - 1) / kMarkEvery] == data_end);
            // the position digests (how dxyWindow's passes compare two files' site lists before printing anything): one per
            // 65536 rows; independent of the thread count; blind to everything but the position column ("+7" and "7" agree);
            // any position changed, two positions swapped or a row's position missing changes its block's digest and no other
            std::vector<uint64_t> dig, dig1, dig2;
            Runs r2;
            std::vector<const char *> m2;
            CHECK(scan_runs_and_marks(b, e, r2, m2, &data_end, &dig) == n);
            CHECK(dig.size() == (n + kMarkEvery - 1) / kMarkEvery);
            setenv("PGT_HOST_THREADS", "1", 1);
            r2 = Runs{};
            CHECK(scan_runs_and_marks(b, e, r2, m2, &data_end, &dig1) == n && dig1 == dig);
            setenv("PGT_HOST_THREADS", "7", 1);
            if (n > 70000) {
                std::string other = text;  // same rows, other columns behind the position, a '+' before one position
                for (size_t i = 0; i + 1 < other.size(); ++i)
                    if (other[i] == '\t' && other[i + 1] == '0' && other[i + 2] == '.') other[i + 3] = '7';
                const size_t at = line_start[66000] + (size_t)(std::strchr(b + line_start[66000], '\t') - (b + line_start[66000])) + 1;
                other.insert(at, "+");
                r2 = Runs{};
                CHECK(scan_runs_and_marks(other.data(), other.data() + other.size(), r2, m2, &data_end, &dig2) == n && dig2 == dig);
                std::string moved = text;   // one position differs, in block 1
                moved[at] = moved[at] == '9' ? '8' : '9';
                r2 = Runs{};
                CHECK(scan_runs_and_marks(moved.data(), moved.data() + moved.size(), r2, m2, &data_end, &dig2) == n);
                CHECK(dig2.size() == dig.size() && dig2[0] == dig[0] && dig2[1] != dig[1]);
                for (size_t k = 2; k < dig.size(); ++k) CHECK(dig2[k] == dig[k]);
            }
        }
        unsetenv("PGT_HOST_THREADS");
    }
    {   // 6b. ScoreTable (ihsWindow / xpehhWindow, extreme_common.h): `<chr>_<id> pos f0 ..`, the chromosome is the locus id up to its
        // first '_' (extractChr, ihsWindow.cpp:80-92), the score numeric field 4 (iHS) or 6 (XP-EHH) behind the position; one chunk
        // and several; a bad score and a missing field are errors with their line number; read_chrlen keeps the first entry of a name
        const size_t rows_s = 120000;
        for (int field : {4, 6}) {
            std::string norm;
            std::vector<double> want(rows_s);
            for (size_t i = 0; i < rows_s; ++i) {
                char line[200];
                want[i] = (double)(int64_t)(rng() % 8000000 - 4000000) / 1e6;
                const size_t c = i * 5 / rows_s + 1;
                int n = std::snprintf(line, sizeof line, i % 3 ? "chr%zu_%zu\t%zu" : "chr%zu_%zu_x\t%zu", c, i + 1, i + 1);
                for (int k = 0; k <= field + 1; ++k) n += std::snprintf(line + n, sizeof line - (size_t)n, "\t%.6f", k == field ? want[i] : 0.25 * k);
                line[n++] = '\n';
                norm.append(line, (size_t)n);
            }
            for (int threads : {1, 5}) {
                setenv("PGT_HOST_THREADS", std::to_string(threads).c_str(), 1);
                std::string padded = norm;
                if (threads > 1) padded.append(std::string((1u << 20) + 7, ' '));
                ScoreTable tab;
                tab.score_field = field;
                Runs runs;
                const size_t got = parse_table(padded.data(), padded.data() + norm.size(), tab, runs, "check", "mem", 1);
                CHECK(got == rows_s && runs.name.size() == 5 && runs.name[0] == "chr1" && runs.name[4] == "chr5");
                for (size_t i = 0; i < rows_s && got == rows_s; i += 499) CHECK(tab.score[i] == want[i] && tab.pos[i] == i + 1);
            }
            std::string bad = norm.substr(0, norm.find('\n', norm.size() / 2) + 1) + "chr5_9\t9\t0.1\n";  // too few fields on the last line
            ScoreTable tab;
            tab.score_field = field;
            Runs runs;
            std::string error;
            (void)parse_table(bad.data(), bad.data() + bad.size(), tab, runs, "check", "mem", 1, &error);
            CHECK(!error.empty() && error.find("line ") != std::string::npos);
        }
        const std::string dirl = argc > 3 ? argv[3] : "/tmp";
        put_file(dirl + "/chrlen.txt", "chr1\t100\nchr2 250\nchr1\t999\n\nbroken\nchr3\tx\n");
        const auto lens = read_chrlen((dirl + "/chrlen.txt").c_str());
        CHECK(lens.size() == 2 && lens.at("chr1") == 100 && lens.at("chr2") == 250);
        unsetenv("PGT_HOST_THREADS");
    }
    {   // 7. HostBuf: malloc below 8 MiB, a 2-MiB-aligned mapping of whole huge pages from there on; every byte writable; release and re-use
        HostBuf h;
        for (size_t bytes : {(size_t)1, (size_t)4096, ((size_t)8 << 20) - 1, (size_t)8 << 20, ((size_t)8 << 20) + 1, ((size_t)33 << 20) + 12345}) {
            h.alloc(bytes);
            CHECK(h.p != nullptr);
            const bool big = bytes >= ((size_t)8 << 20);
            CHECK(big == (h.mapped != 0));
            if (big) {
                CHECK(reinterpret_cast<uintptr_t>(h.p) % ((size_t)2 << 20) == 0);
                CHECK(h.mapped >= bytes && h.mapped % ((size_t)2 << 20) == 0 && h.mapped < bytes + ((size_t)2 << 20));
            }
            std::memset(h.p, 0x5a, bytes);
            CHECK(static_cast<unsigned char *>(h.p)[bytes - 1] == 0x5a && static_cast<unsigned char *>(h.p)[0] == 0x5a);
        }
        h.release();
        CHECK(h.p == nullptr && h.mapped == 0);
        Column<double> c;
        c.alloc((size_t)3 << 20);  // 24 MiB
        c[((size_t)3 << 20) - 1] = 2.5;
        CHECK(c.data()[((size_t)3 << 20) - 1] == 2.5);
    }
    {   // 8. resident or in passes (resident_limit_for): 10^9 fstWindow lines fit a 288-GB MI355X, 10^10 do not
        const size_t free_b = (size_t)280 << 30;
        CHECK(resident_limit_for((size_t)33e9, 33.0, 20, free_b, 1) == 0);           // 33 GB of text + 22 GB of columns
        const uint64_t lim = resident_limit_for((size_t)330e9, 33.0, 20, free_b, 1);  // 10^10 lines
        CHECK(lim > 2000000000ull && lim < 4000000000ull);
        CHECK((double)lim * (33.0 + 22.0) < 0.61 * (double)free_b);
        CHECK(resident_limit_for((size_t)105e9, 35.0, 32, free_b, 2) > 0);           // an all-sites MAF pair of a 3-Gb genome
        CHECK(resident_limit_for((size_t)35e9, 35.0, 32, free_b, 2) == 0);
        CHECK(resident_limit_for((size_t)1 << 40, 8.0, 20, (size_t)1 << 20, 1) >= 1);  // never 0 when passes are needed
    }
    std::printf(fails ? "host_parse_check: %d FAILURES\n" : "host_parse_check: all equal (%d)\n", fails);
    return fails ? 1 : 0;
}
