"""GPU: device-side text ingest (pgt_ingest_text) against a literal Python restatement of the host parser's
rules (host_common.h: token splitting, from_chars conversions, blank line = end of data, first bad line =
error): same rows, same bits in every column, same chromosome runs — on generated tables and on lines that
exercise the slow path (long mantissas, exponents, signs, inf/nan, \\r\\n, extra columns, missing tokens)."""
import re

import numpy as np
import pytest

import synth
from popgenomicstools_amd import _lib
from popgenomicstools_amd._lib import (PGT_TOK_CHR, PGT_TOK_CHR_PREFIX, PGT_TOK_F64, PGT_TOK_FREQ, PGT_TOK_I8, PGT_TOK_I32,
                                       PGT_TOK_SKIP, PGT_TOK_U32)

pytestmark = pytest.mark.gpu

FST = [PGT_TOK_CHR, PGT_TOK_U32, PGT_TOK_F64, PGT_TOK_F64]
HET = [PGT_TOK_CHR, PGT_TOK_U32, PGT_TOK_I8]
MAF = [PGT_TOK_CHR, PGT_TOK_U32, PGT_TOK_SKIP, PGT_TOK_SKIP, PGT_TOK_SKIP, PGT_TOK_FREQ, PGT_TOK_I32]
# selscan *.norm: locus id `chr_position`, position, numeric fields; the score is field 4 (iHS) or 6 (XP-EHH) behind the position
IHS = [PGT_TOK_CHR_PREFIX, PGT_TOK_U32] + [PGT_TOK_SKIP] * 4 + [PGT_TOK_F64]
XPEHH = [PGT_TOK_CHR_PREFIX, PGT_TOK_U32] + [PGT_TOK_SKIP] * 6 + [PGT_TOK_F64]

_F64 = re.compile(rb"-?(\d+\.?\d*|\.\d+)([eE][-+]?\d+)?$|-?(inf|infinity|nan)$", re.I)  # std::from_chars, general format
_INT = re.compile(rb"-?\d+$")


def _f64(tok):
    if tok.startswith(b"+"):
        tok = tok[1:]
    m = _F64.match(tok)
    if not m:
        return None
    v = float(tok)
    if m.group(3) is None:  # a number: libstdc++'s from_chars refuses overflow and underflow to zero, accepts denormals
        if v in (float("inf"), float("-inf")) or (v == 0.0 and re.search(rb"[1-9]", re.split(rb"[eE]", tok)[0])):
            return None
    return v


def _int(tok, lo, hi, clamp):
    if tok.startswith(b"+"):
        tok = tok[1:]
    if not _INT.match(tok):
        return None
    v = int(tok)
    if not (-2**63 <= v < 2**63):
        return None
    if clamp:
        return max(lo, min(hi, v))
    return v if lo <= v <= hi else None


def host_model(text: bytes, tokens):
    """-> (rows per stored token, run names, run lengths, bad_line or -1)"""
    cols = {k: [] for k, t in enumerate(tokens) if t not in (PGT_TOK_CHR, PGT_TOK_CHR_PREFIX, PGT_TOK_SKIP)}
    names, lens, bad = [], [], -1
    lines = text.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    for i, line in enumerate(lines):
        toks = [t for t in re.split(rb"[ \t\r]+", line) if t]
        if not toks:
            break
        vals, ok = {}, True
        for k, kind in enumerate(tokens[1:], start=1):
            tok = toks[k] if k < len(toks) else b""
            if kind == PGT_TOK_SKIP:
                continue
            if kind == PGT_TOK_U32:
                v = None
                t = tok[1:] if tok.startswith(b"+") else tok
                if re.match(rb"\d+$", t) and int(t) <= 0xFFFFFFFF:
                    v = int(t)
            elif kind in (PGT_TOK_F64, PGT_TOK_FREQ):
                v = _f64(tok)
                if kind == PGT_TOK_FREQ and v is not None and not (0.0 <= v <= 1.0):
                    v = None
            elif kind == PGT_TOK_I8:
                v = _int(tok, -128, 127, True)
            else:
                v = _int(tok, -2**31, 2**31 - 1, True)
            if v is None:
                ok = False
                break
            vals[k] = v
        if not ok:
            bad = i
            break
        for k, v in vals.items():
            cols[k].append(v)
        name = toks[0].split(b"_", 1)[0] if tokens[0] == PGT_TOK_CHR_PREFIX else toks[0]  # extractChr, ihsWindow.cpp:80-92
        if not names or names[-1] != name:
            names.append(name)
            lens.append(0)
        lens[-1] += 1
    return cols, [n.decode("latin-1") for n in names], lens, bad


def check(ctx, text: bytes, tokens):
    ing = ctx.ingest_text(text, tokens)
    cols, names, lens, bad = host_model(text, tokens)
    n = len(next(iter(cols.values())))
    assert ing.bad_line == bad, (ing.bad_line, bad)
    assert ing.rows == n, (ing.rows, n)
    assert ing.run_names == names and ing.run_len.tolist() == lens
    for k, ref in cols.items():
        got = ing.column_np(k)
        kind = tokens[k]
        if kind in (PGT_TOK_F64, PGT_TOK_FREQ):
            assert got.tobytes() == np.array(ref, dtype=np.float64).tobytes(), k  # bits, incl. -0.0 and nan
        else:
            dt = {PGT_TOK_U32: np.uint32, PGT_TOK_I8: np.int8, PGT_TOK_I32: np.int32}[kind]
            assert np.array_equal(got, np.array(ref, dtype=np.int64).astype(dt)), k
    ing.free()
    return n


def test_generated_tables_every_format(ctx):
    rng = np.random.default_rng(5)
    n = 400_000
    chr_ids, pos = synth.chromosomes(rng, n, 9, equal=False)
    a, b = synth.fst_columns(rng, n)
    text = "".join(f"chr{c}\t{p}\t{x:.6f}\t{y:.6f}\n" for c, p, x, y in zip(chr_ids, pos, a, b)).encode()
    assert check(ctx, text, FST) == n
    assert check(ctx, text[:-1], FST) == n  # no trailing newline
    g = synth.het_column(rng, n)
    text = "".join(f"sc{c} {p} {v}\n" for c, p, v in zip(chr_ids, pos, g)).encode()
    assert check(ctx, text, HET) == n
    p1, _, n1, _ = synth.dxy_columns(rng, n)
    text = "".join(f"chr{c}\t{p}\tA\tC\tA\t{x:.6f}\t{k}\n" for c, p, x, k in zip(chr_ids, pos, p1, n1)).encode()
    assert check(ctx, text, MAF) == n


def test_irregular_lines_take_the_host_conversions(ctx):
    lines = [
        b"c1\t1\t0.1\t0.2", b"c1 2 -0.000012 0.3", b"c1\t3\t+0.5\t.5", b"  c1 \t 4\t1.\t-.25  extra columns 7",
        b"c1\t5\t1.5e-05\t1E5", b"c1\t6\t0.1234567890123456789\t123456789012345678", b"c1\t7\t1e22\t1e23", b"c1\t8\t1e-22\t1e-23",
        b"c1\t9\t4.9e-324\t2.2250738585072014e-308", b"c1\t10\tinf\t-inf", b"c1\t11\tnan\tNaN", b"c1\t12\t-0\t-0.0",
        b"c1\t13\t1e400\t1e-400", b"c1\t+14\t+-1\t0", b"c2\t00000000000000000015\t0.3\t0.4", b"c2\t16\t0.3\t0.4\r",
        b"c2\t4294967295\t9007199254740993\t0.999999999999999", b"c3\t17\t17976931348623157e292\t0.000000000000000000001",
    ]
    text = b"\n".join(lines) + b"\n"
    assert check(ctx, text, FST) == 12  # line 13 holds 1e400, which no double holds: from_chars refuses it -> rows stop at the error line
    good = [ln for ln in lines if b"1e400" not in ln]
    assert check(ctx, b"\n".join(good) + b"\n", FST) == len(good)
    assert check(ctx, b"\r\n".join(good) + b"\r\n", FST) == len(good)
    # end of data at a blank line; what follows is not read, whatever it holds
    assert check(ctx, b"c1\t1\t0.1\t0.2\nc1\t2\t0.1\t0.2\n \t\r\nc1\tgarbage\n", FST) == 2
    assert check(ctx, b"\nc1\t1\t0.1\t0.2\n", FST) == 0
    # error lines: before the end of data they are reported, rows stop there
    for bad in (b"c1\t1\t0.1", b"c1\tx\t0.1\t0.2", b"c1\t-1\t0.1\t0.2", b"c1\t4294967296\t0.1\t0.2", b"c1\t1\t0.1\t0.2x", b"c1\t1\t1e\t2",
                b"c1\t1\t0x10\t2", b"c1\t1\t1,5\t2", b"c1"):
        assert check(ctx, b"c1\t1\t0.1\t0.2\n" + bad + b"\nc1\t3\t0.1\t0.2\n", FST) == 1
    # het: any integer, clamped; MAF: frequency range, nInd clamp, skipped columns
    assert check(ctx, b"c 1 0\nc 2 1\nc 3 2\nc 4 -1\nc 5 -9\nc 6 300\nc 7 +1\nc 8 99999999999999999999\nc 9 1\n", HET) == 7
    assert check(ctx, b"c 1 1.0\n", HET) == 0
    maf = b"c\t1\tA\tC\tA\t0.5\t10\nc\t2\tA\tC\tA\t1\t99999999999\nc\t3\tA\tC\tA\t0\t-99999999999\nc\t4\tA\tC\tA\t1.000001\t3\n"
    assert check(ctx, maf, MAF) == 3
    assert check(ctx, b"c\t1\tA\tC\t0.5\t10\n", MAF) == 0  # a column short: the frequency token is "10", nInd is missing
    assert check(ctx, b"", FST) == 0


def test_random_token_soup(ctx):
    """Random tokens of every flavour in every column: whatever the host rules say, the device path says."""
    rng = np.random.default_rng(77)
    pool_f = [b"0", b"1", b"-1", b"0.5", b"1e5", b"1e-5", b"+2", b"2.", b".2", b"1e23", b"123456789012345.5", b"1234567890123456",
              b"nan", b"inf", b"abc", b"1e", b"--1", b"0.30000000000000004", b"5e-324", b"1.7976931348623157e308", b"1e309"]
    pool_u = [b"0", b"1", b"42", b"+7", b"4294967295", b"4294967296", b"-3", b"1.0", b"x", b"007"]
    for trial in range(30):
        lines = []
        for i in range(int(rng.integers(1, 400))):
            r = rng.random()
            if r < 0.01:
                lines.append(b"")
                continue
            sep = [b"\t", b" ", b"  ", b"\t "][int(rng.integers(0, 4))]
            toks = [b"c%d" % int(i // 40 + rng.integers(0, 2)), pool_u[int(rng.integers(0, len(pool_u)))] if r < 0.2 else b"%d" % (i + 1),
                    pool_f[int(rng.integers(0, len(pool_f)))] if r < 0.6 else b"%.6f" % rng.random(),
                    pool_f[int(rng.integers(0, len(pool_f)))] if r < 0.3 else b"%.6f" % rng.random()]
            if rng.random() < 0.05:
                toks = toks[: int(rng.integers(1, 4))]
            lines.append(sep.join(toks) + (b"\r" if rng.random() < 0.1 else b""))
        text = b"\n".join(lines) + (b"\n" if rng.random() < 0.8 else b"")
        check(ctx, text, FST)


def test_ingested_columns_feed_the_scan(pgt, ctx, oracle):
    """text -> device columns -> window scan without the columns ever visiting the host: rows equal those of
    the host-buffer path on the same values."""
    import ctypes as C
    rng = np.random.default_rng(8)
    n, W, S = 300_000, 20_000, 5_000
    chr_ids, pos = synth.chromosomes(rng, n, 4, equal=False)
    a, b = synth.fst_columns(rng, n)
    text = "".join(f"chr{c}\t{p}\t{x:.6f}\t{y:.6f}\n" for c, p, x, y in zip(chr_ids, pos, a, b)).encode()
    ing = ctx.ingest_text(text, FST)
    win = pgt.build_windows_sites(ing.run_len, W, S)
    rows = np.zeros(win.size, dtype=_lib.FST_ROW_DTYPE)
    lib = _lib.load()
    _lib.check(lib.pgt_fst_reduce_cols(ctx._ctx, ing.column(1).data_ptr(), ing.column(2).data_ptr(), ing.column(3).data_ptr(),
                                       ing.rows, win.ctypes.data, win.size, rows.ctypes.data, rows.nbytes), ctx._ctx)
    ref = ctx.fst_reduce(pos, a, b, pgt.build_windows_sites(pgt.run_lengths(chr_ids), W, S))
    assert rows.tobytes() == ref.tobytes()
    assert C.sizeof(C.c_void_p) == 8


def test_download_is_bounded_by_the_column(ctx):
    import ctypes as C
    ing = ctx.ingest_text(b"c\t1\t0.5\t0.25\nc\t2\t0.5\t0.25\n", FST)
    lib, host = ctx._lib, np.zeros(64, dtype=np.uint8)
    assert lib.pgt_ingest_download(ctx._ctx, ing._h, 2, host.ctypes.data, 16) == _lib.PGT_OK
    assert host[:16].view(np.float64).tolist() == [0.5, 0.5]
    assert lib.pgt_ingest_download(ctx._ctx, ing._h, 2, host.ctypes.data, 24) == _lib.PGT_EARG   # 2 rows of 8 bytes only
    assert lib.pgt_ingest_download(ctx._ctx, ing._h, 0, host.ctypes.data, 8) == _lib.PGT_EARG    # the name token has no column
    ing.free()


def test_blank_line_flag_for_callers_that_cut_a_text_into_pieces(ctx):
    """pgt_ingest_blank_before_end: 1 whenever the data ended at a blank line of the text — also when that is its LAST
    line (a host that hands consecutive pieces to several GPUs must drop the pieces behind it; tests/ingest_fuzz.py found
    the case where the blank line closed a piece) — and 0 for a text that simply ends, with or without a newline."""
    lib = ctx._lib
    cases = [(b"c0  133  1\n \n", 1, 1), (b"c0  133  1\n\n", 1, 1), (b"c0  133  1\n ", 1, 1), (b"c0  133  1\n \t\r\n", 1, 1),
             (b"c0  133  1\n", 1, 0), (b"c0  133  1", 1, 0), (b"c0  133  1\r\n", 1, 0),
             (b"c0  133  1\n \nc1  5  1\n", 1, 1), (b" \nc1 5 1\n", 0, 1), (b"\n", 0, 1),
             (b"c0  48  -9\nc0  62  1\nc0  110  3\n", 3, 0)]
    for text, rows, flag in cases:
        ing = ctx.ingest_text(text, HET)
        assert (ing.rows, int(lib.pgt_ingest_blank_before_end(ing._h)), ing.bad_line) == (rows, flag, -1), text
        ing.free()


def test_ingest_behind_rows_of_the_caller(ctx):
    """pgt_ingest_text_behind: the parsed rows land behind room for `rows_in_front` rows of the caller's own (uploaded with
    pgt_dev_upload to pgt_ingest_column_base) — one contiguous column; pgt_ingest_column / _rows / _download still speak of
    the parsed rows only.  An odd number of rows in front (the f64 rows then start 8 bytes off a 16-byte boundary)."""
    import ctypes as C
    import torch
    lib = ctx._lib
    text = b"".join(b"c%d\t%d\t0.%03d\t%d.5\n" % (i // 4, 10 * i + 1, i, i) for i in range(9))
    toks = (C.c_uint8 * 4)(*FST)
    for front in (0, 1, 3, 1000):
        h = C.c_void_p(0)
        assert lib.pgt_ingest_text_behind(ctx._ctx, text, len(text), toks, 4, front, C.byref(h)) == _lib.PGT_OK
        assert lib.pgt_ingest_rows(h) == 9 and lib.pgt_ingest_bad_line(h) == -1
        for token, dt, tdt, mine in ((1, np.uint32, torch.int32, np.arange(front, dtype=np.uint32) + 7),
                                     (3, np.float64, torch.float64, np.arange(front, dtype=np.float64) * 0.25)):
            base, col = lib.pgt_ingest_column_base(h, token), lib.pgt_ingest_column(h, token)
            assert col - base == front * np.dtype(dt).itemsize
            if front:
                assert lib.pgt_dev_upload(ctx._ctx, base, mine.ctypes.data, mine.nbytes) == _lib.PGT_OK
            whole = torch.empty(front + 9, dtype=tdt, device="cuda:0")
            assert lib.pgt_dev_copy(ctx._ctx, whole.data_ptr(), ctx._ctx, base, whole.numel() * whole.element_size()) == _lib.PGT_OK
            got = whole.cpu().numpy().view(dt)
            want = np.array([10 * i + 1 for i in range(9)], dtype=dt) if token == 1 else np.array([i + 0.5 for i in range(9)], dtype=dt)
            assert np.array_equal(got[:front], mine) and np.array_equal(got[front:], want)
            parsed = np.zeros(9, dtype=dt)
            assert lib.pgt_ingest_download(ctx._ctx, h, token, parsed.ctypes.data, parsed.nbytes) == _lib.PGT_OK
            assert np.array_equal(parsed, want)
        lib.pgt_ingest_free(h)


def test_absurdly_long_lines_are_refused_not_walked(ctx):
    """A line of 300 KB is not one of the tools' tables: the device path refuses the input (PGT_EDOMAIN) instead
    of letting one lane walk through it; the hosts then parse it themselves."""
    text = b"c1\t1\t0.1\t0.2\n" + b"c1\t2\t" + b"7" * 300_000 + b"\t0.2\nc1\t3\t0.1\t0.2\n"
    with pytest.raises(_lib.PgtError) as e:
        ctx.ingest_text(text, FST)
    assert e.value.code == _lib.PGT_EDOMAIN
    ok = b"c1\t1\t0.1\t0.2\n" + b"c1\t2\t0.5\t0.2 " + b"x" * 60_000 + b"\nc1\t3\t0.1\t0.2\n"  # long but under the cap: extra columns
    assert check(ctx, ok, FST) == 3


def test_text_beyond_4_gib(ctx):
    """Maximum sizes: 4.6 GB of text (byte offsets, run-name offsets and block counts beyond 2^32): a 10^6-line
    block with two chromosome names, repeated 160 times with a few irregular lines (slow path) in every block."""
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 20e9:
        pytest.skip("needs ~12 GB of free HBM")
    rng = np.random.default_rng(8)
    m = 1_000_000
    pos = np.arange(1, m + 1, dtype=np.uint32)
    a = np.round(rng.uniform(-0.1, 0.6, m), 6)
    b = np.round(rng.uniform(0.0, 0.3, m), 6)
    lines = [b"chrA\t%d\t%.6f\t%.6f\n" % (p, x, y) if i < m // 2 else b"chrBB\t%d\t%.6f\t%.6f\n" % (p, x, y)
             for i, (p, x, y) in enumerate(zip(pos, a, b))]
    for i in (17, 500_003, 999_999):  # irregular lines: 17 significant digits, exponent form, a '+' sign
        lines[i] = lines[i].split(b"\t")[0] + b"\t%d\t0.12345678901234567\t+2.5e-1\n" % pos[i]
        a[i], b[i] = 0.12345678901234567, 0.25
    block = b"".join(lines)
    reps = 160
    text = block * reps
    assert len(text) > (1 << 32) + (1 << 28)
    ing = ctx.ingest_text(text, FST)
    assert ing.bad_line == -1 and ing.rows == m * reps
    assert ing.run_names == ["chrA", "chrBB"] * reps
    assert np.array_equal(ing.run_len, np.full(2 * reps, m // 2, dtype=np.uint64))
    for k, want in ((1, pos), (2, a), (3, b)):
        got = ing.column_np(k).reshape(reps, m)
        assert np.array_equal(got[0].view(np.uint8), want.view(np.uint8))       # the bits of the first block
        assert (got == got[0]).all()                                            # ... and of every repeat
    ing.free()


def test_locus_id_prefix_names_the_chromosome(ctx):
    """PGT_TOK_CHR_PREFIX (round 4; ihsWindow / xpehhWindow): the run name is the first token up to its first '_'
    (extractChr, ihsWindow.cpp:80-92) — ids that differ only behind the '_' stay in one run, an id without '_' is the name
    as a whole, an id that starts with '_' gives the empty name; everything else as with PGT_TOK_CHR."""
    rng = np.random.default_rng(12)
    n = 300_000
    chr_ids, pos = synth.chromosomes(rng, n, 11, equal=False)
    sc = np.round(rng.normal(0, 1, n), 5)
    text = "".join(f"chr{c}_{p}\t{p}\t0.2\t1.5\t2.5\t0.1\t{v}\t1\n" for c, p, v in zip(chr_ids, pos, sc)).encode()
    assert check(ctx, text, IHS) == n
    ing = ctx.ingest_text(text, IHS)
    assert ing.run_names == [f"chr{c}" for c in range(11)]
    ing.free()
    text = "".join(f"sc{c}_{p}_x\t{p}\t1\t2\t3\t4\t5\t6\t{v}\textra\n" for c, p, v in zip(chr_ids, pos, sc)).encode()
    assert check(ctx, text, XPEHH) == n
    odd = (b"chr1_5\t5\t0\t0\t0\t0\t1.5\n" b"chr1_9_b\t9\t0\t0\t0\t0\t-2\n" b"chr1\t11\t0\t0\t0\t0\t3\n"   # one run: chr1
           b"chr10_1\t1\t0\t0\t0\t0\t1e-3\n" b"_7\t7\t0\t0\t0\t0\t4\n" b"_\t8\t0\t0\t0\t0\t4\n"                 # chr10, then the empty name twice
           b"chrX_1\t1\t0\t0\t0\t0\t+1.25E+1\n" b"chrX_2\t2\t0\t0\t0\t0\tnan\n")
    check(ctx, odd, IHS)
    ing = ctx.ingest_text(odd, IHS)
    assert ing.run_names == ["chr1", "chr10", "", "chrX"] and ing.run_len.tolist() == [3, 1, 2, 2]
    ing.free()
    for bad_tokens in ([PGT_TOK_U32, PGT_TOK_U32], [PGT_TOK_CHR, PGT_TOK_CHR_PREFIX], [PGT_TOK_CHR_PREFIX] + [PGT_TOK_SKIP] * 12):
        with pytest.raises(_lib.PgtError):
            ctx.ingest_text(b"a 1\n", bad_tokens)
