"""GPU: the HIP path (through the C-ABI of libpgtwin.so) against the oracle, the reference-made
golden fixtures and size-independent properties.

Bar (north_star): coordinates, counts and labels bit-exact; floats within 1e-9 relative (plus a
1e-12 absolute floor for ratios whose numerator cancels to ~0).
"""
import numpy as np
import pytest

import helpers
import synth
from popgenomicstools_amd import _lib
from popgenomicstools_amd._lib import FST_ROW_DTYPE, HET_ROW_DTYPE, DXY_ROW_DTYPE, DXY_TOTAL_DTYPE, WIN_DTYPE
from popgenomicstools_amd.window_scan import rows_from_device, windows_to_device

pytestmark = pytest.mark.gpu

REL, ABS = helpers.REL, helpers.ABS


def assert_close(got, ref, what=""):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    bad = np.abs(got - ref) > REL * np.abs(ref) + ABS
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.size} beyond 1e-9 (first: {got[bad][:3]} vs {ref[bad][:3]})"


def check_fst(pgt, ctx, oracle, chr_ids, pos, a, b, W, S):
    ref = oracle.fst_scan(chr_ids, pos, a, b, W, S)
    res = pgt.fst_window(chr_ids, pos, a, b, W, S, ctx=ctx)
    rows, win = res.rows, res.win
    assert rows.size == ref.size
    for f, g in (("start", "start"), ("end", "end"), ("mid", "mid"), ("n", "n")):
        assert np.array_equal(rows[f], ref[g]), f
    assert np.array_equal(win["label_run"], ref["label"])
    assert_close(rows["asum"], ref["num"], "asum")
    assert_close(rows["bsum"], ref["den"], "bsum")
    assert_close(rows["fst"], ref["value"], "fst")
    return rows


def check_het(pgt, ctx, oracle, chr_ids, pos, g, W, S):
    ref = oracle.het_scan(chr_ids, pos, g, W, S)
    res = pgt.het_window(chr_ids, pos, g, W, S, ctx=ctx)
    rows = res.rows
    assert rows.size == ref.size
    for f, gname in (("start", "start"), ("end", "end"), ("mid", "mid"), ("nonmissing", "n")):
        assert np.array_equal(rows[f], ref[gname]), f
    assert np.array_equal(rows["nhet"], ref["num"].astype(np.uint32))
    assert np.array_equal(res.win["label_run"], ref["label"])
    assert np.array_equal(rows["h"], ref["value"])  # integer counts -> the one division is identical
    return rows


def check_dxy(pgt, ctx, oracle, chr_ids, pos, p1, p2, n1, n2, W, S, minind, fixedsite, skip, chr_len=None):
    ref, rtot = oracle.dxy_scan(chr_ids, pos, p1, p2, n1, n2, W, S, minind, fixedsite, skip, chr_len)
    ref = ref[ref["printed"] == 1]
    res = pgt.dxy_window(chr_ids, pos, p1, p2, n1, n2, W, S, minind, fixedsite, chr_len, skip, ctx=ctx)
    rows = res.rows
    assert rows.size == ref.size
    assert np.array_equal(rows["start"], ref["start"]) and np.array_equal(rows["end"], ref["end"])
    assert np.array_equal(rows["neff"], ref["n"]) and np.array_equal(rows["nskip"], ref["nskip"])
    assert np.array_equal(res.win["label_run"], ref["label"])
    assert_close(rows["sum"], ref["value"], "dxy sum")
    assert int(res.total["neff"]) == int(rtot["neff"]) and int(res.total["nskip"]) == int(rtot["nskip"])
    assert_close([res.total["sum"]], [rtot["sum"]], "dxy total")
    return rows


# ---------------------------------------------------------------------------------------------
# fst
# ---------------------------------------------------------------------------------------------
def test_fst_golden_fixtures(pgt, ctx):
    """Rows against the TSV printed by the unmodified reference binary."""
    cases = helpers.load_golden("ref_random.json")["cases"] + helpers.load_golden("ref_kat.json")["cases"]
    n = 0
    for c in cases:
        if c["tool"] != "fstWindow":
            continue
        names, chr_ids, pos, a, b = helpers.parse_table(c["input"], "fst")
        res = pgt.fst_window(chr_ids, pos, a, b, c["W"], c["S"], ctx=ctx)
        helpers.assert_rows_match_tsv(names, res.win, res.rows, "fst", "n", helpers.parse_tsv(c["stdout"]))
        n += 1
    assert n > 60


def test_fst_config1_golden(pgt, ctx):
    g = helpers.load_golden("ref_config1.json")
    rng = np.random.default_rng(g["seed"])
    chr_ids, pos = synth.chromosomes(rng, g["n"], g["n_chr"])
    a, b = synth.fst_columns(rng, g["n"])
    names = [f"chr{i + 1}" for i in range(g["n_chr"])]
    for run in g["runs"]:
        res = pgt.fst_window(chr_ids, pos, a, b, run["W"], run["S"], ctx=ctx)
        helpers.assert_rows_match_tsv(names, res.win, res.rows, "fst", "n", helpers.parse_tsv(run["stdout"]))


@pytest.mark.parametrize("n,n_chr,W,S", [
    (1, 1, 1, 1), (2, 1, 2, 1), (127, 1, 5, 5), (128, 1, 128, 128), (129, 2, 128, 1), (8191, 1, 100, 50),
    (8192, 1, 8192, 8192), (8193, 3, 4096, 4096), (16384, 1, 16384, 1000), (20000, 4, 7, 3),
    (100_000, 20, 50_000, 10_000), (300_000, 7, 50_000, 10_000), (300_000, 7, 1000, 1), (1_000_003, 5, 600_000, 250_000),
])
def test_fst_vs_oracle(pgt, ctx, oracle, n, n_chr, W, S):
    rng = np.random.default_rng(n * 31 + W)
    chr_ids, pos = synth.chromosomes(rng, n, n_chr, equal=False)
    a, b = synth.fst_columns(rng, n)
    rows = check_fst(pgt, ctx, oracle, chr_ids, pos, a, b, W, S)
    if S <= W and n > W:
        assert rows.size > 0


def test_fst_random_small_sweep(pgt, ctx, oracle):
    rng = np.random.default_rng(11)
    for _ in range(150):
        n = int(rng.integers(1, 3000))
        W = int(rng.integers(1, 400))
        S = int(rng.integers(1, W + 1))
        chr_ids, pos = synth.chromosomes(rng, n, int(rng.integers(1, min(n, 6) + 1)), equal=False)
        a, b = synth.fst_columns(rng, n)
        check_fst(pgt, ctx, oracle, chr_ids, pos, a, b, W, S)


def test_fst_edge_values(pgt, ctx, oracle):
    # Q6 zero denominator, negative zero numerator, cancellation, Q4 midpoint wrap
    chr_ids = np.zeros(6, dtype=np.uint32)
    pos = np.array([5, 6, 7, 3_000_000_000, 4_000_000_000, 4_100_000_000], dtype=np.uint64).astype(np.uint32)
    a = np.array([0.1, -0.05, -0.0, -0.0, 0.25, -0.25])
    b = np.array([0.0, 0.0, 0.5, 0.5, 0.0, 0.0])
    for W, S in [(2, 1), (1, 1), (3, 2), (6, 6)]:
        rows = check_fst(pgt, ctx, oracle, chr_ids, pos, a, b, W, S)
        assert not (np.signbit(rows["fst"]) & (rows["fst"] == 0)).any()  # the reference never prints -0
    rows = check_fst(pgt, ctx, oracle, chr_ids, pos, a, b, 2, 1)
    assert rows["mid"][3] == (3_000_000_000 + 4_000_000_000) % 2**32 // 2


def test_fst_empty_inputs(pgt, ctx):
    res = pgt.fst_window(np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(0), np.zeros(0), 5, 2, ctx=ctx)
    assert res.rows.size == 0
    res = pgt.fst_window(np.zeros(3, np.uint32), np.arange(3, dtype=np.uint32), np.ones(3), np.ones(3), 5, 2, ctx=ctx)
    assert res.rows.size == 0  # N <= W-S prints nothing (Q2)


def test_fst_argument_errors(pgt, ctx):
    pos = np.arange(10, dtype=np.uint32)
    win = np.zeros(1, dtype=WIN_DTYPE)
    win["lo"], win["hi"] = 0, 11  # beyond the columns
    with pytest.raises(_lib.PgtError):
        ctx.fst_reduce(pos, np.ones(10), np.ones(10), win)
    win["lo"], win["hi"] = 5, 4
    with pytest.raises(_lib.PgtError):
        ctx.fst_reduce(pos, np.ones(10), np.ones(10), win)
    with pytest.raises(_lib.PgtError):
        pgt.fst_window(np.zeros(10, np.uint32), pos, np.ones(10), np.ones(10), 3, 4, ctx=ctx)  # S > W


# ---------------------------------------------------------------------------------------------
# het
# ---------------------------------------------------------------------------------------------
def test_het_golden_fixtures(pgt, ctx):
    cases = helpers.load_golden("ref_random.json")["cases"] + helpers.load_golden("ref_kat.json")["cases"]
    n = 0
    for c in cases:
        if c["tool"] != "hetWindow":
            continue
        names, chr_ids, pos, g = helpers.parse_table(c["input"], "het")
        res = pgt.het_window(chr_ids, pos, g, c["W"], c["S"], ctx=ctx)
        helpers.assert_rows_match_tsv(names, res.win, res.rows, "h", "nonmissing", helpers.parse_tsv(c["stdout"]))
        n += 1
    assert n > 60


@pytest.mark.parametrize("n,n_chr,W,S", [
    (1, 1, 1, 1), (1023, 1, 10, 3), (1024, 1, 1024, 1024), (1025, 2, 1000, 1), (65535, 1, 5000, 2500),
    (65536, 1, 65536, 65536), (65537, 3, 30000, 10000), (200_000, 20, 50_000, 10_000), (5_000_000, 9, 4_500_000, 250_000),
])
def test_het_vs_oracle(pgt, ctx, oracle, n, n_chr, W, S):
    rng = np.random.default_rng(n + 7 * S)
    chr_ids, pos = synth.chromosomes(rng, n, n_chr, equal=False)
    g = synth.het_column(rng, n)
    g[rng.integers(0, n, size=max(1, n // 50))] = rng.integers(-5, 9, size=max(1, n // 50))  # other codes: >=0 is non-missing
    check_het(pgt, ctx, oracle, chr_ids, pos, g, W, S)


# ---------------------------------------------------------------------------------------------
# dxy
# ---------------------------------------------------------------------------------------------
def test_dxy_known_answers(pgt, ctx):
    k = helpers.load_golden("dxy_kat.json")
    names = [r[0] for r in k["sizes"]]
    chr_ids = np.array([names.index(r[0]) for r in k["pop1"]], dtype=np.uint32)
    pos = np.array([r[1] for r in k["pop1"]], dtype=np.uint32)
    p1 = np.array([r[2] for r in k["pop1"]]); n1 = np.array([r[3] for r in k["pop1"]], dtype=np.int32)
    p2 = np.array([r[2] for r in k["pop2"]]); n2 = np.array([r[3] for r in k["pop2"]], dtype=np.int32)
    chr_len = np.array([r[1] for r in k["sizes"]], dtype=np.uint32)
    for c in k["cases"]:
        res = pgt.dxy_window(chr_ids, pos, p1, p2, n1, n2, c["winsize"], c["stepsize"], k["minind"],
                             c["fixedsite"], chr_len, c["skip_missing"], ctx=ctx)
        lines = "".join(f"{names[int(w['label_run'])]}\t{int(r['start'])}\t{int(r['end'])}\t{helpers.fmt_g(r['sum'])}\t{int(r['neff'])}\t{int(r['nskip'])}\n"
                        for w, r in zip(res.win, res.rows))
        total = f"{helpers.fmt_g(res.total['sum'])}\t{int(res.total['neff'])}\t{int(res.total['nskip'])}\n"
        if c["winsize"] == 0:
            assert total == c["stdout"] and lines == ""
        else:
            assert lines == c["stdout"] and total == c["stderr"]


def test_dxy_rows_against_hand_walked_cases(pgt, ctx):
    """C-ABI rows (window table + dxy kernels) against tests/golden/dxy_hand_walked.json: outputs derived by stepping
    through dxyWindow.cpp:172-209,282-433 on paper (the walk is in the fixture).  The columns handed to the C-ABI are the
    sites both files list (the host's merge); where the reference itself mis-pairs or truncates such input the fixture
    holds the product's documented output (deliberate divergences) next to the reference's."""
    k = helpers.load_golden("dxy_hand_walked.json")
    done = 0
    for c in k["cases"]:
        names = []
        for r in c["pop1"]:
            if not names or names[-1] != r[0]:
                names.append(r[0])
        key2 = {(r[0], r[1]): r for r in c["pop2"]}
        both = [(r, key2[(r[0], r[1])]) for r in c["pop1"] if (r[0], r[1]) in key2]
        if not both:  # H10: no shared site — the host refuses before the C-ABI is reached
            assert c["product"]["rc"] == 255
            continue
        chr_ids = np.array([names.index(a[0]) for a, _ in both], dtype=np.uint32)
        pos = np.array([a[1] for a, _ in both], dtype=np.uint32)
        p1 = np.array([a[2] for a, _ in both]); n1 = np.array([a[3] for a, _ in both], dtype=np.int32)
        p2 = np.array([b[2] for _, b in both]); n2 = np.array([b[3] for _, b in both], dtype=np.int32)
        sizes = dict(c["sizes"] or [])
        for r in c["runs"]:
            want_out, want_err = helpers.hand_walked_product_expectation(c, r)
            chr_len = None if r["fixedsite"] else np.array([sizes[nm] for nm in names], dtype=np.uint32)
            res = pgt.dxy_window(chr_ids, pos, p1, p2, n1, n2, r["winsize"], r["stepsize"], c["minind"], r["fixedsite"], chr_len,
                                 r["skip_missing"], ctx=ctx)
            lines = "".join(f"{names[int(w['label_run'])]}\t{int(q['start'])}\t{int(q['end'])}\t{helpers.fmt_g(q['sum'])}\t{int(q['neff'])}\t{int(q['nskip'])}\n"
                            for w, q in zip(res.win, res.rows))
            total = f"{helpers.fmt_g(res.total['sum'])}\t{int(res.total['neff'])}\t{int(res.total['nskip'])}\n"
            if r["winsize"] == 0:
                assert total == want_out and lines == "", c["name"]
            else:
                assert lines == want_out, (c["name"], r, lines)
                assert total == want_err, (c["name"], r, total)
            done += 1
    assert done >= 11


@pytest.mark.parametrize("fixedsite", [1, 0])
def test_dxy_vs_oracle_random(pgt, ctx, oracle, fixedsite):
    rng = np.random.default_rng(3 + fixedsite)
    for trial in range(60):
        n_runs = int(rng.integers(1, 5))
        pos_l, chr_l, len_l = [], [], []
        for r in range(n_runs):
            L = int(rng.integers(1, 4000))
            k = int(rng.integers(1, min(L, 700) + 1))
            p = np.sort(rng.choice(np.arange(1, L + 1), size=k, replace=False))
            pos_l.append(p); chr_l.append(np.full(k, r)); len_l.append(L)
        pos = np.concatenate(pos_l).astype(np.uint32)
        chr_ids = np.concatenate(chr_l).astype(np.uint32)
        p1, p2, n1, n2 = synth.dxy_columns(rng, pos.size)
        W = int(rng.integers(1, 900)); S = int(rng.integers(1, W + 1))
        check_dxy(pgt, ctx, oracle, chr_ids, pos, p1, p2, n1, n2, W, S, 5, fixedsite, int(rng.integers(0, 2)),
                  np.array(len_l, dtype=np.uint32))


def test_dxy_rows_against_reference_made_cases(pgt, ctx, tmp_path):
    """The C-ABI rows (pgt_build_windows_* + pgt_dxy_reduce) against the unmodified reference dxyWindow's recorded
    stdout / genome-wide line, for the fixture cases whose two MAF files list the same sites (the column interface
    takes synchronised populations; nested sets go through the CLI test).  Skipped where the fixture cannot exist."""
    cases = helpers.dxy_ref_cases(tmp_path, plain_text=True)
    if cases is None:
        pytest.skip("tests/golden/ref_dxy.json absent (no Boost in this image): dxy parity unpinned")

    def maf(path):
        rows = [ln.split() for ln in open(path).read().splitlines()[1:] if ln.strip()]
        return [r[0] for r in rows], np.array([int(r[1]) for r in rows], np.uint32), np.array([float(r[5]) for r in rows]), \
            np.array([int(r[6]) for r in rows], np.int32)
    used = 0
    for c, argv, o in cases:
        if c["rc"] != 0:
            continue
        (c1, q1, f1, k1), (c2, q2, f2, k2) = maf(o["maf1"]), maf(o["maf2"])
        if c1 != c2 or not np.array_equal(q1, q2):
            continue
        names = [c1[0]] + [b for a, b in zip(c1, c1[1:]) if a != b]
        chr_ids = np.cumsum([0] + [int(a != b) for a, b in zip(c1, c1[1:])]).astype(np.uint32)
        chr_len = None
        if not o["fixedsite"]:
            sizes = {}
            for ln in open(o["sizefile"]):
                sizes.setdefault(ln.split()[0], int(ln.split()[1]))  # first entry of a name wins (std::map::insert)
            chr_len = np.array([sizes[nm] for nm in names], np.uint32)
        res = pgt.dxy_window(chr_ids, q1, f1, f2, k1, k2, o["winsize"], o["stepsize"], o["minind"], o["fixedsite"], chr_len,
                             o["skip_missing"], ctx=ctx)
        tsv = helpers.parse_tsv(c["stdout"] if o["winsize"] else "")
        assert len(tsv) == res.rows.size, c["args"]
        for r, w, t in zip(res.rows, res.win, tsv):
            assert names[int(w["label_run"])] == t[0] and int(r["start"]) == int(t[1]) and int(r["end"]) == int(t[2])
            assert int(r["neff"]) == int(t[4]) and int(r["nskip"]) == int(t[5])
            assert abs(float(r["sum"]) - float(t[3])) <= 1e-9 * abs(float(t[3])) + 5e-6 * max(abs(float(t[3])), 1e-4)
        g = (c["stderr"] if o["winsize"] else c["stdout"]).split()
        assert int(res.total["neff"]) == int(g[1]) and int(res.total["nskip"]) == int(g[2])
        used += 1
    assert used >= 20


def test_dxy_large_and_global(pgt, ctx, oracle):
    rng = np.random.default_rng(8)
    n = 400_000
    chr_ids, pos = synth.chromosomes(rng, n, 6, equal=False)
    p1, p2, n1, n2 = synth.dxy_columns(rng, n)
    check_dxy(pgt, ctx, oracle, chr_ids, pos, p1, p2, n1, n2, 50_000, 10_000, 5, 1, 0)
    check_dxy(pgt, ctx, oracle, chr_ids, pos, p1, p2, n1, n2, 0, 0, 5, 1, 0)  # -winsize 0: global only
    chr_len = np.array([int(pos[chr_ids == c].max()) + 17 for c in range(6)], dtype=np.uint32)
    check_dxy(pgt, ctx, oracle, chr_ids, pos, p1, p2, n1, n2, 50_000, 10_000, 5, 0, 1, chr_len)


def test_dxy_site_value_is_bitwise_the_host_formula(pgt, ctx):
    """-winsize 1 -stepsize 1 exposes the per-site value: it must equal p1*(1-p2)+p2*(1-p1) computed
    with separately rounded products (no FMA contraction), bit for bit (dxyWindow.cpp:381)."""
    rng = np.random.default_rng(9)
    n = 20_000
    p1, p2 = rng.uniform(0, 1, n), rng.uniform(0, 1, n)
    n1 = np.full(n, 9, dtype=np.int32)
    res = pgt.dxy_window(np.zeros(n, np.uint32), np.arange(1, n + 1, dtype=np.uint32), p1, p2, n1, n1, 1, 1, 1, 1, ctx=ctx)
    assert np.array_equal(res.rows["sum"], p1 * (1.0 - p2) + p2 * (1.0 - p1))


# ---------------------------------------------------------------------------------------------
# device-resident entry points, properties at scale
# ---------------------------------------------------------------------------------------------
def _device_fst(ctx, pos, a, b, win):
    import torch
    dev = torch.device("cuda:0")
    tp = torch.from_numpy(pos.view(np.int32)).to(dev)
    ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    out, tree = ctx.fst_reduce_dev(tp, ta, tb, windows_to_device(win, dev))
    torch.cuda.synchronize()
    return rows_from_device(out, FST_ROW_DTYPE)


def test_device_api_equals_host_api_bitwise(pgt, ctx):
    rng = np.random.default_rng(21)
    n = 700_001
    chr_ids, pos = synth.chromosomes(rng, n, 5, equal=False)
    a, b = synth.fst_columns(rng, n)
    win = pgt.build_windows_sites(pgt.run_lengths(chr_ids), 50_000, 10_000)
    host = ctx.fst_reduce(pos, a, b, win)
    dev = _device_fst(ctx, pos, a, b, win)
    assert host.tobytes() == dev.tobytes()
    assert _device_fst(ctx, pos, a, b, win).tobytes() == dev.tobytes()  # idempotent, deterministic


def test_profiling_reports_both_kernels(pgt, ctx):
    rng = np.random.default_rng(22)
    n = 2_000_000
    chr_ids, pos = synth.chromosomes(rng, n, 4)
    a, b = synth.fst_columns(rng, n)
    win = pgt.build_windows_sites(pgt.run_lengths(chr_ids), 50_000, 10_000)
    ctx.set_profiling(True)
    try:
        _device_fst(ctx, pos, a, b, win)
        build_ms, query_ms = ctx.last_kernel_ms()
    finally:
        ctx.set_profiling(False)
    assert 0 < build_ms < 50 and 0 < query_ms < 50


def test_pairs_equal_singles_bitwise(pgt, ctx):
    import torch
    rng = np.random.default_rng(23)
    n, n_pairs = 300_000, 5
    chr_ids, pos = synth.chromosomes(rng, n, 3, equal=False)
    cols = [synth.fst_columns(rng, n) for _ in range(n_pairs)]
    win = pgt.build_windows_sites(pgt.run_lengths(chr_ids), 20_000, 5_000)
    dev = torch.device("cuda:0")
    tp = torch.from_numpy(pos.view(np.int32)).to(dev)
    ta = [torch.from_numpy(c[0]).to(dev) for c in cols]
    tb = [torch.from_numpy(c[1]).to(dev) for c in cols]
    out, _ = ctx.fst_reduce_pairs_dev(tp, ta, tb, windows_to_device(win, dev))
    torch.cuda.synchronize()
    rows = rows_from_device(out, FST_ROW_DTYPE).reshape(n_pairs, win.size)
    for p in range(n_pairs):
        single = ctx.fst_reduce(pos, cols[p][0], cols[p][1], win)
        assert rows[p].tobytes() == single.tobytes()


def test_sharded_equals_single_bitwise(pgt, ctx):
    """The multi-GPU plan on one GPU: every shard reduced on its own columns gives, concatenated,
    exactly the bytes of the single-GPU run (shard starts are tree-node aligned)."""
    from popgenomicstools_amd.distributed import shard_windows
    rng = np.random.default_rng(24)
    n = 1_500_000
    chr_ids, pos = synth.chromosomes(rng, n, 11, equal=False)
    a, b = synth.fst_columns(rng, n)
    win = pgt.build_windows_sites(pgt.run_lengths(chr_ids), 50_000, 10_000)
    single = ctx.fst_reduce(pos, a, b, win)
    for world in (2, 3, 8):
        parts = []
        for rank in range(world):
            s, local, _ = shard_windows(win, rank, world)
            lo, hi = int(s["site_lo"]), int(s["site_hi"])
            parts.append(ctx.fst_reduce(pos[lo:hi], a[lo:hi], b[lo:hi], local))
        assert np.concatenate(parts).tobytes() == single.tobytes()


def test_full_size_properties(pgt, ctx):
    """10^8 sites (BASELINE configs[1]) generated on the device: properties that need no oracle.
    (1) scaling a and b by 2 is exact in binary floating point: sums double bit for bit, fst is unchanged;
    (2) non-overlapping windows: the window sums add up to the genome total within 1e-9;
    (3) coordinates: start/end/mid/n follow pos and the table exactly."""
    import torch
    dev = torch.device("cuda:0")
    n, n_chr, W, S = 100_000_000, 20, 50_000, 10_000
    gen = torch.Generator(device=dev).manual_seed(12345)
    b = torch.round(torch.rand(n, generator=gen, device=dev, dtype=torch.float64) * 0.3 * 1e6) / 1e6
    a = torch.round(b * (torch.rand(n, generator=gen, device=dev, dtype=torch.float64) * 0.7 - 0.1) * 1e6) / 1e6
    per = n // n_chr
    pos = torch.randint(1, 60, (n_chr, per), generator=gen, device=dev, dtype=torch.int32).cumsum(1, dtype=torch.int32).reshape(-1)
    run_len = np.full(n_chr, per, dtype=np.uint64)
    win = pgt.build_windows_sites(run_len, W, S)
    wt = windows_to_device(win, dev)
    out, tree = ctx.fst_reduce_dev(pos, a, b, wt)
    out2, _ = ctx.fst_reduce_dev(pos, a * 2.0, b * 2.0, wt)
    torch.cuda.synchronize()
    r1, r2 = rows_from_device(out, FST_ROW_DTYPE), rows_from_device(out2, FST_ROW_DTYPE)
    assert r1.size == win.size and win.size > n_chr * ((per - W) // S)
    assert np.array_equal(r2["asum"], 2.0 * r1["asum"]) and np.array_equal(r2["bsum"], 2.0 * r1["bsum"])
    assert np.array_equal(r2["fst"], r1["fst"])
    # coordinates
    hpos = pos.cpu().numpy().view(np.uint32)
    assert np.array_equal(r1["start"], hpos[win["lo"]]) and np.array_equal(r1["end"], hpos[win["hi"] - 1])
    assert np.array_equal(r1["mid"], ((r1["start"].astype(np.uint64) + r1["end"]) % 2**32 // 2).astype(np.uint32))
    assert np.array_equal(r1["n"], (win["hi"] - win["lo"]).astype(np.uint32))
    # a sample of windows against float64 sums done by torch (independent code path)
    for i in np.linspace(0, win.size - 1, 25).astype(int):
        lo, hi = int(win["lo"][i]), int(win["hi"][i])
        assert_close([r1["asum"][i]], [float(a[lo:hi].sum())], "asum sample")
        assert_close([r1["bsum"][i]], [float(b[lo:hi].sum())], "bsum sample")
    # non-overlapping cover
    win_t = pgt.build_windows_sites(run_len, W, W)
    out_t, _ = ctx.fst_reduce_dev(pos, a, b, windows_to_device(win_t, dev), tree=tree)
    torch.cuda.synchronize()
    rt = rows_from_device(out_t, FST_ROW_DTYPE)
    assert int(rt["n"].astype(np.int64).sum()) == n
    assert_close([rt["bsum"].sum()], [float(b.sum())], "cover bsum")
    assert_close([rt["asum"].sum()], [float(a.sum())], "cover asum")


@pytest.mark.timeout(1200)
def test_config3_exact_workload_sharded_eight_ways(pgt, ctx, oracle):
    """BASELINE configs[3] in its stated shape — fstWindow, ONE genome of 10^9 sites in 40 chromosomes, W = 50000, S = 10000,
    the 8-way pgt_plan_shards plan of the 8-GPU run — on the one GPU: (i) the eight shards, each reduced from its own
    column range [site_lo, site_hi) with its re-based table, give concatenated the bytes of the single-GPU table;
    (ii) the windows inside the first 5x10^7 sites against the oracle's streaming machine (fstWindow.cpp:109-155
    restated) at 1e-9, coordinates and counts exact; (iii) coordinates / n of all 99 996 rows against pos and the table."""
    import torch
    from synth_genome import SynthGenome
    dev = torch.device("cuda:0")
    free, _ = torch.cuda.mem_get_info(dev)
    if free < 60 * (1 << 30):
        pytest.skip("needs 60 GB of free HBM")
    n, n_chr, W, S = 1_000_000_000, 40, 50_000, 10_000
    g = SynthGenome(12345, n, n_chr)
    win = pgt.build_windows_sites(g.run_len, W, S)
    assert win.size == 99_996
    pos, a, b = g.fst_columns_t(0, n, dev)
    with ctx.hints(W, S):
        single = rows_from_device(ctx.fst_reduce_dev(pos, a, b, windows_to_device(win, dev))[0], FST_ROW_DTYPE)
        # (i) the 8-GPU plan, shard by shard
        shards = pgt.plan_shards(win, 8)
        parts = []
        for s_ in shards:
            lo, hi = int(s_["site_lo"]), int(s_["site_hi"])
            local = np.array(win[int(s_["win_begin"]): int(s_["win_end"])], dtype=WIN_DTYPE, copy=True)
            local["lo"] -= np.uint64(lo)
            local["hi"] -= np.uint64(lo)
            assert lo % 65536 == 0 and 0.9 * n / 8 < hi - lo < 1.1 * n / 8 + 2 * W
            parts.append(rows_from_device(ctx.fst_reduce_dev(pos[lo:hi], a[lo:hi], b[lo:hi], windows_to_device(local, dev))[0],
                                          FST_ROW_DTYPE))
    assert np.concatenate(parts).tobytes() == single.tobytes()
    # (iii) coordinates and counts of every row
    lo_t = torch.from_numpy(win["lo"].astype(np.int64)).to(dev)
    hi_t = torch.from_numpy(win["hi"].astype(np.int64)).to(dev)
    start = pos[lo_t].cpu().numpy().view(np.uint32)
    end = pos[hi_t - 1].cpu().numpy().view(np.uint32)
    assert np.array_equal(single["start"], start) and np.array_equal(single["end"], end)
    assert np.array_equal(single["mid"], ((start.astype(np.uint64) + end) % 2**32 // 2).astype(np.uint32))
    assert np.array_equal(single["n"], (win["hi"] - win["lo"]).astype(np.uint32))
    # (ii) the prefix against the oracle (the reference's own streaming loop, restated and pinned for fst)
    m = 50_000_000
    per = n // n_chr
    assert m % per == 0  # the prefix ends on a chromosome boundary: its windows are exactly the table's windows with hi <= m
    ref = oracle.fst_scan(g.chr_ids_np(0, m), pos[:m].cpu().numpy().view(np.uint32), a[:m].cpu().numpy(), b[:m].cpu().numpy(), W, S)
    k = int(np.searchsorted(win["hi"], m, side="right"))
    assert ref.size == k >= 4900 and np.all(win["hi"][:k] <= m)
    for f, r in (("start", "start"), ("end", "end"), ("mid", "mid"), ("n", "n")):
        assert np.array_equal(single[f][:k], ref[r]), f
    assert_close(single["fst"][:k], ref["value"], "fst vs oracle on the 5e7-site prefix")


def test_fused_dxy_het_equals_separate_bitwise(pgt, ctx, oracle):
    """BASELINE config 3 entry point: same bytes as the three separate reductions, and the oracle's
    counts."""
    import torch
    rng = np.random.default_rng(31)
    n = 900_001
    chr_ids, pos = synth.chromosomes(rng, n, 5, equal=False)
    p1, p2, n1, n2 = synth.dxy_columns(rng, n)
    g1 = synth.het_column(rng, n).astype(np.int8)
    g2 = synth.het_column(rng, n).astype(np.int8)
    W, S, minind = 50_000, 10_000, 5
    win = pgt.build_windows_sites(pgt.run_lengths(chr_ids), W, S)
    dev = torch.device("cuda:0")
    t = lambda x: torch.from_numpy(x).to(dev)
    dxy_out, tot, h1, h2, _ = ctx.dxy_het_reduce_dev(t(pos.view(np.int32)), t(p1), t(p2), t(n1), t(n2), t(g1), t(g2),
                                                    minind, windows_to_device(win, dev))
    torch.cuda.synchronize()
    d_sep, tot_sep = ctx.dxy_reduce(pos, p1, p2, n1, n2, minind, win)
    assert rows_from_device(dxy_out, DXY_ROW_DTYPE).tobytes() == d_sep.tobytes()
    assert rows_from_device(tot, DXY_TOTAL_DTYPE).tobytes() == np.array([tot_sep]).tobytes()
    assert rows_from_device(h1, HET_ROW_DTYPE).tobytes() == ctx.het_reduce(pos, g1, win).tobytes()
    assert rows_from_device(h2, HET_ROW_DTYPE).tobytes() == ctx.het_reduce(pos, g2, win).tobytes()
    ref = oracle.het_scan(chr_ids, pos, g2.astype(np.int32), W, S)
    assert np.array_equal(rows_from_device(h2, HET_ROW_DTYPE)["nonmissing"], ref["n"])


def test_rccl_gather_single_rank(pgt, ctx):
    """The bench's gather path on the RCCL backend (one rank is all a 1-GPU box allows): process
    group init with the nccl backend, barrier, and gather_rows through torch.distributed."""
    import torch
    import torch.distributed as dist
    from popgenomicstools_amd.distributed import gather_rows
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", world_size=1, rank=0, device_id=dev)
    try:
        rows = torch.arange(40 * 7, dtype=torch.uint8, device=dev)
        send = torch.zeros(40 * 7, dtype=torch.uint8, device=dev)
        send.copy_(rows)
        recv = [torch.empty_like(send)]
        dist.gather(send, recv, dst=0)  # the collective gather_rows issues when world > 1
        dist.barrier()
        torch.cuda.synchronize()
        assert torch.equal(recv[0], rows)
        assert torch.equal(gather_rows(rows, [7], 40, dst=0), rows)
    finally:
        dist.destroy_process_group()


def test_max_window_hint_only_changes_speed(pgt, ctx):
    """pgt_set_max_window: with the exact bound the rows are the same bytes (the skipped levels were
    never touched); with a bound that is too small the answers are still right (1e-9)."""
    import torch
    rng = np.random.default_rng(41)
    n = 3_000_000
    chr_ids, pos = synth.chromosomes(rng, n, 2)
    a, b = synth.fst_columns(rng, n)
    win = pgt.build_windows_sites(pgt.run_lengths(chr_ids), 1_200_000, 400_000)  # windows contain level-3 nodes
    base = _device_fst(ctx, pos, a, b, win)
    try:
        ctx.set_max_window(1_200_000)
        assert _device_fst(ctx, pos, a, b, win).tobytes() == base.tobytes()
        ctx.set_max_window(1000)  # wrong on purpose
        low = _device_fst(ctx, pos, a, b, win)
    finally:
        ctx.set_max_window(0)
    for f in ("start", "end", "mid", "n"):
        assert np.array_equal(low[f], base[f])
    assert_close(low["fst"], base["fst"], "fst under a too-small hint")
    assert_close(low["asum"], base["asum"], "asum under a too-small hint")


# ---------------------------------------------------------------------------------------------
# allele-frequency front end: WCFst() of betaAFOutlier.R:400-418 + fstWindow
# ---------------------------------------------------------------------------------------------
def _af_rows(ctx, pos, freqs, nsamp, win):
    import torch
    dev = torch.device("cuda:0")
    out, _ = ctx.fst_af_reduce_dev(torch.from_numpy(pos.view(np.int32)).to(dev),
                                   [torch.from_numpy(f).to(dev) for f in freqs], nsamp, windows_to_device(win, dev))
    torch.cuda.synchronize()
    n_pairs = len(freqs) * (len(freqs) - 1) // 2
    return rows_from_device(out, FST_ROW_DTYPE).reshape(n_pairs, win.size)


@pytest.mark.parametrize("n_pops,n,W,S", [(2, 1, 1, 1), (2, 127, 50, 7), (2, 300_000, 50_000, 10_000), (3, 8193, 1000, 1000),
                                          (4, 129, 128, 1), (5, 70_000, 9_000, 4_000), (8, 16_385, 5_000, 2_500),
                                          (8, 700_001, 50_000, 10_000), (8, 1_200_000, 600_000, 300_000),
                                          # round 6 (512-site leaf nodes, ragged sites read as 16-byte pairs, the column's last site alone):
                                          # column lengths around a leaf, a level-2 tile and odd ends, windows sliding by ONE site so that
                                          # every offset of a window's ends inside a pair and inside a leaf occurs
                                          (8, 511, 300, 1), (8, 512, 257, 1), (8, 513, 512, 1), (3, 1025, 513, 1), (8, 8191, 2000, 1),
                                          (2, 8193, 1030, 1), (8, 9217, 4100, 3), (5, 24_577, 8192, 37)])
def test_af_front_end_vs_oracle(pgt, ctx, oracle, n_pops, n, W, S):
    """All pairs from the frequency columns == fstWindow (oracle, sequential sums) run on the (a, a+b)
    columns that the literal restatement of WCFst() produces for that pair."""
    rng = np.random.default_rng(1000 * n_pops + n % 997)
    chr_ids, pos = synth.chromosomes(rng, n, min(n, 3), equal=False)
    base = rng.uniform(0.02, 0.98, n)
    freqs = [np.clip(np.round(base + rng.normal(0, 0.08, n), 6), 0.0, 1.0) for _ in range(n_pops)]
    nsamp = [float(x) for x in rng.integers(5, 40, n_pops)]
    win = pgt.build_windows_sites(pgt.run_lengths(chr_ids), W, S)
    rows = _af_rows(ctx, pos, freqs, nsamp, win)
    p = 0
    for i in range(n_pops):
        for j in range(i + 1, n_pops):
            a, ab = oracle.wcfst_columns(freqs[i], freqs[j], nsamp[i], nsamp[j])
            ref = oracle.fst_scan(chr_ids, pos, a, ab, W, S)
            r = rows[p]
            assert r.size == ref.size
            for f in ("start", "end", "mid", "n"):
                assert np.array_equal(r[f], ref[f])
            # Σ(a+b) is a sum of positives: relative 1e-9.  Σa cancels (a is negative where the
            # populations agree), so its error is measured against Σ(a+b), as is the ratio's.
            assert_close(r["bsum"], ref["den"], f"pair {i},{j} Σ(a+b)")
            assert np.all(np.abs(r["asum"] - ref["num"]) <= 1e-9 * np.abs(ref["den"]) + 1e-12)
            assert np.all(np.abs(r["fst"] - ref["value"]) <= 1e-9)
            p += 1
    assert p == rows.shape[0]


def test_af_front_end_against_exact_rational_evaluation(pgt, ctx):
    """The device AF front end against tests/golden/wcfst_exact.json (betaAFOutlier.R:400-418,440-446 in exact rationals,
    tests/golden/make_wcfst_exact.py): per-site windows give (a, a+b) of every site, one window over all sites gives
    genomeFst — no oracle in between.  Tolerance 1e-9 relative to the component's scale (north_star's bound)."""
    from fractions import Fraction
    k = helpers.load_golden("wcfst_exact.json")
    for c in k["cases"]:
        n = len(c["sites"])
        f1 = np.array([float(Fraction(s["f1"])) for s in c["sites"]])
        f2 = np.array([float(Fraction(s["f2"])) for s in c["sites"]])
        pos = np.arange(1, n + 1, dtype=np.uint32)
        runs = np.array([n], dtype=np.uint64)
        per_site = _af_rows(ctx, pos, [f1, f2], [float(c["n1"]), float(c["n2"])], pgt.build_windows_sites(runs, 1, 1))[0]
        ea = np.array([s["a_f64"] for s in c["sites"]])
        eab = np.array([s["a_plus_b_f64"] for s in c["sites"]])
        scale = np.maximum(np.abs(eab), np.abs(ea))
        assert per_site.size == n
        assert np.all(np.abs(per_site["bsum"] - eab) <= 1e-9 * scale + 1e-15), per_site["bsum"] - eab
        assert np.all(np.abs(per_site["asum"] - ea) <= 1e-9 * scale + 1e-15), per_site["asum"] - ea
        whole = _af_rows(ctx, pos, [f1, f2], [float(c["n1"]), float(c["n2"])], pgt.build_windows_sites(runs, n, n))[0]
        assert whole.size == 1 and int(whole["n"][0]) == n
        assert abs(float(whole["fst"][0]) - c["genome_fst_f64"]) <= 1e-9
        # the same two columns among eight populations: pair (0, 1) of 28 must not change
        rng = np.random.default_rng(5)
        others = [np.round(rng.uniform(0, 1, n), 6) for _ in range(6)]
        eight = _af_rows(ctx, pos, [f1, f2] + others, [float(c["n1"]), float(c["n2"])] + [9.0] * 6, pgt.build_windows_sites(runs, n, n))
        assert abs(float(eight[0]["fst"][0]) - c["genome_fst_f64"]) <= 1e-9


def test_af_front_end_equals_component_path(pgt, ctx, oracle):
    """Feeding the restated (a, a+b) columns through the ordinary fst path gives the same rows (1e-9)."""
    rng = np.random.default_rng(77)
    n = 400_000
    chr_ids, pos = synth.chromosomes(rng, n, 4)
    f1, f2 = np.round(rng.uniform(0, 1, n), 6), np.round(rng.uniform(0, 1, n), 6)
    win = pgt.build_windows_sites(pgt.run_lengths(chr_ids), 50_000, 10_000)
    af = _af_rows(ctx, pos, [f1, f2], [12.0, 20.0], win)[0]
    a, ab = oracle.wcfst_columns(f1, f2, 12.0, 20.0)
    comp = ctx.fst_reduce(pos, a, ab, win)
    assert_close(af["bsum"], comp["bsum"], "Σ(a+b)")
    assert np.all(np.abs(af["fst"] - comp["fst"]) <= 1e-9)


def test_device_entry_points_are_graph_capturable(pgt, ctx):
    """The *_dev calls allocate nothing and never synchronise, so a whole step (build + upper levels +
    query) can be captured into a HIP graph and replayed; replays on new column contents are right."""
    import torch
    rng = np.random.default_rng(51)
    n = 600_000
    chr_ids, pos = synth.chromosomes(rng, n, 3)
    a, b = synth.fst_columns(rng, n)
    win = pgt.build_windows_sites(pgt.run_lengths(chr_ids), 50_000, 10_000)
    dev = torch.device("cuda:0")
    tp = torch.from_numpy(pos.view(np.int32)).to(dev)
    ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    tw = windows_to_device(win, dev)
    out = torch.empty(win.size * FST_ROW_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    tree = torch.empty(ctx.tree_bytes(_lib.PGT_STAT_FST, n), dtype=torch.uint8, device=dev)
    ctx.fst_reduce_dev(tp, ta, tb, tw, out=out, tree=tree)  # warm-up outside the capture
    torch.cuda.synchronize()
    eager = rows_from_device(out, FST_ROW_DTYPE).copy()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ctx.fst_reduce_dev(tp, ta, tb, tw, out=out, tree=tree)  # launched on the capture stream
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert rows_from_device(out, FST_ROW_DTYPE).tobytes() == eager.tobytes()
    ta.mul_(2.0); tb.mul_(4.0)  # new contents, same buffers
    g.replay()
    torch.cuda.synchronize()
    again = rows_from_device(out, FST_ROW_DTYPE)
    assert np.array_equal(again["asum"], 2.0 * eager["asum"]) and np.array_equal(again["bsum"], 4.0 * eager["bsum"])


# ---------------------------------------------------------------------------------------------
# ihsWindow / xpehhWindow
# ---------------------------------------------------------------------------------------------
def _check_extreme(pgt, ctx, oracle, chr_ids, pos, score, W, mode, cutoff, chr_len):
    ref = oracle.extreme_scan(chr_ids, pos, score, W, mode, cutoff, chr_len)
    if mode == 0:
        res = pgt.ihs_window(chr_ids, pos, score, W, cutoff, chr_len, ctx=ctx)
    else:
        res = pgt.xpehh_window(chr_ids, pos, score, cutoff, W, chr_len, ctx=ctx)
    r = res.rows
    assert r.size == ref.size
    for f in ("start", "end", "nsites", "nbig", "position"):
        assert np.array_equal(r[f], ref[f]), f
    assert np.array_equal(r["value"], ref["value"])  # a selection, not arithmetic: bit-exact
    assert np.array_equal(res.win["label_run"], ref["label"])
    return r


@pytest.mark.parametrize("n,W", [(1, 10), (127, 50), (129, 1), (8193, 500), (300_000, 100_000), (1_000_000, 2_000_000), (700_001, 777)])
def test_extreme_vs_oracle(pgt, ctx, oracle, n, W):
    rng = np.random.default_rng(n + W)
    chr_ids, pos = synth.chromosomes(rng, n, min(n, 4), equal=False)
    score = np.round(rng.normal(0, 1.5, n), 6)
    score[rng.integers(0, n, size=max(1, n // 20))] = rng.choice([2.0, -2.0, 0.0, 3.25, -3.25])  # ties, cutoff-equal values
    runs = pgt.run_lengths(chr_ids)
    ends = np.cumsum(runs).astype(np.int64) - 1
    chr_len = (pos[ends].astype(np.int64) + rng.integers(0, 3 * W, size=runs.size)).astype(np.uint32)
    for mode, cutoff in ((0, 2.0), (1, 2.0), (2, -2.0), (1, 0.0)):
        for cl in (chr_len, None):
            _check_extreme(pgt, ctx, oracle, chr_ids, pos, score, W, mode, cutoff, cl)


def _parse_norm(text, tool):
    lines = text.splitlines()
    if tool == "xpehhWindow":
        lines = lines[1:]
    names, chr_ids, pos, score = [], [], [], []
    for ln in lines:
        t = ln.split()
        c = t[0].split("_")[0]
        if not names or names[-1] != c:
            names.append(c)
        chr_ids.append(len(names) - 1)
        pos.append(int(t[1]))
        score.append(float(t[2 + (4 if tool == "ihsWindow" else 6)]))
    return names, np.array(chr_ids, np.uint32), np.array(pos, np.uint32), np.array(score)


def test_extreme_golden_fixtures(pgt, ctx):
    """Rows against the TSV printed by the unmodified ihsWindow / xpehhWindow binaries."""
    n = 0
    for c in helpers.load_golden("ref_extreme.json")["cases"]:
        names, chr_ids, pos, score = _parse_norm(c["files"]["in.norm"], c["tool"])
        a = c["args"]
        W = int(a[a.index("-winsize") + 1])
        chr_len = None
        if "-chrlen" in a:
            lens = dict(ln.split() for ln in c["files"]["len.txt"].splitlines())
            chr_len = np.array([int(lens.get(nm, 0)) for nm in names], dtype=np.uint32)
        if c["tool"] == "ihsWindow":
            res = pgt.ihs_window(chr_ids, pos, score, W, float(a[a.index("-cutoff") + 1]), chr_len, ctx=ctx)
        else:
            res = pgt.xpehh_window(chr_ids, pos, score, float(a[1]), W, chr_len, ctx=ctx)
        out = []
        for w, r in zip(res.win, res.rows):
            head = f"{names[int(w['label_run'])]}\t{int(r['start'])}\t{int(r['end'])}\t"
            if r["nsites"]:
                out.append(head + f"{helpers.fmt_g(r['value'])}\t{int(r['position'])}\t{helpers.fmt_g(int(r['nbig']) / int(r['nsites']))}\t{int(r['nsites'])}")
            else:
                out.append(head + "NA\tNA\tNA\t0")
        assert "\n".join(out) + "\n" == c["stdout"], a
        n += 1
    assert n >= 100


def test_device_api_argument_checks(pgt, ctx):
    """Misaligned columns, undersized workspace and per-call site limits are refused before any launch."""
    import torch
    dev = torch.device("cuda:0")
    n = 10_000
    a = torch.rand(n + 1, dtype=torch.float64, device=dev)
    pos = torch.arange(n, dtype=torch.int32, device=dev)
    win = windows_to_device(pgt.build_windows_sites(np.array([n], dtype=np.uint64), 1000, 500), dev)
    with pytest.raises(_lib.PgtError):  # 8-byte (not 16-byte) aligned column
        ctx.fst_reduce_dev(pos, a[1:], a[:n], win)
    small = torch.empty(256, dtype=torch.uint8, device=dev)
    with pytest.raises(_lib.PgtError):  # workspace too small
        ctx.fst_reduce_dev(pos, a[:n], a[:n], win, tree=small)
    lib = _lib.load()
    g = torch.zeros(1024, dtype=torch.int8, device=dev)
    tree = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
    out = torch.empty(win.numel() // 32 * 32, dtype=torch.uint8, device=dev)
    rc = lib.pgt_het_reduce_dev(ctx._ctx, pos.data_ptr(), g.data_ptr(), 1 << 32, win.data_ptr(), 1, out.data_ptr(), out.numel(),
                                tree.data_ptr(), tree.numel(), None)
    assert rc == _lib.PGT_EARG and b"2^32" in lib.pgt_last_error(ctx._ctx)  # refused before touching memory
    rc = lib.pgt_extreme_reduce_dev(ctx._ctx, pos.data_ptr(), a.data_ptr(), n, 7, 2.0, win.data_ptr(), 1, out.data_ptr(),
                                    out.numel(), tree.data_ptr(), tree.numel(), None)
    assert rc == _lib.PGT_EARG  # unknown mode
    # ABI 4: the C entry points themselves refuse rows that do not fit `out` (in peer mode `out` is another GPU's
    # memory), before any launch — not only the Python wrapper
    n_win = win.numel() // 32
    b = torch.rand(n, dtype=torch.float64, device=dev)
    big_tree = torch.empty(ctx.tree_bytes(_lib.PGT_STAT_DXY, n) + 2 * ctx.tree_bytes(_lib.PGT_STAT_HET, n), dtype=torch.uint8, device=dev)
    canary = torch.full((n_win * 40 + 64,), 0xAB, dtype=torch.uint8, device=dev)
    i32 = torch.ones(n, dtype=torch.int32, device=dev)
    g8 = torch.zeros(n, dtype=torch.int8, device=dev)
    short = n_win * 24 - 1  # one byte short even of the smallest row type
    calls = [
        lambda: lib.pgt_fst_reduce_dev(ctx._ctx, pos.data_ptr(), b.data_ptr(), b.data_ptr(), n, win.data_ptr(), n_win, canary.data_ptr(),
                                       n_win * 40 - 1, big_tree.data_ptr(), big_tree.numel(), None),
        lambda: lib.pgt_het_reduce_dev(ctx._ctx, pos.data_ptr(), g8.data_ptr(), n, win.data_ptr(), n_win, canary.data_ptr(), n_win * 32 - 1,
                                       big_tree.data_ptr(), big_tree.numel(), None),
        lambda: lib.pgt_dxy_reduce_dev(ctx._ctx, pos.data_ptr(), b.data_ptr(), b.data_ptr(), i32.data_ptr(), i32.data_ptr(), n, 1,
                                       win.data_ptr(), n_win, canary.data_ptr(), short, None, big_tree.data_ptr(), big_tree.numel(), None),
        lambda: lib.pgt_dxy_het_reduce_dev(ctx._ctx, pos.data_ptr(), b.data_ptr(), b.data_ptr(), i32.data_ptr(), i32.data_ptr(),
                                           g8.data_ptr(), g8.data_ptr(), n, 1, win.data_ptr(), n_win, canary.data_ptr(), n_win * 24, None,
                                           canary.data_ptr(), canary.data_ptr(), n_win * 32 - 1, big_tree.data_ptr(), big_tree.numel(), None),
        lambda: lib.pgt_extreme_reduce_dev(ctx._ctx, pos.data_ptr(), b.data_ptr(), n, 0, 2.0, win.data_ptr(), n_win, canary.data_ptr(),
                                           n_win * 32 - 1, big_tree.data_ptr(), big_tree.numel(), None),
    ]
    import ctypes as C
    pa = (C.c_void_p * 2)(b.data_ptr(), b.data_ptr())
    two_trees = torch.empty(2 * ctx.tree_bytes(_lib.PGT_STAT_FST, n), dtype=torch.uint8, device=dev)
    calls.append(lambda: lib.pgt_fst_reduce_pairs_dev(ctx._ctx, pos.data_ptr(), pa, pa, 2, n, win.data_ptr(), n_win, canary.data_ptr(),
                                                      2 * n_win * 40 - 1, two_trees.data_ptr(), two_trees.numel(), None))
    for k, call in enumerate(calls):
        assert call() == _lib.PGT_EARG and b"do not fit" in lib.pgt_last_error(ctx._ctx), k
    torch.cuda.synchronize()
    assert bool((canary == 0xAB).all())  # nothing was launched
    hrows = np.zeros(n_win, dtype=_lib.FST_ROW_DTYPE)
    hwin = pgt.build_windows_sites(np.array([n], dtype=np.uint64), 1000, 500)
    assert lib.pgt_fst_reduce_cols(ctx._ctx, pos.data_ptr(), b.data_ptr(), b.data_ptr(), n, hwin.ctypes.data, hwin.size,
                                   hrows.ctypes.data, hrows.nbytes - 1) == _lib.PGT_EARG
    assert lib.pgt_fst_reduce_cols(ctx._ctx, pos.data_ptr(), b.data_ptr(), b.data_ptr(), n, hwin.ctypes.data, hwin.size,
                                   hrows.ctypes.data, hrows.nbytes) == _lib.PGT_OK
    tab = ctx.window_table_sites(np.array([n], dtype=np.uint64), 1000, 500)
    assert lib.pgt_fst_reduce_tab(ctx._ctx, pos.data_ptr(), b.data_ptr(), b.data_ptr(), n, 1, tab._h, hrows.ctypes.data,
                                  hrows.nbytes - 1) == _lib.PGT_EARG
    tab.free()


def test_device_memory_query(ctx):
    """pgt_dev_memory (the hosts decide with it whether a table is reduced in passes): free <= total, total is this GPU's
    HBM, and an allocation of 1 GiB through pgt_dev_alloc shows up in `free`."""
    import ctypes as C
    import torch
    lib = _lib.load()
    free, total = C.c_size_t(0), C.c_size_t(0)
    assert lib.pgt_dev_memory(ctx._ctx, C.byref(free), C.byref(total)) == _lib.PGT_OK
    assert 0 < free.value <= total.value and total.value == torch.cuda.mem_get_info()[1] and total.value > 100e9
    p = C.c_void_p()
    assert lib.pgt_dev_alloc(ctx._ctx, 1 << 30, C.byref(p)) == _lib.PGT_OK
    after = C.c_size_t(0)
    assert lib.pgt_dev_memory(ctx._ctx, C.byref(after), None) == _lib.PGT_OK
    assert free.value - after.value >= (1 << 30) - (64 << 20)
    assert lib.pgt_dev_free(ctx._ctx, p) == _lib.PGT_OK
    assert lib.pgt_dev_memory(None, C.byref(free), C.byref(total)) == _lib.PGT_EARG


def test_fst_beyond_2_32_sites(pgt, ctx):
    """Maximum sizes: 4.4e9 sites (88 GB of columns) — site indices above 2^32 in the table, the tree
    and the kernels' address arithmetic.  Windows at the far end against float64 sums by torch."""
    import torch
    dev = torch.device("cuda:0")
    free, _ = torch.cuda.mem_get_info()
    n = 4_400_000_000
    if free < 110e9:
        pytest.skip("needs ~110 GB of free HBM")
    gen = torch.Generator(device=dev).manual_seed(5)
    a = torch.empty(n, dtype=torch.float64, device=dev)
    b = torch.empty(n, dtype=torch.float64, device=dev)
    step = 400_000_000
    for o in range(0, n, step):
        m = min(step, n - o)
        b[o:o + m] = torch.rand(m, generator=gen, device=dev, dtype=torch.float64)
        a[o:o + m] = b[o:o + m] * 0.25
    pos = torch.ones(n, dtype=torch.int32, device=dev)  # coordinates are not the point here
    W, S = 50_000, 10_000
    win = pgt.build_windows_sites(np.array([n], dtype=np.uint64), W, S)
    assert win["hi"].max() == n and (win["lo"] > 2**32).sum() > 1000
    tail = win[-2000:]  # windows whose site range lies entirely above 2^32
    out, _ = ctx.fst_reduce_dev(pos, a, b, windows_to_device(tail, dev))
    torch.cuda.synchronize()
    rows = rows_from_device(out, FST_ROW_DTYPE)
    assert np.array_equal(rows["n"], (tail["hi"] - tail["lo"]).astype(np.uint32))
    for i in (0, 1, 777, 1998, 1999):
        lo, hi = int(tail["lo"][i]), int(tail["hi"][i])
        assert_close([rows["bsum"][i]], [float(b[lo:hi].sum())], "bsum above 2^32")
        assert_close([rows["asum"][i]], [float(a[lo:hi].sum())], "asum above 2^32")
    assert_close(rows["fst"], np.full(rows.size, 0.25), "fst above 2^32")
    del a, b, pos
    torch.cuda.empty_cache()


def test_sharded_scan_single_rank_rccl(pgt, ctx):
    """distributed.sharded_scan on the RCCL backend with the one rank a 1-GPU box offers: shard plan,
    shard-local columns on the device, reduce through the C-ABI, assembly — equal to the plain call."""
    import torch
    import torch.distributed as dist
    from popgenomicstools_amd.distributed import sharded_scan
    rng = np.random.default_rng(61)
    n = 400_000
    chr_ids, pos = synth.chromosomes(rng, n, 4, equal=False)
    a, b = synth.fst_columns(rng, n)
    win = pgt.build_windows_sites(pgt.run_lengths(chr_ids), 50_000, 10_000)
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29534", world_size=1, rank=0, device_id=dev)
    try:
        def load(lo, hi):
            return (torch.from_numpy(pos[lo:hi].view(np.int32)).to(dev), torch.from_numpy(a[lo:hi]).to(dev),
                    torch.from_numpy(b[lo:hi]).to(dev))
        rows = sharded_scan(win, FST_ROW_DTYPE, load, lambda c, w, out: ctx.fst_reduce_dev(*c, windows_to_device(w, dev), out=out), dev, ctx=ctx)
    finally:
        dist.destroy_process_group()
    assert rows.tobytes() == ctx.fst_reduce(pos, a, b, win).tobytes()


def test_pairs_sharded_equals_single_bitwise(pgt, ctx):
    """BASELINE config 5 in its multi-GPU form (site-range shards, every shard holds all pairs' columns):
    the batched-pairs entry point run shard by shard equals the single-GPU run bit for bit."""
    import torch
    from popgenomicstools_amd.distributed import shard_windows
    rng = np.random.default_rng(71)
    n, n_pairs = 600_000, 6
    chr_ids, pos = synth.chromosomes(rng, n, 5, equal=False)
    cols = [synth.fst_columns(rng, n) for _ in range(n_pairs)]
    win = pgt.build_windows_sites(pgt.run_lengths(chr_ids), 50_000, 10_000)
    dev = torch.device("cuda:0")

    def run(lo, hi, w):
        tp = torch.from_numpy(pos[lo:hi].view(np.int32)).to(dev)
        ta = [torch.from_numpy(c[0][lo:hi]).to(dev) for c in cols]
        tb = [torch.from_numpy(c[1][lo:hi]).to(dev) for c in cols]
        out, _ = ctx.fst_reduce_pairs_dev(tp, ta, tb, windows_to_device(w, dev))
        torch.cuda.synchronize()
        return rows_from_device(out, FST_ROW_DTYPE).reshape(n_pairs, w.size)

    single = run(0, n, win)
    for world in (2, 4, 8):
        parts = []
        for rank in range(world):
            s, local, _ = shard_windows(win, rank, world)
            parts.append(run(int(s["site_lo"]), int(s["site_hi"]), local))
        assert np.concatenate(parts, axis=1).tobytes() == single.tobytes()


# ---------------------------------------------------------------------------------------------
# round 2: the real multi-rank product path on the one-GPU box, and the config / branch holes
# ---------------------------------------------------------------------------------------------
@pytest.mark.timeout(900)
def test_multi_rank_hip_path_two_ranks_one_gpu():
    """Two processes (fresh children of a torch.distributed.run launcher that never touches the GPU),
    both on GPU 0, gloo for the collectives: shard plan -> per-rank site range -> real HIP kernels
    through the C-ABI -> rows to rank 0 by the gather AND by peer stores through hipIpc -> bytes equal
    to the single-GPU call.  fst, batched pairs (config 5), AF front end, extreme scan with windows
    >= 2^20 sites, dxy rows + genome-wide line (sharded_dxy_scan).  See tests/hip_rank_worker.py."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "tests", "hip_rank_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", PYTHONPATH=root, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29541", script],
                       capture_output=True, text=True, env=env, timeout=850)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    for tag in ("fst gather", "fst peer", "fst auto", "pairs gather", "pairs peer", "af peer", "extreme peer", "het peer", "dxy gather", "dxy peer"):
        assert "HIP_RANKS_OK " + tag in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def _genome(seed, n, n_chr):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from synth_genome import SynthGenome
    return SynthGenome(seed, n, n_chr)


def test_full_size_fused_dxy_het_properties(pgt, ctx):
    """BASELINE config 3 at its full size (10^8 sites, generated on the device): properties that need no oracle.
    (1) het counts are exact integers: equal to torch's counts on sampled windows and in total over a cover;
    (2) dxy: neff/nskip equal torch's counts; the window sums of a non-overlapping cover add up to the
        genome-wide line (pgt_dxy_total) within 1e-9 and its counts exactly;
    (3) p -> p/2 ... is not exact for dxy, but swapping the two populations is: d(p1,p2) = d(p2,p1) bit for bit;
    (4) coordinates follow pos and the table exactly; fused == separate entry points bit for bit."""
    import torch
    dev = torch.device("cuda:0")
    n, n_chr, W, S, minind = 100_000_000, 20, 50_000, 10_000, 5
    g = _genome(777, n, n_chr)
    pos = g.pos_t(0, n, dev)
    p1, p2, n1, n2 = g.dxy_columns_t(0, n, dev)
    g1, g2 = g.genotype_t(0, 0, n, dev), g.genotype_t(1, 0, n, dev)
    win = pgt.build_windows_sites(g.run_len, W, S)
    wt = windows_to_device(win, dev)
    dxy_out, tot, h1, h2, tree = ctx.dxy_het_reduce_dev(pos, p1, p2, n1, n2, g1, g2, minind, wt)
    torch.cuda.synchronize()
    d = rows_from_device(dxy_out, DXY_ROW_DTYPE)
    t = rows_from_device(tot, DXY_TOTAL_DTYPE)[0]
    r1, r2 = rows_from_device(h1, HET_ROW_DTYPE), rows_from_device(h2, HET_ROW_DTYPE)
    assert d.size == r1.size == r2.size == win.size
    hpos = pos.cpu().numpy().view(np.uint32)
    for r in (d, r1, r2):
        assert np.array_equal(r["start"], hpos[win["lo"]]) and np.array_equal(r["end"], hpos[win["hi"] - 1])
    assert np.array_equal(r1["mid"], ((r1["start"].astype(np.uint64) + r1["end"]) % 2**32 // 2).astype(np.uint32))
    ok = (n1 >= minind) & (n2 >= minind)
    site = p1 * (1.0 - p2) + p2 * (1.0 - p1)
    for i in np.linspace(0, win.size - 1, 25).astype(int):
        lo, hi = int(win["lo"][i]), int(win["hi"][i])
        assert int(d["neff"][i]) == int(ok[lo:hi].sum()) and int(d["nskip"][i]) == hi - lo - int(ok[lo:hi].sum())
        assert_close([d["sum"][i]], [float(site[lo:hi][ok[lo:hi]].sum())], "dxy window sum")
        for r, gg in ((r1, g1), (r2, g2)):
            assert int(r["nonmissing"][i]) == int((gg[lo:hi] >= 0).sum()) and int(r["nhet"][i]) == int((gg[lo:hi] == 1).sum())
            assert r["h"][i] == int(r["nhet"][i]) / int(r["nonmissing"][i])
    assert int(t["neff"]) == int(ok.sum()) and int(t["nskip"]) == n - int(ok.sum())
    assert_close([t["sum"]], [float(site[ok].sum())], "genome-wide dxy")
    # non-overlapping cover: window sums add up to the totals
    win_t = pgt.build_windows_sites(g.run_len, W, W)
    dc, tc, hc1, _, _ = ctx.dxy_het_reduce_dev(pos, p1, p2, n1, n2, g1, g2, minind, windows_to_device(win_t, dev), tree=tree)
    torch.cuda.synchronize()
    dc, hc1 = rows_from_device(dc, DXY_ROW_DTYPE), rows_from_device(hc1, HET_ROW_DTYPE)
    assert int(dc["neff"].astype(np.int64).sum()) == int(t["neff"]) and int(dc["nskip"].astype(np.int64).sum()) == int(t["nskip"])
    assert_close([dc["sum"].sum()], [t["sum"]], "cover dxy")
    assert int(hc1["nonmissing"].astype(np.int64).sum()) == int((g1 >= 0).sum())
    assert int(hc1["nhet"].astype(np.int64).sum()) == int((g1 == 1).sum())
    # populations swapped: the same bits
    ds, ts, _, _, _ = ctx.dxy_het_reduce_dev(pos, p2, p1, n2, n1, g1, g2, minind, wt, tree=tree)
    torch.cuda.synchronize()
    assert rows_from_device(ds, DXY_ROW_DTYPE).tobytes() == d.tobytes() and rows_from_device(ts, DXY_TOTAL_DTYPE).tobytes() == np.array([t]).tobytes()
    # fused == separate
    so, st, _ = ctx.dxy_reduce_dev(pos, p1, p2, n1, n2, minind, wt)
    ho, _ = ctx.het_reduce_dev(pos, g2, wt)
    torch.cuda.synchronize()
    assert rows_from_device(so, DXY_ROW_DTYPE).tobytes() == d.tobytes() and rows_from_device(ho, HET_ROW_DTYPE).tobytes() == r2.tobytes()


def test_full_size_28_pairs_properties(pgt, ctx):
    """BASELINE config 5 at its full size in the one-GPU form: 8 populations = 28 pairs x 10^8 sites, one
    table, one batched call.  pairs == singles bit for bit on 3 sampled pairs; exact x2 linearity on
    every pair; a non-overlapping cover adds up to each pair's genome total; coordinates."""
    import torch
    dev = torch.device("cuda:0")
    n, n_chr, W, S, n_pairs = 100_000_000, 20, 50_000, 10_000, 28
    g = _genome(888, n, n_chr)
    pos = g.pos_t(0, n, dev)
    al, bl = [], []
    for p in range(n_pairs):
        a_, b_ = g.pair_columns_t(p, 0, n, dev)
        al.append(a_)
        bl.append(b_)
    win = pgt.build_windows_sites(g.run_len, W, S)
    wt = windows_to_device(win, dev)
    out, tree = ctx.fst_reduce_pairs_dev(pos, al, bl, wt)
    torch.cuda.synchronize()
    rows = rows_from_device(out, FST_ROW_DTYPE).reshape(n_pairs, win.size)
    hpos = pos.cpu().numpy().view(np.uint32)
    for p in range(n_pairs):
        assert np.array_equal(rows[p]["start"], hpos[win["lo"]]) and np.array_equal(rows[p]["end"], hpos[win["hi"] - 1])
        assert np.array_equal(rows[p]["n"], (win["hi"] - win["lo"]).astype(np.uint32))
    for p in (0, 13, 27):
        single, _ = ctx.fst_reduce_dev(pos, al[p], bl[p], wt)
        torch.cuda.synchronize()
        assert rows_from_device(single, FST_ROW_DTYPE).tobytes() == rows[p].tobytes()
        for i in np.linspace(0, win.size - 1, 9).astype(int):
            lo, hi = int(win["lo"][i]), int(win["hi"][i])
            assert_close([rows[p]["asum"][i]], [float(al[p][lo:hi].sum())], "asum sample")
            assert_close([rows[p]["bsum"][i]], [float(bl[p][lo:hi].sum())], "bsum sample")
    for t in al + bl:
        t.mul_(2.0)
    out2, _ = ctx.fst_reduce_pairs_dev(pos, al, bl, wt, tree=tree)
    torch.cuda.synchronize()
    rows2 = rows_from_device(out2, FST_ROW_DTYPE).reshape(n_pairs, win.size)
    assert np.array_equal(rows2["asum"], 2.0 * rows["asum"]) and np.array_equal(rows2["bsum"], 2.0 * rows["bsum"])
    assert np.array_equal(rows2["fst"], rows["fst"])
    win_t = pgt.build_windows_sites(g.run_len, W, W)
    outc, _ = ctx.fst_reduce_pairs_dev(pos, al, bl, windows_to_device(win_t, dev), tree=tree)
    torch.cuda.synchronize()
    rc = rows_from_device(outc, FST_ROW_DTYPE).reshape(n_pairs, win_t.size)
    for p in range(n_pairs):
        assert int(rc[p]["n"].astype(np.int64).sum()) == n
        assert_close([rc[p]["bsum"].sum()], [float(bl[p].sum())], f"cover bsum pair {p}")
        assert_close([rc[p]["asum"].sum()], [float(al[p].sum())], f"cover asum pair {p}")


@pytest.mark.parametrize("n_pairs", [33, 70])
def test_more_pairs_than_one_launch_batch(pgt, ctx, n_pairs):
    """More than 32 pairs: launch_fst walks the pairs in batches of 32 (tree and row offsets of the later
    batches); every pair equals its single call bit for bit."""
    import torch
    rng = np.random.default_rng(100 + n_pairs)
    n = 70_001
    chr_ids, pos = synth.chromosomes(rng, n, 3, equal=False)
    cols = [synth.fst_columns(rng, n) for _ in range(n_pairs)]
    win = pgt.build_windows_sites(pgt.run_lengths(chr_ids), 9_000, 2_000)
    dev = torch.device("cuda:0")
    tp = torch.from_numpy(pos.view(np.int32)).to(dev)
    ta = [torch.from_numpy(c[0]).to(dev) for c in cols]
    tb = [torch.from_numpy(c[1]).to(dev) for c in cols]
    out, _ = ctx.fst_reduce_pairs_dev(tp, ta, tb, windows_to_device(win, dev))
    torch.cuda.synchronize()
    rows = rows_from_device(out, FST_ROW_DTYPE).reshape(n_pairs, win.size)
    for p in range(n_pairs):
        assert rows[p].tobytes() == ctx.fst_reduce(pos, cols[p][0], cols[p][1], win).tobytes(), p


def test_extreme_sharded_equals_single_with_long_windows(pgt, ctx):
    """Extreme-score scan shard by shard (pgt_plan_shards) with windows of >= 2^20 sites, so that the 2^14-
    and 2^20-site tree nodes are in play: concatenated shards == the single call, bit for bit."""
    import torch
    from popgenomicstools_amd._lib import EXT_ROW_DTYPE, PGT_EXT_IHS, PGT_EXT_XP_MIN
    from popgenomicstools_amd.distributed import shard_windows
    dev = torch.device("cuda:0")
    n = 6_000_000
    g = _genome(999, n, 1)
    pos, a, _ = g.fst_columns_t(0, n, dev)
    score = a * 40.0 - 2.0
    hpos = pos.cpu().numpy().view(np.uint32)
    win = pgt.build_windows_extreme(hpos, g.run_len, None, 35_000_000)
    assert int((win["hi"] - win["lo"]).max()) >= 1 << 20
    for mode, cut in ((PGT_EXT_IHS, 2.0), (PGT_EXT_XP_MIN, -1.5)):
        single, _ = ctx.extreme_reduce_dev(pos, score, mode, cut, windows_to_device(win, dev))
        torch.cuda.synchronize()
        single = rows_from_device(single, EXT_ROW_DTYPE)
        for world in (2, 3, 5):
            parts = []
            for rank in range(world):
                s, local, _ = shard_windows(win, rank, world)
                if local.size == 0:
                    continue
                lo, hi = int(s["site_lo"]), int(s["site_hi"])
                assert lo % (1 << 20) == 0
                o, _ = ctx.extreme_reduce_dev(pos[lo:hi], score[lo:hi], mode, cut, windows_to_device(local, dev))
                torch.cuda.synchronize()
                parts.append(rows_from_device(o, EXT_ROW_DTYPE))
            assert np.concatenate(parts).tobytes() == single.tobytes(), (mode, world)


def test_device_wrappers_reject_short_buffers(pgt, ctx):
    """A short column or an undersized out / tree buffer is an error in the binding, not a device overrun."""
    import torch
    dev = torch.device("cuda:0")
    n = 10_000
    pos = torch.arange(1, n + 1, dtype=torch.int32, device=dev)
    a = torch.rand(n, dtype=torch.float64, device=dev)
    win = windows_to_device(pgt.build_windows_sites(np.array([n], dtype=np.uint64), 1000, 500), dev)
    with pytest.raises(_lib.PgtError):
        ctx.fst_reduce_dev(pos[:-1], a, a, win)
    with pytest.raises(_lib.PgtError):
        ctx.fst_reduce_dev(pos, a, a, win, out=torch.empty(40, dtype=torch.uint8, device=dev))
    with pytest.raises(_lib.PgtError):
        ctx.fst_reduce_dev(pos, a, a, win, tree=torch.empty(256, dtype=torch.uint8, device=dev))
    # an empty shard (a rank that owns no window) is a no-op, not an error
    e64, e32 = torch.empty(0, dtype=torch.float64, device=dev), torch.empty(0, dtype=torch.int32, device=dev)
    out, _ = ctx.fst_reduce_dev(e32, e64, e64, torch.empty(0, dtype=torch.uint8, device=dev))
    assert out.numel() == 0


def test_entry_points_leave_the_callers_device_alone(pgt, ctx):
    """pgt_open and the reduce calls run on the ctx's device and restore the caller's current device
    (one visible GPU here: the check is that the current device is still 0 and torch keeps working)."""
    import torch
    before = torch.cuda.current_device()
    c2 = pgt.Context(0)
    c2.close()
    assert torch.cuda.current_device() == before
    assert float(torch.ones(4, device="cuda").sum()) == 4.0


# ---------------------------------------------------------------------------------------------
# sliding query (pgt_set_window_step <= 32)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("W,S", [(1, 1), (2, 1), (5, 2), (127, 1), (128, 1), (129, 3), (300, 1), (5_000, 1), (5_000, 7),
                                 (50_000, 1), (50_000, 32), (1_000, 31),
                                 (32, 1), (33, 1), (64, 3), (100, 1), (100, 32)])  # around the per-lane direct path (<= 32 sites, or inside one tile)
def test_sliding_query_vs_oracle(pgt, ctx, oracle, W, S):
    """S << W: the host API derives the step hint from the table and takes the sliding query; every
    statistic against the oracle (ints exact, floats 1e-9), on ragged chromosome layouts (short
    chromosomes exercise the per-window fallback inside a group)."""
    rng = np.random.default_rng(W * 131 + S)
    n = 60_000 if W >= 5_000 else 9_000
    for n_chr in (1, 4, 37):
        chr_ids, pos = synth.chromosomes(rng, n, n_chr, equal=False)
        a, b = synth.fst_columns(rng, n)
        check_fst(pgt, ctx, oracle, chr_ids, pos, a, b, W, S)
        g = synth.het_column(rng, n)
        check_het(pgt, ctx, oracle, chr_ids, pos, g, W, S)
        p1, p2, n1, n2 = synth.dxy_columns(rng, n)
        check_dxy(pgt, ctx, oracle, chr_ids, pos, p1, p2, n1, n2, W, S, 5, 1, 0)


@pytest.mark.parametrize("W,S,n", [(50_000, 33, 250_000), (50_000, 64, 250_000), (50_000, 100, 300_000), (50_000, 500, 400_000),
                                   (50_000, 2000, 600_000), (50_000, 2048, 600_000), (16_384, 100, 120_000), (16_383, 100, 120_000),
                                   (20_000, 777, 200_000), (140_000, 1000, 900_000)])
def test_group_query_vs_oracle(pgt, ctx, oracle, W, S, n):
    """32 < S <= 2048 (`-winsize 50000 -stepsize 100`: fstWindow.cpp:80-83,95-99 re-sums W sites per window there): the host
    API derives the step hint from the table and takes the GROUP query — 64 consecutive windows per wave, level-1 node scans
    and the interior shared.  Every statistic against the oracle (ints exact, floats 1e-9) on ragged chromosome layouts:
    short chromosomes and W = 16383 exercise the per-window fallback inside a group, W = 140000 puts the int8 tree
    (65536-site level-2 tiles) on the group path too."""
    rng = np.random.default_rng(W * 7 + S)
    for n_chr in (1, 4, 23):
        chr_ids, pos = synth.chromosomes(rng, n, n_chr, equal=False)
        a, b = synth.fst_columns(rng, n)
        check_fst(pgt, ctx, oracle, chr_ids, pos, a, b, W, S)
        g = synth.het_column(rng, n)
        check_het(pgt, ctx, oracle, chr_ids, pos, g, W, S)
        p1, p2, n1, n2 = synth.dxy_columns(rng, n)
        check_dxy(pgt, ctx, oracle, chr_ids, pos, p1, p2, n1, n2, W, S, 5, 1, 0)


def test_small_step_reference_goldens(pgt, ctx):
    """The C-ABI rows against runs of the UNMODIFIED reference tools with S << W (tests/golden/ref_small_step.json: seeded
    tables of up to 4x10^5 sites, windows of 3000 .. 140000 sites, steps 1 .. 2500 — the group, sliding and per-window
    queries): the number of rows, and every k-th row of the reference's stdout (labels, coordinates, counts exact; the
    ratio within 1e-9 + the 6-digit print)."""
    n = 0
    for c, cols in helpers.small_step_cases():
        names = [f"chr{i + 1}" for i in range(int(cols["chr_ids"].max()) + 1)]
        if c["tool"] == "fstWindow":
            res = pgt.fst_window(cols["chr_ids"], cols["pos"], cols["a"], cols["b"], c["W"], c["S"], ctx=ctx)
            stat, cnt = "fst", "n"
        else:
            res = pgt.het_window(cols["chr_ids"], cols["pos"], cols["g"], c["W"], c["S"], ctx=ctx)
            stat, cnt = "h", "nonmissing"
        assert res.rows.size == c["n_rows"], (c["tool"], c["W"], c["S"])
        k = c["every"]
        helpers.assert_rows_match_tsv(names, res.win[::k], res.rows[::k], stat, cnt, helpers.parse_tsv("\n".join(c["rows"])))
        n += 1
    assert n == 16


def test_sliding_query_agrees_with_per_window_query_and_is_shard_independent(pgt, ctx):
    """Device API: with the step hint (sliding) and without (one wave per window) the integer columns
    are identical and the sums agree to 1e-9; het rows are bitwise equal (integer sums); and the sliding
    rows do not depend on how the table is cut into shards (bitwise)."""
    import torch
    from popgenomicstools_amd.distributed import shard_windows
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(404)
    n = 700_000
    chr_ids, pos = synth.chromosomes(rng, n, 6, equal=False)
    a, b = synth.fst_columns(rng, n)
    g = synth.het_column(rng, n).astype(np.int8)
    t = lambda x: torch.from_numpy(x).to(dev)
    tp, ta, tb, tg = t(pos.view(np.int32)), t(a), t(b), t(g)
    for W, S in ((50_000, 1), (10_000, 3), (777, 16), (50_000, 100), (30_000, 700), (140_000, 1500), (9_000, 50)):  # the last four: group query
        win = pgt.build_windows_sites(pgt.run_lengths(chr_ids), W, S)
        wt = windows_to_device(win, dev)
        ctx.set_max_window(W)
        ctx.set_window_step(0)
        std, _ = ctx.fst_reduce_dev(tp, ta, tb, wt)
        hstd, _ = ctx.het_reduce_dev(tp, tg, wt)
        ctx.set_window_step(S)
        sl, _ = ctx.fst_reduce_dev(tp, ta, tb, wt)
        hsl, _ = ctx.het_reduce_dev(tp, tg, wt)
        torch.cuda.synchronize()
        r0, r1 = rows_from_device(std, FST_ROW_DTYPE), rows_from_device(sl, FST_ROW_DTYPE)
        for f in ("start", "end", "mid", "n"):
            assert np.array_equal(r0[f], r1[f]), f
        assert_close(r1["asum"], r0["asum"], "asum")
        assert_close(r1["bsum"], r0["bsum"], "bsum")
        assert rows_from_device(hstd, HET_ROW_DTYPE).tobytes() == rows_from_device(hsl, HET_ROW_DTYPE).tobytes()
        for world in (2, 5):
            parts = []
            for rank in range(world):
                s_, local, _ = shard_windows(win, rank, world)
                lo, hi = int(s_["site_lo"]), int(s_["site_hi"])
                o, _ = ctx.fst_reduce_dev(tp[lo:hi], ta[lo:hi], tb[lo:hi], windows_to_device(local, dev))
                torch.cuda.synchronize()
                parts.append(rows_from_device(o, FST_ROW_DTYPE))
            assert np.concatenate(parts).tobytes() == r1.tobytes(), (W, S, world)
    ctx.set_window_step(0)
    ctx.set_max_window(0)


def test_sliding_query_with_batched_pairs_and_fused_statistics(pgt, ctx):
    """The sliding strategy behind the other entry points: batched pairs (grid.y) equal their single calls bit
    for bit at step 1; the fused dxy + het call equals the separate calls."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(909)
    n, n_pairs = 200_000, 3
    chr_ids, pos = synth.chromosomes(rng, n, 4, equal=False)
    cols = [synth.fst_columns(rng, n) for _ in range(n_pairs)]
    t = lambda x: torch.from_numpy(x).to(dev)
    tp = t(pos.view(np.int32))
    ta, tb = [t(c[0]) for c in cols], [t(c[1]) for c in cols]
    for W, S in ((3_000, 1), (20_000, 100), (140_000, 300)):  # sliding; group (dxy in groups, het per window); group for both trees
        win = pgt.build_windows_sites(pgt.run_lengths(chr_ids), W, S)
        wt = windows_to_device(win, dev)
        with ctx.hints(W, S):
            out, _ = ctx.fst_reduce_pairs_dev(tp, ta, tb, wt)
            torch.cuda.synchronize()
            rows = rows_from_device(out, FST_ROW_DTYPE).reshape(n_pairs, win.size)
            for p in range(n_pairs):
                single, _ = ctx.fst_reduce_dev(tp, ta[p], tb[p], wt)
                torch.cuda.synchronize()
                assert rows_from_device(single, FST_ROW_DTYPE).tobytes() == rows[p].tobytes()
            p1, p2, n1, n2 = synth.dxy_columns(rng, n)
            g1 = synth.het_column(rng, n).astype(np.int8)
            g2 = synth.het_column(rng, n).astype(np.int8)
            d, tot, h1, h2, _ = ctx.dxy_het_reduce_dev(tp, t(p1), t(p2), t(n1), t(n2), t(g1), t(g2), 5, wt)
            ds, tots, _ = ctx.dxy_reduce_dev(tp, t(p1), t(p2), t(n1), t(n2), 5, wt)
            hs, _ = ctx.het_reduce_dev(tp, t(g2), wt)
            torch.cuda.synchronize()
            assert rows_from_device(d, DXY_ROW_DTYPE).tobytes() == rows_from_device(ds, DXY_ROW_DTYPE).tobytes()
            assert rows_from_device(tot, DXY_TOTAL_DTYPE).tobytes() == rows_from_device(tots, DXY_TOTAL_DTYPE).tobytes()
            assert rows_from_device(h2, HET_ROW_DTYPE).tobytes() == rows_from_device(hs, HET_ROW_DTYPE).tobytes()


def test_sharded_dxy_scan_single_process(pgt, ctx):
    """distributed.sharded_dxy_scan without a process group (one rank): the rows of the plain call, the total's
    counts exactly and its sum to rounding (the two-rank run of the same function: tests/hip_rank_worker.py)."""
    import torch
    from popgenomicstools_amd._lib import DXY_ROW_DTYPE, DXY_TOTAL_DTYPE
    from popgenomicstools_amd.distributed import sharded_dxy_scan
    dev = torch.device("cuda:0")
    n = 700_001  # 10 whole 2^16-site blocks and a ragged one
    g = _genome(77, n, 3)
    win = pgt.build_windows_sites(g.run_len, 20_000, 5_000)

    def cols(lo, hi):
        return (g.pos_t(lo, hi, dev),) + tuple(g.dxy_columns_t(lo, hi, dev))
    out, tot, _ = ctx.dxy_reduce_dev(*cols(0, n), 5, windows_to_device(win, dev))
    rows, total = sharded_dxy_scan(win, n, cols, ctx, 5, dev)
    assert rows.tobytes() == rows_from_device(out, DXY_ROW_DTYPE).tobytes()
    t = rows_from_device(tot, DXY_TOTAL_DTYPE)[0]
    assert int(total["neff"]) == int(t["neff"]) and int(total["nskip"]) == int(t["nskip"])
    assert_close([float(total["sum"])], [float(t["sum"])], "genome-wide dxy")
    assert int(t["neff"]) + int(t["nskip"]) == n  # every site counted once (minind filter: skipped, not dropped)


def test_device_built_window_table_is_the_host_table(pgt, ctx):
    """pgt_wintab_sites: the table written by the kernel from the per-run plan is, byte for byte, the table of
    pgt_build_windows_sites — random chromosome runs, windows and steps (carry and tail rules, runs shorter than a
    window, step 1), and the run offsets give every row its label."""
    rng = np.random.default_rng(314)
    cases = [([1], 1, 1), ([5], 10, 3), ([10, 1, 1, 30], 4, 4), ([7, 7, 7], 7, 1), ([1000003], 50000, 1), ([300000] * 7, 1000, 999)]
    for _ in range(150):
        n_runs = int(rng.integers(1, 12))
        W = int(rng.integers(1, 60))
        cases.append(([int(x) for x in rng.integers(1, 200, n_runs)], W, int(rng.integers(1, W + 1))))
    total = 0
    for run_len, W, S in cases:
        run_len = np.array(run_len, dtype=np.uint64)
        host = pgt.build_windows_sites(run_len, W, S)
        tab = ctx.window_table_sites(run_len, W, S)
        assert tab.n_win == host.size, (run_len, W, S)
        assert tab.to_host().tobytes() == host.tobytes(), (run_len, W, S)
        assert np.array_equal(tab.labels(), host["label_run"])
        assert tab.first[0] == 0 and tab.first[-1] == host.size and np.all(np.diff(tab.first.astype(np.int64)) >= 0)
        total += host.size
        tab.free()
    assert total > 900_000


def test_reduce_over_a_device_table_equals_reduce_over_the_host_table(pgt, ctx):
    """*_reduce_tab (columns from the host, window table built on the device) against the host-buffer entry points:
    the same bytes, for a per-site table (W = S = 1), a sliding one and an ordinary one."""
    rng = np.random.default_rng(315)
    n = 400_000
    chr_ids, pos = synth.chromosomes(rng, n, 4, equal=False)
    a, b = synth.fst_columns(rng, n)
    g = synth.het_column(rng, n)
    p1, p2, n1, n2 = synth.dxy_columns(rng, n)
    run_len = pgt.run_lengths(chr_ids)
    for W, S in ((1, 1), (300, 1), (5_000, 1_000)):
        win = pgt.build_windows_sites(run_len, W, S)
        tab = ctx.window_table_sites(run_len, W, S)
        assert ctx.fst_reduce_tab(pos, a, b, tab).tobytes() == ctx.fst_reduce(pos, a, b, win).tobytes(), (W, S)
        assert ctx.het_reduce_tab(pos, g, tab).tobytes() == ctx.het_reduce(pos, g, win).tobytes(), (W, S)
        rows_t, tot_t = ctx.dxy_reduce_tab(pos, p1, p2, n1, n2, 5, tab)
        rows_h, tot_h = ctx.dxy_reduce(pos, p1, p2, n1, n2, 5, win)
        assert rows_t.tobytes() == rows_h.tobytes() and tot_t.tobytes() == tot_h.tobytes(), (W, S)
        tab.free()


# ---------------------------------------------------------------------------------------------
# round 6: the advisor's edge cases
# ---------------------------------------------------------------------------------------------
def test_misaligned_column_views_are_refused_by_name(pgt, ctx):
    """ABI 5 reads the i32 count columns by 16-byte loads too: a view that starts 2 sites into an allocation (8-byte
    aligned, fine until ABI 4) is refused by the WRAPPER, naming the tensor and the site offsets that work; the same views
    at 4 sites are accepted and give the rows of freshly allocated copies, bit for bit.  f64 and i8 columns likewise."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(61)
    n = 300_016
    chr_ids, pos = synth.chromosomes(rng, n, 2)
    p1, p2, n1, n2 = synth.dxy_columns(rng, n)
    g = synth.het_column(rng, n).astype(np.int8)
    t = lambda x: torch.from_numpy(x).to(dev)
    tp, t1, t2, tn1, tn2, tg = t(pos.view(np.int32)), t(p1), t(p2), t(n1), t(n2), t(g)

    def table(k):  # the window table of the sites from k on
        return pgt.build_windows_sites(pgt.run_lengths(chr_ids[k:]), 20_000, 5_000)

    for k, name in ((2, "n1"), (2, "n2")):
        cols = {"n1": tn1[4:], "n2": tn2[4:]}
        cols[name] = (tn1 if name == "n1" else tn2)[k:k + n - 4]  # 8 bytes past a 16-byte boundary
        with pytest.raises(_lib.PgtError) as e:
            ctx.dxy_reduce_dev(tp[4:], t1[4:], t2[4:], cols["n1"], cols["n2"], 5, windows_to_device(table(4), dev))
        assert name in str(e.value) and "multiples of 4" in str(e.value)
    with pytest.raises(_lib.PgtError) as e:
        ctx.fst_reduce_dev(tp[1:], t1[1:], t2[1:], windows_to_device(table(1), dev))
    assert "a:" in str(e.value) and "multiples of 2" in str(e.value)
    with pytest.raises(_lib.PgtError) as e:
        ctx.het_reduce_dev(tp[8:], tg[8:], windows_to_device(table(8), dev))
    assert "g:" in str(e.value) and "multiples of 16" in str(e.value)
    # aligned views: 4 sites in for i32 / f64, 16 for i8 — the rows of fresh copies
    win4, win16 = table(4), table(16)
    out, tot, _ = ctx.dxy_reduce_dev(tp[4:], t1[4:], t2[4:], tn1[4:], tn2[4:], 5, windows_to_device(win4, dev))
    ref, rtot, _ = ctx.dxy_reduce_dev(tp[4:].clone(), t1[4:].clone(), t2[4:].clone(), tn1[4:].clone(), tn2[4:].clone(), 5,
                                      windows_to_device(win4, dev))
    torch.cuda.synchronize()
    assert torch.equal(out, ref) and torch.equal(tot, rtot)
    hout, _ = ctx.het_reduce_dev(tp[16:], tg[16:], windows_to_device(win16, dev))
    href, _ = ctx.het_reduce_dev(tp[16:].clone(), tg[16:].clone(), windows_to_device(win16, dev))
    torch.cuda.synchronize()
    assert torch.equal(hout, href)


def test_genome_wide_dxy_line_is_pinned_to_the_exact_sum(pgt, ctx):
    """The genome-wide dxy line is the sum of the build waves' partial sums in wave order (round 5): its low bits follow the
    build grid, not only the data.  Pinned here so that a retune of the launch geometry shows as a test to look at, not as a
    silently different TSV: (1) against the EXACT sum of the per-site values (math.fsum of the host formula, which the device
    reproduces bit for bit per site) to 1e-13 relative — a wave-order sum of <= 2048 partials of pairwise sums stays well inside;
    (2) against the block-ordered total of the sharded scan (2^16-site blocks added in order: the same bits for every rank
    count) to 1e-13; (3) the counts exactly; (4) two calls, same bits (the static grid is a function of the size alone)."""
    import math
    import torch
    from popgenomicstools_amd.distributed import sharded_dxy_scan
    dev = torch.device("cuda:0")
    n = 5_000_003
    g = _genome(99, n, 4)
    win = pgt.build_windows_sites(g.run_len, 50_000, 10_000)

    def cols(lo, hi):
        return (g.pos_t(lo, hi, dev),) + tuple(g.dxy_columns_t(lo, hi, dev))
    c = cols(0, n)
    minind = 5
    _, tot, _ = ctx.dxy_reduce_dev(*c, minind, windows_to_device(win, dev))
    _, tot2, _ = ctx.dxy_reduce_dev(*c, minind, windows_to_device(win, dev))
    torch.cuda.synchronize()
    t = rows_from_device(tot, DXY_TOTAL_DTYPE)[0]
    assert rows_from_device(tot2, DXY_TOTAL_DTYPE).tobytes() == rows_from_device(tot, DXY_TOTAL_DTYPE).tobytes()
    p1, p2, n1, n2 = (x.cpu().numpy() for x in c[1:])
    keep = (n1 >= minind) & (n2 >= minind)
    d = p1 * (1.0 - p2) + p2 * (1.0 - p1)  # dxyWindow.cpp:381, numpy = the host's IEEE arithmetic, no contraction
    exact = math.fsum(d[keep].tolist())
    assert int(t["neff"]) == int(keep.sum()) and int(t["nskip"]) == n - int(keep.sum())
    assert abs(float(t["sum"]) - exact) <= 1e-13 * abs(exact), (float(t["sum"]), exact)
    _, total = sharded_dxy_scan(win, n, cols, ctx, minind, dev)
    assert int(total["neff"]) == int(t["neff"]) and int(total["nskip"]) == int(t["nskip"])
    assert abs(float(total["sum"]) - float(t["sum"])) <= 1e-13 * abs(exact)
    assert abs(float(total["sum"]) - exact) <= 1e-13 * abs(exact)


@pytest.mark.parametrize("n", [8192 * 64, 8192 * 64 + 15, 1024 * 64, 1024 * 64 + 1, 128 * 64, 300_017])
def test_fast_query_paths_equal_the_general_path_at_their_edges(pgt, ctx, n):
    """The two-level fast paths of the per-window query (range_partial_two_levels, HetTraits::sum_two_ranges, the het tree
    built to one level) against the general walk (no hint: every level built, range_partial): windows whose ends sit on and
    around multiples of 16 / 128 / 1024 / 8192 and within the last 15 sites of the column (the ragged-end fallback), sizes
    with exactly 64 level-1 or level-2 nodes, empty left / right sides, and a hint that a longer window violates — het, dxy
    and fst rows byte for byte for max_window in {50000, 65535, 65536} against max_window = 0."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(n)
    pos = np.arange(1, n + 1, dtype=np.uint32)
    a, b = synth.fst_columns(rng, n)
    p1, p2, n1, n2 = synth.dxy_columns(rng, n)
    g = synth.het_column(rng, n).astype(np.int8)
    edges = set()
    for m in (16, 128, 1024, 8192):
        for k in (1, 2, 3, 7, n // m - 1, n // m):
            for d in (-1, 0, 1):
                edges.add(k * m + d)
    edges |= set(range(n - 16, n + 1)) | {0, 1, 15, 17}
    edges = sorted(e for e in edges if 0 <= e <= n)
    spans = []
    for lo in edges:
        for length in (1, 15, 16, 17, 127, 129, 1023, 1025, 8191, 8193, 49_999, 50_000, 65_535, 65_536, 70_000):
            if lo + length <= n:
                spans.append((lo, lo + length))
    for hi in edges:  # windows that END at an edge
        for length in (1, 16, 1000, 50_000):
            if hi - length >= 0:
                spans.append((hi - length, hi))
    spans = sorted(set(spans))
    win = np.zeros(len(spans), dtype=WIN_DTYPE)
    win["lo"] = [s[0] for s in spans]
    win["hi"] = [s[1] for s in spans]
    t = lambda x: torch.from_numpy(x).to(dev)
    tp, ta, tb, t1, t2, tn1, tn2, tg = t(pos.view(np.int32)), t(a), t(b), t(p1), t(p2), t(n1), t(n2), t(g)
    wd = windows_to_device(win, dev)

    def rows():
        f, _ = ctx.fst_reduce_dev(tp, ta, tb, wd)
        h, _ = ctx.het_reduce_dev(tp, tg, wd)
        d, tot, _ = ctx.dxy_reduce_dev(tp, t1, t2, tn1, tn2, 3, wd)
        fused = ctx.dxy_het_reduce_dev(tp, t1, t2, tn1, tn2, tg, tg, 3, wd)
        torch.cuda.synchronize()
        return [x.cpu().numpy().tobytes() for x in (f, h, d, fused[0], fused[2], fused[3])], rows_from_device(tot, DXY_TOTAL_DTYPE)[0]

    try:
        ctx.set_max_window(0)
        base, btot = rows()
        assert base[2] == base[3] and base[1] == base[4] == base[5]  # fused == separate
        # integer truth for the het rows of the general path itself (numpy prefix sums)
        nm = np.concatenate(([0], np.cumsum(g >= 0)))
        nh = np.concatenate(([0], np.cumsum(g == 1)))
        hr = np.frombuffer(base[1], dtype=HET_ROW_DTYPE)
        assert np.array_equal(hr["nonmissing"], (nm[win["hi"].astype(np.int64)] - nm[win["lo"].astype(np.int64)]).astype(np.uint32))
        assert np.array_equal(hr["nhet"], (nh[win["hi"].astype(np.int64)] - nh[win["lo"].astype(np.int64)]).astype(np.uint32))
        for mw in (50_000, 65_535, 65_536):
            ctx.set_max_window(mw)
            got, gtot = rows()
            inside = (win["hi"] - win["lo"]) <= mw  # windows the hint is true for: the same bytes; longer ones: a too-small hint
            # only changes the order of a float sum (test_max_window_hint_only_changes_speed) — integers exact, floats to 1e-9
            for k, (name, dt) in enumerate((("fst", FST_ROW_DTYPE), ("het", HET_ROW_DTYPE), ("dxy", DXY_ROW_DTYPE), ("fused dxy", DXY_ROW_DTYPE),
                                            ("fused het 1", HET_ROW_DTYPE), ("fused het 2", HET_ROW_DTYPE))):
                gr, br = np.frombuffer(got[k], dtype=dt), np.frombuffer(base[k], dtype=dt)
                assert gr[inside].tobytes() == br[inside].tobytes(), (name, mw, n)
                for f in dt.names:
                    if dt[f].kind == "f":
                        assert_close(gr[f][~inside], br[f][~inside], f"{name}.{f} beyond the hint {mw}")
                    else:
                        assert np.array_equal(gr[f][~inside], br[f][~inside]), (name, f, mw, n)
            assert int(gtot["neff"]) == int(btot["neff"]) and int(gtot["nskip"]) == int(btot["nskip"])
    finally:
        ctx.set_max_window(0)


def test_host_buffer_calls_through_the_staging_ring_equal_the_device_calls(pgt, ctx, monkeypatch):
    """The host-buffer entry points INTEGRATION.md binds (pgt_fst_reduce / pgt_het_reduce / pgt_dxy_reduce / pgt_extreme_reduce)
    with inputs large enough to take the pinned staging ring (>= 32 MiB of columns; several 16-MiB pieces per column, ragged last
    pieces): rows bit for bit those of the device-resident calls and of the plain-hipMemcpy path (PGT_UPLOAD=plain), with and
    without pgt_prepare_host_io, and again after calls of other sizes have grown and reused the context's workspace."""
    import torch
    from popgenomicstools_amd._lib import EXT_ROW_DTYPE, PGT_EXT_IHS
    import popgenomicstools_amd as pg
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(71)
    n = 8_000_037  # hetWindow: 40 MB of columns, the smallest of the four, still above the ring threshold of 32 MiB
    chr_ids, pos = synth.chromosomes(rng, n, 3, equal=False)
    a, b = synth.fst_columns(rng, n)
    p1, p2, n1, n2 = synth.dxy_columns(rng, n)
    g = synth.het_column(rng, n).astype(np.int8)
    win = pgt.build_windows_sites(pgt.run_lengths(chr_ids), 50_000, 10_000)
    t = lambda x: torch.from_numpy(x).to(dev)
    wd = windows_to_device(win, dev)
    tp = t(pos.view(np.int32))
    ref_f = rows_from_device(ctx.fst_reduce_dev(tp, t(a), t(b), wd)[0], FST_ROW_DTYPE).tobytes()
    ref_h = rows_from_device(ctx.het_reduce_dev(tp, t(g), wd)[0], HET_ROW_DTYPE).tobytes()
    d_out, d_tot, _ = ctx.dxy_reduce_dev(tp, t(p1), t(p2), t(n1), t(n2), 5, wd)
    ref_d, ref_t = rows_from_device(d_out, DXY_ROW_DTYPE).tobytes(), rows_from_device(d_tot, DXY_TOTAL_DTYPE).tobytes()
    ewin = pgt.build_windows_extreme(pos, pgt.run_lengths(chr_ids), None, 200_000)
    ref_e = rows_from_device(ctx.extreme_reduce_dev(tp, t(a), PGT_EXT_IHS, 0.9, windows_to_device(ewin, dev))[0], EXT_ROW_DTYPE).tobytes()
    torch.cuda.synchronize()

    def host_rows(c):
        d, tot = c.dxy_reduce(pos, p1, p2, n1, n2, 5, win)
        return (c.fst_reduce(pos, a, b, win).tobytes(), c.het_reduce(pos, g, win).tobytes(), d.tobytes(), np.array([tot]).tobytes(),
                c.extreme_reduce(pos, a, PGT_EXT_IHS, 0.9, ewin).tobytes())

    want = (ref_f, ref_h, ref_d, ref_t, ref_e)
    assert host_rows(ctx) == want                      # ring, allocated inside the first call
    small = pgt.build_windows_sites(np.array([1000], dtype=np.uint64), 100, 50)
    assert ctx.fst_reduce(pos[:1000], a[:1000], b[:1000], small).size == small.size   # a small call in between (plain path, workspace reused)
    assert host_rows(ctx) == want                      # again: every buffer of the workspace already large enough
    with pg.Context(0) as c2:                          # a fresh context, prepared up front
        c2.prepare_host_io(1 << 20)                    # a small input announced: the set-up only, no ring yet
        c2.prepare_host_io()                           # size unknown: the ring; idempotent from here on
        c2.prepare_host_io(1 << 30)
        assert host_rows(c2) == want
    monkeypatch.setenv("PGT_UPLOAD", "plain")          # round 5's path: hipMemcpy from the caller's pageable columns
    assert host_rows(ctx) == want
    monkeypatch.setenv("PGT_UPLOAD", "ring")
    monkeypatch.setenv("PGT_UPLOAD_WORKERS", "5")      # read when a ring is created: a new context with an odd geometry
    monkeypatch.setenv("PGT_UPLOAD_CHUNK_MIB", "3")
    with pg.Context(0) as c3:
        assert host_rows(c3) == want
