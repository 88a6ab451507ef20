"""CPU: the product's closed-form window tables (pgt_build_windows_sites / _bp, O(#windows))
against the oracle's site-by-site / slot-by-slot streaming machine."""
import numpy as np
import pytest

import helpers
import synth
from popgenomicstools_amd import _lib


def _random_runs(rng, max_runs=6, max_len=40):
    lens = rng.integers(1, max_len + 1, size=rng.integers(1, max_runs + 1))
    chr_ids = np.repeat(np.arange(lens.size, dtype=np.uint32), lens)
    return lens.astype(np.uint64), chr_ids


def test_site_windows_match_streaming_machine(oracle, pgt):
    rng = np.random.default_rng(1)
    n_windows = 0
    for trial in range(1500):
        W = int(rng.integers(1, 16))
        S = int(rng.integers(1, W + 1))
        lens, chr_ids = _random_runs(rng)
        n = chr_ids.size
        pos = np.arange(1, n + 1, dtype=np.uint32)
        rows = oracle.fst_scan(chr_ids, pos, np.ones(n), np.ones(n), W, S)
        win = pgt.build_windows_sites(lens, W, S)
        assert win.size == rows.size, (W, S, lens)
        assert np.array_equal(win["lo"], rows["lo"]) and np.array_equal(win["hi"], rows["hi"])
        assert np.array_equal(win["label_run"], rows["label"])
        n_windows += win.size
    assert n_windows > 10000


def test_site_windows_large_closed_form(oracle, pgt):
    """Headline geometry (W=50000, S=10000) on ragged chromosomes, including runs that end exactly
    on a full buffer (Q1 carry) and runs shorter than W-S (Q2)."""
    W, S = 50_000, 10_000
    lens = np.array([50_000, 123_457, 39_999, 40_001, 50_000 + 3 * S, 7, 260_000], dtype=np.uint64)
    chr_ids = np.repeat(np.arange(lens.size, dtype=np.uint32), lens.astype(np.int64))
    n = chr_ids.size
    pos = np.arange(n, dtype=np.uint32)
    rows = oracle.fst_scan(chr_ids, pos, np.ones(n), np.ones(n), W, S)
    win = pgt.build_windows_sites(lens, W, S)
    assert np.array_equal(win["lo"], rows["lo"]) and np.array_equal(win["hi"], rows["hi"])
    assert np.array_equal(win["label_run"], rows["label"])


def test_bp_windows_match_slot_machine(oracle, pgt):
    rng = np.random.default_rng(2)
    n_windows = 0
    for trial in range(1200):
        W = int(rng.integers(1, 16))
        S = int(rng.integers(1, W + 1))
        n_runs = int(rng.integers(1, 5))
        pos_l, chr_l, len_l, run_l = [], [], [], []
        for r in range(n_runs):
            L = int(rng.integers(1, 45))
            k = int(rng.integers(1, min(L, 12) + 1))
            p = np.sort(rng.choice(np.arange(1, L + 1), size=k, replace=False))
            if rng.random() < 0.15:  # data beyond the declared chromosome length (dxyWindow.cpp:365 pads first)
                L = int(p[-1]) - int(rng.integers(0, 3))
                L = max(L, 1)
            pos_l.append(p)
            chr_l.append(np.full(k, r))
            len_l.append(L)
            run_l.append(k)
        pos = np.concatenate(pos_l).astype(np.uint32)
        chr_ids = np.concatenate(chr_l).astype(np.uint32)
        n = pos.size
        p1, p2 = rng.uniform(0, 1, n).round(3), rng.uniform(0, 1, n).round(3)
        n1, n2 = rng.integers(0, 5, n).astype(np.int32), rng.integers(0, 5, n).astype(np.int32)
        rows, tot = oracle.dxy_scan(chr_ids, pos, p1, p2, n1, n2, W, S, 2, 0, 0, np.array(len_l, dtype=np.uint32))
        win = pgt.build_windows_bp(pos, np.array(run_l, dtype=np.uint64), np.array(len_l, dtype=np.uint32), W, S)
        assert win.size == rows.size, (W, S, len_l, pos_l)
        assert np.array_equal(win["start"], rows["start"]) and np.array_equal(win["end"], rows["end"])
        assert np.array_equal(win["label_run"], rows["label"])
        assert np.all(win["flags"] == _lib.PGT_WIN_COORDS)
        nonempty = rows["hi"] > rows["lo"]
        assert np.array_equal(win["lo"][nonempty], rows["lo"][nonempty])
        assert np.array_equal(win["hi"][nonempty], rows["hi"][nonempty])
        assert np.all(win["lo"][~nonempty] == win["hi"][~nonempty])
        n_windows += win.size
    assert n_windows > 5000


def test_builders_reject_out_of_domain(pgt):
    rl = np.array([5, 4], dtype=np.uint64)
    for W, S in [(0, 1), (3, 0), (3, 4)]:  # reference: exit 255 / segfault
        with pytest.raises(_lib.PgtError):
            pgt.build_windows_sites(rl, W, S)
    with pytest.raises(_lib.PgtError):
        pgt.build_windows_sites(np.array([3, 0], dtype=np.uint64), 2, 1)
    with pytest.raises(_lib.PgtError):  # unsorted positions inside a chromosome
        pgt.build_windows_bp(np.array([1, 5, 3], dtype=np.uint32), np.array([3], dtype=np.uint64),
                             np.array([10], dtype=np.uint32), 4, 2)
    with pytest.raises(_lib.PgtError):  # 0-based position
        pgt.build_windows_bp(np.array([0, 5], dtype=np.uint32), np.array([2], dtype=np.uint64),
                             np.array([10], dtype=np.uint32), 4, 2)


def test_empty_and_tiny_inputs(pgt):
    assert pgt.build_windows_sites(np.zeros(0, dtype=np.uint64), 5, 2).size == 0
    assert pgt.build_windows_sites(np.array([3], dtype=np.uint64), 5, 2).size == 0  # N <= W-S
    w = pgt.build_windows_sites(np.array([4], dtype=np.uint64), 5, 2)
    assert w.size == 1 and (w["lo"][0], w["hi"][0]) == (0, 4)


def test_golden_window_counts(pgt):
    """Window tables reproduce the row count and labels of every reference-made fixture."""
    for c in helpers.load_golden("ref_random.json")["cases"] + helpers.load_golden("ref_kat.json")["cases"]:
        kind = "fst" if c["tool"] == "fstWindow" else "het"
        parsed = helpers.parse_table(c["input"], kind)
        names, chr_ids, pos = parsed[0], parsed[1], parsed[2]
        win = pgt.build_windows_sites(pgt.run_lengths(chr_ids), c["W"], c["S"])
        tsv = helpers.parse_tsv(c["stdout"])
        assert win.size == len(tsv)
        for w, t in zip(win, tsv):
            assert names[int(w["label_run"])] == t[0]
            assert str(int(pos[int(w["lo"])])) == t[1] and str(int(pos[int(w["hi"]) - 1])) == t[2]


def test_extreme_windows_match_tool_loop(oracle, pgt):
    """pgt_build_windows_extreme against the oracle's run of the ihsWindow/xpehhWindow loop."""
    rng = np.random.default_rng(3)
    n_windows = 0
    for trial in range(1500):
        W = int(rng.choice([1, 2, 5, 10, 37, 100]))
        n_runs = int(rng.integers(1, 5))
        pos_l, chr_l, len_l = [], [], []
        for r in range(n_runs):
            k = int(rng.integers(1, 30))
            p = np.cumsum(rng.integers(0 if rng.random() < 0.2 else 1, max(2, W), size=k)) + 1  # duplicates allowed
            pos_l.append(p)
            chr_l.append(np.full(k, r))
            len_l.append(0 if rng.random() < 0.4 else int(p[-1]) + int(rng.integers(0, 3 * W)))
        pos = np.concatenate(pos_l).astype(np.uint32)
        chr_ids = np.concatenate(chr_l).astype(np.uint32)
        chr_len = np.array(len_l, dtype=np.uint32)
        use_len = chr_len if rng.random() < 0.8 else None
        rows = oracle.extreme_scan(chr_ids, pos, np.zeros(pos.size), W, 0, 2.0, use_len)
        win = pgt.build_windows_extreme(pos, pgt.run_lengths(chr_ids), use_len, W)
        assert win.size == rows.size, (W, pos_l, len_l)
        assert np.array_equal(win["start"], rows["start"]) and np.array_equal(win["end"], rows["end"])
        assert np.array_equal(win["label_run"], rows["label"])
        assert np.array_equal(win["hi"] - win["lo"], rows["nsites"])
        ne = rows["nsites"] > 0
        assert np.array_equal(win["lo"][ne], rows["lo"][ne]) and np.array_equal(win["hi"][ne], rows["hi"][ne])
        n_windows += win.size
    assert n_windows > 10000


def test_extreme_builder_domain(pgt):
    with pytest.raises(_lib.PgtError):  # position beyond a given chromosome length: the reference never terminates
        pgt.build_windows_extreme(np.array([5, 30], dtype=np.uint32), np.array([2], dtype=np.uint64),
                                  np.array([20], dtype=np.uint32), 10)
    with pytest.raises(_lib.PgtError):
        pgt.build_windows_extreme(np.zeros(0, dtype=np.uint32), np.zeros(0, dtype=np.uint64), None, 10)
    with pytest.raises(_lib.PgtError):
        pgt.build_windows_extreme(np.array([5], dtype=np.uint32), np.array([1], dtype=np.uint64), None, 0)


def test_large_tables_take_the_parallel_fill_and_stay_equal(pgt, oracle):
    """More than 2^20 windows (the -stepsize 1 regime): the table is filled by index in parallel chunks; it
    must still be the streaming machine's sequence, and counting (out = NULL) must agree with filling."""
    rng = np.random.default_rng(99)
    n = 2_400_000
    chr_ids, pos = synth.chromosomes(rng, n, 9, equal=False)
    a = np.ones(n)
    for W, S in ((50, 1), (64, 3)):
        win = pgt.build_windows_sites(pgt.run_lengths(chr_ids), W, S)
        ref = oracle.fst_scan(chr_ids, pos, a, a, W, S)
        assert win.size == ref.size and (W, S) != (50, 1) or win.size > (1 << 20)
        assert np.array_equal(win["lo"], ref["lo"]) and np.array_equal(win["hi"], ref["hi"])
        assert np.array_equal(win["label_run"], ref["label"])
    # bp mode, many windows: against the oracle's slot machine
    m = 300_000
    chr_ids, pos = synth.chromosomes(rng, m, 3, equal=False)
    p1, p2, n1, n2 = synth.dxy_columns(rng, m)
    rl = pgt.run_lengths(chr_ids)
    ends = np.cumsum(rl).astype(np.int64) - 1
    chr_len = (pos[ends] + 17).astype(np.uint32)
    win = pgt.build_windows_bp(pos, rl, chr_len, 40, 2)
    ref, _ = oracle.dxy_scan(chr_ids, pos, p1, p2, n1, n2, 40, 2, 1, 0, 0, run_chr_len=chr_len)
    assert win.size == ref.size > (1 << 20)
    ne = ref["hi"] > ref["lo"]  # an empty window's [lo, hi) is any empty range
    assert np.array_equal(win["hi"] > win["lo"], ne)
    assert np.array_equal(win["lo"][ne], ref["lo"][ne]) and np.array_equal(win["hi"][ne], ref["hi"][ne])
    assert np.array_equal(win["start"], ref["start"]) and np.array_equal(win["end"], ref["end"])
    assert np.array_equal(win["label_run"], ref["label"])
