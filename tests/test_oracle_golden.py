"""CPU: the oracle (our restatement) against fixtures produced by the UNMODIFIED reference tools
(tests/golden/make_golden.py) and against the SURVEY §4 dxy known answers."""
import os

import numpy as np
import pytest

import helpers
import oracle_bind


def _run_text(oracle, tool, text, W, S, tmp_path):
    src = tmp_path / "in.txt"
    dst = tmp_path / "out.txt"
    src.write_text(text)
    fn = oracle.fst_text if tool == "fstWindow" else oracle.het_text
    rc = fn(str(src), W, S, str(dst))
    return rc, dst.read_text()


@pytest.mark.parametrize("fixture", ["ref_kat.json", "ref_random.json"])
def test_text_front_end_byte_identical(oracle, tmp_path, fixture):
    cases = helpers.load_golden(fixture)["cases"]
    assert len(cases) >= 10
    for c in cases:
        rc, out = _run_text(oracle, c["tool"], c["input"], c["W"], c["S"], tmp_path)
        assert rc == 0 and c["rc"] == 0
        assert out == c["stdout"], (c.get("note"), c["W"], c["S"])


def test_scan_rows_against_reference_tsv(oracle):
    for c in helpers.load_golden("ref_random.json")["cases"]:
        kind = "fst" if c["tool"] == "fstWindow" else "het"
        if kind == "fst":
            names, chr_ids, pos, a, b = helpers.parse_table(c["input"], kind)
            rows = oracle.fst_scan(chr_ids, pos, a, b, c["W"], c["S"])
        else:
            names, chr_ids, pos, g = helpers.parse_table(c["input"], kind)
            rows = oracle.het_scan(chr_ids, pos, g, c["W"], c["S"])
        tsv = helpers.parse_tsv(c["stdout"])
        assert len(rows) == len(tsv)
        for r, t in zip(rows, tsv):
            assert [names[r["label"]], str(r["start"]), str(r["end"]), str(r["mid"])] == t[:4]
            assert helpers.fmt_g(r["value"]) == t[4] and str(r["n"]) == t[5]
            assert r["hi"] - r["lo"] >= r["n"] if kind == "het" else r["hi"] - r["lo"] == r["n"]


def test_config1_golden(oracle, tmp_path):
    """BASELINE config 1 (100k sites, CPU path): regenerate the seeded input, check its hash, and
    compare the oracle with the reference's stdout byte for byte."""
    import hashlib
    import synth
    g = helpers.load_golden("ref_config1.json")
    rng = np.random.default_rng(g["seed"])
    chr_ids, pos = synth.chromosomes(rng, g["n"], g["n_chr"])
    a, b = synth.fst_columns(rng, g["n"])
    path = tmp_path / "c1.txt"
    oracle.write_fst_text(str(path), chr_ids, pos, a, b)
    assert hashlib.sha256(path.read_bytes()).hexdigest() == g["input_sha256"]
    for run in g["runs"]:
        out = tmp_path / "o.txt"
        assert oracle.fst_text(str(path), run["W"], run["S"], str(out)) == 0
        assert out.read_text() == run["stdout"]
    assert len(g["runs"][0]["stdout"].splitlines()) == 19  # Q2: the 20th chromosome is dropped


def test_small_step_goldens(oracle, tmp_path):
    """S << W (`fstWindow file 50000 100`: the reference re-sums W sites per window, fstWindow.cpp:80-83,95-99), tables too
    large to commit: the input is regenerated from its seed (hash checked) and the oracle's stdout must have the SHA-256
    of the reference's, i.e. be byte-identical over all rows (16 runs, up to 4x10^5 sites, 40 000 rows)."""
    import hashlib
    n = 0
    for c, cols in helpers.small_step_cases():
        text = cols["fst"] if c["tool"] == "fstWindow" else cols["het"]
        assert hashlib.sha256(text.encode()).hexdigest() == c["input_sha256"]
        rc, out = _run_text(oracle, c["tool"], text, c["W"], c["S"], tmp_path)
        assert rc == 0 and c["rc"] == 0
        assert len(out.splitlines()) == c["n_rows"] and out.splitlines()[:: c["every"]] == c["rows"]
        assert hashlib.sha256(out.encode()).hexdigest() == c["stdout_sha256"], (c["tool"], c["W"], c["S"])
        n += 1
    assert n == 16


def _write_maf(path, header, rows):
    with open(path, "w") as f:
        f.write(header + "\n")
        for c, p, fr, n in rows:
            f.write(f"{c}\t{p}\tA\tC\tA\t{fr:.6f}\t{n}\n")


def test_dxy_known_answers(oracle, tmp_path):
    k = helpers.load_golden("dxy_kat.json")
    m1, m2, sz = tmp_path / "p1.mafs", tmp_path / "p2.mafs", tmp_path / "sizes.txt"
    _write_maf(m1, k["header"], k["pop1"])
    _write_maf(m2, k["header"], k["pop2"])
    sz.write_text("".join(f"{c}\t{n}\n" for c, n in k["sizes"]))
    for c in k["cases"]:
        o, e = tmp_path / "o.txt", tmp_path / "e.txt"
        rc = oracle.dxy_text(str(m1), str(m2), None if c["fixedsite"] else str(sz), c["winsize"], c["stepsize"],
                             k["minind"], c["fixedsite"], c["skip_missing"], str(o), str(e))
        assert rc == 0
        assert o.read_text() == c["stdout"]
        assert e.read_text() == c["stderr"]


def test_dxy_hand_walked_cases(oracle, tmp_path):
    """tests/golden/dxy_hand_walked.json: the bp-slot machine (Q1 carry, Q3 leak, ordinary reset, data beyond the size
    file's length) and the two-file sync (pop2 within pop1, pop1 within pop2, the two ways the reference mis-pairs or
    truncates) stepped through dxyWindow.cpp by hand, the walk beside each case.  The oracle restates those lines
    literally, so it must print the walked output byte for byte — INCLUDING where the reference's behaviour is a quirk
    the product does not follow.  An independent derivation, not a reference-made pin (the grade stays 'unpinned')."""
    k = helpers.load_golden("dxy_hand_walked.json")
    assert len(k["cases"]) >= 10 and all(len(c["walk"]) >= 3 for c in k["cases"])
    n = 0
    for c in k["cases"]:
        m1, m2, sz = helpers.write_hand_walked_case(c, k["header"], tmp_path)
        for r in c["runs"]:
            o, e = tmp_path / "o.txt", tmp_path / "e.txt"
            rc = oracle.dxy_text(m1, m2, None if r["fixedsite"] else sz, r["winsize"], r["stepsize"], c["minind"], r["fixedsite"],
                                 r["skip_missing"], str(o), str(e))
            assert rc == 0, c["name"]
            assert o.read_text() == r["stdout"], (c["name"], r, o.read_text())
            assert e.read_text() == r["stderr"], (c["name"], r, e.read_text())
            n += 1
    assert n >= 14


def test_dxy_reference_made_cases(oracle, tmp_path):
    """THE PIN of the dxy restatement, where it exists: stdout, the stderr genome-wide line and the exit code of the
    unmodified reference dxyWindow (both window modes, -minind, -skip_missing, nested site sets, gzip input), byte
    for byte.  Needs tests/golden/ref_dxy.json, which only an image with the real Boost.Iostreams can produce
    (oracle/Makefile + tests/golden/make_golden.py); this image has none, so here the test reports that as a skip."""
    cases = helpers.dxy_ref_cases(tmp_path, plain_text=True)
    if cases is None:
        pytest.skip("tests/golden/ref_dxy.json absent: dxyWindow.cpp is unbuildable here (no Boost) — dxy parity unpinned")
    assert len(cases) >= 100
    for c, argv, o in cases:
        out, err = tmp_path / "o.txt", tmp_path / "e.txt"
        rc = oracle.dxy_text(o["maf1"], o["maf2"], o["sizefile"], o["winsize"], o["stepsize"], o["minind"], o["fixedsite"],
                             o["skip_missing"], str(out), str(err))
        if c["rc"] != 0:
            assert rc != 0, c["args"]
            continue
        assert rc == 0, c["args"]
        assert out.read_text() == c["stdout"], c["args"]
        assert err.read_text() == c["stderr"], c["args"]


def test_dxy_two_file_sync_restatement(oracle, tmp_path):
    """The oracle restates the two-file synchronisation of dxyWindow.cpp:315-331 literally (UNPINNED like the rest of
    the dxy path: no reference-made fixture exists in this image).  What can be checked without the reference:
    (i) on nested site sets (pop2's sites a subset of pop1's — the inputs make_golden.dxy_case generates for the
    future pin) the result is that of the two files cut down to their shared sites; (ii) a case walked by hand through
    the reference lines: pop2 lists an extra site at the END of a chromosome that pop1 lacks -> at the chromosome
    change pop1 is at (cB,3), pop2 still at (cA,9): names differ, pop2's name is the current chromosome, so :325-330
    tries to advance pop2 while 9 < 3 — it does not — and the main loop breaks: cB is never processed."""
    import gzip
    import random
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_golden
    rng = random.Random(5)
    nested = 0
    for k in range(120):
        files, args = make_golden.dxy_case(rng)
        d = tmp_path / f"c{k}"
        d.mkdir()
        path = {}
        for name, raw in files.items():
            plain = name[:-3] if name.endswith(".gz") else name
            (d / plain).write_bytes(gzip.decompress(raw) if name.endswith(".gz") else raw)
            path["@" + name] = str(d / plain)
        argv = [path.get(a, a) for a in args]
        o = {"winsize": 0, "stepsize": 0, "minind": 1, "fixedsite": 0, "skip_missing": 0, "sizefile": None}
        for i in range(0, len(argv) - 2, 2):
            o[argv[i].lstrip("-")] = argv[i + 1] if argv[i] == "-sizefile" else int(argv[i + 1])
        l1, l2 = open(argv[-2]).read().splitlines(), open(argv[-1]).read().splitlines()
        k1, k2 = {tuple(x.split()[:2]) for x in l1[1:]}, {tuple(x.split()[:2]) for x in l2[1:]}
        nested += k1 != k2
        (d / "s1").write_text("\n".join([l1[0]] + [x for x in l1[1:] if tuple(x.split()[:2]) in k2]) + "\n")
        (d / "s2").write_text("\n".join([l2[0]] + [x for x in l2[1:] if tuple(x.split()[:2]) in k1]) + "\n")
        outs = []
        for m1, m2 in ((argv[-2], argv[-1]), (str(d / "s1"), str(d / "s2"))):
            out, err = d / "o", d / "e"
            assert oracle.dxy_text(m1, m2, o["sizefile"], o["winsize"], o["stepsize"], o["minind"], o["fixedsite"],
                                   o["skip_missing"], str(out), str(err)) == 0, args
            outs.append((out.read_text(), err.read_text()))
        assert outs[0] == outs[1], args
    assert nested >= 30
    hdr = "chromo\tposition\tmajor\tminor\tref\tknownEM\tnInd\n"
    (tmp_path / "a").write_text(hdr + "cA\t2\tA\tC\tA\t0.5\t4\ncA\t5\tA\tC\tA\t0.25\t4\ncB\t3\tA\tC\tA\t0.1\t4\n")
    (tmp_path / "b").write_text(hdr + "cA\t2\tA\tC\tA\t0.5\t4\ncA\t5\tA\tC\tA\t0.75\t4\ncA\t9\tA\tC\tA\t0.3\t4\ncB\t3\tA\tC\tA\t0.9\t4\n")
    out, err = tmp_path / "o", tmp_path / "e"
    assert oracle.dxy_text(str(tmp_path / "a"), str(tmp_path / "b"), None, 1, 1, 1, 1, 0, str(out), str(err)) == 0
    assert out.read_text() == "cA\t2\t2\t0.5\t1\t0\ncA\t5\t5\t0.625\t1\t0\n" and err.read_text() == "1.125\t2\t0\n"
    # the mirror image (the extra site is in pop1): pop1 is advanced until its POSITION matches (:318-323) -> all sites match
    assert oracle.dxy_text(str(tmp_path / "b"), str(tmp_path / "a"), None, 1, 1, 1, 1, 0, str(out), str(err)) == 0
    assert [ln.split("\t")[:2] for ln in out.read_text().splitlines()] == [["cA", "2"], ["cA", "5"], ["cB", "3"]]


def test_oracle_rejects_out_of_domain(oracle):
    z = np.zeros(3, dtype=np.uint32)
    d = np.zeros(3)
    with pytest.raises(RuntimeError):
        oracle.fst_scan(z, z, d, d, 2, 3)  # S > W (reference: segfault, Q9)
    with pytest.raises(RuntimeError):
        oracle.fst_scan(z, z, d, d, 0, 1)  # W == 0 (reference: exit 255)


@pytest.mark.skipif(oracle_bind.ref_binary("fstWindow") is None, reason="oracle/_ref not built here")
def test_live_reference_fuzz(oracle, tmp_path):
    """Where the compiled reference is present, fuzz the oracle against it live as well."""
    import random
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_golden
    rng = random.Random(99)
    for i in range(80):
        tool = rng.choice(["fstWindow", "hetWindow"])
        text, W, S = make_golden.random_case(rng, tool)
        src = tmp_path / "f.txt"
        src.write_text(text)
        ref = subprocess.run([oracle_bind.ref_binary(tool), str(src), str(W), str(S)], capture_output=True, timeout=10)
        rc, out = _run_text(oracle, tool, text, W, S, tmp_path)
        assert rc == 0 and out.encode() == ref.stdout


def test_extreme_tools_byte_identical(oracle, tmp_path):
    """ihsWindow / xpehhWindow restatement against the reference binaries' stdout."""
    cases = helpers.load_golden("ref_extreme.json")["cases"]
    assert len(cases) >= 100
    for c in cases:
        src, W, cutoff, chrlen = helpers.extreme_case_args(c, tmp_path)
        out = tmp_path / "o.txt"
        if c["tool"] == "ihsWindow":
            rc = oracle.ihs_text(src, W, cutoff, chrlen, str(out))
        else:
            rc = oracle.xpehh_text(src, cutoff, W, chrlen, str(out))
        assert rc == 0 and c["rc"] == 0
        assert out.read_text() == c["stdout"], c["args"]


def test_dxy_fixedsite_windows_equal_the_pinned_fst_machine(oracle):
    """dxyWindow -fixedsite 1 uses the same emission rules as fstWindow (dxyWindow.cpp:357-359,376-378,
    424-426 vs fstWindow.cpp:132-138,150-152).  The fst restatement is pinned to the real binary, so
    equality of the two oracles' window ranges pins the dxy restatement's fixed-site machine to it."""
    rng = np.random.default_rng(17)
    for _ in range(400):
        n_runs = int(rng.integers(1, 6))
        lens = rng.integers(1, 40, n_runs)
        chr_ids = np.repeat(np.arange(n_runs, dtype=np.uint32), lens)
        n = chr_ids.size
        pos = np.arange(1, n + 1, dtype=np.uint32)
        W = int(rng.integers(1, 14)); S = int(rng.integers(1, W + 1))
        f = oracle.fst_scan(chr_ids, pos, np.ones(n), np.ones(n), W, S)
        d, _ = oracle.dxy_scan(chr_ids, pos, np.full(n, 0.5), np.full(n, 0.5), np.full(n, 9, np.int32),
                               np.full(n, 9, np.int32), W, S, 1, 1, 0)
        assert d.size == f.size
        for k in ("label", "start", "end", "lo", "hi"):
            assert np.array_equal(d[k], f[k]), k
        assert np.array_equal(d["n"], f["n"])  # every site is effective here


def _slot_model(runs, W, S):
    """Pure-Python model of dxyWindow's bp mode, written from SURVEY.md §4 ("the same machine run
    over base-pair slots 1..chrlen, data site or placeholder, with the chromosome-change rule: emit only
    if n > W-S; reset only if an emit happened and n < W").  runs = [(L, {pos: value})]."""
    out, buf = [], []

    def emit(label):
        vals = [v for _, v in buf]
        out.append((label, buf[0][0], buf[-1][0], sum(v for v in vals if v is not None and v >= 0),
                    sum(1 for v in vals if v is not None and v >= 0), sum(1 for v in vals if v == -9)))

    for r, (L, data) in enumerate(runs):
        for slot in range(1, L + 1):
            if len(buf) == W:
                emit(r)
                del buf[:S]
            buf.append((slot, data.get(slot)))
        last = r == len(runs) - 1
        if not last and len(buf) > W - S:
            full = len(buf) == W
            emit(r)
            if full:
                del buf[:S]
            else:
                buf.clear()
        if last and W - S < len(buf) <= W:
            emit(r)
    return out


def test_dxy_bp_mode_equals_independent_slot_model(oracle):
    rng = np.random.default_rng(23)
    n_win = 0
    for _ in range(500):
        W = int(rng.integers(1, 15)); S = int(rng.integers(1, W + 1)); minind = 3
        runs, chr_l, pos_l, p1_l, p2_l, n1_l, n2_l, len_l = [], [], [], [], [], [], [], []
        for r in range(int(rng.integers(1, 5))):
            L = int(rng.integers(1, 40))
            k = int(rng.integers(1, min(L, 10) + 1))
            pp = np.sort(rng.choice(np.arange(1, L + 1), size=k, replace=False))
            f1, f2 = rng.uniform(0, 1, k).round(3), rng.uniform(0, 1, k).round(3)
            m1, m2 = rng.integers(0, 7, k), rng.integers(0, 7, k)
            data = {}
            for q in range(k):
                ok = m1[q] >= minind and m2[q] >= minind
                data[int(pp[q])] = f1[q] * (1.0 - f2[q]) + f2[q] * (1.0 - f1[q]) if ok else -9
            runs.append((L, data))
            chr_l.append(np.full(k, r)); pos_l.append(pp); p1_l.append(f1); p2_l.append(f2); n1_l.append(m1); n2_l.append(m2)
            len_l.append(L)
        rows, tot = oracle.dxy_scan(np.concatenate(chr_l), np.concatenate(pos_l), np.concatenate(p1_l), np.concatenate(p2_l),
                                    np.concatenate(n1_l).astype(np.int32), np.concatenate(n2_l).astype(np.int32), W, S,
                                    minind, 0, 0, np.array(len_l, dtype=np.uint32))
        model = _slot_model(runs, W, S)
        assert len(model) == rows.size
        for m, r in zip(model, rows):
            assert (m[0], m[1], m[2], m[4], m[5]) == (r["label"], r["start"], r["end"], r["n"], r["nskip"])
            assert abs(m[3] - r["value"]) <= 1e-12
        n_win += rows.size
    assert n_win > 3000


def test_wcfst_restatement_against_exact_rational_evaluation(oracle):
    """orc_wcfst_site (the C restatement of betaAFOutlier.R:400-418 the AF front end is checked against) vs
    tests/golden/wcfst_exact.json: the same R lines evaluated in exact rational arithmetic (make_wcfst_exact.py).  A third
    derivation, not a reference-made pin (no R here): per site both components to a few ulps of the component's scale
    (`a` cancels towards 0 where the populations agree, so its error is measured against a+b's magnitude), and the
    genome-wide ratio (genomeFst, :440-446) to 1e-12."""
    from fractions import Fraction
    k = helpers.load_golden("wcfst_exact.json")
    assert len(k["cases"]) == 3
    for c in k["cases"]:
        f1 = np.array([float(Fraction(s["f1"])) for s in c["sites"]])
        f2 = np.array([float(Fraction(s["f2"])) for s in c["sites"]])
        a, ab = oracle.wcfst_columns(f1, f2, c["n1"], c["n2"])
        ea = np.array([s["a_f64"] for s in c["sites"]])
        eab = np.array([s["a_plus_b_f64"] for s in c["sites"]])
        for s in c["sites"]:  # the stored doubles are the correctly rounded fractions
            assert float(Fraction(s["a"])) == s["a_f64"] and float(Fraction(s["a_plus_b"])) == s["a_plus_b_f64"]
        scale = np.maximum(np.abs(eab), np.abs(ea)) + 1e-300
        assert np.all(np.abs(ab - eab) <= 1e-12 * scale + 1e-15), (c["n1"], c["n2"], ab - eab)
        assert np.all(np.abs(a - ea) <= 1e-12 * scale + 1e-15), (c["n1"], c["n2"], a - ea)
        assert abs(a.sum() / ab.sum() - c["genome_fst_f64"]) <= 1e-12
        assert float(Fraction(c["sum_a"]) / Fraction(c["sum_a_plus_b"])) == c["genome_fst_f64"]


def test_dxy_oracle_against_the_independent_stream_model(oracle, tmp_path):
    """The C oracle's dxyWindow restatement against tests/dxy_stream_model.py — the same reference lines
    (dxyWindow.cpp:172-209, 282-433) restated a second time, in Python, as the reference runs them (line readers that run
    dry, the catch-up loops, slot padding, calcWindow's static) — on 1 500 random file pairs: identical site lists, pop2
    within pop1, pop1 within pop2 and NON-nested lists (where the reference's behaviour is a quirk, but a defined one:
    both restatements must reproduce it), 1-4 chromosomes, both window modes, -winsize 0, -skip_missing, chromosomes
    missing from the size file.  stdout, stderr and the exit status byte for byte.  Pins nothing (the reference cannot be
    built here); it removes transcription slips that one restatement alone could hide."""
    import random

    import dxy_stream_model as model
    rng = random.Random(20261004)
    hdr = "chromo\tposition\tmajor\tminor\tref\tknownEM\tnInd"
    kinds = {"same": 0, "pop2_in_pop1": 0, "pop1_in_pop2": 0, "other": 0}
    n_rows = errors = 0
    for trial in range(1500):
        n_chr = rng.randint(1, 4)
        rows1, rows2, sizes = [], [], {}
        kind = rng.choice(list(kinds))
        for c in range(n_chr):
            L = rng.randint(1, 30)
            sites = sorted(rng.sample(range(1, L + 1), rng.randint(1, min(L, 8))))
            name = f"c{c}"
            if rng.random() < 0.95:
                sizes[name] = L + (rng.randint(0, 3) if rng.random() < 0.3 else 0) - (1 if rng.random() < 0.05 and L > 1 else 0)
            for p in sites:
                in1 = kind in ("same", "pop2_in_pop1") or rng.random() < 0.7
                in2 = kind in ("same", "pop1_in_pop2") or rng.random() < 0.7
                if in1:
                    rows1.append((name, p, round(rng.random(), 6), rng.randint(0, 6)))
                if in2:
                    rows2.append((name, p, round(rng.random(), 6), rng.randint(0, 6)))
        if not rows1 or not rows2:
            continue
        kinds[kind] += 1
        mode = rng.randint(0, 3)
        W = rng.randint(1, 9)
        S = rng.randint(1, W)
        if mode == 0:
            W, S, fixed = 0, 0, 1
        else:
            fixed = 1 if mode == 1 else 0
        minind, skip = rng.randint(1, 4), rng.randint(0, 1)
        f1, f2, fs = tmp_path / "p1.mafs", tmp_path / "p2.mafs", tmp_path / "sizes.txt"
        _write_maf(f1, hdr, rows1)
        _write_maf(f2, hdr, rows2)
        fs.write_text("".join(f"{c}\t{n}\n" for c, n in sizes.items()))
        o, e = tmp_path / "o.txt", tmp_path / "e.txt"
        rc = oracle.dxy_text(str(f1), str(f2), None if fixed else str(fs), W, S, minind, fixed, skip, str(o), str(e))
        mrc, mout, merr = model.maf2dxy(rows1, rows2, W, S, minind, fixed, sizes, skip)
        if mrc != 0:  # "Chromosomes in MAF files differ" / "Unable to determine size for X": the reference exits 255 (after some
            assert rc != 0, (trial, kind, mode)  # rows, in the second case); the oracle reports a status, no text
            errors += 1
            continue
        assert rc == 0, (trial, kind, mode, e.read_text())
        assert o.read_text() == mout, (trial, kind, mode, W, S, rows1, rows2, sizes, o.read_text(), mout)
        assert e.read_text() == merr, (trial, kind, mode, e.read_text(), merr)
        n_rows += mout.count("\n")
    assert min(kinds.values()) > 200 and n_rows > 6000 and 50 < errors < 400, (kinds, n_rows, errors)


def test_stream_model_reproduces_the_hand_walked_cases():
    """tests/dxy_stream_model.py on tests/golden/dxy_hand_walked.json: the paper walks and the Python restatement agree."""
    import dxy_stream_model as model
    k = helpers.load_golden("dxy_hand_walked.json")
    for c in k["cases"]:
        r1, r2 = [tuple(x) for x in c["pop1"]], [tuple(x) for x in c["pop2"]]
        for r in c["runs"]:
            assert model.maf2dxy(r1, r2, r["winsize"], r["stepsize"], c["minind"], r["fixedsite"], dict(c["sizes"] or []),
                                 r["skip_missing"]) == (0, r["stdout"], r["stderr"]), c["name"]
