"""The retained C++ hosts (popgenomicstools_amd/bin/{fstWindow,hetWindow,dxyWindow}): same argv,
TSV and exit codes as the reference tools.  Argument handling is checked on CPU (it happens before
any GPU use); TSV parity against the reference-made goldens needs the GPU."""
import gzip
import os
import subprocess

import numpy as np
import pytest

import helpers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "popgenomicstools_amd", "bin")


@pytest.fixture(scope="module")
def hosts():
    from popgenomicstools_amd import build
    build.build_lib()
    build.build_hosts()
    return {t: os.path.join(BIN, t) for t in ("fstWindow", "hetWindow", "dxyWindow")}


def run(cmd, **kw):
    return subprocess.run(cmd, capture_output=True, text=True, timeout=120, **kw)


def run_all(jobs, workers=3):
    """jobs: [(cmd, env or None), ...] -> their results in order, at most `workers` children at a time.  Most of a small CLI
    run is process and HIP start-up (0.2-0.4 s): the variants of one command (parsers, device lists, cuts) run side by
    side instead of one after the other — the GPU box allows six processes on the card, this keeps to three + pytest."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=workers) as ex:
        return list(ex.map(lambda j: run(j[0]) if j[1] is None else run(j[0], env=j[1]), jobs))


# ---- CPU: command-line behaviour (fstWindow.cpp:37-67,164-170; dxyWindow.cpp:63-139,537-538) ----
def test_usage_exits_zero(hosts):
    for tool in ("fstWindow", "hetWindow"):
        r = run([hosts[tool]])
        assert r.returncode == 0 and "default window size: 1" in r.stdout and "default step size: 1" in r.stdout
    r = run([hosts["dxyWindow"]])
    assert r.returncode == 0 and "-skip_missing" in r.stdout and "[0]" in r.stdout
    assert run([hosts["dxyWindow"], "onlyone.mafs"]).returncode == 0  # argc < 3 -> help, exit 0


def test_bad_arguments_exit_255(hosts, tmp_path):
    f = tmp_path / "in.txt"
    f.write_text("c1\t1\t0.1\t0.2\n")
    for tool in ("fstWindow", "hetWindow"):
        assert run([hosts[tool], str(tmp_path / "missing.txt")]).returncode == 255
        r = run([hosts[tool], str(f), "0"])
        assert r.returncode == 255 and "Window size must be a positive integer" in r.stderr
        r = run([hosts[tool], str(f), "abc", "1"])
        assert r.returncode == 255
        r = run([hosts[tool], str(f), "3", "0"])  # the reference warns, then crashes (Q9)
        assert r.returncode == 255 and "Step size must be a positive integer" in r.stderr
        assert run([hosts[tool], str(f), "3", "4"]).returncode == 255  # S > W: the reference segfaults
    d = hosts["dxyWindow"]
    r = run([d, "-bogus", "1", str(f), str(f)])
    assert r.returncode == 255 and "Unknown command: -bogus" in r.stderr
    r = run([d, "-minind", "0", "-fixedsite", "1", str(f), str(f)])
    assert r.returncode == 255 and "-minind must be at least 1" in r.stderr
    r = run([d, "-winsize", "5", "-fixedsite", "1", str(f), str(f)])
    assert r.returncode == 255 and "Must specify a -stepsize > 0" in r.stderr
    r = run([d, "-winsize", "5", "-stepsize", "1", str(f), str(f)])
    assert r.returncode == 255 and "Must supply size file unless -fixedsite 1" in r.stderr
    r = run([d, "-fixedsite", "1", str(tmp_path / "nope.mafs"), str(f)])
    assert r.returncode == 255 and "Unable to open Pop1 MAF file" in r.stderr


def test_unparsable_line_is_refused(hosts, tmp_path):
    f = tmp_path / "bad.txt"
    f.write_text("c1\t1\t0.1\t0.2\nc1\t2\tnan_or_header\t0.2\n")
    r = run([hosts["fstWindow"], str(f), "1", "1"])
    assert r.returncode == 255 and "line 2" in r.stderr  # the reference would reuse stale values (Q12)
    # a MAF frequency outside [0, 1]: the reference computes a negative dxy that its windows neither add nor count as skipped
    # (dxyWindow.cpp:180-185) while the genome-wide line adds and counts it (:382-385); here the line is refused, nothing printed
    # 
    hdr = "chromo\tposition\tmajor\tminor\tref\tknownEM\tnInd\n"
    m1, m2 = tmp_path / "p1.mafs", tmp_path / "p2.mafs"
    m1.write_text(hdr + "cA\t1\tA\tC\tA\t0.5\t4\ncA\t2\tA\tC\tA\t1.5\t4\n")
    m2.write_text(hdr + "cA\t1\tA\tC\tA\t0.5\t4\ncA\t2\tA\tC\tA\t0.5\t4\n")
    r = run([hosts["dxyWindow"], "-winsize", "1", "-stepsize", "1", "-fixedsite", "1", str(m1), str(m2)])
    assert r.returncode == 255 and r.stdout == "" and "cannot parse MAF line" in r.stderr and "freq in [0,1]" in r.stderr


def test_no_window_input_prints_nothing(hosts, tmp_path):
    f = tmp_path / "short.txt"
    f.write_text("c\t1\t0.1\t0.2\nc\t2\t0.1\t0.2\nc\t3\t0.1\t0.2\n")
    r = run([hosts["fstWindow"], str(f), "5", "2"])
    assert r.returncode == 0 and r.stdout == ""  # N <= W-S (Q2)
    e = tmp_path / "empty.txt"
    e.write_text("")
    assert run([hosts["hetWindow"], str(e), "3", "1"]).stdout == ""


# ---- GPU: TSV parity -------------------------------------------------------------------------
def tsv_equal(mine, ref, float_col):
    a, b = helpers.parse_tsv(mine), helpers.parse_tsv(ref)
    assert len(a) == len(b), (len(a), len(b))
    for x, y in zip(a, b):
        assert len(x) == len(y)
        for k, (u, v) in enumerate(zip(x, y)):
            if k == float_col and u != v:  # %g keeps 6 digits: allow the last printed digit to differ
                assert abs(float(u) - float(v)) <= 1.01e-5 * abs(float(v)) + 1e-12, (x, y)
            else:
                assert u == v, (x, y)


@pytest.mark.gpu
def test_failed_output_write_is_not_a_success(hosts, tmp_path):
    """A TSV that could not be written (disk full, closed pipe) must not end with exit status 0: /dev/full fails every
    write with ENOSPC."""
    src = tmp_path / "in.txt"
    src.write_text("".join(f"c1\t{i}\t0.1\t0.2\n" for i in range(1, 3000)))
    with open("/dev/full", "w") as full:
        r = subprocess.run([hosts["fstWindow"], str(src), "5", "1"], stdout=full, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 255 and "rror" in r.stderr, (r.returncode, r.stderr)
    ok = run([hosts["fstWindow"], str(src), "5", "1"])
    assert ok.returncode == 0 and len(ok.stdout.splitlines()) == 2995


@pytest.mark.gpu
def test_fst_het_cli_against_reference_goldens(hosts, tmp_path):
    cases = helpers.load_golden("ref_kat.json")["cases"] + helpers.load_golden("ref_random.json")["cases"]
    exact = 0
    cases = [c for i, c in enumerate(cases) if not (i % 3 and "note" not in c)]  # every third random case keeps the test short
    jobs = []
    for i, c in enumerate(cases):
        f = tmp_path / f"in{i}.txt"
        f.write_text(c["input"])
        jobs.append(([hosts[c["tool"]], str(f), str(c["W"]), str(c["S"])], None))
    for c, r in zip(cases, run_all(jobs)):
        assert r.returncode == 0, r.stderr
        tsv_equal(r.stdout, c["stdout"], 4)
        exact += r.stdout == c["stdout"]
    assert exact >= 40  # in practice every case is byte-identical


@pytest.mark.gpu
def test_cli_small_step_reference_goldens(hosts, tmp_path):
    """The hosts on the seeded tables of tests/golden/ref_small_step.json (S << W; reference-made): hetWindow's stdout has the
    SHA-256 of the reference's (byte-identical over all rows), fstWindow's rows agree with every k-th reference row."""
    import hashlib
    for c, cols in helpers.small_step_cases():
        f = tmp_path / "in.txt"
        f.write_text(cols["fst"] if c["tool"] == "fstWindow" else cols["het"])
        r = run([hosts[c["tool"]], str(f), str(c["W"]), str(c["S"])])
        assert r.returncode == 0 and len(r.stdout.splitlines()) == c["n_rows"], (c["tool"], c["W"], c["S"], r.stderr)
        if c["tool"] == "hetWindow":
            assert hashlib.sha256(r.stdout.encode()).hexdigest() == c["stdout_sha256"]
        else:
            tsv_equal("\n".join(r.stdout.splitlines()[:: c["every"]]) + "\n", "\n".join(c["rows"]) + "\n", 4)


@pytest.mark.gpu
def test_fst_cli_config1(hosts, tmp_path, oracle):
    import synth
    g = helpers.load_golden("ref_config1.json")
    rng = np.random.default_rng(g["seed"])
    chr_ids, pos = synth.chromosomes(rng, g["n"], g["n_chr"])
    a, b = synth.fst_columns(rng, g["n"])
    f = tmp_path / "c1.txt"
    oracle.write_fst_text(str(f), chr_ids, pos, a, b)
    for runcfg in g["runs"]:
        r = run([hosts["fstWindow"], str(f), str(runcfg["W"]), str(runcfg["S"])])
        assert r.returncode == 0, r.stderr
        tsv_equal(r.stdout, runcfg["stdout"], 4)


def _write_maf(path, header, rows, gz=False):
    text = header + "\n" + "".join(f"{c}\t{p}\tA\tC\tA\t{fr:.6f}\t{n}\n" for c, p, fr, n in rows)
    if gz:
        with gzip.open(path, "wt") as fh:
            fh.write(text)
    else:
        open(path, "w").write(text)


@pytest.mark.gpu
@pytest.mark.parametrize("gz", [False, True])
def test_dxy_cli_known_answers(hosts, tmp_path, gz):
    k = helpers.load_golden("dxy_kat.json")
    m1, m2, sz = tmp_path / ("p1.mafs.gz" if gz else "p1.mafs"), tmp_path / "p2.mafs", tmp_path / "sizes.txt"
    _write_maf(m1, k["header"], k["pop1"], gz)
    _write_maf(m2, k["header"], k["pop2"])
    sz.write_text("".join(f"{c}\t{n}\n" for c, n in k["sizes"]))
    for c in k["cases"]:
        cmd = [hosts["dxyWindow"], "-winsize", str(c["winsize"]), "-stepsize", str(c["stepsize"]),
               "-minind", str(k["minind"]), "-fixedsite", str(c["fixedsite"]), "-skip_missing", str(c["skip_missing"])]
        if not c["fixedsite"]:
            cmd += ["-sizefile", str(sz)]
        r = run(cmd + [str(m1), str(m2)])
        assert r.returncode == 0, r.stderr
        assert r.stdout == c["stdout"] and r.stderr == c["stderr"]


@pytest.mark.gpu
@pytest.mark.parametrize("env_extra", [{}, {"PGT_GPU_INGEST": "1"}, {"PGT_DEVICES": "0,0"}, {"PGT_MAX_RESIDENT_SITES": "1"}])
def test_dxy_cli_hand_walked_cases(hosts, tmp_path, env_extra):
    """The GPU host on tests/golden/dxy_hand_walked.json (bp-slot machine and two-file sync walked through dxyWindow.cpp on
    paper): stdout and stderr byte for byte — with the host parser, the device parser, two contexts and in passes.  Where
    the reference mis-pairs or truncates (H4, H6) the host prints the documented intersection result instead."""
    k = helpers.load_golden("dxy_hand_walked.json")
    for c in k["cases"]:
        m1, m2, sz = helpers.write_hand_walked_case(c, k["header"], tmp_path)
        for r in c["runs"]:
            want_out, want_err = helpers.hand_walked_product_expectation(c, r)
            cmd = [hosts["dxyWindow"], "-winsize", str(r["winsize"]), "-stepsize", str(r["stepsize"]), "-minind", str(c["minind"]),
                   "-fixedsite", str(r["fixedsite"]), "-skip_missing", str(r["skip_missing"])]
            if not r["fixedsite"]:
                cmd += ["-sizefile", sz]
            got = run(cmd + [m1, m2], env=dict(os.environ, **env_extra))
            if want_out is None:  # refused (H10: no shared site)
                assert got.returncode == 255 and got.stdout == "" and want_err in got.stderr, (c["name"], got.returncode, got.stderr)
                continue
            assert got.returncode == 0, (c["name"], got.stderr)
            assert got.stdout == want_out, (c["name"], r, env_extra, got.stdout)
            assert got.stderr == want_err, (c["name"], r, env_extra, got.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("env_extra", [{}, {"PGT_GPU_INGEST": "1"}, {"PGT_DEVICES": "0,0"}])
def test_dxy_cli_reference_sync_mode_prints_what_the_reference_prints(hosts, tmp_path, env_extra):
    """PGT_DXY_SYNC=reference: the host replays the reference's catch-up loops (dxyWindow.cpp:315-331) instead of intersecting
    the two site lists.  (1) EVERY hand-walked case — also H4 (Pop2's extra site at a chromosome end ends the run), H6 (a
    position coincidence pairs sites across chromosomes) and H10 (no shared site: the first chromosome padded with empty
    windows, `0 0 0`) — prints the bytes the paper walk of the reference gives.  (2) 100 random file pairs (identical, nested
    either way, non-nested; every mode) against tests/dxy_stream_model.py, the line-by-line restatement of the reference's
    loop: stdout and stderr equal (the sum column to the printed digits).  The default mode is unchanged
    (test_dxy_cli_hand_walked_cases)."""
    import random
    import dxy_stream_model as model
    env = dict(os.environ, PGT_DXY_SYNC="reference", **env_extra)
    k = helpers.load_golden("dxy_hand_walked.json")
    for c in k["cases"]:
        m1, m2, sz = helpers.write_hand_walked_case(c, k["header"], tmp_path)
        for r in c["runs"]:
            cmd = [hosts["dxyWindow"], "-winsize", str(r["winsize"]), "-stepsize", str(r["stepsize"]), "-minind", str(c["minind"]),
                   "-fixedsite", str(r["fixedsite"]), "-skip_missing", str(r["skip_missing"])]
            if not r["fixedsite"]:
                cmd += ["-sizefile", sz]
            got = run(cmd + [m1, m2], env=env)
            assert (got.returncode, got.stdout, got.stderr) == (0, r["stdout"], r["stderr"]), (c["name"], r, got.stdout, got.stderr)
    bad = run([hosts["dxyWindow"], "-winsize", "2", "-stepsize", "1", "-fixedsite", "1", m1, m2], env=dict(env, PGT_DXY_SYNC="both"))
    assert bad.returncode == 255 and "PGT_DXY_SYNC" in bad.stderr
    if env_extra:
        return  # the random pairs once (host parser, one context): the pairing happens on the host whatever parsed the files
    rng = random.Random(55)
    hdr = "chromo\tposition\tmajor\tminor\tref\tknownEM\tnInd"
    jobs, want = [], []
    while len(jobs) < 100:
        n_chr = rng.randint(1, 4)
        rows1, rows2, sizes = [], [], {}
        kind = rng.choice(["same", "pop2_in_pop1", "pop1_in_pop2", "other", "other"])
        for ci in range(n_chr):
            L = rng.randint(1, 30)
            name = f"c{ci}"
            sizes[name] = L + (rng.randint(0, 3) if rng.random() < 0.3 else 0)
            for p_ in sorted(rng.sample(range(1, L + 1), rng.randint(1, min(L, 8)))):
                if kind in ("same", "pop2_in_pop1") or rng.random() < 0.7:
                    rows1.append((name, p_, round(rng.random(), 6), rng.randint(0, 6)))
                if kind in ("same", "pop1_in_pop2") or rng.random() < 0.7:
                    rows2.append((name, p_, round(rng.random(), 6), rng.randint(0, 6)))
        if not rows1 or not rows2:
            continue
        mode = rng.randint(0, 3)
        W = rng.randint(1, 9)
        S = rng.randint(1, W)
        W, S, fixed = (0, 0, 1) if mode == 0 else (W, S, 1 if mode == 1 else 0)
        minind, skip = rng.randint(1, 4), rng.randint(0, 1)
        d = tmp_path / f"pair{len(jobs)}"
        d.mkdir()
        _write_maf(d / "p1.mafs", hdr, rows1)
        _write_maf(d / "p2.mafs", hdr, rows2)
        (d / "sizes.txt").write_text("".join(f"{c_}\t{n_}\n" for c_, n_ in sizes.items()))
        cmd = [hosts["dxyWindow"], "-winsize", str(W), "-stepsize", str(S), "-minind", str(minind), "-fixedsite", str(fixed), "-skip_missing", str(skip)]
        jobs.append((cmd + ([] if fixed else ["-sizefile", str(d / "sizes.txt")]) + [str(d / "p1.mafs"), str(d / "p2.mafs")], env))
        want.append((kind, model.maf2dxy(rows1, rows2, W, S, minind, fixed, sizes, skip)))
    compared = refused = 0
    for (cmd, _), (kind, (mrc, mout, merr)), got in zip(jobs, want, run_all(jobs)):
        if mrc != 0:  # the first chromosomes differ: both refuse
            assert got.returncode == 255, (cmd, got.stderr)
            continue
        if got.returncode == 255 and "increase strictly" in got.stderr:  # a mis-pairing left positions out of order inside a chromosome: the
            refused += 1                                       # base-pair machine of the product refuses that
            continue
        assert got.returncode == 0, (cmd, kind, got.stderr)
        tsv_equal(got.stdout, mout, 3)
        tsv_equal(got.stderr, merr, 0)
        compared += 1
    assert compared >= 85 and refused <= 6, (compared, refused)


@pytest.mark.gpu
def test_dxy_cli_against_reference_made_cases(hosts, tmp_path):
    """The GPU host against the unmodified reference dxyWindow's recorded runs (gzip members included), when the
    fixture exists (an image with Boost; see tests/golden/make_golden.py): coordinates, counts, labels and the exit
    code exact, the dxy column within 1e-9 + the 6-digit print."""
    cases = helpers.dxy_ref_cases(tmp_path)
    if cases is None:
        pytest.skip("tests/golden/ref_dxy.json absent (no Boost in this image): dxy parity unpinned")
    for c, argv, _ in cases:
        r = run([hosts["dxyWindow"]] + argv)
        assert r.returncode == (0 if c["rc"] == 0 else 255), (c["args"], r.stderr)
        if c["rc"] == 0:
            tsv_equal(r.stdout, c["stdout"], 3)
            tsv_equal(r.stderr, c["stderr"], 0)


@pytest.mark.gpu
def test_dxy_cli_intersects_site_sets(hosts, tmp_path, oracle):
    """pop2 lists extra sites (nested set): rows equal the oracle run on the shared sites."""
    rng = np.random.default_rng(4)
    rows1, rows2, shared = [], [], []
    for c, L in (("cA", 900), ("cB", 500)):
        p2 = np.sort(rng.choice(np.arange(1, L + 1), size=200, replace=False))
        keep = np.sort(rng.choice(p2, size=150, replace=False))
        for p in p2:
            r2 = (c, int(p), round(float(rng.uniform(0, 1)), 6), int(rng.integers(0, 9)))
            rows2.append(r2)
            if p in keep:
                r1 = (c, int(p), round(float(rng.uniform(0, 1)), 6), int(rng.integers(0, 9)))
                rows1.append(r1)
                shared.append((r1, r2))
    hdr = helpers.load_golden("dxy_kat.json")["header"]
    m1, m2, s1, s2, sz = (tmp_path / n for n in ("a.mafs", "b.mafs", "sa.mafs", "sb.mafs", "sizes.txt"))
    _write_maf(m1, hdr, rows1); _write_maf(m2, hdr, rows2)
    _write_maf(s1, hdr, [x for x, _ in shared]); _write_maf(s2, hdr, [y for _, y in shared])
    sz.write_text("cA\t900\ncB\t500\n")
    for fixed in (1, 0):
        o, e = tmp_path / "o.txt", tmp_path / "e.txt"
        assert oracle.dxy_text(str(s1), str(s2), None if fixed else str(sz), 60, 20, 3, fixed, 1, str(o), str(e)) == 0
        cmd = [hosts["dxyWindow"], "-winsize", "60", "-stepsize", "20", "-minind", "3", "-fixedsite", str(fixed), "-skip_missing", "1"]
        if not fixed:
            cmd += ["-sizefile", str(sz)]
        r = run(cmd + [str(m1), str(m2)])
        assert r.returncode == 0, r.stderr
        tsv_equal(r.stdout, o.read_text(), 3)
        tsv_equal(r.stderr, e.read_text(), 0)


@pytest.mark.gpu
def test_dxy_cli_random_modes_vs_oracle_text(hosts, tmp_path, oracle):
    """dxyWindow end to end (MAF text -> parse -> site or bp windows -> GPU -> TSV + the genome-wide line) against the
    oracle's text front end, on random layouts and option mixes: fixed-site and bp windows, window = step and
    window > step, per-site (1/1), global (-winsize 0), -skip_missing, minind above and below the counts."""
    rng = np.random.default_rng(77)
    hdr = helpers.load_golden("dxy_kat.json")["header"]
    m1, m2, sz, o, e = (tmp_path / n for n in ("a.mafs", "b.mafs", "sizes.txt", "o.txt", "e.txt"))
    checked = 0
    for trial in range(24):
        rows1, rows2, sizes = [], [], []
        for c in range(int(rng.integers(1, 5))):
            L = int(rng.integers(5, 400))
            k = int(rng.integers(1, min(L, 120) + 1))
            ps = np.sort(rng.choice(np.arange(1, L + 1), size=k, replace=False))
            sizes.append((f"chr{c + 1}", L + int(rng.integers(0, 30))))
            for p_ in ps:
                rows1.append((f"chr{c + 1}", int(p_), round(float(rng.uniform(0, 1)), 6), int(rng.integers(0, 12))))
                rows2.append((f"chr{c + 1}", int(p_), round(float(rng.uniform(0, 1)), 6), int(rng.integers(0, 12))))
        _write_maf(m1, hdr, rows1); _write_maf(m2, hdr, rows2)
        sz.write_text("".join(f"{c}\t{n}\n" for c, n in sizes))
        fixed = int(rng.integers(0, 2))
        W = int(rng.choice([0, 1, 2, 7, 25, 60])) if fixed else int(rng.choice([1, 3, 10, 40, 100]))
        S = 0 if W == 0 else int(rng.integers(1, W + 1))
        minind, skip = int(rng.integers(1, 9)), int(rng.integers(0, 2))
        assert oracle.dxy_text(str(m1), str(m2), None if fixed else str(sz), W, S, minind, fixed, skip, str(o), str(e)) == 0
        cmd = [hosts["dxyWindow"], "-winsize", str(W), "-stepsize", str(S), "-minind", str(minind), "-fixedsite", str(fixed),
               "-skip_missing", str(skip)]
        if not fixed:
            cmd += ["-sizefile", str(sz)]
        r = run(cmd + [str(m1), str(m2)])
        assert r.returncode == 0, (cmd, r.stderr)
        tsv_equal(r.stdout, o.read_text(), 3)
        tsv_equal(r.stderr, e.read_text(), 0)
        checked += len(r.stdout.splitlines())
    assert checked > 300


@pytest.mark.gpu
def test_cli_step_one_regime(hosts, tmp_path, oracle):
    """S=1 (one window per site): O(N·W) for the reference, O(N·64·log W) here; the
    block-parallel TSV writer must keep row order.  Checked against the oracle's text front end."""
    import synth
    rng = np.random.default_rng(6)
    n, W = 250_000, 3_000
    chr_ids, pos = synth.chromosomes(rng, n, 3, equal=False)
    a, b = synth.fst_columns(rng, n)
    g = synth.het_column(rng, n)
    f, h, o = tmp_path / "fst.txt", tmp_path / "het.txt", tmp_path / "o.txt"
    oracle.write_fst_text(str(f), chr_ids, pos, a, b)
    oracle.write_het_text(str(h), chr_ids, pos, g)
    r = run([hosts["fstWindow"], str(f), str(W), "1"])
    assert r.returncode == 0, r.stderr
    assert oracle.fst_text(str(f), W, 1, str(o)) == 0
    assert len(r.stdout.splitlines()) > n - 3 * W
    tsv_equal(r.stdout, o.read_text(), 4)
    r = run([hosts["hetWindow"], str(h), str(W), "1"])
    assert r.returncode == 0, r.stderr
    assert oracle.het_text(str(h), W, 1, str(o)) == 0
    assert r.stdout == o.read_text()  # integer counts: byte-identical


def test_extreme_cli_arguments(hosts_ext, tmp_path):
    """ihsWindow.cpp:37-79 / xpehhWindow.cpp:42-84: usage exits 1, bad arguments exit 255."""
    f = tmp_path / "x.norm"
    f.write_text("chr1_5\t5\t0.3\t1\t2\t0.5\t1.5\t0\n")
    r = run([hosts_ext["ihsWindow"]])
    assert r.returncode == 1 and "Must supply iHS input file" in r.stderr and "-cutoff FLOAT" in r.stdout
    r = run([hosts_ext["xpehhWindow"], str(f)])
    assert r.returncode == 1 and "Must supply XP-EHH file and cutoff value" in r.stderr
    assert run([hosts_ext["ihsWindow"], str(tmp_path / "none.norm")]).returncode == 255
    assert run([hosts_ext["ihsWindow"], str(f), "-winsize", "0"]).returncode == 255
    assert run([hosts_ext["ihsWindow"], str(f), "-cutoff", "-1"]).returncode == 255
    r = run([hosts_ext["xpehhWindow"], str(f), "2", "-nope", "1"])
    assert r.returncode == 255 and "Unknown argument -nope" in r.stderr


@pytest.fixture(scope="module")
def hosts_ext(hosts):
    return {t: os.path.join(BIN, t) for t in ("ihsWindow", "xpehhWindow")}


def _extreme_argv(hosts_ext, c, tmp_path, i):
    """The argv of one ref_extreme.json case, its files written under a directory of its own (the cases run side by side)."""
    d = tmp_path / f"case{i}"
    d.mkdir()
    paths = {}
    for name, text in c["files"].items():
        (d / name).write_text(text)
        paths[name] = str(d / name)
    return [hosts_ext[c["tool"]]] + [paths[a[1:]] if a.startswith("@") else a for a in c["args"]]


@pytest.mark.gpu
def test_extreme_cli_against_reference_goldens(hosts_ext, tmp_path):
    cases = helpers.load_golden("ref_extreme.json")["cases"]
    cases = cases[::2]
    for c, r in zip(cases, run_all([(_extreme_argv(hosts_ext, c, tmp_path, i), None) for i, c in enumerate(cases)])):
        assert r.returncode == 0, r.stderr
        assert r.stdout == c["stdout"], c["args"]  # selections and integer ratios: byte-identical


@pytest.mark.gpu
@pytest.mark.parametrize("env_extra", [{"PGT_GPU_INGEST": "1"}, {"PGT_DEVICES": "0,0"}, {"PGT_DEVICES": "0,0,0", "PGT_GPU_INGEST": "1"}])
def test_extreme_cli_device_parser_and_several_gpus_print_the_reference_tsv(hosts_ext, tmp_path, env_extra):
    """ihsWindow / xpehhWindow with the table parsed on the GPU (the locus id's `chr_` prefix is the device parser's
    PGT_TOK_CHR_PREFIX; only the positions come back for the window rules) and over two / three contexts (text cut per
    context, window blocks from pgt_plan_shards): all 140 reference-made cases, stdout byte for byte as with the host
    parser on one GPU (ihsWindow.cpp:123-221, xpehhWindow.cpp:126-232)."""
    cases = helpers.load_golden("ref_extreme.json")["cases"]
    assert len(cases) >= 100
    if len(env_extra) == 1:  # all 140 cases with both features together, every third with one of them (the suite's time budget)
        cases = cases[::3]
    env = dict(os.environ, **env_extra)
    for c, r in zip(cases, run_all([(_extreme_argv(hosts_ext, c, tmp_path, i), env) for i, c in enumerate(cases)])):
        assert r.returncode == 0, (c["args"], r.stderr)
        assert r.stdout == c["stdout"], (c["args"], env_extra)


@pytest.mark.gpu
def test_extreme_cli_large_table_all_paths_agree_and_errors_keep_their_line_numbers(hosts_ext, tmp_path):
    """A generated 2 * 10^6-line *.norm table (7 chromosomes, ties, runs of empty windows): host parser on one GPU ==
    device parser == two and three contexts with either parser, for both tools; a bad line deep in the table is reported
    with its line number by every path (xpehhWindow counts its header line); an input beyond the device's memory is
    refused with nothing printed (the extreme-score tools have no passes mode)."""
    rng = np.random.default_rng(31)
    n = 2_000_000
    chrom = np.sort(rng.integers(1, 8, n))
    pos = np.zeros(n, dtype=np.int64)
    for c in range(1, 8):
        m = chrom == c
        pos[m] = np.cumsum(rng.integers(1, 400, int(m.sum())))
    score = np.round(rng.normal(0, 1.2, n), 4)
    score[rng.integers(0, n, 2000)] = 2.5   # ties: the first occurrence wins
    lines = [f"chr{c}_{p}\t{p}\t0.3\t1.1\t2.2\t0.5\t{s}\t0\n" for c, p, s in zip(chrom.tolist(), pos.tolist(), score.tolist())]
    ihs = tmp_path / "big.ihs.norm"
    ihs.write_text("".join(lines))
    xp = tmp_path / "big.xpehh.norm"
    xp.write_text("id\tpos\tgpos\tp1\tihh1\tp2\tihh2\txpehh\tnormxpehh\tcrit\n" +
                  "".join(f"chr{c}_{p}\t{p}\t0.1\t0.3\t1.1\t0.4\t2.2\t0.5\t{s}\t0\n" for c, p, s in zip(chrom.tolist(), pos.tolist(), score.tolist())))
    envs = [{"PGT_GPU_INGEST": "0"}, {"PGT_GPU_INGEST": "1"}, {"PGT_GPU_INGEST": "0", "PGT_DEVICES": "0,0"},
            {"PGT_GPU_INGEST": "1", "PGT_DEVICES": "0,0"}, {"PGT_GPU_INGEST": "1", "PGT_DEVICES": "0,0,0"}]
    for argv in ([hosts_ext["ihsWindow"], str(ihs), "-winsize", "50000", "-cutoff", "2"],
                 [hosts_ext["xpehhWindow"], str(xp), "2", "-winsize", "30000"]):
        outs = [run(argv, env=dict(os.environ, **e)) for e in envs]
        assert outs[0].returncode == 0 and len(outs[0].stdout.splitlines()) > 1000
        for e, r in zip(envs[1:], outs[1:]):
            assert (r.returncode, r.stdout) == (0, outs[0].stdout), (argv[0], e, r.stderr[-500:])
    # a bad line at line 1 500 001 of the data
    bad_at = 1_500_000
    lines_bad = list(lines)
    lines_bad[bad_at] = lines_bad[bad_at].replace("\t0.5\t", "\t0.5\tnot_a_number_")
    ihs.write_text("".join(lines_bad))
    for e in envs:
        r = run([hosts_ext["ihsWindow"], str(ihs), "-winsize", "50000"], env=dict(os.environ, **e))
        assert r.returncode == 255 and r.stdout == "" and f"line {bad_at + 1} of" in r.stderr, (e, r.stderr[-300:])
    ihs.write_text("".join(lines))
    # PGT_MAX_RESIDENT_SITES is a limit PER GPU compared with the table's real line count: beyond it the tools refuse (they
    # have no passes mode) and say which limit it was; a limit the table stays under changes nothing
    r = run([hosts_ext["ihsWindow"], str(ihs)], env=dict(os.environ, PGT_MAX_RESIDENT_SITES="100000"))
    assert r.returncode == 255 and r.stdout == "" and "no passes mode" in r.stderr
    assert f"{len(lines)} lines" in r.stderr and "PGT_MAX_RESIDENT_SITES=100000 x 1 GPU" in r.stderr
    r = run([hosts_ext["ihsWindow"], str(ihs)], env=dict(os.environ, PGT_MAX_RESIDENT_SITES=str(len(lines) // 2 - 1), PGT_DEVICES="0,0"))
    assert r.returncode == 255 and r.stdout == "" and "x 2 GPUs" in r.stderr
    plain = run([hosts_ext["ihsWindow"], str(ihs)])
    for env in ({"PGT_MAX_RESIDENT_SITES": str(len(lines))}, {"PGT_MAX_RESIDENT_SITES": str(len(lines) // 2), "PGT_DEVICES": "0,0"},
                {"PGT_MAX_RESIDENT_SITES": "10000000"}):
        r = run([hosts_ext["ihsWindow"], str(ihs)], env=dict(os.environ, **env))
        assert (r.returncode, r.stdout) == (0, plain.stdout) and plain.returncode == 0, (env, r.stderr[-300:])


@pytest.mark.gpu
def test_cli_input_format_edge_cases(hosts, tmp_path):
    """Text details: CRLF line ends, a last line without newline (the reference loops forever on it,
    SURVEY Q8), exponent notation / leading '+', extra columns, and the stop at the first empty line
    (fstWindow.cpp:125) all give the rows of the plain file."""
    plain = "".join(f"c1\t{10 * i}\t{0.01 * i:.6f}\t0.2\n" for i in range(1, 9)) + "".join(f"c2\t{7 * i}\t0.05\t0.25\n" for i in range(1, 6))
    f = tmp_path / "plain.txt"
    f.write_text(plain)
    ref = run([hosts["fstWindow"], str(f), "4", "2"])
    assert ref.returncode == 0 and len(ref.stdout.splitlines()) >= 4
    variants = {
        "crlf": plain.replace("\n", "\r\n"),
        "no_final_newline": plain.rstrip("\n"),
        "exponent_and_plus": plain.replace("\t0.2\n", "\t+2e-1\n").replace("\t0.25\n", "\t2.5E-1\n"),
        "extra_columns": plain.replace("\n", "\tignored\t1\n"),
        "stops_at_empty_line": plain + "\nc9\t1\t0.5\t0.5\nc9\t2\t0.5\t0.5\nc9\t3\t0.5\t0.5\nc9\t4\t0.5\t0.5\nc9\t5\t0.5\t0.5\n",
        "spaces_for_tabs": plain.replace("\t", "  "),
    }
    for name, text in variants.items():
        p = tmp_path / f"{name}.txt"
        p.write_bytes(text.encode())
        r = run([hosts["fstWindow"], str(p), "4", "2"])
        assert r.returncode == 0, (name, r.stderr)
        assert r.stdout == ref.stdout, name
    # gzip input is read transparently by every host (zlib), not only by dxyWindow
    gz = tmp_path / "plain.txt.gz"
    with gzip.open(gz, "wt") as fh:
        fh.write(plain)
    assert run([hosts["fstWindow"], str(gz), "4", "2"]).stdout == ref.stdout
    # ... and so is bgzf (ANGSD's gzip flavour: inflated block-parallel), with blocks cut inside lines
    bg = tmp_path / "plain.bgzf.gz"
    helpers.write_bgzf(bg, plain.encode(), block=37)
    assert gzip.open(bg, "rb").read() == plain.encode()  # the file is a valid multi-member gzip file
    assert run([hosts["fstWindow"], str(bg), "4", "2"]).stdout == ref.stdout


# ---- binary column cache (PGT_COLUMN_CACHE) ----------------------------------------------
def test_column_cache_is_written_and_tolerates_garbage(hosts, tmp_path):
    """CPU (an input without windows never touches the GPU): the cache file appears on the first run, is
    accepted on the second, and a truncated / foreign file under its name is ignored, not trusted."""
    f = tmp_path / "short.txt"
    f.write_text("c\t1\t0.1\t0.2\nc\t2\t0.1\t0.2\nc\t3\t0.1\t0.2\n")
    cache = tmp_path / "cache"
    cache.mkdir()
    env = dict(os.environ, PGT_COLUMN_CACHE=str(cache), PGT_HOST_TIMING="1")
    r = run([hosts["fstWindow"], str(f), "5", "2"], env=env)
    assert r.returncode == 0 and r.stdout == "" and "parse" in r.stderr and "cache write" in r.stderr
    files = list(cache.glob("*.pgtcols"))
    assert len(files) == 1 and files[0].read_bytes()[:8] == b"PGTCOLS1"
    r = run([hosts["fstWindow"], str(f), "5", "2"], env=env)
    assert r.returncode == 0 and "cache map" in r.stderr and "parse" not in r.stderr
    files[0].write_bytes(b"PGTCOLS1" + b"\xff" * 40)
    r = run([hosts["fstWindow"], str(f), "5", "2"], env=env)
    assert r.returncode == 0 and "parse" in r.stderr  # garbage is not trusted: parsed again (and rewritten)
    # a directory that does not exist only disables the cache
    r = run([hosts["fstWindow"], str(f), "5", "2"], env=dict(env, PGT_COLUMN_CACHE=str(tmp_path / "nope")))
    assert r.returncode == 0


@pytest.mark.gpu
def test_column_cache_gives_the_same_tsv(hosts, tmp_path):
    """fstWindow / hetWindow / dxyWindow (one MAF gzipped): without the cache, writing it, and mapping it back
    print the same bytes; an edited input gets a new key and is parsed again."""
    rng = np.random.default_rng(12)
    n = 30_000
    import synth
    chr_ids, pos = synth.chromosomes(rng, n, 4, equal=False)
    a, b = synth.fst_columns(rng, n)
    g = synth.het_column(rng, n)
    p1, p2, n1, n2 = synth.dxy_columns(rng, n)
    ffst, fhet = tmp_path / "fst.txt", tmp_path / "het.txt"
    ffst.write_text("".join(f"chr{c}\t{p}\t{x:.6f}\t{y:.6f}\n" for c, p, x, y in zip(chr_ids, pos, a, b)))
    fhet.write_text("".join(f"chr{c}\t{p}\t{v}\n" for c, p, v in zip(chr_ids, pos, g)))
    m1, m2 = tmp_path / "p1.mafs.gz", tmp_path / "p2.mafs"
    hdr = "chromo\tposition\tmajor\tminor\tref\tknownEM\tnInd\n"
    with gzip.open(m1, "wt") as fh:
        fh.write(hdr + "".join(f"chr{c}\t{p}\tA\tC\tA\t{x:.6f}\t{k}\n" for c, p, x, k in zip(chr_ids, pos, p1, n1)))
    m2.write_text(hdr + "".join(f"chr{c}\t{p}\tA\tC\tA\t{x:.6f}\t{k}\n" for c, p, x, k in zip(chr_ids, pos, p2, n2)))
    cache = tmp_path / "cache"
    cache.mkdir()
    env = dict(os.environ, PGT_COLUMN_CACHE=str(cache))
    cmds = [[hosts["fstWindow"], str(ffst), "500", "100"], [hosts["hetWindow"], str(fhet), "500", "100"],
            [hosts["dxyWindow"], "-winsize", "500", "-stepsize", "100", "-minind", "5", "-fixedsite", "1", str(m1), str(m2)]]
    outputs = []
    for cmd in cmds:
        plain = run(cmd)
        outputs.append(plain.stdout)
        first = run(cmd, env=env)
        second = run(cmd, env=env)
        assert plain.returncode == first.returncode == second.returncode == 0, plain.stderr + first.stderr + second.stderr
        assert plain.stdout and plain.stdout == first.stdout == second.stdout
        assert plain.stderr == first.stderr == second.stderr  # dxy's genome-wide line
    assert len(list(cache.glob("*.pgtcols"))) == 4  # fst, het, two MAF files
    # edited input: same path, new content -> new key, new answer
    ffst.write_text("".join(f"chr{c}\t{p}\t{y:.6f}\t{x + 0.5:.6f}\n" for c, p, x, y in zip(chr_ids, pos, a, b)))
    os.utime(ffst, ns=(1, 1))
    changed = run(cmds[0], env=env)
    assert changed.returncode == 0 and changed.stdout == run(cmds[0]).stdout and changed.stdout != outputs[0]
    assert len(list(cache.glob("*.pgtcols"))) == 5


# ---- device-side ingest (PGT_GPU_INGEST) against the host parser ---------------------------------------
@pytest.mark.gpu
def test_gpu_ingest_cli_equals_host_parser(hosts, tmp_path):
    """Every host, the same command with the text parsed on the GPU (PGT_GPU_INGEST=1) and by the host parser
    (=0): identical stdout, stderr and exit code — reference-made goldens, inputs with irregular numbers,
    CRLF, blank-line stops, no trailing newline, and errors (the message carries the line number)."""
    def both(cmd):
        a, b = run_all([(cmd, dict(os.environ, PGT_GPU_INGEST="1")), (cmd, dict(os.environ, PGT_GPU_INGEST="0"))])
        assert (a.returncode, a.stdout, a.stderr) == (b.returncode, b.stdout, b.stderr), (cmd, a.stderr[-300:], b.stderr[-300:])
        return a

    cases = helpers.load_golden("ref_kat.json")["cases"] + helpers.load_golden("ref_random.json")["cases"][::4]
    jobs = []
    for i, c in enumerate(cases):
        f = tmp_path / f"in{i}.txt"
        f.write_text(c["input"])
        cmd = [hosts[c["tool"]], str(f), str(c["W"]), str(c["S"])]
        jobs += [(cmd, dict(os.environ, PGT_GPU_INGEST="1")), (cmd, dict(os.environ, PGT_GPU_INGEST="0"))]
    res = run_all(jobs)
    for c, a, b in zip(cases, res[0::2], res[1::2]):
        assert (a.returncode, a.stdout, a.stderr) == (b.returncode, b.stdout, b.stderr) and a.returncode == 0, (c["tool"], a.stderr[-300:], b.stderr[-300:])
        tsv_equal(a.stdout, c["stdout"], 4)
    f = tmp_path / "odd.txt"
    body = ("c1\t1\t0.1\t0.2\nc1 2 -0.000012 0.3\r\nc1\t3\t+0.5\t.5\n  c1 \t 4\t1.\t-.25  extra 7\nc1\t5\t1.5e-05\t1E5\n"
            "c1\t6\t0.1234567890123456789\t123456789012345678\nc2\t7\t1e22\t1e23\nc2\t8\t4.9e-324\t2.2250738585072014e-308\n"
            "c2\t9\tinf\t1\nc2\t10\tnan\t1\nc2\t+11\t+-1\t1\nc3\t00000000000000000012\t0.3\t0.4")
    f.write_text(body)  # no trailing newline
    assert both([hosts["fstWindow"], str(f), "3", "1"]).returncode == 0
    f.write_text(body + "\n\nc9\tgarbage after the blank line\n")
    assert both([hosts["fstWindow"], str(f), "3", "1"]).returncode == 0
    for bad in ("c1\t1\t0.1\n", "c1\tx\t0.1\t0.2\n", "c1\t4294967296\t0.1\t0.2\n", "c1\t1\t1e400\t2\n", "c1\n"):
        f.write_text("c1\t1\t0.1\t0.2\nc1\t2\t0.1\t0.2\n" + bad + "c1\t4\t0.1\t0.2\n")
        r = both([hosts["fstWindow"], str(f), "2", "1"])
        assert r.returncode == 255 and "line 3" in r.stderr
    g = tmp_path / "het.txt"
    g.write_text("cA 10 1\ncA 20 0\r\ncA 30 -1\ncA 40 2\ncA 50 +1\ncB 5 -9\ncB 9 300\n")
    assert both([hosts["hetWindow"], str(g), "3", "2"]).returncode == 0
    g.write_text("cA 10 1\ncA 20 1.5\n")
    assert both([hosts["hetWindow"], str(g), "1", "1"]).returncode == 255
    # dxyWindow: known answers (identical site sets -> columns stay on the GPU) and differing site sets (host merge)
    k = helpers.load_golden("dxy_kat.json")
    m1, m2, sz = tmp_path / "p1.mafs.gz", tmp_path / "p2.mafs", tmp_path / "sizes.txt"
    _write_maf(m1, k["header"], k["pop1"], True)
    _write_maf(m2, k["header"], k["pop2"])
    sz.write_text("".join(f"{c}\t{n}\n" for c, n in k["sizes"]))
    for c in k["cases"]:
        cmd = [hosts["dxyWindow"], "-winsize", str(c["winsize"]), "-stepsize", str(c["stepsize"]),
               "-minind", str(k["minind"]), "-fixedsite", str(c["fixedsite"]), "-skip_missing", str(c["skip_missing"])]
        if not c["fixedsite"]:
            cmd += ["-sizefile", str(sz)]
        r = both(cmd + [str(m1), str(m2)])
        assert r.returncode == 0 and r.stdout == c["stdout"] and r.stderr == c["stderr"]
    sub = [row for i, row in enumerate(k["pop2"]) if i != 1]  # pop2 lacks one site of pop1
    _write_maf(m2, k["header"], sub)
    r = both([hosts["dxyWindow"], "-winsize", "2", "-stepsize", "1", "-minind", "2", "-fixedsite", "1", str(m1), str(m2)])
    assert r.returncode == 0 and r.stdout
    _write_maf(m2, k["header"], [k["pop2"][0], ["cA", 3, 1.5, 4]])  # frequency outside [0,1]
    r = both([hosts["dxyWindow"], "-winsize", "2", "-stepsize", "1", "-fixedsite", "1", str(m1), str(m2)])
    assert r.returncode == 255 and "line 3" in r.stderr


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_cli_on_several_gpus_prints_the_single_gpu_tsv(hosts, tmp_path, oracle):
    """PGT_DEVICES=0,0 and 0,0,0 — several contexts on the one GPU of this box, one host thread each: the window table is
    cut by pgt_plan_shards, every context reduces its block from the columns of its own site range, and the main thread
    prints.  stdout, stderr and the exit code must be those of the single-device run, byte for byte: on the
    reference-made goldens (host parser and device parser: with the latter the TEXT is cut at line starts, one piece per
    context, and a context gathers its shard's columns from the pieces), on inputs with a blank-line stop or a bad line in
    a later piece, and on a 3 * 10^6-line table for fstWindow (steps 10000 and 100: per-window and group query) and hetWindow."""
    import synth

    def runs(cmd, **env):
        lists = ("0,0", "0,0,0")
        one, *several = run_all([(cmd, dict(os.environ, **env))] + [(cmd, dict(os.environ, PGT_DEVICES=devs, **env)) for devs in lists])
        for devs, many in zip(lists, several):
            assert (many.returncode, many.stdout, many.stderr) == (one.returncode, one.stdout, one.stderr), (cmd, devs, env, many.stderr[-300:])
        return one

    cases = helpers.load_golden("ref_kat.json")["cases"] + helpers.load_golden("ref_random.json")["cases"][::9]
    for c in cases:
        f = tmp_path / "in.txt"
        f.write_text(c["input"])
        for ingest in ("0", "1"):
            r = runs([hosts[c["tool"]], str(f), str(c["W"]), str(c["S"])], PGT_GPU_INGEST=ingest)
            assert r.returncode == 0
            tsv_equal(r.stdout, c["stdout"], 4)
    # the data ends at a blank line in the first piece; a bad line in the last piece (the message carries the GLOBAL line number)
    f = tmp_path / "stop.txt"
    body = "".join(f"c{1 + i // 40}\t{i + 1}\t0.0{i % 7}\t0.{1 + i % 5}\n" for i in range(200))
    lines = body.splitlines(True)
    f.write_text("".join(lines[:40]) + "\n" + "".join(lines[40:]))  # a blank line after line 40: the data ends there
    r = runs([hosts["fstWindow"], str(f), "7", "3"], PGT_GPU_INGEST="1")
    assert r.returncode == 0 and r.stdout.splitlines()[-1].split("\t")[2] == "40"
    f.write_text("".join(lines[:190]) + "c5\t191\tx\t0.1\n" + "".join(lines[191:]))
    r = runs([hosts["fstWindow"], str(f), "7", "3"], PGT_GPU_INGEST="1")
    assert r.returncode == 255 and "line 191" in r.stderr
    # a blank line at every line of a short table, also as the LAST line of a context's piece (tests/ingest_fuzz.py found that
    # one: the parser of such a piece sees nothing unusual, yet the data ends there for the pieces behind it)
    g = tmp_path / "stop.het.txt"
    rows_h = [f"c{i // 5}  {10 * i + 3}  {i % 3 - 1}\n" for i in range(12)]
    for at in range(13):  # (every position: with 2 and 3 pieces of 12 lines the piece ends fall on 4, 6, 8; tests/ingest_fuzz.py varies the rest)
        for blank in ((" \n",) if at % 2 else ("\n",)):
            g.write_text("".join(rows_h[:at]) + blank + "".join(rows_h[at:]))
            for W, S in (((340, 136),) if at % 3 else ((340, 136), (3, 2))):
                r = runs([hosts["hetWindow"], str(g), str(W), str(S)], PGT_GPU_INGEST="1")
                assert r.returncode == 0
    # 3 * 10^6 lines (10^7 until round 4, 4 * 10^6 until round 5: the driver gives the whole GPU suite 900 s)
    rng = np.random.default_rng(31)
    n = 3_000_000
    chr_ids, pos = synth.chromosomes(rng, n, 7, equal=False)
    a, b = synth.fst_columns(rng, n)
    big = tmp_path / "big.fst.txt"
    oracle.write_fst_text(str(big), chr_ids, pos, a, b)
    for W, S in ((50_000, 10_000), (50_000, 100)):
        for ingest in ("1", "0"):
            r = runs([hosts["fstWindow"], str(big), str(W), str(S)], PGT_GPU_INGEST=ingest)
            assert r.returncode == 0 and len(r.stdout.splitlines()) > n // S - 7 * (W // S + 1)
    os.unlink(big)
    g = synth.het_column(rng, n)
    bigh = tmp_path / "big.het.txt"
    oracle.write_het_text(str(bigh), chr_ids, pos, g)
    r = runs([hosts["hetWindow"], str(bigh), "50000", "10000"], PGT_GPU_INGEST="1")
    assert r.returncode == 0 and len(r.stdout.splitlines()) > n // 10_000 - 7 * 6


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_dxy_cli_on_several_gpus_prints_the_single_gpu_tsv(hosts, tmp_path, oracle):
    """dxyWindow with PGT_DEVICES=0,0 and 0,0,0 (several contexts on this box's one GPU): the window table is cut by
    pgt_plan_shards, the genome-wide line is summed from 65536-site blocks of the owned site ranges; with the device
    parser file 1 is parsed by the first context and file 2 by the second, and a context copies its slice of the columns
    device to device.  stdout, stderr and the exit code are those of the single-device run: known answers, random option
    mixes (fixed-site, bp, global, per-site, -skip_missing), nested site sets, a bad line in either file, and 3 * 10^6-site
    files in every mode."""
    import synth

    def runs(cmd, **env):
        lists = ("0,0", "0,0,0")
        one, *several = run_all([(cmd, dict(os.environ, **env))] + [(cmd, dict(os.environ, PGT_DEVICES=devs, **env)) for devs in lists])
        for devs, many in zip(lists, several):
            assert (many.returncode, many.stdout, many.stderr) == (one.returncode, one.stdout, one.stderr), (cmd, devs, env, many.stderr[-300:])
        return one

    k = helpers.load_golden("dxy_kat.json")
    m1, m2, sz = tmp_path / "p1.mafs.gz", tmp_path / "p2.mafs", tmp_path / "sizes.txt"
    _write_maf(m1, k["header"], k["pop1"], True)
    _write_maf(m2, k["header"], k["pop2"])
    sz.write_text("".join(f"{c}\t{n}\n" for c, n in k["sizes"]))
    for c in k["cases"]:
        cmd = [hosts["dxyWindow"], "-winsize", str(c["winsize"]), "-stepsize", str(c["stepsize"]),
               "-minind", str(k["minind"]), "-fixedsite", str(c["fixedsite"]), "-skip_missing", str(c["skip_missing"])]
        if not c["fixedsite"]:
            cmd += ["-sizefile", str(sz)]
        for ingest in ("0", "1"):
            r = runs(cmd + [str(m1), str(m2)], PGT_GPU_INGEST=ingest)
            assert r.returncode == 0 and r.stdout == c["stdout"] and r.stderr == c["stderr"]
    sub = [row for i, row in enumerate(k["pop2"]) if i != 1]  # pop2 lacks one site of pop1: host merge
    _write_maf(m2, k["header"], sub)
    for ingest in ("0", "1"):
        r = runs([hosts["dxyWindow"], "-winsize", "2", "-stepsize", "1", "-minind", "2", "-fixedsite", "1", str(m1), str(m2)], PGT_GPU_INGEST=ingest)
        assert r.returncode == 0 and r.stdout
    # a bad line: in Pop2 only, then in both (Pop1's message wins, as in the single-device run)
    _write_maf(m2, k["header"], [k["pop2"][0], ["cA", 3, 1.5, 4]])
    r = runs([hosts["dxyWindow"], "-winsize", "2", "-stepsize", "1", "-fixedsite", "1", str(m1), str(m2)], PGT_GPU_INGEST="1")
    assert r.returncode == 255 and "line 3" in r.stderr and "p2.mafs" in r.stderr
    bad1 = tmp_path / "bad1.mafs"
    _write_maf(bad1, k["header"], [k["pop1"][0], k["pop1"][1], ["cA", 9, -0.5, 4]])
    r = runs([hosts["dxyWindow"], "-winsize", "2", "-stepsize", "1", "-fixedsite", "1", str(bad1), str(m2)], PGT_GPU_INGEST="1")
    assert r.returncode == 255 and "line 4" in r.stderr and "bad1.mafs" in r.stderr
    # 3 * 10^6 sites in 6 uneven chromosomes
    rng = np.random.default_rng(99)
    n = 3_000_000
    chr_ids, pos = synth.chromosomes(rng, n, 6, equal=False)
    p1, p2, n1, n2 = synth.dxy_columns(rng, n)
    f1, f2 = tmp_path / "big1.mafs", tmp_path / "big2.mafs"
    oracle.write_maf_text(str(f1), chr_ids, pos, p1, n1)
    oracle.write_maf_text(str(f2), chr_ids, pos, p2, n2)
    ends = np.cumsum(np.diff(np.concatenate(([0], np.flatnonzero(np.diff(chr_ids)) + 1, [n])))) - 1
    sz.write_text("".join(f"chr{c + 1}\t{int(pos[e]) + 777}\n" for c, e in enumerate(ends)))
    modes = [["-winsize", "50000", "-stepsize", "10000", "-minind", "5", "-fixedsite", "1"],
             ["-winsize", "20000", "-stepsize", "100", "-minind", "5", "-fixedsite", "1", "-skip_missing", "1"],
             ["-winsize", "50000", "-stepsize", "10000", "-minind", "5", "-sizefile", str(sz)],
             ["-winsize", "1000000", "-stepsize", "2000", "-minind", "3", "-sizefile", str(sz)],
             ["-winsize", "0", "-fixedsite", "1", "-minind", "5"],
             ["-winsize", "1", "-stepsize", "1", "-minind", "5", "-fixedsite", "1"]]
    for opts in modes:
        for ingest in ("1", "0"):
            r = runs([hosts["dxyWindow"]] + opts + [str(f1), str(f2)], PGT_GPU_INGEST=ingest)
            assert r.returncode == 0 and (r.stdout if opts[1] != "0" else r.stdout.count("\n") == 1), (opts, r.stderr[-300:])
    # pop2 lists a subset of pop1's sites (host merge of large tables)
    keep = np.sort(rng.choice(n, size=n - 200_000, replace=False))
    oracle.write_maf_text(str(f2), chr_ids[keep], pos[keep], p2[keep], n2[keep])
    for ingest in ("1", "0"):
        r = runs([hosts["dxyWindow"]] + modes[0] + [str(f1), str(f2)], PGT_GPU_INGEST=ingest)
        assert r.returncode == 0 and r.stdout


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_cli_hybrid_ingest_prints_the_same_tsv(hosts, tmp_path, oracle):
    """Large inputs on one GPU: the host threads parse the head of the text beside HIP start-up, the GPU parses the tail, the
    columns meet on the GPU (host_common.h: ingest_hybrid; by default from 2 GiB on, cut near 1 GiB — here moved with
    PGT_HYBRID_HOST_BYTES).  stdout, stderr and the exit code are those of the run without it: reference-made goldens with the
    cut at every few bytes, 3 * 10^6-line tables (fstWindow per-window / group / sliding query, hetWindow) with the cut in
    many places, a blank line in the head, at the cut and in the tail, a bad line in the head and in the tail (global line
    numbers), a chromosome run that continues across the cut."""
    import synth

    def hybrid(cmd, cut, **env):
        return run(cmd, env=dict(os.environ, PGT_GPU_INGEST="1", PGT_HYBRID_HOST_BYTES=str(cut), **env))

    for c in helpers.load_golden("ref_kat.json")["cases"] + helpers.load_golden("ref_random.json")["cases"][::10]:
        f = tmp_path / "in.txt"
        f.write_text(c["input"])
        cuts = list(range(1, len(c["input"]) + 2, max(1, len(c["input"]) // 5)))
        cmd = [hosts[c["tool"]], str(f), str(c["W"]), str(c["S"])]
        for cut, r in zip(cuts, run_all([(cmd, dict(os.environ, PGT_GPU_INGEST="1", PGT_HYBRID_HOST_BYTES=str(cut))) for cut in cuts])):
            assert r.returncode == 0, (cut, r.stderr[-300:])
            tsv_equal(r.stdout, c["stdout"], 4)
    rng = np.random.default_rng(2024)
    n = 3_000_000
    chr_ids, pos = synth.chromosomes(rng, n, 7, equal=False)
    a, b = synth.fst_columns(rng, n)
    big = tmp_path / "big.fst.txt"
    oracle.write_fst_text(str(big), chr_ids, pos, a, b)
    size = os.path.getsize(big)
    for W, S in ((50_000, 10_000), (50_000, 100), (700, 5)):
        cmd = [hosts["fstWindow"], str(big), str(W), str(S)]
        one = run(cmd, env=dict(os.environ, PGT_GPU_INGEST="1", PGT_HYBRID_HOST_BYTES="0"))
        assert one.returncode == 0 and one.stdout
        cuts = (1, 4097, size // 3, size // 2 + 11, size - 40, size, size + 5) if S == 10_000 else (4097, size // 2 + 11, size - 40)
        for cut, r in zip(cuts, run_all([(cmd, dict(os.environ, PGT_GPU_INGEST="1", PGT_HYBRID_HOST_BYTES=str(cut))) for cut in cuts])):
            assert (r.returncode, r.stdout, r.stderr) == (0, one.stdout, one.stderr), (W, S, cut, r.stderr[-300:])
    timed = hybrid([hosts["fstWindow"], str(big), "50000", "10000"], size // 2, PGT_HOST_TIMING="1")
    assert "head on the host" in timed.stderr and "columns joined" in timed.stderr
    lines = big.read_text().splitlines(True)
    at = len("".join(lines[:1_500_000]))  # the cut will fall right behind line 1 500 000
    odd = tmp_path / "odd.fst.txt"
    cmd = [hosts["fstWindow"], str(odd), "50000", "10000"]
    for where in (400_000, 1_500_000, 2_600_000):  # blank line: in the head, first line of the tail, in the tail
        odd.write_text("".join(lines[:where]) + " \n" + "".join(lines[where:]))
        one, r = run(cmd, env=dict(os.environ, PGT_GPU_INGEST="0")), hybrid(cmd, at)
        assert one.returncode == 0 and (r.returncode, r.stdout, r.stderr) == (0, one.stdout, one.stderr), where
    for where in (400_000, 2_600_000):  # a bad line in the head / in the tail
        odd.write_text("".join(lines[:where]) + lines[where].replace("\t", "\tx", 2) + "".join(lines[where + 1:]))
        one, r = run(cmd, env=dict(os.environ, PGT_GPU_INGEST="0")), hybrid(cmd, at)
        assert one.returncode == 255 and f"line {where + 1} " in one.stderr and (r.returncode, r.stdout, r.stderr) == (255, one.stdout, one.stderr), where
    os.unlink(odd)
    os.unlink(big)
    g = synth.het_column(rng, n)
    bigh = tmp_path / "big.het.txt"
    oracle.write_het_text(str(bigh), chr_ids, pos, g)
    size = os.path.getsize(bigh)
    for W, S in ((50_000, 10_000), (200_000, 64)):
        cmd = [hosts["hetWindow"], str(bigh), str(W), str(S)]
        one = run(cmd, env=dict(os.environ, PGT_GPU_INGEST="1", PGT_HYBRID_HOST_BYTES="0"))
        for cut in (1, size // 2, size - 9):
            r = hybrid(cmd, cut)
            assert (r.returncode, r.stdout, r.stderr) == (0, one.stdout, one.stderr), (W, S, cut, r.stderr[-300:])


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_dxy_cli_in_passes_prints_the_resident_tsv(hosts, tmp_path, oracle):
    """dxyWindow with PGT_MAX_RESIDENT_SITES: both MAF texts are scanned for runs and row marks, then reduced block by block
    (the text of a block's rows of both files -> device parser -> reduce -> print; base-pair windows after one extra pass over
    file 1 for the positions), the genome-wide line from the blocks' 65536-site rows.  stdout, stderr and the exit code are
    the resident run's: known answers, 3 * 10^6-site files in every mode and at several limits, a bad line in either file;
    files whose site lists differ — in their runs or only in their positions — run through the resident path (same output)."""
    import synth

    def passes(cmd, limit):
        return run(cmd, env=dict(os.environ, PGT_MAX_RESIDENT_SITES=str(limit)))

    k = helpers.load_golden("dxy_kat.json")
    m1, m2, sz = tmp_path / "p1.mafs.gz", tmp_path / "p2.mafs", tmp_path / "sizes.txt"
    _write_maf(m1, k["header"], k["pop1"], True)
    _write_maf(m2, k["header"], k["pop2"])
    sz.write_text("".join(f"{c}\t{n}\n" for c, n in k["sizes"]))
    for c in k["cases"]:
        cmd = [hosts["dxyWindow"], "-winsize", str(c["winsize"]), "-stepsize", str(c["stepsize"]),
               "-minind", str(k["minind"]), "-fixedsite", str(c["fixedsite"]), "-skip_missing", str(c["skip_missing"])]
        if not c["fixedsite"]:
            cmd += ["-sizefile", str(sz)]
        r = passes(cmd + [str(m1), str(m2)], 1)
        assert r.returncode == 0 and r.stdout == c["stdout"] and r.stderr == c["stderr"], (c, r.stderr)
    sub = [row for i, row in enumerate(k["pop2"]) if i != 1]  # pop2 lacks one site of pop1: the resident path's host merge
    _write_maf(m2, k["header"], sub)
    cmd = [hosts["dxyWindow"], "-winsize", "2", "-stepsize", "1", "-minind", "2", "-fixedsite", "1", str(m1), str(m2)]
    one, r = run(cmd), passes(cmd, 1)
    assert one.returncode == 0 and (r.returncode, r.stdout, r.stderr) == (0, one.stdout, one.stderr)
    moved = [list(row) for row in k["pop2"]]  # the same number of sites per chromosome, one position differs
    assert moved[0][0] == moved[1][0] and moved[1][1] - 1 > moved[0][1]
    moved[1][1] -= 1
    _write_maf(m2, k["header"], moved)
    one, r = run(cmd), passes(cmd, 1)  # seen in the first scan (position digests), before a row is printed: the resident path runs
    assert one.returncode == 0 and (r.returncode, r.stdout, r.stderr) == (0, one.stdout, one.stderr)
    rng = np.random.default_rng(123)
    n = 3_000_000
    chr_ids, pos = synth.chromosomes(rng, n, 6, equal=False)
    p1, p2, n1, n2 = synth.dxy_columns(rng, n)
    f1, f2 = tmp_path / "big1.mafs", tmp_path / "big2.mafs"
    oracle.write_maf_text(str(f1), chr_ids, pos, p1, n1)
    oracle.write_maf_text(str(f2), chr_ids, pos, p2, n2)
    ends = np.cumsum(np.diff(np.concatenate(([0], np.flatnonzero(np.diff(chr_ids)) + 1, [n])))) - 1
    sz.write_text("".join(f"chr{c + 1}\t{int(pos[e]) + 777}\n" for c, e in enumerate(ends)))
    modes = [["-winsize", "50000", "-stepsize", "10000", "-minind", "5", "-fixedsite", "1"],
             ["-winsize", "20000", "-stepsize", "100", "-minind", "5", "-fixedsite", "1", "-skip_missing", "1"],
             ["-winsize", "50000", "-stepsize", "10000", "-minind", "5", "-sizefile", str(sz)],
             ["-winsize", "1000000", "-stepsize", "2000", "-minind", "3", "-sizefile", str(sz)],
             ["-winsize", "0", "-fixedsite", "1", "-minind", "5"]]
    for opts in modes:
        cmd = [hosts["dxyWindow"]] + opts + [str(f1), str(f2)]
        one = run(cmd)
        assert one.returncode == 0
        timed = run(cmd, env=dict(os.environ, PGT_MAX_RESIDENT_SITES="300000", PGT_HOST_TIMING="1"))
        assert "scan runs" in timed.stderr and "passes" in timed.stderr and timed.stdout == one.stdout
        for limit in (1, 300_000, 1_500_000, 10**9):
            r = passes(cmd, limit)
            assert (r.returncode, r.stdout, r.stderr) == (0, one.stdout, one.stderr), (opts, limit, r.stderr[-300:])
    # a bad line in a later block of file 2, then of file 1 as well (Pop1's message first where both are in one block)
    lines2 = f2.read_text().splitlines(True)
    bad2 = tmp_path / "bad2.mafs"
    bad2.write_text("".join(lines2[:2_000_000]) + lines2[2_000_000].rstrip("\n") + "x\n" + "".join(lines2[2_000_001:]))
    cmd = [hosts["dxyWindow"]] + modes[0] + [str(f1), str(bad2)]
    one, r = run(cmd), passes(cmd, 300_000)
    assert one.returncode == 255 and "line 2000001 " in one.stderr and (r.returncode, r.stderr) == (255, one.stderr)
    full = run([hosts["dxyWindow"]] + modes[0] + [str(f1), str(f2)]).stdout
    assert r.stdout and full.startswith(r.stdout) and len(r.stdout) < len(full)


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_cli_in_passes_prints_the_resident_tsv(hosts, tmp_path, oracle):
    """PGT_MAX_RESIDENT_SITES=<n>: the table is reduced block by block (first scan of the text on the host for runs and
    row marks, then per block: text of its rows -> GPU parser -> reduce -> print), as for an input larger than the GPU's
    memory.  stdout and the exit code must be those of the resident run, byte for byte: reference-made goldens, a
    blank-line stop, a bad line (global line number; the rows of earlier blocks are out by then), and 3 * 10^6-line tables
    in 7 uneven chromosomes for fstWindow (per-window, group and sliding query) and hetWindow at several limits."""
    import synth

    def passes(cmd, limit, **env):
        return run(cmd, env=dict(os.environ, PGT_MAX_RESIDENT_SITES=str(limit), **env))

    cases = helpers.load_golden("ref_kat.json")["cases"] + helpers.load_golden("ref_random.json")["cases"][::3]
    jobs = []
    for i, c in enumerate(cases):
        f = tmp_path / f"in{i}.txt"
        f.write_text(c["input"])
        jobs.append(([hosts[c["tool"]], str(f), str(c["W"]), str(c["S"])], dict(os.environ, PGT_MAX_RESIDENT_SITES="1")))
    for c, r in zip(cases, run_all(jobs)):
        assert r.returncode == 0, r.stderr[-300:]
        tsv_equal(r.stdout, c["stdout"], 4)
    import hashlib
    for c, cols in helpers.small_step_cases():  # S << W, reference-made
        f = tmp_path / "in.txt"
        f.write_text(cols["fst"] if c["tool"] == "fstWindow" else cols["het"])
        r = passes([hosts[c["tool"]], str(f), str(c["W"]), str(c["S"])], 1)
        assert r.returncode == 0 and len(r.stdout.splitlines()) == c["n_rows"], (c["tool"], c["W"], c["S"], r.stderr)
        if c["tool"] == "hetWindow":
            assert hashlib.sha256(r.stdout.encode()).hexdigest() == c["stdout_sha256"]
        else:
            tsv_equal("\n".join(r.stdout.splitlines()[:: c["every"]]) + "\n", "\n".join(c["rows"]) + "\n", 4)
    rng = np.random.default_rng(77)
    n = 3_000_000
    chr_ids, pos = synth.chromosomes(rng, n, 7, equal=False)
    a, b = synth.fst_columns(rng, n)
    big = tmp_path / "big.fst.txt"
    oracle.write_fst_text(str(big), chr_ids, pos, a, b)
    for W, S in ((50_000, 10_000), (50_000, 100), (700, 5), (131_072, 65_536), (1, 1)):
        if S == 1:
            small = tmp_path / "small.fst.txt"
            small.write_text("".join(big.read_text().splitlines(True)[:400_000]))
            cmd = [hosts["fstWindow"], str(small), str(W), str(S)]
        else:
            cmd = [hosts["fstWindow"], str(big), str(W), str(S)]
        one = run(cmd)
        assert one.returncode == 0 and one.stdout
        timed = passes(cmd, 300_000, PGT_HOST_TIMING="1")  # the phases say which path ran
        assert "scan runs" in timed.stderr and "passes" in timed.stderr and timed.stdout == one.stdout
        assert "scan runs" not in run(cmd, env=dict(os.environ, PGT_HOST_TIMING="1")).stderr
        limits = (1, 300_000, 1_000_000, 2_999_999, 10**9) if S == 10_000 else (1, 1_000_000)  # (tests/ingest_fuzz.py draws more)
        lists = ("0,0", "0,0,0") if S == 10_000 else ("0,0,0",)  # the blocks go round several contexts and are printed in order
        res = run_all([(cmd, dict(os.environ, PGT_MAX_RESIDENT_SITES=str(limit))) for limit in limits] +
                      [(cmd, dict(os.environ, PGT_MAX_RESIDENT_SITES="300000", PGT_DEVICES=devs)) for devs in lists])
        for what, r in zip(limits + lists, res):
            assert (r.returncode, r.stderr) == (0, one.stderr), (W, S, what, r.stderr[-300:])
            assert r.stdout == one.stdout, (W, S, what)
    # no window at all (every run shorter than the window), and a bad line: still an error, as in the resident run
    short = tmp_path / "short.fst.txt"
    short.write_text("".join(f"c{1 + i // 50}\t{i + 1}\t0.01\t0.2\n" for i in range(200)))
    for body_fix in (lambda t: t, lambda t: t.replace("c3\t120\t0.01", "c3\t120\tzero")):
        short.write_text(body_fix(short.read_text()))
        one, r = run([hosts["fstWindow"], str(short), "60", "10"]), passes([hosts["fstWindow"], str(short), "60", "10"], 1)
        assert (r.returncode, r.stdout, r.stderr) == (one.returncode, one.stdout, one.stderr)
    assert one.returncode == 255 and "line 120 " in one.stderr
    # the data ends at a blank line in a later block; then a bad line there
    lines = big.read_text().splitlines(True)
    stop = tmp_path / "stop.fst.txt"
    stop.write_text("".join(lines[:2_100_000]) + "  \n" + "".join(lines[2_100_000:]))
    cmd = [hosts["fstWindow"], str(stop), "50000", "10000"]
    one = run(cmd)
    r = passes(cmd, 300_000)
    assert one.returncode == 0 and (r.returncode, r.stdout) == (0, one.stdout)
    stop.write_text("".join(lines[:2_100_000]) + lines[2_100_000].split("\t")[0] + "\t5\tzero\t1\n" + "".join(lines[2_100_001:]))
    one = run(cmd)
    r = passes(cmd, 300_000)
    assert one.returncode == 255 and "line 2100001 " in one.stderr and one.stdout == ""
    assert r.returncode == 255 and r.stderr == one.stderr
    full = run([hosts["fstWindow"], str(big), "50000", "10000"]).stdout
    assert r.stdout and full.startswith(r.stdout) and len(r.stdout) < len(full)  # what was printed before the bad block is right
    r = passes(cmd, 300_000, PGT_DEVICES="0,0,0")
    assert r.returncode == 255 and r.stderr == one.stderr and full.startswith(r.stdout) and len(r.stdout) < len(full)
    os.unlink(stop)
    os.unlink(big)
    g = synth.het_column(rng, n)
    bigh = tmp_path / "big.het.txt"
    oracle.write_het_text(str(bigh), chr_ids, pos, g)
    for W, S in ((50_000, 10_000), (200_000, 64), (1000, 1000)):
        cmd = [hosts["hetWindow"], str(bigh), str(W), str(S)]
        one = run(cmd)
        assert one.returncode == 0 and one.stdout
        for limit in (1, 500_000):
            r = passes(cmd, limit)
            assert (r.returncode, r.stdout, r.stderr) == (0, one.stdout, one.stderr), (W, S, limit, r.stderr[-300:])
        r = passes(cmd, 200_000, PGT_DEVICES="0,0")
        assert (r.returncode, r.stdout, r.stderr) == (0, one.stdout, one.stderr), (W, S, r.stderr[-300:])


@pytest.mark.gpu
def test_cli_live_against_the_reference_binaries(hosts, hosts_ext, tmp_path):
    """Where oracle/_ref holds the compiled, unmodified reference tools (this container, and the GPU box, which
    receives them with the snapshot): fresh random inputs — other seeds than the committed goldens — through the
    reference binary and through the host, stdout compared (fstWindow: the %g column to its last printed digit;
    hetWindow, ihsWindow, xpehhWindow: byte for byte)."""
    import importlib.util
    import random
    import oracle_bind
    if not all(oracle_bind.ref_binary(t) for t in ("fstWindow", "hetWindow", "ihsWindow", "xpehhWindow")):
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    rng = random.Random(20261004)
    exact = total = 0
    for tool in ("fstWindow", "hetWindow"):
        for _ in range(15):
            text, W, S = mg.random_case(rng, tool, max_chr=6, max_sites=60, max_w=25)
            f = tmp_path / "in.txt"
            f.write_text(text)
            ref, mine = run_all([([oracle_bind.ref_binary(tool), str(f), str(W), str(S)], None), ([hosts[tool], str(f), str(W), str(S)], None)])
            assert mine.returncode == 0 and ref.returncode == 0, (tool, W, S, mine.stderr)
            if tool == "fstWindow":
                tsv_equal(mine.stdout, ref.stdout, 4)
            else:
                assert mine.stdout == ref.stdout, (tool, W, S)
            exact += mine.stdout == ref.stdout
            total += 1
    for tool in ("ihsWindow", "xpehhWindow"):
        for _ in range(12):
            files, args = mg.extreme_case(rng, tool, max_chr=5, max_sites=60)
            paths = {}
            for name, text in files.items():
                p = tmp_path / name
                p.write_text(text)
                paths[name] = str(p)
            argv = [paths[a[1:]] if a.startswith("@") else a for a in args]
            ref, mine = run_all([([oracle_bind.ref_binary(tool)] + argv, None), ([hosts_ext[tool]] + argv, None)])
            assert mine.returncode == 0 and ref.returncode == 0, (tool, args, mine.stderr)
            assert mine.stdout == ref.stdout, (tool, args)
            exact += 1
            total += 1
    assert exact >= total - 3  # fstWindow rows may differ in the sixth digit of a ratio; in practice none does
    # S << W with windows of two and more level-2 tiles: the group query (steps 1 .. 1024), the sliding query (short
    # windows, steps up to 32) and one wave per window, each against the reference's own re-summation
    # (fstWindow.cpp:80-83,95-99) — larger tables than the fixtures can hold
    import synth
    nrng = np.random.default_rng(20261004)
    for n, n_chr, W, S in ((60_000, 3, 20_000, 1), (90_000, 2, 16_384, 7), (150_000, 4, 50_000, 100), (150_000, 1, 24_577, 64),
                           (200_000, 5, 30_000, 1000), (30_000, 2, 3_000, 5), (200_000, 3, 50_000, 2500), (400_000, 2, 140_000, 300)):
        chr_ids, pos = synth.chromosomes(nrng, n, n_chr, equal=False)
        a_, b_ = synth.fst_columns(nrng, n)
        g_ = synth.het_column(nrng, n)
        f = tmp_path / "big.txt"
        f.write_text("".join(f"chr{c}\t{p}\t{x:.6f}\t{y:.6f}\n" for c, p, x, y in zip(chr_ids, pos, a_, b_)))
        fh_ = tmp_path / "big.het.txt"
        fh_.write_text("".join(f"chr{c}\t{p}\t{v}\n" for c, p, v in zip(chr_ids, pos, g_)))
        ref, mine, refh, mineh = run_all([([oracle_bind.ref_binary("fstWindow"), str(f), str(W), str(S)], None), ([hosts["fstWindow"], str(f), str(W), str(S)], None),
                                          ([oracle_bind.ref_binary("hetWindow"), str(fh_), str(W), str(S)], None), ([hosts["hetWindow"], str(fh_), str(W), str(S)], None)],
                                         workers=4)  # two of the four are the CPU reference
        assert mine.returncode == 0 and ref.returncode == 0 and len(ref.stdout.splitlines()) > 10, (W, S, mine.stderr)
        tsv_equal(mine.stdout, ref.stdout, 4)
        assert mineh.returncode == 0 and mineh.stdout == refh.stdout, ("hetWindow", W, S)


@pytest.mark.gpu
def test_cli_device_window_table_equals_host_table(hosts, tmp_path, oracle):
    """From 2^20 windows on the hosts let the GPU write the site-window table (pgt_wintab_sites) and find the rows'
    chromosome names from the run offsets.  PGT_DEVICE_WINTAB=1 forces that path on small inputs: reference-made
    goldens, a multi-chromosome -stepsize 1 run and dxyWindow's fixed-site modes must print the same bytes as with
    the host table (=0); one run above the threshold takes the path by itself."""
    import synth
    on, off = dict(os.environ, PGT_DEVICE_WINTAB="1"), dict(os.environ, PGT_DEVICE_WINTAB="0")
    cases = helpers.load_golden("ref_kat.json")["cases"] + helpers.load_golden("ref_random.json")["cases"][::4]
    jobs = []
    for i, c in enumerate(cases):
        f = tmp_path / f"in{i}.txt"
        f.write_text(c["input"])
        cmd = [hosts[c["tool"]], str(f), str(c["W"]), str(c["S"])]
        jobs += [(cmd, on), (cmd, off)]
    res = run_all(jobs)
    for c, a, b in zip(cases, res[0::2], res[1::2]):
        assert a.returncode == b.returncode == 0 and a.stdout == b.stdout, (c["tool"], c["W"], c["S"])
        tsv_equal(a.stdout, c["stdout"], 4)
    k = helpers.load_golden("dxy_kat.json")
    m1, m2 = tmp_path / "p1.mafs", tmp_path / "p2.mafs"
    _write_maf(m1, k["header"], k["pop1"]); _write_maf(m2, k["header"], k["pop2"])
    for c in k["cases"]:
        if not c["fixedsite"]:
            continue
        cmd = [hosts["dxyWindow"], "-winsize", str(c["winsize"]), "-stepsize", str(c["stepsize"]), "-minind", str(k["minind"]),
               "-fixedsite", "1", "-skip_missing", str(c["skip_missing"]), str(m1), str(m2)]
        a = run(cmd, env=on)
        assert a.returncode == 0 and a.stdout == c["stdout"] and a.stderr == c["stderr"]
    # above the threshold: 1.2e6 sites in 5 chromosomes, one window per site
    rng = np.random.default_rng(8)
    n = 1_200_000
    chr_ids, pos = synth.chromosomes(rng, n, 5, equal=False)
    g = synth.het_column(rng, n)
    h = tmp_path / "het.txt"
    oracle.write_het_text(str(h), chr_ids, pos, g)
    a = run([hosts["hetWindow"], str(h), "40", "1"], env=dict(os.environ, PGT_HOST_TIMING="1"))
    b = run([hosts["hetWindow"], str(h), "40", "1"], env=off)
    assert a.returncode == b.returncode == 0 and a.stdout == b.stdout and len(a.stdout.splitlines()) > n - 5 * 40
