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
