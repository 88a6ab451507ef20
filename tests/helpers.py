"""Shared helpers for the parity tests."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")

# float columns are compared numerically: |x - y| <= REL * |y| + ABS (north_star: 1e-9 relative;
# the absolute floor covers ratios of sums that cancel to ~0, SURVEY.md §7 hard parts)
REL, ABS = 1e-9, 1e-12


def load_golden(name):
    return json.load(open(os.path.join(GOLDEN, name)))


def parse_table(text, kind):
    """fst / het text -> (names per run, chr run ids, pos, cols...) exactly as the tools tokenise it."""
    names, chr_ids, pos, c1, c2 = [], [], [], [], []
    for line in text.splitlines():
        if not line.strip():
            break
        tok = line.split()
        if not names or names[-1] != tok[0]:
            names.append(tok[0])
        chr_ids.append(len(names) - 1)
        pos.append(int(tok[1]))
        if kind == "fst":
            c1.append(float(tok[2]))
            c2.append(float(tok[3]))
        else:
            c1.append(int(tok[2]))
    chr_ids = np.array(chr_ids, dtype=np.uint32)
    pos = np.array(pos, dtype=np.uint64).astype(np.uint32)
    if kind == "fst":
        return names, chr_ids, pos, np.array(c1, dtype=np.float64), np.array(c2, dtype=np.float64)
    return names, chr_ids, pos, np.array(c1, dtype=np.int32)


def parse_tsv(stdout):
    return [ln.split("\t") for ln in stdout.splitlines() if ln]


def close(x, y):
    return abs(x - y) <= REL * abs(y) + ABS


def fmt_g(x):
    """std::cout << double with default precision == printf('%g')."""
    s = "%g" % x
    return s


def assert_rows_match_tsv(names, win, rows, stat_field, count_field, tsv_rows, with_mid=True):
    """Integers, coordinates and labels byte-exact against the reference TSV; the float column
    numerically (the TSV carries 6 significant digits, so the tolerance is half a unit of the
    6th digit on top of REL)."""
    assert len(rows) == len(tsv_rows), (len(rows), len(tsv_rows))
    for w, r, t in zip(win, rows, tsv_rows):
        assert names[int(w["label_run"])] == t[0]
        assert str(int(r["start"])) == t[1] and str(int(r["end"])) == t[2]
        k = 3
        if with_mid:
            assert str(int(r["mid"])) == t[3]
            k = 4
        ref = float(t[k])
        got = float(r[stat_field])
        assert abs(got - ref) <= 5.1e-6 * abs(ref) + 1e-12, (got, ref)
        assert str(int(r[count_field])) == t[k + 1]


def extreme_case_args(case, tmp_path):
    """Materialise a ref_extreme.json case: returns (in_path, W, cutoff, chrlen_path or None)."""
    paths = {}
    for name, text in case["files"].items():
        p = tmp_path / name
        p.write_text(text)
        paths[name] = str(p)
    a = case["args"]
    W = int(a[a.index("-winsize") + 1])
    chrlen = paths[a[a.index("-chrlen") + 1][1:]] if "-chrlen" in a else None
    cutoff = float(a[a.index("-cutoff") + 1]) if case["tool"] == "ihsWindow" else float(a[1])
    return paths["in.norm"], W, cutoff, chrlen


def write_bgzf(path, data: bytes, block=0xff00, level=6):
    """`data` as a bgzf file (what ANGSD writes its .mafs.gz with): independent gzip members of at most 64 KiB,
    each with the "BC" extra field giving its size, then the empty end-of-file member."""
    import struct
    import zlib
    out = bytearray()
    chunks = [data[o: o + block] for o in range(0, len(data), block)] + [b""]
    for c in chunks:
        z = zlib.compressobj(level, zlib.DEFLATED, -15)
        d = z.compress(c) + z.flush()
        out += b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", 18 + len(d) + 8 - 1)
        out += d + struct.pack("<II", zlib.crc32(c) & 0xFFFFFFFF, len(c))
    with open(path, "wb") as fh:
        fh.write(bytes(out))


def dxy_ref_cases(tmp_path, plain_text=False):
    """The reference-made dxyWindow fixtures (tests/golden/ref_dxy.json: written by make_golden.py only on an image
    with the real Boost.Iostreams, where oracle/Makefile builds the unmodified dxyWindow.cpp).  None when absent —
    the dxy oracle is then "parity unpinned".  Yields (case, argv tail with real paths, options dict); with
    plain_text the gzip members are inflated first (the oracle's text front end reads plain text only)."""
    import base64
    import gzip
    path = os.path.join(GOLDEN, "ref_dxy.json")
    if not os.path.exists(path):
        return None
    out = []
    for k, c in enumerate(json.load(open(path))["cases"]):
        d = tmp_path / f"dxyref{k}"
        d.mkdir()
        names = {}
        for name, b64 in c["files"].items():
            raw = base64.b64decode(b64)
            if plain_text and name.endswith(".gz"):
                raw, new = gzip.decompress(raw), name[:-3]
            else:
                new = name
            (d / new).write_bytes(raw)
            names["@" + name] = str(d / new)
        argv = [names.get(a, a) for a in c["args"]]
        opt = {"winsize": 0, "stepsize": 0, "minind": 1, "fixedsite": 0, "skip_missing": 0, "sizefile": None}
        for i in range(0, len(argv) - 2, 2):
            key = argv[i].lstrip("-")
            opt[key] = argv[i + 1] if key == "sizefile" else int(argv[i + 1])
        opt["maf1"], opt["maf2"] = argv[-2], argv[-1]
        out.append((c, argv, opt))
    return out


def small_step_cases():
    """tests/golden/ref_small_step.json: reference-made runs with S << W whose inputs are regenerated from a seed
    (tests/golden/make_golden.py: small_step_input).  Yields (case, columns dict, fst text, het text) per distinct input."""
    import sys
    sys.path.insert(0, GOLDEN)
    import make_golden
    cases = load_golden("ref_small_step.json")["cases"]
    seen = {}
    for c in cases:
        key = (c["seed"], c["n"], c["n_chr"])
        if key not in seen:
            chr_ids, pos, a, b, g, fst, het = make_golden.small_step_input(*key)
            seen[key] = {"chr_ids": chr_ids, "pos": pos, "a": a, "b": b, "g": g, "fst": fst, "het": het}
        yield c, seen[key]


def write_hand_walked_case(case, header, tmp_path):
    """Materialise one case of tests/golden/dxy_hand_walked.json -> (maf1 path, maf2 path, size file path or None)."""
    d = tmp_path / case["name"]
    d.mkdir(exist_ok=True)
    paths = []
    for key in ("pop1", "pop2"):
        p = d / (key + ".mafs")
        with open(p, "w") as f:
            f.write(header + "\n")
            for c, pos, fr, n in case[key]:
                f.write(f"{c}\t{pos}\tA\tC\tA\t{fr:.6f}\t{n}\n")
        paths.append(str(p))
    sz = None
    if case["sizes"]:
        sz = d / "sizes.txt"
        sz.write_text("".join(f"{c}\t{n}\n" for c, n in case["sizes"]))
        sz = str(sz)
    return paths[0], paths[1], sz


def hand_walked_product_expectation(case, run):
    """What the GPU host / C-ABI must give for a run of a hand-walked case: the reference's output, or — for the
    documented deliberate divergences — the output of the machine on the sites both files list."""
    if case["product"] == "same":
        return run["stdout"], run["stderr"]
    if case["product"].get("rc", 0) != 0:  # the product refuses this input (exit status, a phrase of the message); nothing on stdout
        return None, case["product"]["stderr_contains"]
    return case["product"]["stdout"], case["product"]["stderr"]
