#!/usr/bin/env python3
"""Randomised differential fuzzer of the device-side ingest at CLI level: random fst / het / MAF / selscan *.norm files (random
separators, number formats, CRLF, blank-line stops, missing newline at the end, a bad line now and then) through
the hosts with PGT_GPU_INGEST=1 and =0 — once more with two or three contexts (PGT_DEVICES=0,0[,0]: the multi-GPU paths, both
parsers), and in passes (PGT_MAX_RESIDENT_SITES): stdout, stderr and exit code must be identical.  ihsWindow / xpehhWindow (round 4):
well-formed files additionally against the unmodified reference binaries (oracle/_ref/), the refusal beyond the resident limit.
usage: python tests/ingest_fuzz.py [seconds] [seed]"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "popgenomicstools_amd", "bin")


def fmt_float(rng, x):
    k = rng.integers(0, 10)
    if k < 5:
        return "%.6f" % x
    if k == 5:
        return "%.3e" % x
    if k == 6:
        return "%.17g" % x
    if k == 7:
        return ("+" if x >= 0 else "") + "%.4f" % x
    if k == 8:
        return "%g" % x
    return "%.10f" % x


def make_file(rng, path, kind):
    n = int(rng.choice([rng.integers(1, 50), rng.integers(50, 5000), rng.integers(5000, 200000)]))
    sep = ["\t", " ", "  ", " \t"][int(rng.integers(0, 4))]
    eol = "\r\n" if rng.random() < 0.2 else "\n"
    n_chr = int(rng.integers(1, 6))
    chrs = np.sort(rng.integers(0, n_chr, n))
    pos = np.cumsum(rng.integers(1, 60, n))
    bad_at = int(rng.integers(0, n)) if rng.random() < 0.15 else -1
    blank_at = int(rng.integers(0, n)) if rng.random() < 0.15 else -1
    lines = []
    if kind == "maf":
        lines.append("chromo\tposition\tmajor\tminor\tref\tknownEM\tnInd")
    if kind == "xpehh":
        lines.append("id\tpos\tgpos\tp1\tihh1\tp2\tihh2\txpehh\tnormxpehh\tcrit")
    id_style = int(rng.integers(0, 3))  # selscan locus ids: chr_pos, chr_pos_tag, or (now and then) an id without '_
    for i in range(n):
        if i == blank_at:
            lines.append(" " if rng.random() < 0.5 else "")
        if kind == "fst":
            toks = ["c%d" % chrs[i], str(pos[i]), fmt_float(rng, rng.normal(0, 0.05)), fmt_float(rng, rng.random() * 0.3)]
        elif kind == "het":
            toks = ["c%d" % chrs[i], str(pos[i]), str(int(rng.choice([0, 1, 2, -1, 3, -9])))]
        elif kind in ("ihs", "xpehh"):  # <chr>_<id> pos f0 .. ; the score is numeric field 4 (iHS) / 6 (XP-EHH) behind the position
            lid = "c%d" % chrs[i] if id_style == 2 else ("c%d_%d" % (chrs[i], pos[i]) if id_style == 0 else "c%d_%d_x%d" % (chrs[i], pos[i], i % 7))
            score = float(np.round(rng.normal(0, 1.3), 4)) if rng.random() < 0.97 else 2.5  # ties now and then
            nf = 6 if kind == "ihs" else 8
            fields = [fmt_float(rng, rng.random()) for _ in range(nf)]
            fields[4 if kind == "ihs" else 6] = fmt_float(rng, score)
            toks = [lid, str(pos[i])] + fields
        else:
            toks = ["c%d" % chrs[i], str(pos[i]), "A", "C", "A", fmt_float(rng, rng.random()), str(int(rng.integers(0, 21)))]
        if i == bad_at:
            j = int(rng.integers(1, len(toks)))
            toks[j] = ["x", "1e", "--1", "", "0x1", "1e999"][int(rng.integers(0, 6))]
        if rng.random() < 0.02:
            toks.append("extra")
        lines.append(sep.join(toks))
    text = eol.join(lines) + (eol if rng.random() < 0.85 else "")
    with open(path, "w", newline="") as f:
        f.write(text)
    make_file.well_formed = bad_at < 0 and blank_at < 0 and eol == "\n" and text.endswith("\n")
    return n


def keep_failure(d, cmd, runs):
    """the inputs and every run's output of a mismatch -> gpurun_out/fuzz_fail/ (comes back from the GPU box)"""
    import shutil
    out = os.path.join(ROOT, "gpurun_out", "fuzz_fail")
    os.makedirs(out, exist_ok=True)
    for fn in os.listdir(d):
        if os.path.getsize(os.path.join(d, fn)) < (48 << 20):
            shutil.copy(os.path.join(d, fn), out)
    with open(os.path.join(out, "cmd.txt"), "w") as fh:
        fh.write(" ".join(cmd) + "\n")
    for name, r in runs.items():
        with open(os.path.join(out, name + ".stdout"), "wb") as fh:
            fh.write(r.stdout)
        with open(os.path.join(out, name + ".stderr"), "wb") as fh:
            fh.write(r.stderr)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    d = tempfile.mkdtemp(prefix="pgt_ingest_fuzz_")
    t0, trials, last = time.time(), 0, time.time()
    counts = {"fst": 0, "het": 0, "maf": 0, "ihs": 0, "xpehh": 0, "errors": 0}
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import oracle_bind  # the compiled reference binaries (oracle/_ref/, where they travelled): the live check of the extreme-score hosts
    while time.time() - t0 < budget:
        trials += 1
        if time.time() - last > 30:
            print(f"  ... {trials} trials, {time.time() - t0:.0f} s", flush=True)
            last = time.time()
        kind = ["fst", "het", "maf", "ihs", "xpehh"][int(rng.integers(0, 5))]
        W = int(rng.integers(1, 400))
        S = int(rng.integers(1, W + 1))
        if kind == "maf":
            f1, f2 = os.path.join(d, "p1.mafs"), os.path.join(d, "p2.mafs")
            state = rng.bit_generator.state
            make_file(rng, f1, "maf")
            if rng.random() < 0.7:  # same sites, other frequencies
                rng2 = np.random.default_rng(int(rng.integers(0, 2**31)))
                rng.bit_generator.state = state
                make_file(rng, f2, "maf")
                del rng2
            else:
                make_file(rng, f2, "maf")
            mode = int(rng.integers(0, 4))  # fixed-site windows, base-pair windows (two sizes of them), the global line only
            opts = ["-minind", str(int(rng.integers(1, 12))), "-skip_missing", str(int(rng.integers(0, 2)))]
            if mode == 0:
                opts += ["-winsize", str(W), "-stepsize", str(S), "-fixedsite", "1"]
            elif mode == 3:
                opts += ["-winsize", "0", "-fixedsite", "1"]
            else:
                fs = os.path.join(d, "sizes.txt")
                with open(fs, "w") as fh:  # every chromosome name the generator can produce, longer than any position it writes
                    fh.write("".join("c%d\t%d\n" % (c, 200000 * 60 + 1000 + c) for c in range(6)))
                Wb = int(rng.choice([W * 30, W * 3000]))
                opts += ["-winsize", str(Wb), "-stepsize", str(max(1, Wb // int(rng.integers(1, 6)))), "-sizefile", fs]
            cmd = [os.path.join(BIN, "dxyWindow")] + opts + [f1, f2]
        elif kind in ("ihs", "xpehh"):
            f = os.path.join(d, kind + ".norm")
            make_file(rng, f, kind)
            Wb, cutoff = int(rng.choice([W * 30, W * 3000])), "%.2f" % (rng.random() * 3)
            cmd = ([os.path.join(BIN, "ihsWindow"), f, "-winsize", str(Wb), "-cutoff", cutoff] if kind == "ihs" else
                   [os.path.join(BIN, "xpehhWindow"), f, cutoff, "-winsize", str(Wb)])
            ref = oracle_bind.ref_binary(kind + "Window")
            if ref and make_file.well_formed:  # the unmodified reference on the same file: byte-identical stdout
                r0 = subprocess.run([ref] + cmd[1:], capture_output=True, timeout=120)
                h0 = subprocess.run(cmd, capture_output=True, env=dict(os.environ, PGT_GPU_INGEST="1"), timeout=120)
                if r0.returncode == 0 and (h0.returncode, h0.stdout) != (0, r0.stdout):
                    print("MISMATCH against the reference binary", cmd, h0.returncode, h0.stderr[-200:], "files kept in", d)
                    keep_failure(d, cmd, {"reference": r0, "host_device_parser": h0})
                    sys.exit(1)
                counts["vs_reference"] = counts.get("vs_reference", 0) + 1
        else:
            f = os.path.join(d, kind + ".txt")
            make_file(rng, f, kind)
            cmd = [os.path.join(BIN, kind + "Window"), f, str(W), str(S)]
        res = [subprocess.run(cmd, capture_output=True, env=dict(os.environ, PGT_GPU_INGEST=m), timeout=120) for m in ("1", "0")]
        a, b = res
        # several contexts on the one GPU: fst / het: the text cut at line starts, shards gathered from the pieces; dxy: one
        # file per context, the genome-wide line from 65536-site blocks
        devs = "0,0" if rng.random() < 0.5 else "0,0,0"
        for m in ("1", "0"):
            c = subprocess.run(cmd, capture_output=True, env=dict(os.environ, PGT_GPU_INGEST=m, PGT_DEVICES=devs), timeout=120)
            if (c.returncode, c.stdout, c.stderr) != (a.returncode, a.stdout, a.stderr):
                print("MISMATCH (PGT_DEVICES=%s, PGT_GPU_INGEST=%s)" % (devs, m), cmd, a.returncode, c.returncode, a.stderr[-200:], c.stderr[-200:],
                      "files kept in", d)
                again = subprocess.run(cmd, capture_output=True, env=dict(os.environ, PGT_GPU_INGEST=m, PGT_DEVICES=devs), timeout=120)
                print("the same run again:", "as before" if again.stdout == c.stdout else ("as the single-device run" if again.stdout == a.stdout else "a third output"))
                keep_failure(d, cmd, {"single": a, "multi": c, "multi_again": again})
                sys.exit(1)
        counts["multi"] = counts.get("multi", 0) + 1
        if kind != "maf":  # the head of the text on the host parser, the tail on the GPU, the columns joined there
            cut = int(rng.integers(1, max(2, os.path.getsize(f) + 2)))
            c = subprocess.run(cmd, capture_output=True, env=dict(os.environ, PGT_GPU_INGEST="1", PGT_HYBRID_HOST_BYTES=str(cut)), timeout=120)
            if (c.returncode, c.stdout, c.stderr) != (a.returncode, a.stdout, a.stderr):
                print("MISMATCH (PGT_HYBRID_HOST_BYTES=%d)" % cut, cmd, a.returncode, c.returncode, a.stderr[-200:], c.stderr[-200:], "files kept in", d)
                keep_failure(d, cmd, {"single": a, "hybrid": c})
                sys.exit(1)
            counts["hybrid"] = counts.get("hybrid", 0) + 1
        if kind in ("ihs", "xpehh"):  # no passes mode: an input REALLY beyond the per-GPU limit is refused, nothing printed; one under it runs as ever
            with open(f, "rb") as fh:
                data = fh.read()
            n_lines = data.count(b"\n") + (1 if data and not data.endswith(b"\n") else 0) - (1 if kind == "xpehh" else 0)  # (xpehh: the header line is not a SNP)
            for limit, gpus in ((max(n_lines - 1, 1), 1), (n_lines, 1), (n_lines + 5, 1), ((n_lines + 1) // 2, 2), (max((n_lines - 1) // 2, 1), 2)):
                env = dict(os.environ, PGT_MAX_RESIDENT_SITES=str(limit))
                if gpus == 2:
                    env["PGT_DEVICES"] = "0,0"
                c = subprocess.run(cmd, capture_output=True, env=env, timeout=120)
                if n_lines > limit * gpus:
                    ok = c.returncode == 255 and c.stdout == b"" and b"no passes mode" in c.stderr and b"PGT_MAX_RESIDENT_SITES=" in c.stderr
                else:
                    ok = (c.returncode, c.stdout, c.stderr) == (a.returncode, a.stdout, a.stderr)
                if not ok:
                    print("MISMATCH (resident limit %d x %d GPUs, %d lines)" % (limit, gpus, n_lines), cmd, c.returncode, c.stderr[-200:], "files kept in", d)
                    sys.exit(1)
        elif True:  # in passes (as for a table larger than the GPU): same rows; a bad line ends the run after the earlier blocks' rows
            limit = int(rng.choice([1, 70000, 150000]))
            env = dict(os.environ, PGT_MAX_RESIDENT_SITES=str(limit))
            if rng.random() < 0.5:
                env["PGT_DEVICES"] = devs  # fst / het: the blocks go round the contexts (dxyWindow: the first one)
            c = subprocess.run(cmd, capture_output=True, env=env, timeout=120)
            ok = (c.returncode, c.stderr) == (a.returncode, a.stderr) and (c.stdout == a.stdout if a.returncode == 0 else True)
            if not ok:
                print("MISMATCH (PGT_MAX_RESIDENT_SITES=%d)" % limit, cmd, a.returncode, c.returncode, a.stderr[-200:], c.stderr[-200:], "files kept in", d)
                sys.exit(1)
            counts["passes"] = counts.get("passes", 0) + 1
        if (a.returncode, a.stdout, a.stderr) != (b.returncode, b.stdout, b.stderr):
            keep = os.path.join(d, "FAILED")
            print("MISMATCH", cmd, a.returncode, b.returncode, a.stderr[-200:], b.stderr[-200:], "files kept in", d)
            sys.exit(1)
        counts[kind] += 1
        counts["errors"] += a.returncode != 0
    print(f"ingest_fuzz: {trials} trials in {time.time() - t0:.0f} s, device ingest == host parser everywhere: {counts}")


if __name__ == "__main__":
    main()
