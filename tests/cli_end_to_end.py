#!/usr/bin/env python3
"""End-to-end wall time of the retained hosts vs the unmodified reference binaries on the same text
input (parse + window scan + TSV), on this box.  Markdown on stdout."""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))  # this script lives in tests/: it uses the oracle
import oracle_bind  # noqa: E402
import synth  # noqa: E402

BIN = os.path.join(ROOT, "popgenomicstools_amd", "bin")


def wall(cmd, env=None):
    t = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, env=env)
    return time.perf_counter() - t, r


def bgzf_copy(src, dst, block=0xff00):
    """src as a bgzf file (members of <= 64 KiB with the BC size field + the empty end member); 8 threads (zlib drops the GIL)"""
    import struct
    import zlib
    from concurrent.futures import ThreadPoolExecutor

    def member(c):
        z = zlib.compressobj(6, zlib.DEFLATED, -15)
        d = z.compress(c) + z.flush()
        return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", 18 + len(d) + 8 - 1) + d +
                struct.pack("<II", zlib.crc32(c) & 0xFFFFFFFF, len(c)))
    with open(src, "rb") as fi, open(dst, "wb") as fo, ThreadPoolExecutor(8) as ex:
        while True:
            big = fi.read(block * 2048)
            if not big:
                break
            for m in ex.map(member, [big[o: o + block] for o in range(0, len(big), block)]):
                fo.write(m)
        fo.write(member(b""))


def gzip_copy(src, dst):
    import zlib
    z = zlib.compressobj(6, zlib.DEFLATED, 31)
    with open(src, "rb") as fi, open(dst, "wb") as fo:
        while True:
            big = fi.read(1 << 24)
            if not big:
                break
            fo.write(z.compress(big))
        fo.write(z.flush())


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
    orc = oracle_bind.load()
    rng = np.random.default_rng(5)
    chr_ids, pos = synth.chromosomes(rng, n, 20)
    a, b = synth.fst_columns(rng, n)
    g = synth.het_column(rng, n)
    d = tempfile.mkdtemp(prefix="pgt_e2e_")
    f_fst, f_het = os.path.join(d, "fst.txt"), os.path.join(d, "het.txt")
    orc.write_fst_text(f_fst, chr_ids, pos, a, b)
    orc.write_het_text(f_het, chr_ids, pos, g)
    print(f"{n:.0e} sites, W=50000 S=10000, host has {os.cpu_count()} logical cores\n")
    print("| tool | reference (1 thread) s | this host s | speed-up | rows | TSV identical | host phases |")
    print("|---|---|---|---|---|---|---|")
    for tool, path in (("fstWindow", f_fst), ("hetWindow", f_het)):
        env = dict(os.environ, PGT_HOST_TIMING="1")
        wall([os.path.join(BIN, tool), path, "50000", "10000"], env)  # warm the page cache / GPU runtime
        # first the host parser (PGT_GPU_INGEST=0), then the default: text parsed on the GPU above 8 MiB
        t_h, r_h = wall([os.path.join(BIN, tool), path, "50000", "10000"], dict(env, PGT_GPU_INGEST="0"))
        ph = "; ".join(ln.replace("[pgt-host]", "").strip() for ln in r_h.stderr.decode().splitlines() if "pgt-host" in ln)
        t_new, r_new = wall([os.path.join(BIN, tool), path, "50000", "10000"], env)
        print(f"| {tool}, host parser (PGT_GPU_INGEST=0) | | {t_h:.2f} | | {len(r_h.stdout.splitlines())} | {r_h.stdout == r_new.stdout} (vs device parse) | {ph} |")
        ref = oracle_bind.ref_binary(tool)
        if ref:
            t_ref, r_ref = wall([ref, path, "50000", "10000"])
            same = r_ref.stdout == r_new.stdout
            t_ref_s, sp = f"{t_ref:.2f}", f"{t_ref / t_new:.1f}x"
        else:
            same, t_ref_s, sp = "n/a", "n/a", "n/a"
        phases = "; ".join(ln.replace("[pgt-host]", "").strip() for ln in r_new.stderr.decode().splitlines() if "pgt-host" in ln)
        print(f"| {tool} | {t_ref_s} | {t_new:.2f} | {sp} | {len(r_new.stdout.splitlines())} | {same} | {phases} |")
        # the same run in passes (PGT_MAX_RESIDENT_SITES): how an input larger than the GPU's memory is reduced — an eighth of
        # the table resident at a time, rows printed block by block
        t_p, r_p = wall([os.path.join(BIN, tool), path, "50000", "10000"], dict(env, PGT_MAX_RESIDENT_SITES=str(max(n // 8, 1))))
        phases = "; ".join(ln.replace("[pgt-host]", "").strip() for ln in r_p.stderr.decode().splitlines() if "pgt-host" in ln)
        print(f"| {tool}, in passes of <= {max(n // 8, 1):.3g} resident sites (PGT_MAX_RESIDENT_SITES) | {t_ref_s} | {t_p:.2f} | "
              f"{(t_ref / t_p if ref else 0):.1f}x | {len(r_p.stdout.splitlines())} | {r_p.stdout == r_new.stdout} | {phases} |")
        # the same run with the binary column cache (PGT_COLUMN_CACHE): first run writes it, second maps it
        cdir = os.path.join(d, "cache_" + tool)
        os.mkdir(cdir)
        cenv = dict(env, PGT_COLUMN_CACHE=cdir)
        t_w, r_w = wall([os.path.join(BIN, tool), path, "50000", "10000"], cenv)
        t_c, r_c = wall([os.path.join(BIN, tool), path, "50000", "10000"], cenv)
        phases = "; ".join(ln.replace("[pgt-host]", "").strip() for ln in r_c.stderr.decode().splitlines() if "pgt-host" in ln)
        print(f"| {tool}, PGT_COLUMN_CACHE (2nd run; the 1st, writing the cache, took {t_w:.2f} s) | {t_ref_s} | {t_c:.2f} | "
              f"{(t_ref / t_c if ref else 0):.1f}x | {len(r_c.stdout.splitlines())} | {r_c.stdout == r_new.stdout} | {phases} |")
        for fn in os.listdir(cdir):
            os.unlink(os.path.join(cdir, fn))
        os.rmdir(cdir)
    # dxyWindow: the reference source needs Boost (absent here), so the CPU side of this row is the oracle's
    # text front end — our restatement of dxyWindow.cpp's streaming loop ("port"), single-threaded.
    nd = min(n, 20_000_000)  # two MAF files of ~35 B per line: bounded so that the scratch disk is enough
    full = (chr_ids, pos)
    chr_ids, pos = chr_ids[:nd], pos[:nd]
    p1, p2, n1, n2 = synth.dxy_columns(rng, nd)
    f_m1, f_m2, f_sz = os.path.join(d, "pop1.mafs"), os.path.join(d, "pop2.mafs"), os.path.join(d, "sizes.txt")
    orc.write_maf_text(f_m1, chr_ids, pos, p1, n1)
    orc.write_maf_text(f_m2, chr_ids, pos, p2, n2)
    lens = np.diff(np.concatenate(([0], np.flatnonzero(np.diff(chr_ids)) + 1, [nd])))
    ends = np.cumsum(lens) - 1
    with open(f_sz, "w") as fh:
        for c, e in enumerate(ends):
            fh.write(f"chr{c + 1}\t{int(pos[e]) + 1000}\n")
    for label, opts in (("dxyWindow -fixedsite 1 (50000/10000 sites)", ["-winsize", "50000", "-stepsize", "10000", "-minind", "5", "-fixedsite", "1"]),
                        ("dxyWindow bp (50 kb / 10 kb)", ["-winsize", "50000", "-stepsize", "10000", "-minind", "5", "-sizefile", f_sz])):
        env = dict(os.environ, PGT_HOST_TIMING="1")
        cmd = [os.path.join(BIN, "dxyWindow")] + opts + [f_m1, f_m2]
        wall(cmd, env)
        t_new, r_new = wall(cmd, env)
        o_out, o_err = os.path.join(d, "orc.out"), os.path.join(d, "orc.err")
        t0 = time.perf_counter()
        rc = orc.dxy_text(f_m1, f_m2, f_sz if "-sizefile" in opts else None, 50000, 10000, 5, 1 if "-fixedsite" in opts else 0, 0, o_out, o_err)
        t_ref = time.perf_counter() - t0
        assert rc == 0 and r_new.returncode == 0, (rc, r_new.returncode, r_new.stderr[-300:])
        ref_rows = [ln.split("\t") for ln in open(o_out).read().splitlines()]
        new_rows = [ln.split("\t") for ln in r_new.stdout.decode().splitlines()]
        same = len(ref_rows) == len(new_rows) and all(a_[:3] == b_[:3] and a_[4:] == b_[4:] and
                                                      abs(float(a_[3]) - float(b_[3])) <= 5.1e-6 * abs(float(a_[3])) + 1e-12
                                                      for a_, b_ in zip(ref_rows, new_rows))
        phases = "; ".join(ln.replace("[pgt-host]", "").strip() for ln in r_new.stderr.decode().splitlines() if "pgt-host" in ln)
        print(f"| {label}, {nd:.0e} sites x 2 files | {t_ref:.2f} (oracle port, not the reference binary) | {t_new:.2f} | {t_ref / t_new:.1f}x | {len(new_rows)} | "
              f"{same} (ints exact, sums to 6 digits) | {phases} |")
        os.unlink(o_out)
        os.unlink(o_err)
    # the same MAF pair gzipped: as bgzf (what ANGSD writes; inflated block-parallel) and as ordinary one-member gzip (gzread)
    opts = ["-winsize", "50000", "-stepsize", "10000", "-minind", "5", "-fixedsite", "1"]
    _, r_plain = wall([os.path.join(BIN, "dxyWindow")] + opts + [f_m1, f_m2])
    for label, writer in (("bgzf", bgzf_copy), ("one-member gzip", gzip_copy)):
        z1, z2 = f_m1 + ".gz", f_m2 + ".gz"
        writer(f_m1, z1)
        writer(f_m2, z2)
        env = dict(os.environ, PGT_HOST_TIMING="1")
        cmd = [os.path.join(BIN, "dxyWindow")] + opts + [z1, z2]
        wall(cmd, env)
        t_new, r_new = wall(cmd, env)
        phases = "; ".join(ln.replace("[pgt-host]", "").strip() for ln in r_new.stderr.decode().splitlines() if "pgt-host" in ln)
        print(f"| dxyWindow -fixedsite 1, both inputs {label} ({os.path.getsize(z1) / 1e6:.0f} + {os.path.getsize(z2) / 1e6:.0f} MB) | n/a | {t_new:.2f} | | "
              f"{len(r_new.stdout.splitlines())} | {r_new.stdout == r_plain.stdout} (vs the plain-text run) | {phases} |")
        os.unlink(z1)
        os.unlink(z2)
    for p_ in (f_m1, f_m2, f_sz):
        os.unlink(p_)
    chr_ids, pos = full
    # S = 1: one window per site.  The reference re-sums W entries per site (O(N*W)); it is timed on
    # a 10^5-site sample only.
    m, W1 = 2_000_000, 50_000
    f_s1, f_ref = os.path.join(d, "fst_s1.txt"), os.path.join(d, "fst_s1_ref.txt")
    orc.write_fst_text(f_s1, chr_ids[:m], pos[:m], a[:m], b[:m])
    orc.write_fst_text(f_ref, chr_ids[:100_000], pos[:100_000], a[:100_000], b[:100_000])
    env = dict(os.environ, PGT_HOST_TIMING="1")
    t_new, r_new = wall([os.path.join(BIN, "fstWindow"), f_s1, str(W1), "1"], env)
    phases = "; ".join(ln.replace("[pgt-host]", "").strip() for ln in r_new.stderr.decode().splitlines() if "pgt-host" in ln)
    print(f"\nS=1, W={W1}: this host, {m:.0e} sites -> {len(r_new.stdout.splitlines())} rows in {t_new:.2f} s "
          f"({m / t_new:.3e} sites/s; {phases})")
    ref = oracle_bind.ref_binary("fstWindow")
    if ref:
        t_ref, r_ref = wall([ref, f_ref, str(W1), "1"])
        print(f"S=1, W={W1}: reference, 1e5-site sample -> {len(r_ref.stdout.splitlines())} rows in {t_ref:.2f} s ({1e5 / t_ref:.3e} sites/s)")
    for p in (f_fst, f_het, f_s1, f_ref):
        os.unlink(p)
    os.rmdir(d)


if __name__ == "__main__":
    main()
