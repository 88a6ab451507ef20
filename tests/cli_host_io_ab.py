#!/usr/bin/env python3
"""End-to-end wall time of the hosts' HOST-PARSER path with and without round 6's host-I/O preparation, alternating on one box
(markdown on stdout):  A = PGT_UPLOAD=plain PGT_PREPARE_HOST_IO=0 (round 5: hipMemcpy from the parser's pageable columns, every
first-use cost inside "gpu reduce"),  B = the default (pinned staging ring + first-copy set-up on the thread that opens the device,
beside the parse).      python tests/cli_host_io_ab.py [sites=1e8] [runs=7]"""
import os
import statistics
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_bind  # noqa: E402
import synth  # noqa: E402

BIN = os.path.join(ROOT, "popgenomicstools_amd", "bin")


def phases(stderr):
    out = {}
    for ln in stderr.decode().splitlines():
        if "[pgt-host]" in ln:
            body = ln.replace("[pgt-host]", "").strip()
            name, ms = body.rsplit(None, 2)[0].strip(), body.rsplit(None, 2)[1]
            try:
                out[name] = out.get(name, 0.0) + float(ms)
            except ValueError:
                pass
    return out


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
    runs = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    orc = oracle_bind.load()
    rng = np.random.default_rng(5)
    chr_ids, pos = synth.chromosomes(rng, n, 20)
    d = tempfile.mkdtemp(prefix="pgt_ioab_")
    f_fst, f_het = os.path.join(d, "fst.txt"), os.path.join(d, "het.txt")
    a, b = synth.fst_columns(rng, n)
    orc.write_fst_text(f_fst, chr_ids, pos, a, b)
    del a, b
    orc.write_het_text(f_het, chr_ids, pos, synth.het_column(rng, n))
    nd = n // 5
    p1, p2, n1, n2 = synth.dxy_columns(rng, nd)
    m1, m2 = os.path.join(d, "p1.mafs"), os.path.join(d, "p2.mafs")
    orc.write_maf_text(m1, chr_ids[:nd], pos[:nd], p1, n1)
    orc.write_maf_text(m2, chr_ids[:nd], pos[:nd], p2, n2)
    jobs = [("hetWindow", [os.path.join(BIN, "hetWindow"), f_het, "50000", "10000"]),
            ("dxyWindow -fixedsite 1", [os.path.join(BIN, "dxyWindow"), "-winsize", "50000", "-stepsize", "10000", "-minind", "5", "-fixedsite", "1", m1, m2]),
            ("fstWindow", [os.path.join(BIN, "fstWindow"), f_fst, "50000", "10000"])]
    base = dict(os.environ, PGT_HOST_TIMING="1", PGT_GPU_INGEST="0")
    envs = {"A (round 5: plain hipMemcpy, set-up inside the reduce)": dict(base, PGT_UPLOAD="plain", PGT_PREPARE_HOST_IO="0"),
            "B (round 6: staging ring, set-up beside the parse)": base}
    print(f"{n:.0e} sites (dxy: {nd:.0e} x 2 files), host parser, {runs} alternating runs each, medians; this box has {os.cpu_count()} logical cores\n")
    print("| tool | variant | wall ms | total ms (host timer) | parse ms | wait for HIP ms | gpu reduce ms | stdout identical |")
    print("|---|---|---|---|---|---|---|---|")
    for tool, cmd in jobs:
        subprocess.run(cmd, capture_output=True, env=base)  # page cache
        res = {k: [] for k in envs}
        outs = set()
        for _ in range(runs):
            for k, env in envs.items():
                t = time.perf_counter()
                r = subprocess.run(cmd, capture_output=True, env=env)
                w = (time.perf_counter() - t) * 1e3
                assert r.returncode == 0, r.stderr[-500:]
                outs.add(r.stdout)
                ph = phases(r.stderr)
                res[k].append((w, ph.get("total", float("nan")), ph.get("parse", float("nan")), ph.get("wait for HIP", float("nan")), ph.get("gpu reduce", float("nan"))))
        for k, v in res.items():
            med = [statistics.median(x[i] for x in v) for i in range(5)]
            print(f"| {tool} | {k} | {med[0]:.1f} | {med[1]:.1f} | {med[2]:.1f} | {med[3]:.1f} | {med[4]:.1f} | {len(outs) == 1} |")
    for f in (f_fst, f_het, m1, m2):
        os.unlink(f)
    os.rmdir(d)


if __name__ == "__main__":
    main()
