#!/usr/bin/env python3
"""Where the device parser starts to pay: fstWindow end to end (wall time of the process) with the host parser, with the
device parser alone and with the default (from 2 GiB of text on the head goes to the host parser beside HIP start-up,
the tail to the GPU), alternating, at several table sizes.  The host parser runs beside HIP start-up and uploads 20 B of columns per line; the
device parser has to wait for HIP and uploads the ~33 B of text per line.  Markdown on stdout.
usage: python tests/ingest_crossover.py [lines ...]"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))  # this script lives in tests/: it writes its tables with the oracle's text writer
import oracle_bind  # noqa: E402
import synth  # noqa: E402


def main():
    sizes = [int(float(x)) for x in sys.argv[1:]] or [10_000_000, 30_000_000, 60_000_000, 100_000_000]
    orc = oracle_bind.load()
    exe = os.path.join(ROOT, "popgenomicstools_amd", "bin", "fstWindow")
    d = tempfile.mkdtemp(prefix="pgt_cross_")
    f = os.path.join(d, "fst.txt")
    print("| lines | text MB | host parser: wall s | in-process ms | device parser alone: wall s | in-process ms | default (from 2 GiB on: head on the host, tail on the GPU): wall s | in-process ms |")
    print("|---|---|---|---|---|---|---|---|")
    for n in sizes:
        rng = np.random.default_rng(5)
        chr_ids, pos = synth.chromosomes(rng, n, 20)
        a, b = synth.fst_columns(rng, n)
        orc.write_fst_text(f, chr_ids, pos, a, b)
        del chr_ids, pos, a, b
        envs = {"0": {"PGT_GPU_INGEST": "0"}, "1": {"PGT_GPU_INGEST": "1", "PGT_HYBRID_HOST_BYTES": "0"}, "d": {}}
        res = {k: [] for k in envs}
        for rep in range(6):
            for m in envs:
                t = time.perf_counter()
                r = subprocess.run([exe, f, "50000", "10000"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                                   env=dict(os.environ, PGT_HOST_TIMING="1", **envs[m]))
                dt = time.perf_counter() - t
                tot = [float(ln.split()[-2]) for ln in r.stderr.decode().splitlines() if " total " in ln][0]
                if rep:
                    res[m].append((dt, tot))
        med = {m: (np.median([x[0] for x in v]), np.median([x[1] for x in v])) for m, v in res.items()}
        print(f"| {n:.0e} | {os.path.getsize(f) / 1e6:.0f} | {med['0'][0]:.3f} | {med['0'][1]:.0f} | {med['1'][0]:.3f} | {med['1'][1]:.0f} | "
              f"{med['d'][0]:.3f} | {med['d'][1]:.0f} |", flush=True)
    os.unlink(f)
    os.rmdir(d)


if __name__ == "__main__":
    main()
