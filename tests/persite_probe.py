"""dxyWindow per site (`-winsize 1 -stepsize 1`, the mode dxyWindow.cpp:47 documents) end to end: 2x10^7 sites in two MAF
files, stdout to /dev/null, in-process phases.  Uses the oracle's text writer (tests/oracle_bind.py) to make the input."""
import os, subprocess, sys, tempfile, time
import numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import oracle_bind, synth
orc = oracle_bind.load()
n = 20_000_000
rng = np.random.default_rng(5)
chr_ids, pos = synth.chromosomes(rng, n, 20)
p1, p2, n1, n2 = synth.dxy_columns(rng, n)
d = tempfile.mkdtemp()
f1, f2 = d + "/p1.mafs", d + "/p2.mafs"
orc.write_maf_text(f1, chr_ids, pos, p1, n1); orc.write_maf_text(f2, chr_ids, pos, p2, n2)
exe = "/root/repo/popgenomicstools_amd/bin/dxyWindow"
for args in (["-winsize", "1", "-stepsize", "1", "-fixedsite", "1", "-minind", "5"], ["-winsize", "100", "-stepsize", "1", "-fixedsite", "1", "-minind", "5"]):
    for rep in range(2):
        t = time.perf_counter()
        r = subprocess.run([exe] + args + [f1, f2], stdout=open("/dev/null", "w"), stderr=subprocess.PIPE, env=dict(os.environ, PGT_HOST_TIMING="1"))
        dt = time.perf_counter() - t
    print(args, f"{dt:.2f} s", r.returncode)
    print("   ", "; ".join(l.replace("[pgt-host]", "").strip() for l in r.stderr.decode().splitlines() if "pgt-host" in l and "ingest:" not in l))
