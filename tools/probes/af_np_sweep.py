#!/usr/bin/env python3
"""AF front end, 2 … 8 populations: build kernel with two waves per SIMD against one (PGT_AF_ONE_WAVE_FROM), each setting in a
process of its own (the knob is read once), same box, 10^8 sites.   python tools/probes/af_np_sweep.py [child <from>]"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child():
    import torch
    import popgenomicstools_amd as pgt
    from popgenomicstools_amd.window_scan import windows_to_device
    from synth_genome import SynthGenome
    dev = torch.device("cuda", 0)
    n = 100_000_000
    g = SynthGenome(12345, n, 20)
    win = windows_to_device(pgt.build_windows_sites(g.run_len, 50_000, 10_000), dev)
    pos = g.pos_t(0, n, dev)
    fr = [g.freq_t(k, 0, n, dev) for k in range(8)]
    ctx = pgt.Context(0)
    ctx.set_max_window(50_000)
    ctx.set_profiling(True)
    for NP in range(2, 9):
        tree = torch.empty(int(pgt._lib.load().pgt_af_tree_bytes(NP, n)), dtype=torch.uint8, device=dev)
        b, q = [], []
        for _ in range(12):
            ctx.fst_af_reduce_dev(pos, fr[:NP], [10.0 + k for k in range(NP)], win, tree=tree)
            x = ctx.last_kernel_ms()
            b.append(x[0])
            q.append(x[1])
        bm, qm = float(np.median(b[2:])), float(np.median(q[2:]))
        print(f"NP {NP} build_ms {bm:.4f} query_ms {qm:.4f} frac {8.0 * NP * n / bm / 8e9 * 1e-0 / 1e0 / 1e3:.4f}", flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        return child()
    res = {}
    for frm in ("9", "2"):
        for rep in range(2):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], capture_output=True, text=True,
                               env=dict(os.environ, PGT_AF_ONE_WAVE_FROM=frm), timeout=600)
            for ln in r.stdout.splitlines():
                if ln.startswith("NP "):
                    f = ln.split()
                    res.setdefault((int(f[1]), frm), []).append((float(f[3]), float(f[5])))
            if r.returncode:
                print(r.stderr[-800:])
    print("| populations | two waves per SIMD: build ms (% of 8 TB/s), query ms | one wave per SIMD: build ms (%), query ms |\n|---|---|---|")
    for NP in range(2, 9):
        cells = []
        for frm in ("9", "2"):
            v = res.get((NP, frm), [])
            if v:
                bm = min(x[0] for x in v)
                cells.append(f"{bm:.4f} ({8.0 * NP * 1e8 / (bm * 1e-3) / 8e12 * 100:.1f} %), {min(x[1] for x in v):.4f}")
            else:
                cells.append("-")
        print(f"| {NP} | {cells[0]} | {cells[1]} |")


if __name__ == "__main__":
    main()
