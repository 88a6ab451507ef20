#!/usr/bin/env python3
"""fst build kernel time against the number of sites: t = t0 + bytes / BW.  Shows the fixed cost per
launch that makes 10^8-site runs less efficient than 10^9-site ones.  Markdown on stdout."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import popgenomicstools_amd as pgt  # noqa: E402
from popgenomicstools_amd.window_scan import windows_to_device  # noqa: E402
from synth_genome import SynthGenome  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    nmax = 1_000_000_000
    g = SynthGenome(1, nmax, 40)
    pos, a, b = g.fst_columns_t(0, nmax, dev)
    ctx = pgt.Context(0)
    ctx.set_max_window(50_000)
    ctx.set_profiling(True)
    tree = torch.empty(ctx.tree_bytes(0, nmax), dtype=torch.uint8, device=dev)
    sizes = [int(float(x)) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else
                                     "1e6,1e7,2e7,5e7,1e8,1.25e8,2e8,5e8,1e9".split(","))]
    # tuning build only: the product kernel template with other parameters, "stage:loads in flight:workgroups" (PGT_TUNE_FST)
    variants = os.environ.get("SWEEP_VARIANTS", "").split(",") if os.environ.get("SWEEP_VARIANTS") else [""]
    print("| variant | sites | build ms (median of 15) | GB/s | % of 8 TB/s |")
    print("|---|---|---|---|---|")
    fit = {v: ([], []) for v in variants}
    for n in sizes:
        win = windows_to_device(pgt.build_windows_sites(np.array([n], dtype=np.uint64), 50_000, 10_000), dev)
        t = {v: [] for v in variants}
        for r in range(18):
            for v in variants:
                if v:
                    os.environ["PGT_TUNE_FST"] = v
                ctx.fst_reduce_dev(pos[:n], a[:n], b[:n], win, tree=tree)
                bm, _ = ctx.last_kernel_ms()
                if r >= 3:
                    t[v].append(bm)
        for v in variants:
            med = float(np.median(t[v]))
            fit[v][0].append(16.0 * n)
            fit[v][1].append(med * 1e-3)
            print(f"| {v or 'product'} | {n:.3g} | {med:.4f} | {16 * n / med / 1e6:.0f} | {16 * n / med / 1e6 / 80:.1f} |", flush=True)
    for v in variants:
        xs, ys = fit[v]
        A = np.vstack([np.ones(len(xs)), xs]).T
        t0, inv_bw = np.linalg.lstsq(A, np.array(ys), rcond=None)[0]
        print(f"\n{v or 'product'}: least squares t = {t0 * 1e6:.1f} us + bytes / {1 / inv_bw / 1e12:.2f} TB/s")


if __name__ == "__main__":
    main()
