#!/usr/bin/env python3
"""Interleaved A/B of the straight build kernel's knobs (fst_build_straight_kernel) in ONE process (cdna_hip_programming.md §5.4 rule 24).

Variants are selected through the PGT_TUNE_BUILD_* environment variables that csrc/pgt_kernels.hip
reads at launch time.  Prints median / min kernel time (HIP events around the build pass) per
variant and size, as markdown.   usage: python tools/tune_build.py [rounds]
Needs libpgtwin.so built with -DPGT_TUNING_BUILD (PGT_EXTRA_HIPCC_FLAGS=-DPGT_TUNING_BUILD python -m
popgenomicstools_amd.build --force); the product build carries only the chosen variant.
"""
import itertools
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import popgenomicstools_amd as pgt  # noqa: E402
from popgenomicstools_amd._lib import FST_ROW_DTYPE, PGT_STAT_FST  # noqa: E402
from popgenomicstools_amd.window_scan import windows_to_device  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    dev = torch.device("cuda", 0)
    n = 1_000_000_000
    pos, a, b, run_len = bench.synth_columns(n, 40, 12345, dev)
    ctx = pgt.Context(0)
    tree = torch.empty(ctx.tree_bytes(PGT_STAT_FST, n), dtype=torch.uint8, device=dev)
    sizes = {}
    for m, chroms in ((1_000_000_000, 40), (100_000_000, 20)):
        rl = np.full(chroms, m // chroms, dtype=np.uint64)
        win = windows_to_device(pgt.build_windows_sites(rl, 50_000, 10_000), dev)
        out = torch.empty(win.numel() // 32 * FST_ROW_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        sizes[m] = (win, out)
    variants = [dict(blocks=bl, unroll=u, nt=nt)
                for bl, u, nt in itertools.product((1024, 2048, 4096, 0), (4, 8), (0, 1))]
    ctx.set_profiling(True)
    times = {(m, i): [] for m in sizes for i in range(len(variants))}
    for r in range(rounds + 1):
        for m, (win, out) in sizes.items():
            for i, v in enumerate(variants):
                os.environ["PGT_TUNE_BUILD_STRAIGHT"] = "1"  # these knobs belong to the straight (first) kernel
                os.environ["PGT_TUNE_BUILD_BLOCKS"] = str(v["blocks"])
                os.environ["PGT_TUNE_BUILD_UNROLL"] = str(v["unroll"])
                os.environ["PGT_TUNE_BUILD_NT"] = str(v["nt"])
                ctx.fst_reduce_dev(pos[:m], a[:m], b[:m], win, out=out, tree=tree)
                bm, qm = ctx.last_kernel_ms()
                if r > 0:  # round 0 is warm-up
                    times[(m, i)].append(bm)
    print(f"| sites | blocks cap | unroll (x2 loads in flight) | nt | median ms | min ms | median GB/s | % of 8 TB/s |")
    print("|---|---|---|---|---|---|---|---|")
    for m in sizes:
        for i, v in enumerate(variants):
            t = np.array(times[(m, i)])
            med = float(np.median(t))
            print(f"| {m:.0e} | {v['blocks'] or 'none'} | {v['unroll']} | {v['nt']} | {med:.4f} | {t.min():.4f} | "
                  f"{16.0 * m / med / 1e6:.0f} | {16.0 * m / med / 1e6 / 80:.1f} |")
    ctx.close()


if __name__ == "__main__":
    main()
