#!/usr/bin/env python3
"""Timing-only ablation: fst_build_kernel with the cross-lane butterfly removed (results wrong,
loads identical) vs the real kernel, interleaved in one process.  Needs a library built with
-DPGT_TUNING_BUILD (the product build does not contain the ablated kernel)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import popgenomicstools_amd as pgt
from popgenomicstools_amd._lib import FST_ROW_DTYPE, PGT_STAT_FST
from popgenomicstools_amd.window_scan import windows_to_device

dev = torch.device("cuda", 0)
n = 1_000_000_000
pos, a, b, run_len = bench.synth_columns(n, 40, 12345, dev)
ctx = pgt.Context(0)
tree = torch.empty(ctx.tree_bytes(PGT_STAT_FST, n), dtype=torch.uint8, device=dev)
ctx.set_profiling(True)
os.environ["PGT_TUNE_BUILD_NT"] = "1"
os.environ["PGT_TUNE_BUILD_STRAIGHT"] = "1"  # the ablation applies to the straight (first) kernel
for m, chroms in ((1_000_000_000, 40), (100_000_000, 20)):
    rl = np.full(chroms, m // chroms, dtype=np.uint64)
    win = windows_to_device(pgt.build_windows_sites(rl, 50_000, 10_000), dev)
    out = torch.empty(win.numel() // 32 * FST_ROW_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    t = {0: [], 1: []}
    for r in range(13):
        for ab in (0, 1):
            if ab: os.environ["PGT_TUNE_BUILD_ABLATE"] = "1"
            else: os.environ.pop("PGT_TUNE_BUILD_ABLATE", None)
            ctx.fst_reduce_dev(pos[:m], a[:m], b[:m], win, out=out, tree=tree)
            bm, _ = ctx.last_kernel_ms()
            if r: t[ab].append(bm)
    for ab in (0, 1):
        med = float(np.median(t[ab]))
        print(f"{m:.0e} sites  {'ablated (no cross-lane)' if ab else 'real kernel (nt)'}: median {med:.4f} ms  min {min(t[ab]):.4f}  {16*m/med/1e6:.0f} GB/s")
