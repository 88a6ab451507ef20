#!/usr/bin/env python3
"""A short run of every build kernel for the rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE): few torch
dispatches (counter collection serialises and slows every dispatch), three launches of each library kernel (or the
second argument: ten under `--kernel-trace --stats`, where the average should be that of warm back-to-back launches) at
10^8 sites (or the first argument), a progress line per phase.  The trimmed counter rows go to profiles/rNN/pmc_counters_all.csv.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR -o fetch -- python3 tools/pmc_kernels.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d DIR -o write -- python3 tools/pmc_kernels.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import popgenomicstools_amd as pgt  # noqa: E402
from popgenomicstools_amd._lib import PGT_EXT_IHS  # noqa: E402
from popgenomicstools_amd.window_scan import windows_to_device  # noqa: E402


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3  # launches per kernel: 3 under --pmc (every dispatch is serialised), 10 under --stats
    dev = torch.device("cuda", 0)
    say = lambda *a: print(*a, flush=True)  # noqa: E731
    cols = [torch.rand(n, dtype=torch.float64, device=dev) for _ in range(12)]
    n1 = torch.randint(0, 21, (n,), dtype=torch.int32, device=dev)
    n2 = torch.randint(0, 21, (n,), dtype=torch.int32, device=dev)
    g1 = torch.randint(-1, 3, (n,), dtype=torch.int8, device=dev)
    g2 = torch.randint(-1, 3, (n,), dtype=torch.int8, device=dev)
    pos = torch.arange(1, n + 1, dtype=torch.int32, device=dev)
    run_len = np.full(20, n // 20, dtype=np.uint64)
    win = windows_to_device(pgt.build_windows_sites(run_len, 50_000, 10_000), dev)
    ewin_h = pgt.build_windows_extreme(np.arange(1, n + 1, dtype=np.uint32), run_len, None, 100_000)
    ewin = windows_to_device(ewin_h, dev)
    ctx = pgt.Context(0)
    ctx.set_max_window(50_000)
    say("data ready")
    for _ in range(reps):
        ctx.fst_reduce_dev(pos, cols[0], cols[1], win)
    say("fst done")
    for _ in range(reps):
        ctx.dxy_reduce_dev(pos, cols[0], cols[1], n1, n2, 5, win)
    say("dxy done")
    for _ in range(reps):
        ctx.het_reduce_dev(pos, g1, win)
    say("het done")
    for _ in range(reps):
        ctx.dxy_het_reduce_dev(pos, cols[0], cols[1], n1, n2, g1, g2, 5, win)
    say("fused dxy+het done")
    for _ in range(reps):
        ctx.fst_reduce_pairs_dev(pos, cols[0:12:2], cols[1:12:2], win)
    say("6 pairs done")
    for _ in range(reps):
        ctx.fst_af_reduce_dev(pos, cols[:8], [10.0 + k for k in range(8)], win)
    say("AF 8 populations done")
    ctx.set_max_window(int((ewin_h["hi"] - ewin_h["lo"]).max()))
    for _ in range(reps):
        ctx.extreme_reduce_dev(pos, cols[2], PGT_EXT_IHS, 0.9, ewin)
    torch.cuda.synchronize()
    say("extreme done")
    ctx.close()


if __name__ == "__main__":
    main()
