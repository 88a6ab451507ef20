#!/usr/bin/env python3
"""Does the build rate depend on WHICH physical memory the columns got?  Successive processes of one box read 80.8, 81.6, 86.3,
86.4, 86.3 % with the same binary (profiles/r04/bench_spread.txt).  Here, inside ONE process: column set after column set is
allocated (the earlier ones kept alive at first, so that new memory is used; later freed, so that memory is recycled) and the
fst build is timed on each.

    python tools/placement_probe2.py [sites=1e9] [sets=8]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import popgenomicstools_amd as pgt  # noqa: E402
from popgenomicstools_amd._lib import FST_ROW_DTYPE, PGT_STAT_FST  # noqa: E402
from popgenomicstools_amd.window_scan import windows_to_device  # noqa: E402
from synth_genome import SynthGenome  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
sets = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0)
g = SynthGenome(12345, n, 40 if n > 200_000_000 else 20)
win_h = pgt.build_windows_sites(g.run_len, 50_000, 10_000)
win = windows_to_device(win_h, dev)
ctx = pgt.Context(0)
ctx.set_max_window(50_000)
ctx.set_window_step(10_000)
ctx.set_profiling(True)
tree = torch.empty(ctx.tree_bytes(PGT_STAT_FST, n), dtype=torch.uint8, device=dev)
out = torch.empty(win_h.size * FST_ROW_DTYPE.itemsize, dtype=torch.uint8, device=dev)
keep = []
print("| set | kept alive before it | a.data_ptr | build ms (median of 30) | % of 8 TB/s |\n|---|---|---|---|---|", flush=True)
for k in range(sets):
    pos, a, b = g.fst_columns_t(0, n, dev)
    t = []
    for _ in range(35):
        ctx.fst_reduce_dev(pos, a, b, win, out=out, tree=tree)
        t.append(ctx.last_kernel_ms()[0])
    ms = float(np.median(t[5:]))
    print(f"| {k} | {len(keep)} | {a.data_ptr():#x} | {ms:.4f} | {16.0 * n / ms / 1e6 / 80:.1f} |", flush=True)
    if k < 4:
        keep.append((pos, a, b))  # the next set must take other memory
    else:
        keep.clear()              # from here on memory is recycled
        del pos, a, b
        torch.cuda.empty_cache()
