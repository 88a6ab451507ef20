#!/usr/bin/env python3
"""Build-kernel time of the 10^9-site scan against wall time since the process first touched the GPU, on a
box that was idle before: shows how long the device takes to reach its steady-state memory throughput."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import popgenomicstools_amd as pgt  # noqa: E402
from popgenomicstools_amd.window_scan import windows_to_device  # noqa: E402
from synth_genome import SynthGenome  # noqa: E402

t_start = time.perf_counter()
dev = torch.device("cuda", 0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
g = SynthGenome(1, n, 40)
pos, a, b = g.fst_columns_t(0, n, dev)
ctx = pgt.Context(0)
ctx.set_max_window(50_000)
ctx.set_profiling(True)
tree = torch.empty(ctx.tree_bytes(0, n), dtype=torch.uint8, device=dev)
win = windows_to_device(pgt.build_windows_sites(g.run_len, 50_000, 10_000), dev)
torch.cuda.synchronize()
print(f"columns ready {time.perf_counter() - t_start:.1f} s after start")
print("| s since first launch | launches so far | build ms (median of the last 20) | % of 8 TB/s |")
print("|---|---|---|---|")
t0 = time.perf_counter()
k, nxt, recent = 0, 0.0, []
while time.perf_counter() - t0 < secs:
    ctx.fst_reduce_dev(pos, a, b, win, tree=tree)
    bm, _ = ctx.last_kernel_ms()
    recent = (recent + [bm])[-20:]
    k += 1
    el = time.perf_counter() - t0
    if el >= nxt:
        med = float(np.median(recent))
        print(f"| {el:.1f} | {k} | {med:.4f} | {16 * n / med / 1e6 / 80:.1f} |", flush=True)
        nxt += 1.0 if el < 10 else 5.0
