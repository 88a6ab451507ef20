#!/usr/bin/env python3
"""Does the headline step run slower right after the GPU was idle?  After a synchronise (+ an optional sleep) a burst of
steps is launched with one event per step; prints the duration of step k for the first steps and the mean of later groups.
(Round 4: bench.py's 20-step timed region read 2.577 ms/step where 1010 back-to-back steps read 2.488.)

    python tools/ramp_probe.py [sites=1e9] [idle seconds=0.5] [steps=400]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import popgenomicstools_amd as pgt  # noqa: E402
from popgenomicstools_amd._lib import FST_ROW_DTYPE, PGT_STAT_FST  # noqa: E402
from popgenomicstools_amd.window_scan import windows_to_device  # noqa: E402
from synth_genome import SynthGenome  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
idle = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
K = int(sys.argv[3]) if len(sys.argv) > 3 else 400
dev = torch.device("cuda", 0)
g = SynthGenome(12345, n, 40 if n > 200_000_000 else 20)
pos, a, b = g.fst_columns_t(0, n, dev)
win_h = pgt.build_windows_sites(g.run_len, 50_000, 10_000)
win = windows_to_device(win_h, dev)
ctx = pgt.Context(0)
ctx.set_max_window(50_000)
ctx.set_window_step(10_000)
tree = torch.empty(ctx.tree_bytes(PGT_STAT_FST, n), dtype=torch.uint8, device=dev)
out = torch.empty(win_h.size * FST_ROW_DTYPE.itemsize, dtype=torch.uint8, device=dev)
step = lambda: ctx.fst_reduce_dev(pos, a, b, win, out=out, tree=tree)  # noqa: E731
for trial in range(3):
    torch.cuda.synchronize()
    time.sleep(idle)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    ev[0].record()
    for k in range(K):
        step()
        ev[k + 1].record()
    torch.cuda.synchronize()
    d = np.array([ev[k].elapsed_time(ev[k + 1]) for k in range(K)])
    t = np.cumsum(d)
    groups = [(0, 1), (1, 2), (2, 5), (5, 10), (10, 20), (20, 40), (40, 80), (80, 160), (160, K)]
    print(f"trial {trial} (idle {idle} s before): " + "  ".join(f"steps {lo}-{hi - 1}: {d[lo:hi].mean():.4f} ms (t={t[hi - 1]:.0f} ms)" for lo, hi in groups if hi <= K), flush=True)
