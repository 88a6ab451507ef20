#!/usr/bin/env python3
"""Does the build kernel's rate depend on where the two columns lie relative to each other?  a and b are
views into ONE allocation, b starting `skew` bytes after the end of a; each skew is timed in the same
process (median of 12 build launches).  Markdown on stdout."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import popgenomicstools_amd as pgt  # noqa: E402
from popgenomicstools_amd.window_scan import windows_to_device  # noqa: E402
from synth_genome import SynthGenome  # noqa: E402

dev = torch.device("cuda", 0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
g = SynthGenome(1, n, 40)
pos, a0, b0 = g.fst_columns_t(0, n, dev)
ctx = pgt.Context(0)
ctx.set_max_window(50_000)
ctx.set_profiling(True)
tree = torch.empty(ctx.tree_bytes(0, n), dtype=torch.uint8, device=dev)
win = windows_to_device(pgt.build_windows_sites(g.run_len, 50_000, 10_000), dev)
slack = 64 << 20
pool = torch.empty(16 * n + slack, dtype=torch.uint8, device=dev)
a = pool[: 8 * n].view(torch.float64)
a.copy_(a0)


def timed(aa, bb):
    t = []
    for r in range(15):
        ctx.fst_reduce_dev(pos, aa, bb, win, tree=tree)
        bm, _ = ctx.last_kernel_ms()
        if r >= 3:
            t.append(bm)
    return float(np.median(t))


print(f"{n:.0e} sites; separately allocated columns: a at {a0.data_ptr():#x}, b at {b0.data_ptr():#x} (b - a = {b0.data_ptr() - a0.data_ptr():#x})")
m = timed(a0, b0)
print(f"separate allocations: {m:.4f} ms = {16 * n / m / 1e6 / 80:.1f} %\n")
print("| skew of b after the end of a (bytes) | build ms | % of 8 TB/s |")
print("|---|---|---|")
for skew in [0, 256, 1024, 4096, 16384, 65536 + 4096, 1 << 20, (1 << 20) + 4096, (2 << 20) + 8192, (32 << 20) + 12288]:
    b = pool[8 * n + skew: 16 * n + skew].view(torch.float64)
    b.copy_(b0)
    m = timed(a, b)
    print(f"| {skew} | {m:.4f} | {16 * n / m / 1e6 / 80:.1f} |", flush=True)
