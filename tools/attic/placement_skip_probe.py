#!/usr/bin/env python3
"""Does it matter WHERE in a process's VRAM the columns lie?  (profiles/r04: the same kernel reads 80.7 ... 86.4 % of the HBM peak
from process to process; inside one process the first column set was the slow one.)  One process per measurement: optionally a
dummy allocation of SKIP GB first (kept alive), then the 10^9-site fst columns, 40 launches of the build alone, median ms.

    python tools/attic/placement_skip_probe.py <skip GB> [sites]     (run it several times, alternating the argument)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import popgenomicstools_amd as pgt  # noqa: E402
from popgenomicstools_amd._lib import PGT_STAT_FST  # noqa: E402
from synth_genome import SynthGenome  # noqa: E402


def main():
    skip = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
    n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000_000
    dev = torch.device("cuda", 0)
    dummy = torch.empty(int(skip * (1 << 30)), dtype=torch.uint8, device=dev) if skip > 0 else None
    if dummy is not None:
        dummy.fill_(1)
    g = SynthGenome(12345, n, 40)
    pos, a, b = g.fst_columns_t(0, n, dev)
    ctx = pgt.Context(0)
    ctx.set_max_window(50_000)
    tree = torch.empty(ctx.tree_bytes(PGT_STAT_FST, n), dtype=torch.uint8, device=dev)
    nowin = torch.empty(0, dtype=torch.uint8, device=dev)
    out = torch.empty(64, dtype=torch.uint8, device=dev)
    for _ in range(100):
        ctx.fst_reduce_dev(pos, a, b, nowin, out=out, tree=tree)
    torch.cuda.synchronize()
    ms = []
    for _ in range(40):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ctx.fst_reduce_dev(pos, a, b, nowin, out=out, tree=tree)
        e1.record()
        e1.synchronize()
        ms.append(e0.elapsed_time(e1) / 5)
    med = float(np.median(ms))
    print(f"skip {skip:5.1f} GB  a at {a.data_ptr():#x}  build {med:.4f} ms  {16.0 * n / med / 8e7:.1f} % of 8 TB/s", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
