#!/usr/bin/env python3
"""Query-kernel time for S << W on one MI355X: fstWindow, 10^8 sites, W = 50000, S in {1, 8, 32, 64, 100, 500, 1000, 2048, 10000},
with the per-window query (step hint 0) and with the strategy the step hint S selects (S <= 32 sliding, 32 < S <= 2048
group, above that the hint changes nothing).  Markdown on stdout.
usage: measure_query.py [sites [steps,comma,separated [winsize]]]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import popgenomicstools_amd as pgt  # noqa: E402
from popgenomicstools_amd._lib import FST_ROW_DTYPE  # noqa: E402
from popgenomicstools_amd.window_scan import windows_to_device  # noqa: E402
from synth_genome import SynthGenome  # noqa: E402


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
    steps = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 4, 8, 16, 32, 64, 100, 250, 500, 1000, 2048, 10_000]
    dev = torch.device("cuda", 0)
    g = SynthGenome(12345, n, 20)
    pos, a, b = g.fst_columns_t(0, n, dev)
    ctx = pgt.Context(0)
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 50_000
    ctx.set_max_window(W)
    ctx.set_profiling(True)
    tree = torch.empty(ctx.tree_bytes(0, n), dtype=torch.uint8, device=dev)
    print(f"fstWindow query kernel, {n:.0e} sites in 20 chromosomes, W = {W}; build kernel for reference")
    print("| S | windows | strategy | query ms | bytes/window at 6 TB/s equiv | build ms | max rel. diff of sums vs per-window |")
    print("|---|---|---|---|---|---|---|")
    for S in steps:
        win_h = pgt.build_windows_sites(g.run_len, W, S)
        win = windows_to_device(win_h, dev)
        nw = win_h.size
        del win_h
        # The hint selects the strategy, and every strategy answers any table correctly: so each can be timed at every S by
        # passing a hint inside its range (sliding: <= 32 with the longest window unknown, group: 1 .. 1024) — the product passes the true step.
        cases = [("per-window", 0)]
        if S <= 32:
            cases.append(("sliding", S))
        cases.append(("group", min(S, 1024)))
        out = [torch.empty(nw * FST_ROW_DTYPE.itemsize, dtype=torch.uint8, device=dev) for _ in range(len(cases))]
        chosen = "group" if (S <= 1024 and W >= 16384) else ("sliding" if S <= 32 else "per-window")
        for k, (name, hint) in enumerate(cases):
            ctx.set_window_step(hint)
            ctx.set_max_window(0 if name == "sliding" else W)  # an unknown longest window rules the group query out: the sliding one runs
            q, bms = [], []
            for r in range(4):
                ctx.fst_reduce_dev(pos, a, b, win, out=out[k], tree=tree)
                bm, qm = ctx.last_kernel_ms()
                if r:
                    q.append(qm)
                    bms.append(bm)
            qm = float(np.median(q))
            diff = ""
            if k > 0:
                r0 = out[0].view(torch.float64).view(-1, 5)
                r1 = out[k].view(torch.float64).view(-1, 5)
                d = ((r1[:, 3:] - r0[:, 3:]).abs() / r0[:, 3:].abs().clamp_min(1e-300)).max().item()
                ints_equal = bool(torch.equal(out[0].view(torch.int32).view(-1, 10)[:, :4], out[k].view(torch.int32).view(-1, 10)[:, :4]))
                diff = f"{d:.2e} (coordinates/counts equal: {ints_equal})"
            mark = " **(the product's choice)**" if name == chosen else ""
            print(f"| {S} | {nw} | {name}{mark} | {qm:.3f} | {qm * 1e-3 * 6e12 / nw:.0f} | {float(np.median(bms)):.3f} | {diff} |", flush=True)
        del win, out
    ctx.close()


if __name__ == "__main__":
    main()
