#!/usr/bin/env python3
"""Where a small input's step goes (round 6): het / extreme-score / fst steps at 1e5 … 1e8 sites — whole step
(events around back-to-back calls), build kernel and query kernel (the library's own events) — and the straight line
time = a + b * bytes through the sizes: `a` is what a launch costs whatever it reads (launch + ramp-up + drain), `b` the
streaming rate.  Markdown on stdout."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import popgenomicstools_amd as pgt  # noqa: E402
from popgenomicstools_amd._lib import PGT_EXT_IHS  # noqa: E402
from popgenomicstools_amd.window_scan import windows_to_device  # noqa: E402


def timed(ctx, call, reps=30):
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    step = e0.elapsed_time(e1) / reps * 1e3
    ctx.set_profiling(True)
    b, q = [], []
    for _ in range(reps):
        call()
        x = ctx.last_kernel_ms()
        b.append(x[0] * 1e3)
        q.append(x[1] * 1e3)
    ctx.set_profiling(False)
    return step, float(np.median(b)), float(np.median(q))


def main():
    dev = torch.device("cuda", 0)
    ctx = pgt.Context(0)
    sizes = [100_000, 1_000_000, 10_000_000, 30_000_000, 100_000_000]
    nmax = sizes[-1]
    g = torch.randint(-1, 3, (nmax,), dtype=torch.int8, device=dev)
    a, b = torch.rand(nmax, dtype=torch.float64, device=dev), torch.rand(nmax, dtype=torch.float64, device=dev) + 0.5
    pos = torch.arange(1, nmax + 1, dtype=torch.int32, device=dev)
    print("| statistic | sites | bytes streamed | step us | build kernel us | query kernel us | step - build - query us |")
    print("|---|---|---|---|---|---|---|")
    fits = {}
    for name, bps in (("het", 1), ("extreme", 8), ("fst", 16)):
        xs, ys, yb = [], [], []
        for n in sizes:
            W = min(50_000, n // 2)
            ctx.set_max_window(W if name != "extreme" else 0)
            if name == "extreme":
                wh = pgt.build_windows_extreme(np.arange(1, n + 1, dtype=np.uint32), np.array([n], dtype=np.uint64), None, 100_000)
                ctx.set_max_window(int((wh["hi"] - wh["lo"]).max()))
            else:
                wh = pgt.build_windows_sites(np.array([n], dtype=np.uint64), W, max(W // 5, 1))
            wd = windows_to_device(wh, dev)
            if name == "het":
                tree = torch.empty(ctx.tree_bytes(pgt._lib.PGT_STAT_HET, n), dtype=torch.uint8, device=dev)
                out = torch.empty(wh.size * 32, dtype=torch.uint8, device=dev)
                call = lambda: ctx.het_reduce_dev(pos[:n], g[:n], wd, out=out, tree=tree)  # noqa: E731
            elif name == "extreme":
                tree = torch.empty(ctx.tree_bytes(pgt._lib.PGT_STAT_EXT, n), dtype=torch.uint8, device=dev)
                out = torch.empty(wh.size * pgt._lib.EXT_ROW_DTYPE.itemsize, dtype=torch.uint8, device=dev)
                call = lambda: ctx.extreme_reduce_dev(pos[:n], a[:n], PGT_EXT_IHS, 0.9, wd, out=out, tree=tree)  # noqa: E731
            else:
                tree = torch.empty(ctx.tree_bytes(pgt._lib.PGT_STAT_FST, n), dtype=torch.uint8, device=dev)
                out = torch.empty(wh.size * 40, dtype=torch.uint8, device=dev)
                call = lambda: ctx.fst_reduce_dev(pos[:n], a[:n], b[:n], wd, out=out, tree=tree)  # noqa: E731
            step, bu, qu = timed(ctx, call)
            print(f"| {name} | {n:.0e} | {bps * n / 1e6:.1f} MB | {step:.1f} | {bu:.1f} | {qu:.1f} | {step - bu - qu:+.1f} |")
            xs.append(bps * n)
            ys.append(step)
            yb.append(bu)
        A = np.vstack([np.ones(len(xs)), np.array(xs, dtype=float)]).T
        (a0, b0), *_ = np.linalg.lstsq(A, np.array(yb), rcond=None)
        fits[name] = (a0, b0)
    print("\nbuild kernel us = a + bytes / rate (least squares over the five sizes; the library's events around the build launch):\n")
    for name, (a0, b0) in fits.items():
        print(f"- {name}: a = {a0:.1f} us per launch whatever it reads, rate = {1e-6 / b0:.2f} TB/s = {1e-6 / b0 / 8 * 100:.0f} % of 8 TB/s")
    ctx.close()


if __name__ == "__main__":
    main()
