#!/usr/bin/env python3
"""Build-kernel time / HBM rate of the other BASELINE configs on one MI355X (markdown on stdout):
config 2  fstWindow 10^8 sites              16 B/site
config 3  dxyWindow + hetWindow 10^8 sites  26 B/site (fused entry point) and each tool alone
config 5  fstWindow, 28 population pairs x 10^8 sites, one table, 448 B/site (all pairs on ONE GPU here)
plus the host-buffer entry point (PCIe included) for the record."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import popgenomicstools_amd as pgt  # noqa: E402
from synth_genome import SynthGenome  # noqa: E402
from popgenomicstools_amd.window_scan import windows_to_device  # noqa: E402


def timed(ctx, fn, reps=12):
    """-> (build kernel ms, query kernel ms, whole step ms), medians.  Kernel times: the library's HIP events
    around the build and the query launches; whole step: torch events on the launch stream around one call
    (round 1 divided a host-timed loop of 12 calls by 12, which charged the idle GPU's wake-up and the
    host's launch latency of the first call to every step: 0.57 ms for a 0.29 ms step)."""
    ctx.set_profiling(True)
    b, q = [], []
    for r in range(reps + 2):
        fn()
        bm, qm = ctx.last_kernel_ms()
        if r >= 2:
            b.append(bm); q.append(qm)
    ctx.set_profiling(False)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for e0, e1 in ev:
        e0.record()
        fn()
        e1.record()
    torch.cuda.synchronize()
    return float(np.median(b)), float(np.median(q)), float(np.median([e0.elapsed_time(e1) for e0, e1 in ev]))


def main():
    dev = torch.device("cuda", 0)
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
    big = n > 200_000_000  # at 10^9 sites the 28-pair and host-buffer legs are skipped (memory / time)
    chroms, W, S = (40 if big else 20), 50_000, 10_000
    ctx = pgt.Context(0)
    gen = torch.Generator(device=dev).manual_seed(7)
    genome = SynthGenome(12345, n, chroms)
    run_len = genome.run_len
    pos, a, b = genome.fst_columns_t(0, n, dev)
    win = windows_to_device(pgt.build_windows_sites(run_len, W, S), dev)
    print("| config | bytes/site | build ms | GB/s | % of 8 TB/s | query ms | whole step ms | sites/s |")
    print("|---|---|---|---|---|---|---|---|")

    def row(name, bps, res):
        bm, qm, step = res
        print(f"| {name} | {bps} | {bm:.4f} | {bps * n / bm / 1e6:.0f} | {bps * n / bm / 1e6 / 80:.1f} | {qm:.4f} | {step:.4f} | {n / step * 1e3:.3e} |")

    from popgenomicstools_amd._lib import PGT_STAT_FST
    n_pairs = 2 if big else 28
    tree = torch.empty(n_pairs * ctx.tree_bytes(PGT_STAT_FST, n), dtype=torch.uint8, device=dev)  # reused by every config
    out = torch.empty(28 * 40 * (win.numel() // 32), dtype=torch.uint8, device=dev)
    ctx.set_max_window(W)
    row(f"2: fstWindow {n:.0e}", 16, timed(ctx, lambda: ctx.fst_reduce_dev(pos, a, b, win, out=out, tree=tree)))
    p1 = torch.round(torch.rand(n, generator=gen, device=dev, dtype=torch.float64) * 1e6) / 1e6
    p2 = torch.round(torch.rand(n, generator=gen, device=dev, dtype=torch.float64) * 1e6) / 1e6
    n1 = torch.randint(0, 21, (n,), generator=gen, device=dev, dtype=torch.int32)
    n2 = torch.randint(0, 21, (n,), generator=gen, device=dev, dtype=torch.int32)
    g1 = (torch.randint(0, 20, (n,), generator=gen, device=dev, dtype=torch.int32) % 4 - 1).to(torch.int8)
    g2 = (torch.randint(0, 20, (n,), generator=gen, device=dev, dtype=torch.int32) % 4 - 1).to(torch.int8)
    row(f"3a: dxyWindow alone {n:.0e}", 24, timed(ctx, lambda: ctx.dxy_reduce_dev(pos, p1, p2, n1, n2, 5, win, out=out, tree=tree)))
    row(f"3b: hetWindow alone {n:.0e}", 1, timed(ctx, lambda: ctx.het_reduce_dev(pos, g1, win, out=out, tree=tree)))
    row(f"3: dxy + het x2 fused {n:.0e}", 26, timed(ctx, lambda: ctx.dxy_het_reduce_dev(pos, p1, p2, n1, n2, g1, g2, 5, win, tree=tree)))
    del p1, p2, n1, n2, g1, g2
    # ihsWindow-style extreme-score scan: one f64 score column, 100 kb non-overlapping windows
    from popgenomicstools_amd._lib import EXT_ROW_DTYPE
    hpos = pos.cpu().numpy().view(np.uint32)
    ewin_h = pgt.build_windows_extreme(hpos, run_len, None, 100_000)
    ewin = windows_to_device(ewin_h, dev)
    eout = torch.empty(ewin_h.size * EXT_ROW_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    lib = pgt._lib.load()
    score = a * 40.0 - 2.0

    def ext_call():
        ctx.extreme_reduce_dev(pos, score, 0, 2.0, ewin, out=eout, tree=tree)
    ctx.set_max_window(int((ewin_h["hi"] - ewin_h["lo"]).max()))
    row(f"ihsWindow-style extreme scan {n:.0e} ({ewin_h.size} windows)", 8, timed(ctx, ext_call))
    ctx.set_max_window(W)
    del score
    if big:
        print(f"\n({n:.0e} sites: 28-pair and host-buffer legs skipped)")
        ctx.close()
        return
    al = [a] + [a.roll(1000 * (k + 1)) for k in range(n_pairs - 1)]
    bl = [b] + [b.roll(1000 * (k + 1)) for k in range(n_pairs - 1)]
    row("5: fstWindow 28 pairs x 1e8 (one GPU)", 448, timed(ctx, lambda: ctx.fst_reduce_pairs_dev(pos, al, bl, win, out=out, tree=tree), reps=6))
    del al, bl
    # the same 28 pairs from 8 allele-frequency columns: 64 B/site streamed
    fr = [torch.round(torch.rand(n, generator=gen, device=dev, dtype=torch.float64) * 1e6) / 1e6 for _ in range(8)]
    nsamp = [10.0 + k for k in range(8)]
    af_tree = torch.empty(int(pgt._lib.load().pgt_af_tree_bytes(8, n)), dtype=torch.uint8, device=dev)
    row("5': 28 pairs from 8 frequency columns x 1e8 (pgt_fst_af_reduce_dev)", 64,
        timed(ctx, lambda: ctx.fst_af_reduce_dev(pos, fr, nsamp, win, out=out, tree=af_tree), reps=6))
    row("AF front end, 2 populations x 1e8", 16,
        timed(ctx, lambda: ctx.fst_af_reduce_dev(pos, fr[:2], nsamp[:2], win, out=out, tree=af_tree)))
    del fr, af_tree
    # host-buffer entry point: H2D of 20 B/site + kernels + D2H, synchronous
    m = 50_000_000
    hp, ha, hb = pos[:m].cpu().numpy().view(np.uint32), a[:m].cpu().numpy(), b[:m].cpu().numpy()
    hw = pgt.build_windows_sites(np.full(10, m // 10, dtype=np.uint64), W, S)
    import time
    ctx.fst_reduce(hp, ha, hb, hw)
    t0 = time.perf_counter()
    for _ in range(3):
        ctx.fst_reduce(hp, ha, hb, hw)
    dt = (time.perf_counter() - t0) / 3
    print(f"\nhost-buffer pgt_fst_reduce (pageable host memory, PCIe included), {m:.0e} sites: {dt * 1e3:.1f} ms = {m / dt:.3e} sites/s = {20 * m / dt / 1e9:.1f} GB/s over the link")
    ctx.close()


if __name__ == "__main__":
    main()
