#!/usr/bin/env python3
"""What the last, PARTIAL level-2 tile of a build costs (round 6): the fst / dxy / extreme builds at sizes that differ only in
their last tile — a whole number of 8192-site tiles, the same plus 1 / 13 / 44 / 63 leaves of 128 sites — interleaved in one
process, build kernel by the library's events (median of 15).  10^9 sites end in a tile of 44 leaves, 10^8 in one of 2."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import popgenomicstools_amd as pgt  # noqa: E402
from popgenomicstools_amd.window_scan import windows_to_device  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    ctx = pgt.Context(0)
    ctx.set_max_window(50_000)
    bases = [int(float(x)) for x in sys.argv[1:]] or [100_000_000, 1_000_000_000]
    print("| statistic | whole tiles | + leaves of the last tile | sites | build kernel us (median of 15) | against whole tiles |")
    print("|---|---|---|---|---|---|")
    for base in bases:
        tiles = base // 8192
        nmax = tiles * 8192 + 63 * 128 + 77
        a = torch.rand(nmax, dtype=torch.float64, device=dev)
        b = torch.rand(nmax, dtype=torch.float64, device=dev) + 0.5
        n1 = torch.randint(0, 12, (nmax,), dtype=torch.int32, device=dev)
        pos = torch.arange(1, nmax + 1, dtype=torch.int32, device=dev)
        sizes = [(0, tiles * 8192), (1, tiles * 8192 + 128), (13, tiles * 8192 + 13 * 128), (44, tiles * 8192 + 44 * 128), (63, tiles * 8192 + 63 * 128 + 77)]
        for stat in ("fst", "dxy"):
            calls = {}
            for leaves, n in sizes:
                wd = windows_to_device(pgt.build_windows_sites(np.array([n], dtype=np.uint64), 50_000, 10_000), dev)
                if stat == "fst":
                    tree = torch.empty(ctx.tree_bytes(pgt._lib.PGT_STAT_FST, n), dtype=torch.uint8, device=dev)
                    calls[leaves] = (n, (lambda n=n, wd=wd, tree=tree: ctx.fst_reduce_dev(pos[:n], a[:n], b[:n], wd, tree=tree)))
                else:
                    tree = torch.empty(ctx.tree_bytes(pgt._lib.PGT_STAT_DXY, n), dtype=torch.uint8, device=dev)
                    calls[leaves] = (n, (lambda n=n, wd=wd, tree=tree: ctx.dxy_reduce_dev(pos[:n], a[:n], b[:n], n1[:n], n1[:n], 5, wd, tree=tree)))
            ctx.set_profiling(True)
            t = {k: [] for k in calls}
            for _ in range(17):
                for k, (n, fn) in calls.items():
                    fn()
                    t[k].append(ctx.last_kernel_ms()[0] * 1e3)
            ctx.set_profiling(False)
            ref = float(np.median(t[0][2:]))
            for k, (n, _) in calls.items():
                m = float(np.median(t[k][2:]))
                print(f"| {stat} | {tiles} | {k} | {n} | {m:.1f} | {m - ref:+.1f} us ({100 * (m - ref) / ref:+.2f} %) |", flush=True)
        del a, b, n1, pos
        torch.cuda.empty_cache()
    ctx.close()


if __name__ == "__main__":
    main()
