#!/usr/bin/env python3
"""Interleaved A/B: fst_build_straight_kernel (register loads) vs fst_build_lds_kernel (LDS-DMA ring) in one
process; also checks that both produce the same bytes.  Needs a library built with -DPGT_TUNING_BUILD
(PGT_EXTRA_HIPCC_FLAGS=-DPGT_TUNING_BUILD python -m popgenomicstools_amd.build --force)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import popgenomicstools_amd as pgt  # noqa: E402
from popgenomicstools_amd._lib import FST_ROW_DTYPE, PGT_STAT_FST  # noqa: E402
from popgenomicstools_amd.window_scan import windows_to_device  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device("cuda", 0)
    n = 1_000_000_000
    pos, a, b, run_len = bench.synth_columns(n, 40, 12345, dev)
    ctx = pgt.Context(0)
    ctx.set_max_window(50_000)
    tree = torch.empty(ctx.tree_bytes(PGT_STAT_FST, n), dtype=torch.uint8, device=dev)
    sizes = {}
    for m, chroms in ((1_000_000_000, 40), (100_000_000, 20)):
        rl = np.full(chroms, m // chroms, dtype=np.uint64)
        win = windows_to_device(pgt.build_windows_sites(rl, 50_000, 10_000), dev)
        out = torch.empty(win.numel() // 32 * FST_ROW_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        sizes[m] = (win, out)
    variants = [("straight kernel, register loads", "straight", None), ("straight kernel, permlane-swap butterfly", "perm", None)] + [(f"LDS-DMA ring, {2 * g} loads in flight", g, bl)
                                                        for g, bl in ((4, 1024), (4, 2048))] + [
        ("LDS-DMA ring, 8 loads in flight, permlane-swap butterfly", "ldsperm", 1024),
        ("LDS-DMA ring, 8 loads in flight, permlane-swap butterfly", "ldsperm", 2048)]
    def select(v):
        os.environ.pop("PGT_TUNE_BUILD_LDS", None)
        os.environ.pop("PGT_TUNE_BUILD_BLOCKS", None)
        os.environ.pop("PGT_TUNE_BUILD_PERMLANE", None)
        os.environ.pop("PGT_TUNE_BUILD_STRAIGHT", None)
        if v[1] == "straight":
            os.environ["PGT_TUNE_BUILD_STRAIGHT"] = "1"
        elif v[1] == "perm":
            os.environ["PGT_TUNE_BUILD_STRAIGHT"] = "1"
            os.environ["PGT_TUNE_BUILD_PERMLANE"] = "1"
        elif v[1] == "ldsperm":
            os.environ["PGT_TUNE_BUILD_PERMLANE"] = "1"
            os.environ["PGT_TUNE_BUILD_LDS"] = "4"
            os.environ["PGT_TUNE_BUILD_BLOCKS"] = str(v[2])
        elif v[1] is not None:
            os.environ["PGT_TUNE_BUILD_LDS"] = str(v[1])
            os.environ["PGT_TUNE_BUILD_BLOCKS"] = str(v[2])
    # correctness first: same bytes
    for m, (win, out) in sizes.items():
        ref = None
        for v in variants:
            select(v)
            out.zero_()
            ctx.fst_reduce_dev(pos[:m], a[:m], b[:m], win, out=out, tree=tree)
            torch.cuda.synchronize()
            got = out.cpu().numpy().tobytes()
            if ref is None:
                ref = got
            assert got == ref, f"variant {v} differs at {m}"
    print("all variants produce identical rows\n")
    ctx.set_profiling(True)
    times = {(m, i): [] for m in sizes for i in range(len(variants))}
    for r in range(rounds + 1):
        for m, (win, out) in sizes.items():
            for i, v in enumerate(variants):
                select(v)
                ctx.fst_reduce_dev(pos[:m], a[:m], b[:m], win, out=out, tree=tree)
                bm, _ = ctx.last_kernel_ms()
                if r:
                    times[(m, i)].append(bm)
    print("| sites | variant | workgroups | median ms | min ms | GB/s | % of 8 TB/s |")
    print("|---|---|---|---|---|---|---|")
    for m in sizes:
        for i, v in enumerate(variants):
            t = np.array(times[(m, i)])
            med = float(np.median(t))
            print(f"| {m:.0e} | {v[0]} | {v[2] or 2048} | {med:.4f} | {t.min():.4f} | {16.0 * m / med / 1e6:.0f} | {16.0 * m / med / 1e6 / 80:.1f} |")
    ctx.close()


if __name__ == "__main__":
    main()
