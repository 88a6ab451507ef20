#!/usr/bin/env python3
"""Interleaved A/B of the extreme-score build kernel variants (tuning build only:
PGT_EXTRA_HIPCC_FLAGS=-DPGT_TUNING_BUILD python -m popgenomicstools_amd.build --force).
PGT_EXT_VARIANT_NOW selects the variant per call: 0 product (stage 16, 8 waves/CU), 1 direct stores,
2 stage 8 / 16 waves per CU, 3 stage 4 / 32 waves per CU, 4 product with 16 loads in flight,
5 product with 4 loads in flight, 6 direct stores with 16 loads in flight; round 3 (the short queue): stage 16 / 8 waves
per CU with 7: 4, 8: 8, 9: 16, 10: 2 loads in flight.  Rows must be identical.  EXT_VARIANTS=0,7,8 selects."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import popgenomicstools_amd as pgt  # noqa: E402
from popgenomicstools_amd._lib import PGT_EXT_IHS  # noqa: E402
from popgenomicstools_amd.window_scan import windows_to_device  # noqa: E402
from synth_genome import SynthGenome  # noqa: E402


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    dev = torch.device("cuda", 0)
    g = SynthGenome(5, n, 40 if n > 2e8 else 20)
    pos, a, _ = g.fst_columns_t(0, n, dev)
    score = a * 40.0 - 2.0
    del a
    hpos = pos.cpu().numpy().view(np.uint32)
    win_h = pgt.build_windows_extreme(hpos, g.run_len, None, 100_000)
    win = windows_to_device(win_h, dev)
    ctx = pgt.Context(0)
    ctx.set_max_window(int((win_h["hi"] - win_h["lo"]).max()))
    ctx.set_profiling(True)
    tree = torch.empty(ctx.tree_bytes(3, n), dtype=torch.uint8, device=dev)
    variants = [int(x) for x in os.environ.get("EXT_VARIANTS", "0,1,2,3,4,5,6").split(",")]
    t = {v: [] for v in variants}
    ref = None
    for r in range(reps + 1):
        for v in variants:
            os.environ["PGT_EXT_VARIANT_NOW"] = str(v)
            out, _ = ctx.extreme_reduce_dev(pos, score, PGT_EXT_IHS, 2.0, win, tree=tree)
            bm, _ = ctx.last_kernel_ms()
            if r:
                t[v].append(bm)
            else:
                b = out.cpu().numpy().tobytes()
                ref = ref or b
                assert b == ref, f"variant {v} differs"
    print(f"extreme-score build, {n:.0e} sites, {win_h.size} windows, median of {reps}; rows identical across variants")
    print("| variant | build ms | GB/s | % of 8 TB/s |")
    print("|---|---|---|---|")
    for v in variants:
        med = float(np.median(t[v]))
        print(f"| {v} | {med:.4f} | {8 * n / med / 1e6:.0f} | {8 * n / med / 1e6 / 80:.1f} |")


if __name__ == "__main__":
    main()
