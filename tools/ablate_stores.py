#!/usr/bin/env python3
"""Where do the tree-node stores of fst_build_kernel cost time, and what recovers it?  Interleaved A/B
in one process, on the tuning build (PGT_EXTRA_HIPCC_FLAGS=-DPGT_TUNING_BUILD python -m
popgenomicstools_amd.build --force).  Every real variant is first checked to give the same bytes
as the product kernel; the timing-only variants (stores or butterfly removed) give wrong results
by construction."""
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

ALL_ENVS = ("PGT_TUNE_BUILD_PHASED", "PGT_TUNE_BUILD_STRAIGHT", "PGT_TUNE_BUILD_BLOCKS", "PGT_TUNE_BUILD_STORE_MODE", "PGT_TUNE_BUILD_ABLATE", "PGT_TUNE_BUILD_PIPE", "PGT_TUNE_BUILD_SW",
            "PGT_TUNE_BUILD_DEFER")
ST = {"PGT_TUNE_BUILD_STRAIGHT": "1"}  # the first product kernel: node rows stored after every tile
REAL = {  # variants that must reproduce the product's bytes
    "product: stores deferred, 16 tiles staged, nt, 512 WGs": {},
    "product kernel, one launch per stage (4 launches at 1e9)": {"PGT_TUNE_BUILD_PHASED": "1"},
    "deferred: 16 staged, plain stores": {"PGT_TUNE_BUILD_DEFER": "16:4:0:512"},
    "deferred: 16 staged, 16+16 loads in flight": {"PGT_TUNE_BUILD_DEFER": "16:8:1:512"},
    "deferred: 32 staged, 256 WGs": {"PGT_TUNE_BUILD_DEFER": "32:4:1:256"},
    "deferred: 8 staged, 1024 WGs": {"PGT_TUNE_BUILD_DEFER": "8:4:1:1024"},
    "deferred: 4 staged, 2048 WGs": {"PGT_TUNE_BUILD_DEFER": "4:4:1:2048"},
    "straight kernel (stores after every tile), 2048 WGs": dict(ST),
    "straight kernel, 512 WGs (control: fewer waves only)": dict(ST, PGT_TUNE_BUILD_BLOCKS="512"),
    "straight kernel, nt stores": dict(ST, PGT_TUNE_BUILD_STORE_MODE="4"),
    "store wave (3 compute + 1 storing wave)": {"PGT_TUNE_BUILD_SW": "1"},
    "pipelined across tile boundaries": {"PGT_TUNE_BUILD_PIPE": "1"},
}
TIMING_ONLY = {
    "[timing only] straight, no node stores": dict(ST, PGT_TUNE_BUILD_STORE_MODE="1"),
    "[timing only] straight, no level-1 store": dict(ST, PGT_TUNE_BUILD_STORE_MODE="2"),
    "[timing only] straight, no level-2 store": dict(ST, PGT_TUNE_BUILD_STORE_MODE="3"),
    "[timing only] straight, no cross-lane butterfly": dict(ST, PGT_TUNE_BUILD_ABLATE="1"),
}


def select(env):
    for e in ALL_ENVS:
        os.environ.pop(e, None)
    os.environ.update(env)


def main():
    dev = torch.device("cuda", 0)
    n = 1_000_000_000
    pos, a, b, _ = bench.synth_columns(n, 40, 12345, dev)
    ctx = pgt.Context(0)
    ctx.set_max_window(50_000)
    tree = torch.empty(ctx.tree_bytes(PGT_STAT_FST, n), dtype=torch.uint8, device=dev)

    def table(m):
        chroms = 40 if m == n else 20
        w = windows_to_device(pgt.build_windows_sites(np.full(chroms, m // chroms, dtype=np.uint64), 50_000, 10_000), dev)
        return w, torch.empty(w.numel() // 32 * FST_ROW_DTYPE.itemsize, dtype=torch.uint8, device=dev)

    # correctness: same bytes as the product kernel, on an aligned and on a ragged size
    for m in (n, 123_456_789):
        w, o = table(m if m == n else 100_000_000)
        ref = None
        for name, env in REAL.items():
            select(env)
            o.zero_()
            ctx.fst_reduce_dev(pos[:m], a[:m], b[:m], w[: 32 * 2000], out=o[: 40 * 2000], tree=tree)
            torch.cuda.synchronize()
            got = o[: 40 * 2000].cpu().numpy().tobytes()
            ref = got if ref is None else ref
            assert got == ref, f"{name} differs from the product kernel at {m} sites"
    print("every real variant reproduces the product kernel's rows bit for bit\n")

    ctx.set_profiling(True)
    variants = {**REAL, **TIMING_ONLY}
    for m in (n, 100_000_000):
        w, o = table(m)
        t = {k: [] for k in variants}
        for r in range(13):
            for k, env in variants.items():
                select(env)
                ctx.fst_reduce_dev(pos[:m], a[:m], b[:m], w, out=o, tree=tree)
                bm, _ = ctx.last_kernel_ms()
                if r:
                    t[k].append(bm)
        print(f"--- {m:.0e} sites: median / min build-kernel ms, GB/s, % of 8 TB/s")
        for k in variants:
            med = float(np.median(t[k]))
            print(f"{k:>56}: {med:.4f} / {min(t[k]):.4f}  {16 * m / med / 1e6:5.0f}  {16 * m / med / 1e6 / 80:.1f}")
    select({})
    ctx.close()


if __name__ == "__main__":
    main()
