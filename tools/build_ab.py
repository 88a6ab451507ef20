#!/usr/bin/env python3
"""Tuning build only: variants of the streaming build launches against the product, interleaved in ONE process
(box-to-box and process-to-process differences on this pool exceed most effects), rows checked identical.

    PGT_EXTRA_HIPCC_FLAGS=-DPGT_TUNING_BUILD python tools/build_ab.py [sizes ...]
    AB_VARIANTS="name:ENV=V,ENV=V;name:ENV=V"   variants (default: the deeper queue in a wave's last tile)
    AB_CONFIGS=fst,dxy,fused,ext                 which build kernels
    AB_LIB=tools/_ab/libpgtwin_tuning.so         load this (prebuilt) tuning library instead of rebuilding the tree's

Environment knobs of the tuning build (tools/pgt_kernels_tuning.hip): PGT_TUNE_FST =
<stage>:<loads in flight>:<workgroups>[:<tail loads>:<tail scope>] (fst build), PGT_TUNE_FST_PAIR = <loads in flight> (fst build, two
waves per tile), PGT_EXT_VARIANT_NOW = 0..10 (extreme-score build)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import popgenomicstools_amd as pgt  # noqa: E402
from popgenomicstools_amd import _lib  # noqa: E402
from popgenomicstools_amd._lib import PGT_EXT_IHS, PGT_STAT_FST  # noqa: E402

if os.environ.get("AB_LIB"):  # a tuning library built beforehand (e.g. tools/_ab/libpgtwin_tuning.so, cross-compiled where there is no GPU)
    _lib.LIB_PATH = os.path.abspath(os.environ["AB_LIB"])
from popgenomicstools_amd.window_scan import windows_to_device  # noqa: E402
from synth_genome import SynthGenome  # noqa: E402

KNOBS = ("PGT_TUNE_FST", "PGT_TUNE_FST_PAIR", "PGT_EXT_VARIANT_NOW")
sizes = [int(float(x)) for x in sys.argv[1:]] or [100_000_000, 125_000_000, 1_000_000_000]
spec = os.environ.get("AB_VARIANTS", "last tile with 16 loads in flight:PGT_TUNE_FST=16:4:512:16:2;"
                                      "last b column with 16 loads in flight:PGT_TUNE_FST=16:4:512:16:1")
variants = [("product", {})]
for item in spec.split(";"):
    name, _, envs = item.partition(":")
    variants.append((name, dict(kv.split("=", 1) for kv in envs.split(",") if kv)))
want = os.environ.get("AB_CONFIGS", "fst").split(",")
ROUNDS = int(os.environ.get("AB_ROUNDS", "12"))

dev = torch.device("cuda", 0)
ctx = pgt.Context(0)
ctx.set_profiling(True)
print(f"{ROUNDS} interleaved rounds per variant; build phase by the library's HIP events (median); % of 8 TB/s on the algorithmic bytes\n")
print("| sites | build | variant | build ms | % of 8 TB/s | vs product | rows = product |")
print("|---|---|---|---|---|---|---|")
for n in sizes:
    g = SynthGenome(12345, n, 40 if n > 200_000_000 else 20)
    pos, a, b = g.fst_columns_t(0, n, dev)
    W, S = 50_000, 10_000
    win = windows_to_device(pgt.build_windows_sites(g.run_len, W, S), dev)
    tree = torch.empty(2 * ctx.tree_bytes(PGT_STAT_FST, n), dtype=torch.uint8, device=dev)
    out = torch.empty(4 * 40 * (win.numel() // 32), dtype=torch.uint8, device=dev)
    configs = []
    if "fst" in want:
        configs.append(("fst", 16, lambda: ctx.fst_reduce_dev(pos, a, b, win, out=out, tree=tree), 50_000))
    if "dxy" in want or "fused" in want:
        p1, p2, n1, n2 = g.dxy_columns_t(0, n, dev)
    if "dxy" in want:
        configs.append(("dxy", 24, lambda: ctx.dxy_reduce_dev(pos, p1, p2, n1, n2, 5, win, out=out, tree=tree), 50_000))
    if "fused" in want:
        g1, g2 = g.genotype_t(0, 0, n, dev), g.genotype_t(1, 0, n, dev)
        configs.append(("dxy + het x2 fused", 26, lambda: ctx.dxy_het_reduce_dev(pos, p1, p2, n1, n2, g1, g2, 5, win, tree=tree), 50_000))
    if "ext" in want:
        score = a * 40.0 - 2.0
        ewin_h = pgt.build_windows_extreme(pos.cpu().numpy().view(np.uint32), g.run_len, None, 100_000)
        ewin = windows_to_device(ewin_h, dev)
        emax = int((ewin_h["hi"] - ewin_h["lo"]).max())
        configs.append(("extreme score", 8, lambda: ctx.extreme_reduce_dev(pos, score, PGT_EXT_IHS, 2.0, ewin, out=out, tree=tree), emax))
    for cname, bps, call, maxw in configs:
        ctx.set_max_window(maxw)
        t = {name: [] for name, _ in variants}
        same, ref = {}, None
        for r in range(ROUNDS + 1):
            for name, env in variants:
                for k in KNOBS:
                    os.environ.pop(k, None)
                os.environ.update(env)
                res = call()
                rows = res[0] if isinstance(res, tuple) else res
                bm, _ = ctx.last_kernel_ms()
                if r:
                    t[name].append(bm)
                else:
                    rows = rows[0] if isinstance(rows, (tuple, list)) else rows
                    if ref is None:
                        ref = rows.clone()
                    same[name] = bool(torch.equal(rows, ref))
        base = float(np.median(t["product"]))
        for name, _ in variants:
            med = float(np.median(t[name]))
            print(f"| {n:.3g} | {cname} | {name} | {med:.4f} | {bps * n / med / 1e6 / 80:.1f} | {100 * (base / med - 1):+.1f} % | {same[name]} |", flush=True)
    for k in KNOBS:
        os.environ.pop(k, None)
    del pos, a, b, tree, out, configs
    torch.cuda.empty_cache()
