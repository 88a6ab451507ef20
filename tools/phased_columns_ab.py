"""Tuning build only: the fst build with ONE column stream at a time per wave (PGT_TUNE_BUILD_PHASED_COLUMNS =
"<loads in flight>:<leaves per phase>:<stage tiles>:<workgroups>"; 4:64:16:512 is the product since round 3) against the
product kernel and the interleaved kernel of rounds 1-2 (PGT_TUNE_BUILD_INTERLEAVED), in one process, rows checked identical.
The files profiles/r03/phased_columns_ab_*.md were taken while the product still WAS the interleaved kernel.
    PGT_EXTRA_HIPCC_FLAGS=-DPGT_TUNING_BUILD python tools/phased_columns_ab.py [sizes]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import popgenomicstools_amd as pgt
from popgenomicstools_amd._lib import PGT_STAT_FST
from popgenomicstools_amd.window_scan import windows_to_device
sizes = [int(float(x)) for x in sys.argv[1:]] or [100_000_000, 125_000_000, 1_000_000_000]
dev = torch.device("cuda", 0)
ctx = pgt.Context(0); ctx.set_max_window(50_000); ctx.set_profiling(True)
variants = [("product", None), ("interleaved a/b, 8 loads in flight (rounds 1-2, long inputs)", "I4"),
            ("interleaved a/b, 16 loads in flight (rounds 1-2, short inputs)", "I8")] + [("phased u:phase:stage:blocks = " + v, v)
                                 for v in os.environ.get("PHASED_VARIANTS", "4:64:16:512,2:64:16:512,1:64:16:512,4:32:16:512,2:32:16:512,4:64:8:1024,2:64:8:1024,4:64:4:2048,2:64:4:2048,8:64:16:512").split(",")]
print("| sites | variant | build ms (median of 12) | % of 8 TB/s | rows = product |")
print("|---|---|---|---|---|")
for n in sizes:
    a = torch.rand(n, dtype=torch.float64, device=dev); b = torch.rand(n, dtype=torch.float64, device=dev)
    pos = torch.arange(n, dtype=torch.int32, device=dev)
    win = windows_to_device(pgt.build_windows_sites(np.array([n], dtype=np.uint64), 50_000, 10_000), dev)
    tree = torch.empty(ctx.tree_bytes(PGT_STAT_FST, n), dtype=torch.uint8, device=dev)
    t = {name: [] for name, _ in variants}
    ref, same = None, {}
    for r in range(13):
        for name, env in variants:
            os.environ.pop("PGT_TUNE_BUILD_PHASED_COLUMNS", None)
            os.environ.pop("PGT_TUNE_BUILD_INTERLEAVED", None)
            if env is not None and env.startswith("I"): os.environ["PGT_TUNE_BUILD_INTERLEAVED"] = env[1:]
            elif env is not None: os.environ["PGT_TUNE_BUILD_PHASED_COLUMNS"] = env
            out, _ = ctx.fst_reduce_dev(pos, a, b, win, tree=tree)
            bm, _ = ctx.last_kernel_ms()
            if r: t[name].append(bm)
            if r == 0:
                if ref is None: ref = out.clone()
                same[name] = bool(torch.equal(out, ref))
    for name, _ in variants:
        med = float(np.median(t[name]))
        print(f"| {n:.3g} | {name} | {med:.4f} | {16.0 * n / med / 1e6 / 80:.1f} | {same[name]} |", flush=True)
    del a, b, pos, tree
