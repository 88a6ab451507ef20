"""Where does the fixed cost of a build launch go?  Tuning build only (PGT_EXTRA_HIPCC_FLAGS=-DPGT_TUNING_BUILD):
runs the stamped copy of the fst build kernel (tools/pgt_build_experiments.inc) and prints per input size the distribution over waves of the end time (relative to the earliest wave start,
microseconds, s_memrealtime at 100 MHz), the tiles a wave took, and the mean end time per XCD.

    python tools/wave_timeline.py [sizes...]      default 1e8 1.25e8 1e9"""
import os, sys, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import popgenomicstools_amd as pgt
from popgenomicstools_amd._lib import PGT_STAT_FST
from popgenomicstools_amd.window_scan import windows_to_device

sizes = [int(float(x)) for x in sys.argv[1:]] or [100_000_000, 125_000_000, 1_000_000_000]
dev = torch.device("cuda", 0)
ctx = pgt.Context(0)
ctx.set_max_window(50_000)
path = os.path.join(tempfile.mkdtemp(), "stamps.bin")
print("| sites | mode | waves | tiles/wave min/mean/max | first tile p50 | end p1 / p50 / p99 / max (us) | kernel ms (events) | % of 8 TB/s | mean end per XCD (us) |")
print("|---|---|---|---|---|---|---|---|---|")
for n in sizes:
    a = torch.rand(n, dtype=torch.float64, device=dev)
    b = torch.rand(n, dtype=torch.float64, device=dev)
    pos = torch.arange(n, dtype=torch.int32, device=dev)
    win = windows_to_device(pgt.build_windows_sites(np.array([n], dtype=np.uint64), 50_000, 10_000), dev)
    tree = torch.empty(ctx.tree_bytes(PGT_STAT_FST, n), dtype=torch.uint8, device=dev)
    ref = None
    for mode in ("product", "stamped"):
        if mode == "product":
            os.environ.pop("PGT_TUNE_BUILD_STAMPS", None)
            ctx.set_profiling(True)
            ev = []
            for _ in range(8):
                out, _ = ctx.fst_reduce_dev(pos, a, b, win, tree=tree)
                ev.append(ctx.last_kernel_ms()[0])
            ctx.set_profiling(False)
            ref = out.clone()
            ms = float(np.median(ev[2:]))
            print(f"| {n:.3g} | product (unstamped) | | | | | {ms:.4f} | {16.0 * n / ms / 1e6 / 80:.1f} | |", flush=True)
            continue
        os.environ["PGT_TUNE_BUILD_STAMPS"] = path
        kms = []
        for _ in range(6):
            out, _ = ctx.fst_reduce_dev(pos, a, b, win, tree=tree)
            torch.cuda.synchronize()
            raw = np.fromfile(path, dtype=np.uint64)
            kms.append(float(raw[2:3].view(np.float64)[0]))
        same = bool(torch.equal(out, ref))
        nw = int(raw[0])
        st = raw[3:].reshape(nw, 6)
        t = st[:, :4].astype(np.int64)
        us = (t - t[:, 0].min()) / 100.0
        done = st[:, 4].astype(np.int64)
        xcc = (st[:, 5] >> np.uint64(32)).astype(np.int64) & 0xF
        q = lambda x, p: float(np.percentile(x, p))
        per_xcd = " ".join(f"{us[xcc == k, 3].mean():.0f}" for k in range(8) if (xcc == k).any())
        ms = float(np.median(kms[1:]))
        print(f"| {n:.3g} | {mode}{'' if same is None else (' (rows = product)' if same else ' (ROWS DIFFER)')} | {nw} | {done.min()} / {done.mean():.2f} / {done.max()} | {q(us[:,1],50):.1f} | "
              f"{q(us[:,3],1):.1f} / {q(us[:,3],50):.1f} / {q(us[:,3],99):.1f} / {us[:,3].max():.1f} | {ms:.4f} | {16.0 * n / ms / 1e6 / 80:.1f} | {per_xcd} |", flush=True)
    os.environ.pop("PGT_TUNE_BUILD_STAMPS", None)
    del a, b, pos, tree
