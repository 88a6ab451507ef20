"""AF build kernel timing at 10^8 sites (history: the first version with scattered per-tile stores
measured 59.6 % at 8 populations, a timing-only build without them 76.9 %)."""
import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import popgenomicstools_amd as pgt
from popgenomicstools_amd.window_scan import windows_to_device
dev = torch.device("cuda", 0); n = 100_000_000
gen = torch.Generator(device=dev).manual_seed(3)
pos = torch.arange(n, dtype=torch.int32, device=dev)
fr = [torch.round(torch.rand(n, generator=gen, device=dev, dtype=torch.float64) * 1e6) / 1e6 for _ in range(8)]
win = windows_to_device(pgt.build_windows_sites(np.full(20, n // 20, dtype=np.uint64), 50_000, 10_000), dev)
ctx = pgt.Context(0); ctx.set_max_window(50_000); ctx.set_profiling(True)
tree = torch.empty(int(pgt._lib.load().pgt_af_tree_bytes(8, n)), dtype=torch.uint8, device=dev)
out = torch.empty(28 * 40 * (win.numel() // 32), dtype=torch.uint8, device=dev)
names = {0: "product", 1: "no reduce-scatter (timing only)", 2: "no level-1 stores (timing only)", 3: "no per-site arithmetic (timing only)",
         20: "128-site leaves, 64 per level-2 tile (the layout before; timing only)"}
variants = ([int(x) for x in os.environ.get("AF_VARIANTS", "0,1,2,3").split(",")]
            if "-DPGT_TUNING_BUILD" in os.environ.get("PGT_EXTRA_HIPCC_FLAGS", "") else [0])
caps = [int(x) for x in os.environ.get("AF_CAPS", "2048").split(",")]
for npop, cap in [(p, c) for p in (8, 4, 2) for c in caps]:
    os.environ["PGT_AF_CAP"] = str(cap)
    print(f"--- grid cap {cap} workgroups")
    t = {v: [] for v in variants}
    tq = []
    for r in range(9):
        for ab in variants:
            os.environ["PGT_AF_ABLATE"] = str(ab)
            ctx.fst_af_reduce_dev(pos, fr[:npop], [10.0 + k for k in range(npop)], win, out=out, tree=tree)
            bm, qm = ctx.last_kernel_ms()
            if r: t[ab].append(bm)
            if r and ab == 0: tq.append(qm)
    print(f"NP={npop} query kernel (product): {float(np.median(tq)):.4f} ms")
    for ab in variants:
        med = float(np.median(t[ab])); print(f"NP={npop} {names[ab]}: {med:.4f} ms  {8*npop*n/med/1e6:.0f} GB/s  {8*npop*n/med/1e6/80:.1f} %")
