"""AF build kernel timing at 10^8 sites, 8 / 4 / 2 populations (product library).
History (profiles/r02/af_*.txt, taken at commits that still carried the variants as template modes of the product
kernel; they were removed from csrc/ in round 3): scattered per-tile stores 59.6 % at 8 populations, a timing-only
build without level-1 stores 76.9 %, 128-site leaves 64-68 % against 70-77 % for the 256-site leaves of the product."""
import sys, numpy as np, torch
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
for npop in (8, 4, 2):
    t, tq = [], []
    for r in range(9):
        ctx.fst_af_reduce_dev(pos, fr[:npop], [10.0 + k for k in range(npop)], win, out=out, tree=tree)
        bm, qm = ctx.last_kernel_ms()
        if r:
            t.append(bm); tq.append(qm)
    med = float(np.median(t))
    print(f"NP={npop} build {med:.4f} ms  {8*npop*n/med/1e6:.0f} GB/s  {8*npop*n/med/1e6/80:.1f} %   query {float(np.median(tq)):.4f} ms")
