"""Does replaying a captured HIP graph of one step (build + upper levels + query) beat launching the step's kernels
one by one?  Whole-step time per size, eager against graph replay, on the product build."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '/root/repo')
import popgenomicstools_amd as pgt
from popgenomicstools_amd._lib import PGT_STAT_FST, FST_ROW_DTYPE
from popgenomicstools_amd.window_scan import windows_to_device
from synth_genome import SynthGenome

dev = torch.device("cuda", 0)
ctx = pgt.Context(0)
ctx.set_max_window(50_000)
print("| sites | eager ms/step | graph replay ms/step | gain |\n|---|---|---|---|")
for n in [int(float(x)) for x in (sys.argv[1:] or ["1e8", "1.25e8", "1e9"])]:
    g = SynthGenome(12345, n, 40 if n >= 500_000_000 else 20)
    pos, a, b = g.fst_columns_t(0, n, dev)
    win = windows_to_device(pgt.build_windows_sites(g.run_len, 50_000, 10_000), dev)
    out = torch.empty(win.numel() // 32 * FST_ROW_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    tree = torch.empty(ctx.tree_bytes(PGT_STAT_FST, n), dtype=torch.uint8, device=dev)
    step = lambda: ctx.fst_reduce_dev(pos, a, b, win, out=out, tree=tree)  # noqa: E731
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    res = {}
    K = 300 if n <= 200_000_000 else 60
    for rep in range(3):
        for name, fn in (("eager", step), ("graph", graph.replay)):
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(K):
                fn()
            torch.cuda.synchronize()
            res.setdefault(name, []).append((time.perf_counter() - t) / K * 1e3)
    e, r = float(np.median(res["eager"])), float(np.median(res["graph"]))
    print(f"| {n:.3g} | {e:.4f} | {r:.4f} | {100 * (e / r - 1):+.1f} % |", flush=True)
    del pos, a, b, win, out, tree, graph
    torch.cuda.empty_cache()
