#!/usr/bin/env python3
"""Does the build kernel's rate depend on WHERE inside an allocation its columns start?  (round 6, the two modes of item 2)
One process, one 10^9-site genome's worth of a / b values written at several byte offsets inside two big buffers; the fst build
timed at each (events around 12 back-to-back launches, twice, interleaved).  If the rate moved with the offset, fine-grained
placement (which HBM channels the waves of a round hit together) would be a candidate for the process-to-process modes."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import popgenomicstools_amd as pgt  # noqa: E402
from popgenomicstools_amd._lib import PGT_STAT_FST  # noqa: E402
from popgenomicstools_amd.window_scan import windows_to_device  # noqa: E402


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
    dev = torch.device("cuda", 0)
    ctx = pgt.Context(0)
    ctx.set_max_window(50_000)
    slack = 8 << 20
    bufa = torch.empty(8 * n + slack, dtype=torch.uint8, device=dev)
    bufb = torch.empty(8 * n + slack, dtype=torch.uint8, device=dev)
    src = torch.rand(n, dtype=torch.float64, device=dev)
    pos = torch.arange(1, n + 1, dtype=torch.int32, device=dev)
    tree = torch.empty(ctx.tree_bytes(PGT_STAT_FST, n), dtype=torch.uint8, device=dev)
    nowin = windows_to_device(pgt.build_windows_sites(np.array([n], dtype=np.uint64), 50_000, 10_000), dev)[:0]
    print(f"{n:.0e} sites; buffer a at 0x{bufa.data_ptr():x}, b at 0x{bufb.data_ptr():x}, tree at 0x{tree.data_ptr():x}")
    offs = [(0, 0), (4096, 4096), (16384, 16384), (65536, 65536), (262144, 262144), (1 << 20, 1 << 20), (2 << 20, 2 << 20), (3 << 20, 3 << 20),
            (0, 4096), (0, 16384), (0, 65536), (0, 1 << 20), (0, 4 << 20), (4 << 20, 0)]
    res = {o: [] for o in offs}
    for rep in range(2):
        for oa, ob in offs:
            a = bufa[oa: oa + 8 * n].view(torch.float64)
            b = bufb[ob: ob + 8 * n].view(torch.float64)
            a.copy_(src)
            b.copy_(src)
            for _ in range(3):
                ctx.fst_reduce_dev(pos, a, b, nowin, tree=tree)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(12):
                ctx.fst_reduce_dev(pos, a, b, nowin, tree=tree)
            e1.record()
            e1.synchronize()
            res[(oa, ob)].append(e0.elapsed_time(e1) / 12)
    print("| offset of a (bytes) | offset of b | build ms (1st pass) | build ms (2nd pass) | % of 8 TB/s (mean) |\n|---|---|---|---|---|")
    for (oa, ob), v in res.items():
        print(f"| {oa} | {ob} | {v[0]:.4f} | {v[1]:.4f} | {16.0 * n / (sum(v) / len(v) * 1e-3) / 8e12 * 100:.2f} |")
    ctx.close()


if __name__ == "__main__":
    main()
