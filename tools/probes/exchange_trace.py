#!/usr/bin/env python3
"""The 8-GPU shard's step with the rows left in place and with the one-rank RCCL gather, for a kernel trace:

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/xchg -o x -- python3 tools/probes/exchange_trace.py

40 steps 'local', a marker (torch.zeros fill kernel of 12345 elements), 40 steps 'gather'.  What to read off the trace: the
build kernel's duration in both halves, the RCCL kernel's name / duration / where it sits relative to the next build."""
import os
import socket
import sys
from datetime import timedelta

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import popgenomicstools_amd as pgt  # noqa: E402
from popgenomicstools_amd._lib import FST_ROW_DTYPE, PGT_STAT_FST, WIN_DTYPE  # noqa: E402
from popgenomicstools_amd.distributed import RowExchange  # noqa: E402
from popgenomicstools_amd.window_scan import windows_to_device  # noqa: E402
from synth_genome import SynthGenome  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, timeout=timedelta(seconds=60))
    group = dist.new_group(backend="nccl")
    x = torch.ones(1, device=dev)
    dist.all_reduce(x, group=group)
    n, W, S = 125_040_000, 50_000, 10_000
    g = SynthGenome(12345, 1_000_000_000, 40)
    win = pgt.build_windows_sites(g.run_len, W, S)
    sh = pgt.plan_shards(win, 8)[0]
    hi = int(sh["site_hi"])
    loc = np.array(win[int(sh["win_begin"]): int(sh["win_end"])], dtype=WIN_DTYPE, copy=True)
    pos, a, b = g.fst_columns_t(0, hi, dev)
    wd = windows_to_device(loc, dev)
    ctx = pgt.Context(0)
    ctx.set_max_window(W)
    ctx.set_window_step(S)
    tree = torch.empty(ctx.tree_bytes(PGT_STAT_FST, hi), dtype=torch.uint8, device=dev)
    for to_self in (False, True):
        ex = RowExchange(ctx, [loc.size], FST_ROW_DTYPE.itemsize, dev, group=group, mode="gather", coll_device=dev, gather_to_self=to_self)
        for _ in range(steps):
            out = ex.begin()
            ctx.fst_reduce_dev(pos, a, b, wd, out=out, tree=tree)
            ex.end()
        ex.flush()
        ex.finish()
        ex.close()
        torch.zeros(12345, device=dev)  # marker between the halves
        torch.cuda.synchronize()
    ctx.close()
    dist.destroy_process_group()
    print("ok")


if __name__ == "__main__":
    main()
