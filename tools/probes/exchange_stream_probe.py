#!/usr/bin/env python3
"""Does the per-step cost of the one-rank RCCL gather depend on WHICH stream the scan is launched on?  (round 6)
The kernel trace (profiles/r06/exchange_trace_summary.md) shows the gather's copy kernel serialised between query k and build
k+1 although it sits on torch's own NCCL stream: as if both streams fed one hardware queue.  Here: the 8-GPU shard's step,
local vs gather, with the scan on torch's default (NULL) stream and on a stream of its own.

    python3 tools/probes/exchange_stream_probe.py [steps]"""
import os
import socket
import sys
import time
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
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, timeout=timedelta(seconds=60))
    group = dist.new_group(backend="nccl")
    x = torch.ones(1, device=dev)
    dist.all_reduce(x, group=group)
    W, S = 50_000, 10_000
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
    side = [torch.cuda.Stream(device=dev) for _ in range(6)]
    print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', '(default)')}; {steps} steps per figure, {hi} sites, {loc.size} rows per step")

    def run(to_self, stream):
        ex = RowExchange(ctx, [loc.size], FST_ROW_DTYPE.itemsize, dev, group=group, mode="gather", coll_device=dev, gather_to_self=to_self)

        def loop(k):
            for _ in range(k):
                out = ex.begin()
                ctx.fst_reduce_dev(pos, a, b, wd, out=out, tree=tree)
                ex.end()
        ctxm = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.default_stream(dev))
        with ctxm:
            loop(30)
            ex.flush()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            loop(steps)
            ex.flush()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        ex.finish()
        ex.close()
        return dt / steps * 1e3

    for name, st in [("default (NULL) stream", None)] + [(f"own stream #{k}", side[k]) for k in range(6)]:
        res = []
        for _ in range(2):
            res.append((run(False, st), run(True, st)))
        l_, g_ = min(r[0] for r in res), min(r[1] for r in res)
        print(f"scan on the {name}: local {l_:.4f} ms/step, gather {g_:.4f} ms/step, overhead {1e3 * (g_ - l_):+.1f} us ({100 * (g_ - l_) / l_:+.2f} %)", flush=True)
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
