"""world_size-2 worker for tests/test_shards.py (gloo, CPU).  The per-rank reduce is a plain
sequential numpy sum standing in for the GPU (absent here); everything around it — shard plan,
re-based window tables, halo columns, the gather — is the product's multi-GPU path."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_bind  # noqa: E402
import synth  # noqa: E402
from popgenomicstools_amd import build_windows_sites, run_lengths  # noqa: E402
from popgenomicstools_amd._lib import FST_ROW_DTYPE  # noqa: E402
from popgenomicstools_amd.distributed import RowGatherer, gather_rows, shard_windows, sharded_scan  # noqa: E402


def cpu_reduce(pos, a, b, win):
    rows = np.zeros(win.size, dtype=FST_ROW_DTYPE)
    for i, w in enumerate(win):
        lo, hi = int(w["lo"]), int(w["hi"])
        asum = bsum = 0.0
        for k in range(lo, hi):
            asum += a[k]
            bsum += b[k]
        rows[i] = (pos[lo], pos[hi - 1], (int(pos[lo]) + int(pos[hi - 1])) % 2**32 // 2, hi - lo,
                   asum / bsum if bsum != 0 else 0.0, asum, bsum)
    return rows


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    rng = np.random.default_rng(77)
    n, W, S = 300_000, 5_000, 2_000
    chr_ids, pos = synth.chromosomes(rng, n, 7, equal=False)
    a, b = synth.fst_columns(rng, n)
    win = build_windows_sites(run_lengths(chr_ids), W, S)
    shard, local_win, shards = shard_windows(win, rank, world)
    lo, hi = int(shard["site_lo"]), int(shard["site_hi"])
    rows = cpu_reduce(pos[lo:hi], a[lo:hi], b[lo:hi], local_win)  # this rank only touches its columns
    counts = (shards["win_end"] - shards["win_begin"]).astype(np.int64)
    packed = torch.from_numpy(rows.view(np.uint8).copy())
    allrows = gather_rows(packed, counts, FST_ROW_DTYPE.itemsize, dst=0)
    parts = RowGatherer(counts, FST_ROW_DTYPE.itemsize, torch.device("cpu"), dst=0)(packed)  # the bench's form
    if rank == 0:
        assert torch.equal(torch.cat(parts), allrows)
        got = np.frombuffer(allrows.numpy().tobytes(), dtype=FST_ROW_DTYPE)
        ref = oracle_bind.load().fst_scan(chr_ids, pos, a, b, W, S)
        assert got.size == ref.size == win.size
        assert np.array_equal(got["start"], ref["start"]) and np.array_equal(got["end"], ref["end"])
        assert np.array_equal(got["mid"], ref["mid"]) and np.array_equal(got["n"], ref["n"])
        assert np.array_equal(got["fst"], ref["value"])  # same sequential order -> same bits
        assert counts.min() > 0
        print("GLOO_OK", win.size, counts.tolist())
    # the one-call form of the same path
    got2 = sharded_scan(win, FST_ROW_DTYPE, lambda lo_, hi_: (pos[lo_:hi_], a[lo_:hi_], b[lo_:hi_]),
                        lambda c, w: torch.from_numpy(cpu_reduce(*c, w).view(np.uint8).copy()), torch.device("cpu"))
    if rank == 0:
        assert got2.tobytes() == got.tobytes()
        print("GLOO_OK sharded_scan")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
