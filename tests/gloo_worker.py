"""world_size-N worker for tests/test_shards.py (gloo, CPU).  The per-rank reduce is a plain
sequential numpy sum standing in for the GPU (absent here); everything around it — shard plan,
re-based window tables, halo columns, the row exchange (gather transport) — is the product's
multi-GPU path.  Covered: uneven blocks, a rank without windows (world > #windows), several tables
per window (population pairs), and a sub-group whose `dst` is not global rank 0."""
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


def put(out, rows):
    out.copy_(torch.from_numpy(rows.view(np.uint8).copy()))


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    cpu = torch.device("cpu")
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
    parts = RowGatherer(counts, FST_ROW_DTYPE.itemsize, cpu, dst=0)(packed)
    ref = oracle_bind.load().fst_scan(chr_ids, pos, a, b, W, S)
    if rank == 0:
        assert torch.equal(torch.cat(parts), allrows)
        got = np.frombuffer(allrows.numpy().tobytes(), dtype=FST_ROW_DTYPE)
        assert got.size == ref.size == win.size
        assert np.array_equal(got["start"], ref["start"]) and np.array_equal(got["end"], ref["end"])
        assert np.array_equal(got["mid"], ref["mid"]) and np.array_equal(got["n"], ref["n"])
        assert np.array_equal(got["fst"], ref["value"])  # same sequential order -> same bits
        assert counts.min() > 0
        print("GLOO_OK", win.size, counts.tolist())

    # the one-call form of the same path (RowExchange, gather transport)
    load = lambda lo_, hi_: (pos[lo_:hi_], a[lo_:hi_], b[lo_:hi_])  # noqa: E731
    got2 = sharded_scan(win, FST_ROW_DTYPE, load, lambda c, w, out: put(out, cpu_reduce(*c, w)), cpu)
    if rank == 0:
        assert got2.tobytes() == got.tobytes()
        print("GLOO_OK sharded_scan")

    # two tables per window (population pairs): table 1 = the same statistic on (b, a)
    def two(c, w, out):
        p_, a_, b_ = c
        put(out, np.concatenate([cpu_reduce(p_, a_, b_, w), cpu_reduce(p_, b_, a_, w)]))
    got3 = sharded_scan(win, FST_ROW_DTYPE, load, two, cpu, tables=2)
    if rank == 0:
        assert got3.size == 2 * win.size and got3[: win.size].tobytes() == got.tobytes()
        assert np.array_equal(got3[win.size:]["asum"], got["bsum"]) and np.array_equal(got3[win.size:]["n"], got["n"])
        print("GLOO_OK tables")

    # more ranks than windows: some rank owns nothing, must not call load/reduce and must not hang the others
    few = win[:1] if world >= 2 else win
    calls = []

    def guarded(c, w, out):
        calls.append(len(w))
        put(out, cpu_reduce(*c, w))
    got4 = sharded_scan(few, FST_ROW_DTYPE, load, guarded, cpu)
    _, lw, _ = shard_windows(few, rank, world)
    assert len(calls) == (1 if lw.size else 0)
    if rank == 0:
        assert got4.tobytes() == got[: few.size].tobytes()
        print("GLOO_OK empty-shard")

    # a sub-group whose destination is its LAST member (group rank != global rank)
    if world >= 3:
        members = list(range(1, world))
        grp = dist.new_group(members)
        if rank in members:
            gdst = len(members) - 1
            got5 = sharded_scan(win, FST_ROW_DTYPE, load, lambda c, w, out: put(out, cpu_reduce(*c, w)), cpu, dst=gdst, group=grp)
            if dist.get_rank(grp) == gdst:
                assert np.array_equal(got5["fst"], ref["value"]) and np.array_equal(got5["n"], ref["n"])
                print("GLOO_OK subgroup")
            else:
                assert got5 is None
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
