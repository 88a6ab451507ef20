"""CPU: the multi-GPU shard plan, and the N>1 path over gloo with world_size 2."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plan_shards_partitions_windows(pgt):
    rng = np.random.default_rng(5)
    for W, S in [(50_000, 10_000), (1000, 1000), (300_000, 7)]:
        lens = rng.integers(1, 400_000, size=23).astype(np.uint64)
        win = pgt.build_windows_sites(lens, W, S)
        n = int(lens.sum())
        for R in (1, 2, 3, 4, 8):
            sh = pgt.plan_shards(win, R)
            assert sh[0]["win_begin"] == 0 and sh[-1]["win_end"] == win.size
            assert np.all(sh["win_begin"][1:] == sh["win_end"][:-1])
            for s in sh:
                if s["win_end"] == s["win_begin"]:
                    continue
                blk = win[int(s["win_begin"]): int(s["win_end"])]
                assert s["site_lo"] <= blk["lo"].min() and s["site_hi"] >= blk["hi"].max()
                assert s["site_lo"] % 65536 == 0 and s["site_hi"] <= n
            if R > 1 and win.size > 50 * R:
                spans = (sh["site_hi"] - sh["site_lo"]).astype(np.float64)
                assert spans.max() < 1.6 * n / R + 2 * max(W, 1 << 19)


def test_plan_shards_alignment_follows_window_length(pgt):
    lens = np.array([5_000_000], dtype=np.uint64)
    win = pgt.build_windows_sites(lens, 1_000_000, 250_000)  # windows contain 2^19-site tree nodes
    sh = pgt.plan_shards(win, 4)
    for s in sh:
        assert s["site_lo"] % (1 << 19) == 0


def test_table_hints_of_site_and_ragged_tables(pgt):
    """pgt_table_hints: (W, W, S) for a site-window table whatever its chromosome layout; medians for a ragged table;
    an explicit 'no step' (2^64-1, not 0 = unknown) where windows share their starts."""
    from popgenomicstools_amd.window_scan import table_hints
    rng = np.random.default_rng(8)
    for W, S in [(50_000, 10_000), (1000, 100), (300, 1)]:
        lens = rng.integers(2 * W, 40 * W, size=9).astype(np.uint64)
        win = pgt.build_windows_sites(lens, W, S)
        assert table_hints(win) == (W, W, S)
    win = np.zeros(5, dtype=win.dtype)
    win["lo"] = [0, 10, 30, 60, 100]
    win["hi"] = [40, 90, 35, 200, 100]
    assert table_hints(win) == (140, 80, 30)  # longest; upper medians of the lengths of windows 0..3 (5, 40, 80, 140) and of their start distances (10, 20, 30, 40)
    win["lo"] = 7
    win["hi"] = 7
    win["flags"] = 1
    assert table_hints(win) == (1, 0, 2**64 - 1)
    assert table_hints(win[:0]) == (1, 0, 2**64 - 1)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,port", [(2, 29531), (3, 29532)])
def test_gloo_ranks_match_single(tmp_path, world, port):
    """world_size 2 and 3 over gloo on CPU: each rank reduces its shard (the per-rank reduce is played
    by a numpy loop here — the GPU is absent), rows travel to the destination rank through the
    product's RowExchange (gather transport), and the assembled table equals the oracle's; also a
    rank without windows, two tables per window, and (3 ranks) a sub-group with dst != global rank 0."""
    script = os.path.join(ROOT, "tests", "gloo_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), script, str(tmp_path)],
                       capture_output=True, text=True, env=env, timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    for tag in ["GLOO_OK sharded_scan", "GLOO_OK tables", "GLOO_OK empty-shard"] + (["GLOO_OK subgroup"] if world >= 3 else []):
        assert tag in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.timeout(120)
def test_row_exchange_gathers_to_itself_on_one_rank():
    """RowExchange(gather_to_self=True) on a group of ONE rank (gloo here, RCCL in bench.py's exchange_overhead leg): the
    double-buffered asynchronous gather runs as it does between GPUs and delivers each scan's rows; without the flag one
    rank keeps the 'local' mode.  In a child process: the process group is process-wide state."""
    code = r"""
import numpy as np, torch, torch.distributed as dist
from popgenomicstools_amd.distributed import RowExchange
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29541", rank=0, world_size=1)
cpu = torch.device("cpu")
for tables in (1, 3):
    counts = [7 * tables]
    assert RowExchange(None, counts, 40, cpu, tables=tables).mode == "local"
    ex = RowExchange(None, counts, 40, cpu, tables=tables, gather_to_self=True)
    assert ex.mode == "gather"
    for k in range(5):
        out = ex.begin()
        assert out.numel() == 7 * tables * 40
        out.copy_(torch.full((out.numel(),), k + 1, dtype=torch.uint8))
        ex.end()
    got = ex.finish()
    assert got.shape == (7 * tables * 40,) and np.all(got == 5), got[:8]
    ex.close()
dist.destroy_process_group()
print("SELF_GATHER_OK")
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, PYTHONPATH=ROOT), timeout=100)
    assert r.returncode == 0 and "SELF_GATHER_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
