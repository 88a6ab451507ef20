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


@pytest.mark.timeout(300)
def test_two_rank_gloo_gather_matches_single(tmp_path):
    """world_size 2 over gloo on CPU: each rank reduces its shard (the per-rank reduce is played by
    the oracle here — the GPU is absent), rows are gathered to rank 0 exactly as bench.py does,
    and the assembled table equals the single-rank one."""
    script = os.path.join(ROOT, "tests", "gloo_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29531", script, str(tmp_path)],
                       capture_output=True, text=True, env=env, timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "GLOO_OK" in r.stdout
