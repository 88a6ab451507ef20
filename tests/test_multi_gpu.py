"""GPU, boxes with AT LEAST TWO GPUs only (skipped on the one-GPU pool): the first code paths that really cross devices —
RCCL between two ranks on two GPUs, peer stores and hipIpc mappings over xGMI, device-to-device column copies and
PGT_DEVICES=0,1 in all five hosts.  Everything here also runs, with both ranks / contexts on GPU 0 and gloo for the
collectives, in the one-GPU suite (test_gpu_parity.py::test_multi_rank_hip_path_two_ranks_one_gpu, test_bench_script.py,
test_cli.py::*_several_gpus_*): these tests are the same drivers pointed at distinct devices, so that a driver box with more
than one GPU exercises SURVEY §8(e) on real hardware without being asked.

torch.cuda.device_count() does not initialise the GPU on this image; the ranks are child processes of launchers that never
touch it."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "popgenomicstools_amd", "bin")


def _n_gpus():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:  # noqa: BLE001
        return 0


# Rehearsal on a one-GPU box (how these tests were debugged; never set by the driver): PGT_TEST_PRETEND_TWO_GPUS=1 runs the
# same bodies with "GPU 1" = GPU 0 again and gloo in place of RCCL (which refuses two ranks on one device).
PRETEND = os.environ.get("PGT_TEST_PRETEND_TWO_GPUS") == "1"
SECOND = 0 if PRETEND else 1
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(_n_gpus() < 2 and not PRETEND, reason="needs two GPUs (the gpurun pool has one per box)")]


def _env(**kw):
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", **kw)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PGT_BENCH_BACKEND", "PGT_BENCH_SHARE_GPU"):
        env.pop(k, None)
    return env


@pytest.mark.timeout(900)
def test_bench_two_gpus_rccl_gather_and_peer_stores():
    """`python bench.py --gpus 2 --sites 2e8 --exchange both` exactly as a user would type it: two ranks on two GPUs, the row
    collectives on RCCL, both transports timed, each assembled table bitwise the single-GPU table of the same genome."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--sites", "2e8", "--chroms", "8", "--steps", "5",
                        "--warmup", "2", "--exchange", "both"], capture_output=True, text=True,
                       env=dict(_env(), PGT_BENCH_BACKEND="gloo", PGT_BENCH_SHARE_GPU="1") if PRETEND else _env(), timeout=850)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ok"] is True and line["degraded"] is False
    assert line["config"]["collective_backend"] == ("nccl (RCCL)" if not PRETEND else "gloo (REHEARSAL: PGT_BENCH_BACKEND)"), line["config"]["collective_backend"]
    assert line["rows_check"].startswith("bitwise equal")
    assert line["extra"]["exchange_gather"]["rows_check"] == "bitwise equal"
    assert line["extra"]["exchange_peer"].get("rows_check") == "bitwise equal", line["extra"]["exchange_peer"]
    assert len(set(line["config"]["sites_resident_per_gpu"])) >= 1 and sum(line["config"]["sites_resident_per_gpu"]) < 2.1e8


@pytest.mark.timeout(900)
def test_two_ranks_two_gpus_every_statistic_both_transports():
    """tests/hip_rank_worker.py with rank r on GPU r and the rows on an RCCL group: fst, batched pairs, the AF front end, the
    extreme-score scan, het, dxy rows + genome-wide line; gather and peer stores (hipIpc mapping of rank 0's buffer, stores
    across xGMI after the self-test); bytes equal to the single-GPU call."""
    script = os.path.join(ROOT, "tests", "hip_rank_worker.py")
    env = _env(MASTER_ADDR="127.0.0.1", MASTER_PORT="29571", **({} if PRETEND else {"PGT_TEST_DEVICE_PER_RANK": "1", "PGT_TEST_BACKEND": "nccl"}))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29571", script], capture_output=True, text=True, env=env, timeout=850)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    for tag in ("fst gather", "fst peer", "fst auto", "pairs gather", "pairs peer", "af peer", "extreme peer", "het peer", "dxy gather", "dxy peer"):
        assert "HIP_RANKS_OK " + tag in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_peer_access_stores_and_copies_between_device_0_and_1():
    """One process, one context per GPU: pgt_peer_access both ways; a kernel on GPU 1 stores the self-test pattern into a
    buffer that lives on GPU 0 (pgt_rowbuf_fill: what the query kernels do in peer mode) and GPU 0 reads it back;
    pgt_dev_copy moves a column GPU 0 -> GPU 1 and back; a scan on GPU 1 of columns copied from GPU 0 gives GPU 0's rows."""
    import torch
    import popgenomicstools_amd as pgt
    from popgenomicstools_amd import _lib
    from popgenomicstools_amd._lib import FST_ROW_DTYPE
    from popgenomicstools_amd.window_scan import RowBuffer, rows_from_device, windows_to_device
    sys.path.insert(0, ROOT)
    from synth_genome import SynthGenome

    c0, c1 = pgt.Context(0), pgt.Context(SECOND)
    c0.peer_access(SECOND)
    c1.peer_access(0)
    lib = _lib.load()
    d0, d1 = torch.device("cuda", 0), torch.device("cuda", SECOND)
    # stores across the link: GPU 1's kernel writes GPU 0's memory
    words = 1 << 16
    buf0 = torch.zeros(words, dtype=torch.int64, device=d0)
    torch.cuda.synchronize(d0)
    c1.rowbuf_fill(RowBuffer(buf0.data_ptr(), words * 8), 99)
    torch.cuda.synchronize(d1)
    assert np.array_equal(c0.rowbuf_read(RowBuffer(buf0.data_ptr(), words * 8)).view(np.uint64), pgt.Context.pattern_words(words, 99))
    # columns GPU 0 -> GPU 1 (pgt_dev_copy, what dxyWindow's several-GPU path does), the scan there, rows equal
    n, W, S = 3_000_000, 50_000, 10_000
    g = SynthGenome(77, n, 4)
    pos0, a0, b0 = g.fst_columns_t(0, n, d0)
    win = pgt.build_windows_sites(g.run_len, W, S)
    rows0 = rows_from_device(c0.fst_reduce_dev(pos0, a0, b0, windows_to_device(win, d0))[0], FST_ROW_DTYPE)
    pos1, a1, b1 = torch.empty_like(pos0, device=d1), torch.empty_like(a0, device=d1), torch.empty_like(b0, device=d1)
    torch.cuda.synchronize(d0)
    for dst, src in ((pos1, pos0), (a1, a0), (b1, b0)):
        assert lib.pgt_dev_copy(c1._ctx, dst.data_ptr(), c0._ctx, src.data_ptr(), src.numel() * src.element_size()) == _lib.PGT_OK
    assert torch.equal(a1.cpu(), a0.cpu()) and torch.equal(pos1.cpu(), pos0.cpu())
    rows1 = rows_from_device(c1.fst_reduce_dev(pos1, a1, b1, windows_to_device(win, d1))[0], FST_ROW_DTYPE)
    assert rows1.tobytes() == rows0.tobytes()
    back = torch.empty_like(b0)
    assert lib.pgt_dev_copy(c0._ctx, back.data_ptr(), c1._ctx, b1.data_ptr(), b1.numel() * 8) == _lib.PGT_OK
    assert torch.equal(back.cpu(), b0.cpu())
    # a tensor of the other GPU is refused by the wrapper before any launch
    if not PRETEND:
        with pytest.raises(_lib.PgtError):
            c1.fst_reduce_dev(pos0, a0, b0, windows_to_device(win, d1))
    c0.close()
    c1.close()


@pytest.mark.timeout(900)
def test_all_five_hosts_on_two_gpus_print_the_single_gpu_tsv(tmp_path):
    """PGT_DEVICES=0,1 (and 1,0): fstWindow, hetWindow, dxyWindow, ihsWindow, xpehhWindow — host parser and device parser —
    print the bytes of the single-GPU run: reference-made goldens plus one table per tool large enough for every GPU to own
    windows."""
    import oracle_bind
    import synth
    from popgenomicstools_amd import build
    build.build_lib()
    build.build_hosts()
    orc = oracle_bind.load()

    def run(cmd, **env):
        return subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))

    def same_on_two(cmd):
        for ingest in ("0", "1"):
            one = run(cmd, PGT_GPU_INGEST=ingest)
            assert one.returncode == 0, one.stderr[-500:]  # (some golden cases print no window at all)
            for devs in (("0,1", "1,0") if not PRETEND else ("0,0",)):
                two = run(cmd, PGT_GPU_INGEST=ingest, PGT_DEVICES=devs)
                assert (two.returncode, two.stdout, two.stderr) == (one.returncode, one.stdout, one.stderr), (cmd, devs, ingest, two.stderr[-500:])
        return one

    for c in helpers.load_golden("ref_kat.json")["cases"] + helpers.load_golden("ref_random.json")["cases"][::10]:
        f = tmp_path / "in.txt"
        f.write_text(c["input"])
        r = same_on_two([os.path.join(BIN, c["tool"]), str(f), str(c["W"]), str(c["S"])])
        assert [ln.split("\t")[:4] for ln in r.stdout.splitlines()] == [ln.split("\t")[:4] for ln in c["stdout"].splitlines()]  # the reference's rows
    rng = np.random.default_rng(123)
    n = 1_500_000
    chr_ids, pos = synth.chromosomes(rng, n, 5, equal=False)
    a, b = synth.fst_columns(rng, n)
    f = tmp_path / "big.fst.txt"
    orc.write_fst_text(str(f), chr_ids, pos, a, b)
    r = same_on_two([os.path.join(BIN, "fstWindow"), str(f), "50000", "10000"])
    assert len(r.stdout.splitlines()) > 100
    g = synth.het_column(rng, n)
    h = tmp_path / "big.het.txt"
    orc.write_het_text(str(h), chr_ids, pos, g)
    same_on_two([os.path.join(BIN, "hetWindow"), str(h), "50000", "10000"])
    p1, p2, n1, n2 = synth.dxy_columns(rng, n)
    m1, m2 = tmp_path / "p1.mafs", tmp_path / "p2.mafs"
    orc.write_maf_text(str(m1), chr_ids, pos, p1, n1)
    orc.write_maf_text(str(m2), chr_ids, pos, p2, n2)
    same_on_two([os.path.join(BIN, "dxyWindow"), "-winsize", "50000", "-stepsize", "10000", "-minind", "5", "-fixedsite", "1", str(m1), str(m2)])
    score = np.round(rng.normal(0, 1.2, n), 4)
    ihs = tmp_path / "big.ihs.norm"
    ihs.write_text("".join(f"chr{c}_{p}\t{p}\t0.3\t1.1\t2.2\t0.5\t{s}\t0\n" for c, p, s in zip(chr_ids.tolist(), pos.tolist(), score.tolist())))
    same_on_two([os.path.join(BIN, "ihsWindow"), str(ihs), "-winsize", "50000", "-cutoff", "2"])
    xp = tmp_path / "big.xpehh.norm"
    xp.write_text("id\tpos\tgpos\tp1\tihh1\tp2\tihh2\txpehh\tnormxpehh\tcrit\n" +
                  "".join(f"chr{c}_{p}\t{p}\t0.1\t0.3\t1.1\t0.4\t2.2\t0.5\t{s}\t0\n" for c, p, s in zip(chr_ids.tolist(), pos.tolist(), score.tolist())))
    same_on_two([os.path.join(BIN, "xpehhWindow"), str(xp), "2", "-winsize", "30000"])


@pytest.mark.timeout(900)
def test_bench_two_gpus_pairs_workload_rccl_gather_and_peer_stores():
    """BASELINE configs[4] on two GPUs as a user would type it: `python bench.py --gpus 2 --workload pairs` — 28 population
    pairs batched per launch, site ranges sharded, 28 x windows rows per rank delivered by the RCCL gather and by peer stores,
    each assembled (re-interleaved: table-major over ALL windows) table bitwise the single-GPU call's."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "pairs", "--pairs", "28", "--sites", "2e7",
                        "--chroms", "6", "--steps", "5", "--warmup", "2", "--exchange", "both"], capture_output=True, text=True,
                       env=dict(_env(), PGT_BENCH_BACKEND="gloo", PGT_BENCH_SHARE_GPU="1") if PRETEND else _env(), timeout=850)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ok"] is True and line["degraded"] is False
    assert "28 population pairs" in line["metric"]
    assert line["config"]["collective_backend"] == ("nccl (RCCL)" if not PRETEND else "gloo (REHEARSAL: PGT_BENCH_BACKEND)"), line["config"]["collective_backend"]
    assert line["rows_check"].startswith("bitwise equal")
    assert line["extra"]["exchange_gather"]["rows_check"] == "bitwise equal"
    assert line["extra"]["exchange_peer"].get("rows_check") == "bitwise equal", line["extra"]["exchange_peer"]
    assert line["roofline"]["algorithmic_bytes_per_launch"] == 16.0 * 28 * line["config"]["sites_resident_per_gpu"][0]


@pytest.mark.timeout(900)
def test_hosts_reduce_in_passes_over_two_gpus(tmp_path):
    """PGT_DEVICES=0,1 together with PGT_MAX_RESIDENT_SITES (inputs beyond the GPUs' memory: the table is reduced block by
    block, the blocks dealt to the GPUs, one host thread and context each): fstWindow, hetWindow and dxyWindow print the
    bytes of the single-GPU single-pass run, for limits that give every GPU several passes and for one that gives the second
    GPU nothing to do."""
    import oracle_bind
    import synth
    from popgenomicstools_amd import build
    build.build_lib()
    build.build_hosts()
    orc = oracle_bind.load()
    rng = np.random.default_rng(321)
    n = 1_200_000
    chr_ids, pos = synth.chromosomes(rng, n, 4, equal=False)
    a, b = synth.fst_columns(rng, n)
    g = synth.het_column(rng, n)
    p1, p2, n1, n2 = synth.dxy_columns(rng, n)
    f, h, m1, m2 = (tmp_path / x for x in ("fst.txt", "het.txt", "p1.mafs", "p2.mafs"))
    orc.write_fst_text(str(f), chr_ids, pos, a, b)
    orc.write_het_text(str(h), chr_ids, pos, g)
    orc.write_maf_text(str(m1), chr_ids, pos, p1, n1)
    orc.write_maf_text(str(m2), chr_ids, pos, p2, n2)
    cmds = [[os.path.join(BIN, "fstWindow"), str(f), "50000", "10000"],
            [os.path.join(BIN, "hetWindow"), str(h), "50000", "10000"],
            [os.path.join(BIN, "dxyWindow"), "-winsize", "50000", "-stepsize", "10000", "-minind", "5", "-fixedsite", "1", str(m1), str(m2)]]
    lists = ("0,1", "1,0") if not PRETEND else ("0,0",)
    for cmd in cmds:
        one = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        assert one.returncode == 0 and len(one.stdout.splitlines()) > 50, one.stderr[-500:]
        for limit in ("150000", "400000", "2000000"):  # 8 passes, 3 passes, one pass (the whole table fits one GPU)
            for devs in lists:
                two = subprocess.run(cmd, capture_output=True, text=True, timeout=300,
                                     env=dict(os.environ, PGT_DEVICES=devs, PGT_MAX_RESIDENT_SITES=limit))
                assert (two.returncode, two.stdout) == (0, one.stdout), (cmd[0], limit, devs, two.stderr[-500:])
                assert two.stderr == one.stderr, (cmd[0], limit, devs)  # dxyWindow: the genome-wide line
