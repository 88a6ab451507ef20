"""GPU: bench.py itself, small — the single-GPU line and the N > 1 path (two ranks sharing GPU 0 over gloo,
the rehearsal knobs of bench.py), both row transports: the sharded table must be the single-GPU table."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--sites", "3e6", "--chroms", "5", "--steps", "3", "--warmup", "1", "--headline-only", "--no-cpu"]


def _line(r):
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.mark.timeout(900)
@pytest.mark.parametrize("workload", [[], ["--workload", "pairs", "--pairs", "3"]])
def test_bench_single_and_two_rank_tables_agree(workload):
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    one = _line(subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + workload, capture_output=True,
                               text=True, env=env, timeout=400))
    assert one["n_gpus"] == 1 and one["unit"] == "sites/s" and one["value"] > 0 and 0 < one["roofline"]["frac"] < 1
    assert one["roofline"]["traffic"] is None and one["config"]["row_exchange"] == "local" and one["rows_check"] is None
    for mode, port in (("peer", "29561"), ("gather", "29562")):
        env2 = dict(env, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, PGT_BENCH_BACKEND="gloo", PGT_BENCH_SHARE_GPU="1")
        two = _line(subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                                    "--master-addr", "127.0.0.1", "--master-port", port, os.path.join(ROOT, "bench.py"),
                                    "--gpus", "2", "--exchange", mode] + SMALL + workload,
                                   capture_output=True, text=True, env=env2, timeout=400))
        assert two["n_gpus"] == 2 and two["scaling"] == "strong" and two["config"]["row_exchange"] == mode
        assert two["rows_check"].startswith("bitwise equal")
        assert two["rows_sha256"] == one["rows_sha256"]
        assert sum(two["config"]["sites_resident_per_gpu"]) < 1.2 * one["config"]["sites_total"]
        assert two["extra"]["exchange_" + mode]["rows_check"] == "bitwise equal" and two["extra"]["exchange_" + mode]["headline"]


@pytest.mark.timeout(600)
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` exactly as the driver calls it for N = 1 (no launcher, no WORLD_SIZE): the parent starts
    the two ranks as a child process group, relays rank 0's line and exits with the child's code.  The default (--exchange
    auto) is the gather alone and every rank announces every phase; with --exchange both, BOTH transports are timed, each
    table is checked bit for bit, the headline is one of the verified ones."""
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", PGT_BENCH_BACKEND="gloo", PGT_BENCH_SHARE_GPU="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    one = _line(subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + SMALL, capture_output=True, text=True,
                               env=env, timeout=400))
    # the default (--exchange auto) is the gather alone: the transport north_star names, nothing opt-in in an unattended run
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL, capture_output=True, text=True, env=env,
                       timeout=500)
    dflt = _line(r)
    assert dflt["n_gpus"] == 2 and dflt["ok"] is True and dflt["rows_sha256"] == one["rows_sha256"]
    assert dflt["config"]["row_exchange"] == "gather" and "exchange_peer" not in dflt["extra"]
    assert "REHEARSAL" in dflt["config"]["collective_backend"]
    for phase in ("init", "columns", "gather timed", "verify", "roofline"):   # every rank announces every phase on stderr
        for rk in (0, 1):
            assert f"[bench r{rk}/2" in r.stderr and f"] {phase}: done in" in r.stderr, (phase, r.stderr[-3000:])
        assert dflt["extra"]["phase_seconds"][phase] >= 0
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--exchange", "both"] + SMALL, capture_output=True,
                       text=True, env=env, timeout=500)
    two = _line(r)
    assert two["n_gpus"] == 2 and two["rows_check"].startswith("bitwise equal") and two["rows_sha256"] == one["rows_sha256"]
    assert two["ok"] is True
    ex = two["extra"]
    assert ex["exchange_gather"]["rows_check"] == "bitwise equal" and ex["exchange_gather"]["ms_per_step"] > 0
    assert ex["exchange_peer"]["rows_check"] == "bitwise equal" and ex["exchange_peer"]["ms_per_step"] > 0
    assert [ex[k]["headline"] for k in ("exchange_gather", "exchange_peer")].count(True) == 1
    assert two["config"]["row_exchange"] in ("gather", "peer")
    # a failing rank is a failing bench: the parent passes the child's exit code on
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stepsize", "0"] + SMALL, capture_output=True,
                         text=True, env=env, timeout=300)
    assert bad.returncode != 0


def _rehearsal_env(**kw):
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", PGT_BENCH_BACKEND="gloo", PGT_BENCH_SHARE_GPU="1", **kw)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


@pytest.mark.timeout(900)
def test_bench_rank_lost_or_wedged_mid_run_fails_fast_and_names_the_phase():
    """The first N > 1 run on real hardware is unattended: a rank that dies, or one that never returns from a phase, must
    end the whole command with a non-zero code well inside the phase deadline and leave the phase's name on stderr."""
    import time
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL
    # (a) rank 1 dies at the start of the timed gather
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, env=_rehearsal_env(PGT_BENCH_FAULT="1:gather timed:die",
                                                                                PGT_BENCH_DEADLINE_SCALE="0.5"), timeout=400)
    assert r.returncode != 0 and time.time() - t0 < 200
    assert "fault injected: die in phase 'gather timed'" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    # (b) rank 1 wedges in the timed gather: BOTH ranks' watchdogs name the phase (rank 0 waits in the collective)
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, env=_rehearsal_env(PGT_BENCH_FAULT="1:gather timed:hang",
                                                                                PGT_BENCH_DEADLINE_SCALE="0.1"), timeout=400)
    took = time.time() - t0
    assert r.returncode != 0 and took < 200, took
    assert "phase 'gather timed' exceeded its deadline of 6 s" in r.stderr, r.stderr[-3000:]
    assert "[bench r1/2" in r.stderr and "giving up (exit 124)" in r.stderr
    # (c) rank 0 wedges while rebuilding the genome for the check
    r = subprocess.run(cmd, capture_output=True, text=True, env=_rehearsal_env(PGT_BENCH_FAULT="0:verify:hang",
                                                                                PGT_BENCH_DEADLINE_SCALE="0.1"), timeout=400)
    assert r.returncode != 0 and "phase 'verify' exceeded its deadline of 12 s" in r.stderr, r.stderr[-3000:]


@pytest.mark.timeout(600)
def test_bench_second_transport_that_never_returns_degrades_to_the_gather_line():
    """--exchange both: the gather's line is secured before the opt-in peer-store transport starts; a peer phase that
    overruns its deadline makes rank 0 print that line (peer marked unavailable, with the reason) and the command exit 0."""
    one = _line(subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + SMALL, capture_output=True, text=True,
                               env=_rehearsal_env(), timeout=400))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--exchange", "both"] + SMALL, capture_output=True,
                       text=True, env=_rehearsal_env(PGT_BENCH_FAULT="1:peer timed:hang", PGT_BENCH_DEADLINE_SCALE="0.15"), timeout=400)
    two = _line(r)
    assert two["config"]["row_exchange"] == "gather" and two["rows_sha256"] == one["rows_sha256"] and two["ok"] is True
    assert two["extra"]["exchange_gather"]["rows_check"] == "bitwise equal" and two["extra"]["exchange_gather"]["headline"]
    assert two["extra"]["exchange_peer"]["available"] is False and "deadline" in two["extra"]["exchange_peer"]["why"]
    assert "second transport abandoned" in two["extra"]["note"]
    assert "degrading to the result secured before it" in r.stderr


@pytest.mark.timeout(900)
def test_bench_falls_back_to_cpu_staged_rows_when_rccl_cannot_be_brought_up():
    """Two ranks on ONE GPU without the rehearsal backend knob: RCCL refuses (or never finishes) a communicator with two
    ranks on the same device, so this is the real thing the fallback exists for — the probe all-reduce fails or overruns its
    60 s on the helper thread, the ranks agree over the gloo control group, the rows are staged through the CPU, the line
    says so, the table is still the single-GPU table bit for bit."""
    env = _rehearsal_env()
    env.pop("PGT_BENCH_BACKEND")
    one = _line(subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + SMALL, capture_output=True, text=True,
                               env=env, timeout=400))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL, capture_output=True, text=True, env=env,
                       timeout=600)
    two = _line(r)
    assert two["rows_sha256"] == one["rows_sha256"] and two["rows_check"].startswith("bitwise equal")
    assert "FALLBACK" in two["config"]["collective_backend"], two["config"]["collective_backend"]
    assert "RCCL unusable" in r.stderr
    # a CPU-staged number is not the RCCL result the run was asked for: said at the top of the line, not only in config
    assert two["degraded"] is True and two["ok"] is False and "DEGRADED" in two["metric"] and one["degraded"] is False


@pytest.mark.timeout(600)
def test_bench_rccl_probe_that_never_returns_still_falls_back():
    """The probe all-reduce HANGS on rank 1 (fault injected on the helper thread, as an RCCL bootstrap that never completes
    would): the main threads give it its share of the init deadline, agree over gloo, stage the rows through the CPU — the
    advertised fallback, not the watchdog's exit 124 — and the line says degraded."""
    env = _rehearsal_env()
    env.pop("PGT_BENCH_BACKEND")
    env.update(PGT_BENCH_FAULT="1:rccl probe:hang", PGT_BENCH_DEADLINE_SCALE="0.25")  # init: 30 s, the probe gets 15 s of it
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL, capture_output=True, text=True, env=env,
                       timeout=500)
    two = _line(r)
    assert two["degraded"] is True and two["ok"] is False and two["rows_check"].startswith("bitwise equal")
    assert "fault injected: hang in phase 'rccl probe'" in r.stderr and "did not return in time" in r.stderr


@pytest.mark.timeout(600)
def test_rccl_probe_brings_up_a_communicator_beside_the_gloo_control_group():
    """bench.bring_up_collectives as the ranks of an N > 1 run call it — gloo default group, RCCL as a second group created
    on the main thread, probed on a helper thread — with the one rank a 1-GPU box allows: the API sequence (new_group(backend="nccl")
    on top of a gloo default group, all-reduce, agreement over gloo) is what the 8-GPU run executes."""
    code = """
import os, sys, json
sys.argv = ['bench.py']
import torch, bench
import torch.distributed as dist
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
ph = bench.Phases(0, 1)
with ph('init'):
    group, coll_dev, desc = bench.bring_up_collectives(1, 0, dev, ph)
t = torch.full((4,), 3.0, device=coll_dev)
dist.all_reduce(t, group=group)
w = dist.barrier(group=group)
dist.barrier(group=group, device_ids=[0])  # the form bench.py and RowExchange use on RCCL: the rank's device is named
out = [torch.zeros(8, dtype=torch.uint8, device=coll_dev)]
dist.gather(torch.arange(8, dtype=torch.uint8, device=coll_dev), out, dst=0, group=group)
print(json.dumps({'desc': desc, 'dev': str(coll_dev), 'sum': float(t.sum().item()), 'gathered': out[0].cpu().tolist()}), flush=True)
os._exit(0)
"""
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               MASTER_ADDR="127.0.0.1", MASTER_PORT="29581")
    env.pop("PGT_BENCH_BACKEND", None)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=500, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    got = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert got["desc"] == "nccl (RCCL)" and got["dev"] == "cuda:0" and got["sum"] == 12.0 and got["gathered"] == list(range(8)), (got, r.stderr[-1500:])
