#!/usr/bin/env python3
"""bench.py — genomic sites/s of the 2-population FST window scan on MI355X (BASELINE.json metric).

One step = one pass of the hot path over the columns resident in this rank's HBM: build the range
tree (the streaming pass over a,b: 16 B/site) + answer every window (W=50000, S=10000 sites) +, for
N>1, gather the window rows to rank 0 over RCCL.  Inputs are synthetic (BASELINE.md definition) and
already in HBM when the timed region starts.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Scaling is WEAK: every rank holds --sites sites (default 10^9 = 20 GB of columns) in --chroms
chromosomes; the job is N x that.  value = all sites of all ranks / max-over-ranks time.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import popgenomicstools_amd as pgt  # noqa: E402
from popgenomicstools_amd._lib import FST_ROW_DTYPE, PGT_STAT_FST  # noqa: E402
from popgenomicstools_amd.distributed import RowGatherer  # noqa: E402
from popgenomicstools_amd.window_scan import rows_from_device, windows_to_device  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy ceiling)
BYTES_PER_SITE = 16.0  # algorithmic: a,b f64 read once by the tree-build kernel


def synth_columns(n, n_chr, seed, dev):
    """BASELINE.md synthetic table on the device: pos = running sum of U{1..59} gaps per chromosome,
    b ~ U(0,.3), a = b*U(-.1,.6), both rounded to 6 decimals.  Generated per chromosome to bound
    temporaries.  torch is plumbing here (device RNG), not the product."""
    gen = torch.Generator(device=dev).manual_seed(seed)
    a = torch.empty(n, dtype=torch.float64, device=dev)
    b = torch.empty(n, dtype=torch.float64, device=dev)
    pos = torch.empty(n, dtype=torch.int32, device=dev)
    base = n // n_chr
    lens = [base + (1 if c < n - base * n_chr else 0) for c in range(n_chr)]
    o = 0
    for L in lens:
        bb = torch.round(torch.rand(L, generator=gen, device=dev, dtype=torch.float64) * 0.3e6) / 1e6
        u = torch.rand(L, generator=gen, device=dev, dtype=torch.float64) * 0.7 - 0.1
        a[o:o + L] = torch.round(bb * u * 1e6) / 1e6
        b[o:o + L] = bb
        pos[o:o + L] = torch.randint(1, 60, (L,), generator=gen, device=dev, dtype=torch.int32).cumsum(0, dtype=torch.int32)
        o += L
        del bb, u
    return pos, a, b, np.array(lens, dtype=np.uint64)


def cpu_baseline(pos, a, b, run_len, W, S, n_sample):
    """The reference CPU path on this box's host cores, on a bounded sample of the same workload:
    the first n_sample sites written as the tool's text input, then the UNMODIFIED reference binary
    (oracle/_ref/fstWindow, kind "reference") — or, if that binary did not travel, our restatement
    (oracle/liboracle.so, kind "port") — timed end to end, single-threaded like the reference."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_bind
    orc = oracle_bind.load()
    n_sample = int(min(n_sample, a.numel()))
    hp = pos[:n_sample].cpu().numpy().view(np.uint32)
    ha, hb = a[:n_sample].cpu().numpy(), b[:n_sample].cpu().numpy()
    chr_ids = np.repeat(np.arange(run_len.size, dtype=np.uint32), run_len.astype(np.int64))[:n_sample]
    tmpdir = tempfile.mkdtemp(prefix="pgt_bench_")
    path = os.path.join(tmpdir, "sample.fst.txt")
    orc.write_fst_text(path, chr_ids, hp, ha, hb)
    ref = oracle_bind.ref_binary("fstWindow")
    t0 = time.perf_counter()
    if ref:
        with open(os.devnull, "w") as devnull:
            subprocess.run([ref, path, str(W), str(S)], stdout=devnull, check=True)
        kind = "reference"
    else:
        assert orc.fst_text(path, W, S, os.devnull) == 0
        kind = "port"
    dt = time.perf_counter() - t0
    os.unlink(path)
    os.rmdir(tmpdir)
    return {"value": n_sample / dt, "unit": "sites/s", "cores": 1, "kind": kind,
            "sample": f"first {n_sample} sites of the workload as text ({'oracle/_ref/fstWindow' if ref else 'oracle port'} "
                      f"{W} {S}, parse included, stdout to /dev/null, {dt:.2f} s); host has {os.cpu_count()} logical cores, "
                      f"the reference is single-threaded"}


def measured_traffic(n_sites):
    """HBM bytes per build launch from the PMC pass committed under profiles/ (collected separately:
    rocprofv3 --pmc cannot run inside this process).  None when no such file exists for this size."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(p):
        try:
            t = json.load(open(p))
            return t.get(str(int(n_sites)))
        except Exception:
            return None
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--sites", type=float, default=1e9, help="sites per GPU (weak scaling)")
    ap.add_argument("--chroms", type=int, default=40, help="chromosomes per GPU")
    ap.add_argument("--winsize", type=int, default=50_000)
    ap.add_argument("--stepsize", type=int, default=10_000)
    ap.add_argument("--cpu-sites", type=float, default=5e7,
                    help="sample size of the CPU baseline leg (5e7 sites = 1.6 GB of text, ~10 s of the reference tool)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--extra", action="store_true",
                    help="also time the 10^8-site configuration (BASELINE configs[1]) on the same buffers; off by "
                         "default so that a rocprofv3 --stats average of the default command covers one size only")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # Rehearsal knobs for a one-GPU box (never set by the driver): PGT_BENCH_BACKEND=gloo moves the
    # collectives to CPU staging, PGT_BENCH_SHARE_GPU=1 puts every rank on GPU 0.  The measured
    # configuration is always the default: one GPU per rank, backend nccl (= RCCL over xGMI).
    backend = os.environ.get("PGT_BENCH_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("PGT_BENCH_SHARE_GPU") == "1" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend)

    n, W, S = int(args.sites), args.winsize, args.stepsize
    pos, a, b, run_len = synth_columns(n, args.chroms, 12345 + rank, dev)
    win = pgt.build_windows_sites(run_len, W, S)  # host, O(#windows)
    win_d = windows_to_device(win, dev)
    ctx = pgt.Context(dev_index)
    ctx.set_max_window(int((win["hi"] - win["lo"]).max()))  # = W: tree levels above 8192 sites are not needed
    tree = torch.empty(ctx.tree_bytes(PGT_STAT_FST, n), dtype=torch.uint8, device=dev)
    out = torch.empty(win.size * FST_ROW_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    counts = [win.size] * world  # every rank has the same geometry
    gather = RowGatherer(counts, FST_ROW_DTYPE.itemsize, coll_dev, dst=0) if world > 1 else None

    # Opt-in (PGT_BENCH_ASYNC_GATHER=1, never set by the driver): rows are double-buffered and the
    # gather of step k is only waited for before step k+2 reuses its buffer, so it overlaps the build
    # of step k+1.  Default: the gather completes inside the step that produced the rows.
    async_gather = gather is not None and os.environ.get("PGT_BENCH_ASYNC_GATHER") == "1"
    outs = [out, torch.empty_like(out)] if async_gather else [out]
    pending = [None, None]
    counter = [0]

    def step():
        k = counter[0] & 1 if async_gather else 0
        counter[0] += 1
        if pending[k] is not None:
            pending[k].wait()  # the gather that last read outs[k] must be done before it is overwritten
            pending[k] = None
        ctx.fst_reduce_dev(pos, a, b, win_d, out=outs[k], tree=tree)
        if gather is None:
            return outs[k]
        rows = outs[k] if coll_dev is dev else outs[k].cpu()
        if async_gather:
            pending[k] = (gather.start(rows), rows)[0]
            return None
        return gather(rows)  # one RCCL gather of 40 B/window to rank 0

    def fence():
        for k in (0, 1):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # --- roofline of the dominant kernel: HIP events on the launch stream, around the build pass only
    ctx.set_profiling(True)
    build_ms, query_ms = [], []
    for _ in range(max(5, min(args.steps, 20))):
        ctx.fst_reduce_dev(pos, a, b, win_d, out=out, tree=tree)
        bm, qm = ctx.last_kernel_ms()
        build_ms.append(bm)
        query_ms.append(qm)
    ctx.set_profiling(False)
    build_avg = float(np.mean(build_ms))
    achieved = BYTES_PER_SITE * n / (build_avg * 1e-3) / 1e9  # GB/s

    # --- the 10^8-site configuration (BASELINE configs[1]) on the same buffers, for the record
    extra = {}
    if args.extra and rank == 0 and n > 100_000_000:
        n8 = 100_000_000
        rl8 = np.full(20, n8 // 20, dtype=np.uint64)
        win8 = windows_to_device(pgt.build_windows_sites(rl8, W, S), dev)
        out8 = torch.empty(win8.numel() // 32 * FST_ROW_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        for _ in range(3):
            ctx.fst_reduce_dev(pos[:n8], a[:n8], b[:n8], win8, out=out8, tree=tree)
        torch.cuda.synchronize()
        # per-step time from events on the launch stream (= torch's current stream); median of 50,
        # so that a one-off stall inside the loop does not pass for a per-step cost
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
        for e0, e1 in ev:
            e0.record()
            ctx.fst_reduce_dev(pos[:n8], a[:n8], b[:n8], win8, out=out8, tree=tree)
            e1.record()
        torch.cuda.synchronize()
        d8 = float(np.median([e0.elapsed_time(e1) for e0, e1 in ev])) * 1e-3
        ctx.set_profiling(True)
        b8 = []
        for _ in range(10):
            ctx.fst_reduce_dev(pos[:n8], a[:n8], b[:n8], win8, out=out8, tree=tree)
            b8.append(ctx.last_kernel_ms()[0])
        ctx.set_profiling(False)
        extra["sites_1e8"] = {"value": n8 / d8, "ms_per_step": d8 * 1e3, "build_kernel_ms": float(np.mean(b8)),
                              "roofline_frac": BYTES_PER_SITE * n8 / (float(np.mean(b8)) * 1e-3) / 1e9 / HBM_PEAK_GBS}

    # --- sanity: a sample of windows against float64 sums taken by torch (independent path)
    rows = rows_from_device(out, FST_ROW_DTYPE)
    for i in np.linspace(0, win.size - 1, 7).astype(int):
        lo, hi = int(win["lo"][i]), int(win["hi"][i])
        ref = float(a[lo:hi].sum()) / float(b[lo:hi].sum())
        assert abs(rows["fst"][i] - ref) <= 1e-9 * abs(ref) + 1e-12, (i, rows["fst"][i], ref)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline(pos, a, b, run_len, W, S, args.cpu_sites)

    if rank == 0:
        total_sites = float(n) * world
        line = {
            "metric": "genomic sites/sec for 2-pop FST window scan",
            "value": total_sites * args.steps / dt,
            "unit": "sites/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"fstWindow 2 pops x {n:.0e} sites per GPU in {args.chroms} chromosomes, "
                                   f"window {W} sites / step {S} sites, {win.size} windows per GPU, columns resident in HBM"
                                   + (", rows gathered to rank 0 over RCCL" if world > 1 else ""),
                       "baseline_config": "the 10^9-site two-population FST window scan north_star's target is quoted on "
                                          "(it fits one GPU: 20 GB); BASELINE configs[1] (10^8 sites) is measured with --extra",
                       "sites_per_gpu": n, "winsize": W, "stepsize": S, "windows_per_gpu": int(win.size),
                       "parallelism": (f"site-range shards x{world}" + (", async gather" if async_gather else "")
                                       + ("" if backend == "nccl" else f" (REHEARSAL: backend {backend})"))
                                      if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": "fst_build_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(n),
                         "kernel_ms": build_avg, "query_kernel_ms": float(np.mean(query_ms)),
                         "algorithmic_bytes_per_launch": BYTES_PER_SITE * n},
            "cpu_baseline": cpu,
            "extra": extra,
        }
        print(json.dumps(line), flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
