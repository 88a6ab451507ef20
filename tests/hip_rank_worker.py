"""Multi-rank worker with the REAL HIP kernels, for the one-GPU box: started by tests/test_gpu_parity.py as
    python -m torch.distributed.run --nproc-per-node R tests/hip_rank_worker.py
(gloo for the collectives, every rank on GPU 0 — the launcher runs before anything touches the GPU).
On a box with several GPUs tests/test_multi_gpu.py starts the same worker with PGT_TEST_DEVICE_PER_RANK=1 (rank r on GPU r)
and PGT_TEST_BACKEND=nccl (the row collectives on an RCCL group, device tensors): the real transport, over xGMI.
Each rank materialises only its own site range of one counter-based synthetic genome, reduces its
block of the window table through the C-ABI device entry points, and the rows travel to rank 0 through
popgenomicstools_amd.distributed (both transports: the gather, and peer stores into rank 0's row
buffer through hipIpc).  Rank 0 demands the bytes of the single-GPU call on the whole input.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import popgenomicstools_amd as pgt  # noqa: E402
from popgenomicstools_amd._lib import (DXY_ROW_DTYPE, DXY_TOTAL_DTYPE, EXT_ROW_DTYPE, FST_ROW_DTYPE, HET_ROW_DTYPE, PGT_EXT_IHS,  # noqa: E402
                                       WIN_DTYPE)
from popgenomicstools_amd.distributed import TOTAL_BLOCK, sharded_dxy_scan, sharded_scan  # noqa: E402
from popgenomicstools_amd.window_scan import rows_from_device, windows_to_device  # noqa: E402
from synth_genome import SynthGenome  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev_index = int(os.environ.get("LOCAL_RANK", "0")) if os.environ.get("PGT_TEST_DEVICE_PER_RANK") == "1" else 0
    torch.cuda.set_device(dev_index)
    dev, cpu = torch.device("cuda", dev_index), torch.device("cpu")
    group = None
    if os.environ.get("PGT_TEST_BACKEND") == "nccl":  # RCCL for the rows (device tensors); gloo stays the control plane
        group = dist.new_group(backend="nccl")
        cpu = dev  # `cpu` names the device the collectives' tensors live on, below
    ctx = pgt.Context(dev_index)
    n, W, S = 3_000_000, 50_000, 10_000
    g = SynthGenome(4242, n, 5)
    win = pgt.build_windows_sites(g.run_len, W, S)
    ctx.set_max_window(W)

    def whole(fn):  # rank 0: the single-GPU answer on the whole input
        if rank != 0:
            return None
        return fn()

    # ---- fstWindow, one pair ----------------------------------------------------------------
    def fst_reduce(c, w, out):
        ctx.fst_reduce_dev(*c, windows_to_device(w, dev), out=out)
    ref = whole(lambda: rows_from_device(ctx.fst_reduce_dev(*g.fst_columns_t(0, n, dev), windows_to_device(win, dev))[0],
                                         FST_ROW_DTYPE).tobytes())
    for mode in ("gather", "peer", "peer_or_gather", "auto"):
        got = sharded_scan(win, FST_ROW_DTYPE, lambda lo, hi: g.fst_columns_t(lo, hi, dev), fst_reduce, dev,
                           ctx=ctx, mode=mode, coll_device=cpu, group=group)
        if rank == 0:
            assert got.size == win.size and got.tobytes() == ref, f"fst {mode}"
            print(f"HIP_RANKS_OK fst {mode}", flush=True)

    # ---- BASELINE config 5: population pairs batched over one table (tables = pairs) ----------
    n_pairs = 5

    def pair_cols(lo, hi):
        cols = [g.pair_columns_t(p, lo, hi, dev) for p in range(n_pairs)]
        return g.pos_t(lo, hi, dev), [c[0] for c in cols], [c[1] for c in cols]

    def pairs_reduce(c, w, out):
        ctx.fst_reduce_pairs_dev(c[0], c[1], c[2], windows_to_device(w, dev), out=out)
    ref = whole(lambda: rows_from_device(ctx.fst_reduce_pairs_dev(*pair_cols(0, n), windows_to_device(win, dev))[0],
                                         FST_ROW_DTYPE).tobytes())
    for mode in ("gather", "peer"):
        got = sharded_scan(win, FST_ROW_DTYPE, pair_cols, pairs_reduce, dev, tables=n_pairs, ctx=ctx, mode=mode, coll_device=cpu, group=group)
        if rank == 0:
            assert got.size == n_pairs * win.size and got.tobytes() == ref, f"pairs {mode}"
            print(f"HIP_RANKS_OK pairs {mode}", flush=True)

    # ---- the same pairs from allele frequencies (pgt_fst_af_reduce_dev) -------------------------
    n_pops, nsamp = 4, [10.0, 12.0, 9.0, 20.0]

    def af_cols(lo, hi):
        return g.pos_t(lo, hi, dev), [g.freq_t(k, lo, hi, dev) for k in range(n_pops)]

    def af_reduce(c, w, out):
        ctx.fst_af_reduce_dev(c[0], c[1], nsamp, windows_to_device(w, dev), out=out)
    ref = whole(lambda: rows_from_device(ctx.fst_af_reduce_dev(*af_cols(0, n), nsamp, windows_to_device(win, dev))[0],
                                         FST_ROW_DTYPE).tobytes())
    got = sharded_scan(win, FST_ROW_DTYPE, af_cols, af_reduce, dev, tables=n_pops * (n_pops - 1) // 2, ctx=ctx,
                       mode="peer", coll_device=cpu, group=group)
    if rank == 0:
        assert got.tobytes() == ref, "af"
        print("HIP_RANKS_OK af peer", flush=True)

    # ---- ihsWindow-style extreme scan with windows >= 2^20 sites (upper tree levels in play) -----
    ctx.set_max_window(0)
    g1 = SynthGenome(4243, n, 1)  # one chromosome of ~9e7 bp
    hp = g1.fst_columns_np(0, n)[0]
    ewin = pgt.build_windows_extreme(hp, g1.run_len, None, 40_000_000)  # ~1.3e6 sites per window

    def ext_cols(lo, hi):
        p, a, _ = g1.fst_columns_t(lo, hi, dev)
        return p, (a * 40.0 - 2.0)

    def ext_reduce(c, w, out):
        # position/value refer to GLOBAL sites only through pos[] and score[], both local here: fine
        ctx.extreme_reduce_dev(c[0], c[1], PGT_EXT_IHS, 2.0, windows_to_device(w, dev), out=out)
    ref = whole(lambda: rows_from_device(ctx.extreme_reduce_dev(*ext_cols(0, n), PGT_EXT_IHS, 2.0, windows_to_device(ewin, dev))[0],
                                         EXT_ROW_DTYPE).tobytes())
    got = sharded_scan(ewin, EXT_ROW_DTYPE, ext_cols, ext_reduce, dev, ctx=ctx, mode="peer", coll_device=cpu, group=group)
    if rank == 0:
        assert int((ewin["hi"] - ewin["lo"]).max()) >= 1 << 20
        assert got.tobytes() == ref, "extreme"
        print("HIP_RANKS_OK extreme peer", flush=True)

    # ---- hetWindow (int8 column, 1024-site leaves, 65536-site level-2 nodes = the smallest shard alignment) ----
    ctx.set_max_window(W)

    def het_cols(lo, hi):
        return g.pos_t(lo, hi, dev), g.genotype_t(0, lo, hi, dev)

    def het_reduce(c, w, out):
        ctx.het_reduce_dev(c[0], c[1], windows_to_device(w, dev), out=out)
    ref = whole(lambda: rows_from_device(ctx.het_reduce_dev(*het_cols(0, n), windows_to_device(win, dev))[0], HET_ROW_DTYPE).tobytes())
    got = sharded_scan(win, HET_ROW_DTYPE, het_cols, het_reduce, dev, ctx=ctx, mode="peer", coll_device=cpu, group=group)
    if rank == 0:
        assert got.tobytes() == ref, "het"
        print("HIP_RANKS_OK het peer", flush=True)

    # ---- dxyWindow: window rows + the genome-wide line, bits independent of the rank count ---------
    ctx.set_max_window(W)

    def dxy_cols(lo, hi):
        return (g.pos_t(lo, hi, dev),) + tuple(g.dxy_columns_t(lo, hi, dev))
    minind = 5
    if rank == 0:
        out1, tot1, _ = ctx.dxy_reduce_dev(*dxy_cols(0, n), minind, windows_to_device(win, dev))
        ref_rows = rows_from_device(out1, DXY_ROW_DTYPE).tobytes()
        ref_tot = rows_from_device(tot1, DXY_TOTAL_DTYPE)[0]
        # the block-ordered total, formed here by the single-GPU call on a table of 2^16-site blocks
        starts = np.arange(0, n, TOTAL_BLOCK, dtype=np.uint64)
        bw = np.zeros(starts.size, dtype=WIN_DTYPE)
        bw["lo"], bw["hi"], bw["flags"] = starts, np.minimum(starts + np.uint64(TOTAL_BLOCK), np.uint64(n)), 1
        brow = rows_from_device(ctx.dxy_reduce_dev(*dxy_cols(0, n), minind, windows_to_device(bw, dev))[0], DXY_ROW_DTYPE)
        acc = 0.0
        for v in brow["sum"].tolist():
            acc += v
    for mode in ("gather", "peer"):
        rows, total = sharded_dxy_scan(win, n, dxy_cols, ctx, minind, dev, mode=mode, coll_device=cpu, group=group)
        if rank == 0:
            assert rows.tobytes() == ref_rows, f"dxy rows {mode}"
            assert int(total["neff"]) == int(ref_tot["neff"]) and int(total["nskip"]) == int(ref_tot["nskip"])
            assert float(total["sum"]) == acc, (float(total["sum"]), acc)   # bitwise: the block order is the same for any rank count
            assert abs(float(total["sum"]) - float(ref_tot["sum"])) <= 1e-12 * abs(float(ref_tot["sum"]))
            print(f"HIP_RANKS_OK dxy {mode}", flush=True)

    ctx.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
