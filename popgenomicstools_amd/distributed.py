"""Multi-GPU plumbing: one process per GPU, windows sharded by site range, one final gather.

Windows are independent given their [lo,hi), so the data path needs no
collective: rank r holds the site columns [site_lo, site_hi) of its shard (neighbouring shards
overlap by at most one window length; the halo is loaded twice, never exchanged), reduces its
own block of the window table, and the fixed-size rows are gathered to rank 0 — over RCCL/xGMI
when the process group's backend is "nccl", over gloo in the CPU tests.  The gather moves
40 B per window (≈4 MB for 10^9 sites at S=10^4): it is latency-bound, not link-bound.
"""
from __future__ import annotations

import numpy as np

from ._lib import WIN_DTYPE
from .window_scan import plan_shards


def shard_windows(win: np.ndarray, rank: int, world: int):
    """-> (shard record, this rank's windows re-based to its local columns)."""
    shards = plan_shards(win, world)
    s = shards[rank]
    local = np.array(win[int(s["win_begin"]): int(s["win_end"])], dtype=WIN_DTYPE, copy=True)
    local["lo"] -= s["site_lo"]
    local["hi"] -= s["site_lo"]
    return s, local, shards


class RowGatherer:
    """gather_rows with its buffers allocated once (the bench's per-step path): when every rank has
    the same number of windows the local row tensor is sent as it is, no staging copy."""

    def __init__(self, counts, row_bytes: int, device, dst: int = 0, group=None):
        import torch
        import torch.distributed as dist
        self.dist, self.torch = dist, torch
        self.counts = [int(c) for c in counts]
        self.row_bytes, self.dst, self.group = row_bytes, dst, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.width = max(self.counts) * row_bytes
        self.uniform = len(set(self.counts)) == 1
        self.send = None if self.uniform else torch.zeros(self.width, dtype=torch.uint8, device=device)
        self.recv = ([torch.empty(self.width, dtype=torch.uint8, device=device) for _ in range(self.world)]
                     if self.rank == dst else None)

    def start(self, local_rows):
        """Asynchronous form: issues the gather and returns its Work handle; the caller must keep
        `local_rows` untouched until handle.wait() (a stream-level wait) has been issued.  Lets the
        gather of step k overlap the build of step k+1 when the rows are double-buffered."""
        if self.uniform:
            send = local_rows
        else:
            send = self.send
            send[: local_rows.numel()].copy_(local_rows)
        return self.dist.gather(send, self.recv, dst=self.dst, group=self.group, async_op=True)

    def __call__(self, local_rows):
        """Returns on dst the list of per-rank row tensors (views, valid until the next call)."""
        if self.uniform:
            send = local_rows
        else:
            send = self.send
            send[: local_rows.numel()].copy_(local_rows)
        self.dist.gather(send, self.recv, dst=self.dst, group=self.group)
        if self.rank != self.dst:
            return None
        return [self.recv[r][: self.counts[r] * self.row_bytes] for r in range(self.world)]


def gather_rows(local_rows, counts, row_bytes: int, dst: int = 0, group=None):
    """Gather per-rank packed row tensors (uint8, counts[r]*row_bytes bytes on rank r) to `dst`.

    local_rows: torch uint8 tensor on the rank's device (CUDA for nccl/RCCL, CPU for gloo).
    counts: windows per rank, known to every rank from the shard plan (no size exchange needed).
    Returns on dst the concatenated uint8 tensor in window order; None elsewhere.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return local_rows
    width = int(max(counts)) * row_bytes
    send = torch.zeros(width, dtype=torch.uint8, device=local_rows.device)
    send[: local_rows.numel()] = local_rows
    recv = [torch.empty(width, dtype=torch.uint8, device=local_rows.device) for _ in range(world)] if rank == dst else None
    dist.gather(send, recv, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([recv[r][: int(counts[r]) * row_bytes] for r in range(world)])


def sharded_scan(win: np.ndarray, row_dtype: np.dtype, load_columns, reduce_rows, device, dst: int = 0, group=None):
    """The whole multi-GPU path for one input, to be called by every rank of the process group.

      win           the full window table (identical on every rank; built on the host in O(#windows))
      load_columns  callable(site_lo, site_hi) -> whatever reduce_rows needs for sites [site_lo, site_hi):
                    a rank only ever touches its own shard (plus a halo of at most one window)
      reduce_rows   callable(columns, local_win) -> packed uint8 torch tensor of len(local_win) rows on
                    `device` (e.g. lambda c, w: ctx.fst_reduce_dev(*c, windows_to_device(w, dev))[0])
    Returns on `dst` the assembled rows as a numpy structured array in window order, None elsewhere.
    One collective: the gather of the fixed-size rows (RCCL when the backend is nccl).
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    shard, local_win, shards = shard_windows(win, rank, world)
    columns = load_columns(int(shard["site_lo"]), int(shard["site_hi"]))
    rows = reduce_rows(columns, local_win)
    counts = (shards["win_end"] - shards["win_begin"]).astype(np.int64)
    if world == 1:
        packed = rows
    else:
        parts = RowGatherer(counts, row_dtype.itemsize, device, dst=dst, group=group)(rows)
        packed = torch.cat(parts) if rank == dst else None
    if rank != dst:
        return None
    return np.frombuffer(packed.cpu().numpy().tobytes(), dtype=row_dtype)
