"""Multi-GPU plumbing: one process per GPU, windows sharded by site range, rows assembled on one rank.

Windows are independent given their [lo,hi), so the data path needs no
collective: rank r holds the site columns [site_lo, site_hi) of its shard (neighbouring shards
overlap by at most one window length; the halo is loaded twice, never exchanged) and reduces its
own block of the window table.  What has to travel is the fixed-size rows (40 B per fst window,
≈4 MB for 10^9 sites at S=10^4) to the rank that prints them — latency-bound, not link-bound.
Two transports (RowExchange):

  peer    rank `dst` owns ONE row buffer for the whole table (pgt_rowbuf_create), every other rank
          maps it (hipIpc, pgt_rowbuf_open) and hands its slice to the *_dev call as `out`: the
          query kernel's row stores cross xGMI into dst's HBM, no collective per scan, one barrier
          when the caller wants the table.
  gather  torch.distributed.gather of the rows (backend "nccl" = RCCL over xGMI; gloo in the CPU
          tests), asynchronous and double-buffered so that the gather of scan k overlaps scan k+1.

`mode="gather"` is the default and what `mode="auto"` means: peer stores have been rehearsed (several
processes on ONE GPU) but the gpurun boxes have a single GPU, so until a run on real xGMI has proven
them they are opt-in.  `mode="peer"` raises when some rank cannot map the buffer; `mode="peer_or_gather"`
falls back to the gather instead (decided collectively, so all ranks agree).  A freshly mapped buffer is
never trusted blindly: every rank first stores a test pattern through its slice with a kernel
(pgt_rowbuf_fill: the same plain global stores the query kernels use), `dst` reads the whole buffer back
and compares; a mismatch counts as "cannot map".
"""
from __future__ import annotations

import numpy as np

from ._lib import WIN_DTYPE, PgtError
from .window_scan import plan_shards


def shard_windows(win: np.ndarray, rank: int, world: int):
    """-> (shard record, this rank's windows re-based to its local columns, all shard records)."""
    shards = plan_shards(win, world)
    s = shards[rank]
    local = np.array(win[int(s["win_begin"]): int(s["win_end"])], dtype=WIN_DTYPE, copy=True)
    local["lo"] -= s["site_lo"]
    local["hi"] -= s["site_lo"]
    return s, local, shards


def _global_dst(dist, group, dst: int) -> int:
    """`dst` is a rank of `group`; torch.distributed.gather / broadcast want the global rank."""
    return dist.get_global_rank(group, dst) if group is not None else dst


def _align(x: int, a: int = 256) -> int:
    return (x + a - 1) // a * a


class RowGatherer:
    """gather_rows with its buffers allocated once (the per-scan path): when every rank has the same
    number of rows the local row tensor is sent as it is, no staging copy.  `dst` is a rank of `group`."""

    def __init__(self, counts, row_bytes: int, device, dst: int = 0, group=None, slots: int = 1):
        import torch
        import torch.distributed as dist
        self.dist, self.torch = dist, torch
        self.counts = [int(c) for c in counts]
        self.row_bytes, self.dst, self.group = row_bytes, dst, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.gdst = _global_dst(dist, group, dst)
        self.width = max(max(self.counts) * row_bytes, 1)
        self.uniform = len(set(self.counts)) == 1 and self.counts[0] > 0
        self.send = None if self.uniform else torch.zeros(self.width, dtype=torch.uint8, device=device)
        # one receive set PER SLOT: two gathers in flight (RowExchange) never write the same tensors, so the
        # order in which a backend completes them (gloo: worker threads) cannot mix two scans' rows
        self.recvs = ([[torch.empty(self.width, dtype=torch.uint8, device=device) for _ in range(self.world)]
                       for _ in range(max(1, slots))] if self.rank == dst else None)

    def _staged(self, local_rows):
        if self.uniform:
            return local_rows
        if local_rows is not None and local_rows.numel():
            self.send[: local_rows.numel()].copy_(local_rows)
        return self.send

    def start(self, local_rows):
        """Asynchronous form: issues the gather and returns its Work handle; the caller must keep
        `local_rows` untouched until handle.wait() (a stream-level wait) has been issued, and — when the
        counts differ between ranks, i.e. when the rows go through the one staging buffer — must not
        call start() again before that wait.  RowExchange avoids the staging copy with start_padded()."""
        return self.dist.gather(self._staged(local_rows), self._recv(0), dst=self.gdst, group=self.group, async_op=True)

    def _recv(self, slot):
        return self.recvs[slot] if self.recvs is not None else None

    def start_padded(self, padded_rows, slot: int = 0):
        """Asynchronous gather of a caller-owned buffer of exactly `width` bytes (its first counts[rank] *
        row_bytes bytes are the rows) into receive set `slot`: no staging copy, so several gathers may be
        in flight as long as each has its own send buffer and its own slot."""
        assert padded_rows.numel() == self.width
        return self.dist.gather(padded_rows, self._recv(slot), dst=self.gdst, group=self.group, async_op=True)

    def __call__(self, local_rows):
        """Returns on dst the list of per-rank row tensors (views, valid until the next call)."""
        self.dist.gather(self._staged(local_rows), self._recv(0), dst=self.gdst, group=self.group)
        return self.parts()

    def parts(self, slot: int = 0):
        if self.rank != self.dst:
            return None
        return [self.recvs[slot][r][: self.counts[r] * self.row_bytes] for r in range(self.world)]


def gather_rows(local_rows, counts, row_bytes: int, dst: int = 0, group=None):
    """Gather per-rank packed row tensors (uint8, counts[r]*row_bytes bytes on rank r) to `dst` (a rank
    of `group`).  local_rows: torch uint8 tensor on the rank's device (CUDA for nccl/RCCL, CPU for gloo).
    counts: rows per rank, known to every rank from the shard plan (no size exchange needed).
    Returns on dst the concatenated uint8 tensor in window order; None elsewhere."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return local_rows
    parts = RowGatherer(counts, row_bytes, local_rows.device, dst=dst, group=group)(local_rows)
    return torch.cat(parts) if rank == dst else None


class RowExchange:
    """Rows of every rank's window block -> rank `dst`, once per scan, for a fixed shard plan.

        ex = RowExchange(ctx, counts, row_bytes, device)       # collective (every rank of the group)
        for every scan:
            out = ex.begin()                                   # where this scan's rows go (pass as out=)
            ctx.fst_reduce_dev(..., out=out, ...)
            ex.end()                                           # gather mode: starts the async gather
        table = ex.finish()                                    # collective; uint8 numpy on dst, None elsewhere

    counts[r] = rows rank r produces per scan (tables * windows of its shard).  With tables > 1
    (population pairs) a rank's block is table-major over ITS windows; finish() re-interleaves to
    table-major over ALL windows.  coll_device: where collective tensors live (the GPU for nccl,
    CPU for gloo); defaults to `device`.  gather_to_self: with ONE rank, run the gather anyway (default: rows stay local).
    """

    def __init__(self, ctx, counts, row_bytes: int, device, dst: int = 0, group=None, mode: str = "gather",
                 tables: int = 1, coll_device=None, gather_to_self: bool = False):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.ctx = torch, dist, ctx
        self.counts = [int(c) for c in counts]
        self.row_bytes, self.dst, self.group, self.tables = int(row_bytes), dst, group, int(tables)
        self.device = torch.device(device)
        self.coll_device = torch.device(coll_device) if coll_device is not None else self.device
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.my_bytes = self.counts[self.rank] * self.row_bytes
        self.offsets = [0] * self.world  # 256-byte aligned block starts: no cache line is written by two GPUs
        for r in range(1, self.world):
            self.offsets[r] = _align(self.offsets[r - 1] + self.counts[r - 1] * self.row_bytes)
        self.total = _align(self.offsets[-1] + self.counts[-1] * self.row_bytes)
        if mode not in ("auto", "gather", "peer", "peer_or_gather", "local"):
            raise PgtError(1, f"RowExchange: unknown mode {mode!r}")
        # One rank has nothing to exchange ("local": the rows stay where the kernel wrote them) — unless the caller asks for the
        # gather all the same (gather_to_self, with an initialised process group): the whole per-step machinery of the
        # multi-GPU run (torch.distributed.gather enqueue on RCCL, event waits, two send buffers, two receive sets) then runs
        # on a group of one rank, which is how bench.py prices it on a one-GPU box (extra.exchange_overhead).
        to_self = gather_to_self and mode in ("auto", "gather") and dist.is_initialized()
        self.mode = ("gather" if to_self else "local") if self.world == 1 else ("gather" if mode == "auto" else mode)
        self.buf = None       # peer: the shared row buffer (owned on dst, mapped elsewhere)
        self.peer_error = ""
        if self.mode in ("peer_or_gather", "peer"):
            ok = self._setup_peer()
            if not ok and self.mode == "peer":
                raise PgtError(3, "RowExchange: peer mapping of the row buffer failed on some rank: " + self.peer_error)
            self.mode = "peer" if ok else "gather"
        if self.mode in ("gather", "local"):
            depth = 2 if self.mode == "gather" else 1
            self.gatherer = (RowGatherer(self.counts, self.row_bytes, self.coll_device, dst=dst, group=group, slots=depth)
                             if self.mode == "gather" else None)
            # each slot is a full-width send buffer whose head is this rank's rows: the kernel writes the rows
            # where the gather reads them, no staging copy, two gathers in flight never share a buffer
            width = self.gatherer.width if self.gatherer else max(self.my_bytes, 1)
            self.bufs = [torch.zeros(width, dtype=torch.uint8, device=device) for _ in range(depth)]
            self.outs = [b[: self.my_bytes] for b in self.bufs]
            self.pending = [None] * depth
            self.staged = [None] * depth  # rehearsal over gloo: the CPU copy a gather reads, one per slot
            self.k = 0
            self.last = 0

    def _barrier(self):
        """Barrier of the exchange's group.  On RCCL (collective tensors on the GPU) the rank's own device is NAMED: a group
        created without a bound device (lazily, as bench.py does) would otherwise have torch guess it from the rank."""
        if self.coll_device.type == "cuda":
            self.dist.barrier(group=self.group, device_ids=[self.coll_device.index if self.coll_device.index is not None else self.torch.cuda.current_device()])
        else:
            self.dist.barrier(group=self.group)

    # ---- peer ------------------------------------------------------------------------------
    def _setup_peer(self) -> bool:
        dist, torch = self.dist, self.torch
        gdst = _global_dst(dist, self.group, self.dst)
        box = [None]
        ok = 1
        if self.ctx is None:  # no library context to create / map the buffer with: every rank reports failure
            self.peer_error, ok = "no Context given", 0
        elif self.rank == self.dst:
            try:
                self.buf, handle = self.ctx.rowbuf_create(self.total)
                box[0] = (handle, self.ctx.device)  # the other ranks check peer access to this device first
            except PgtError as e:
                self.peer_error, ok = str(e), 0
        # nccl moves the pickled handle through a tensor on `device`: name this rank's own GPU rather than
        # rely on the process-wide current device
        dist.broadcast_object_list(box, src=gdst, group=self.group,
                                   device=self.coll_device if self.coll_device.type == "cuda" else None)
        if self.rank != self.dst and ok:
            if box[0] is None:
                ok = 0
            else:
                try:
                    self.ctx.peer_access(box[0][1])
                    self.buf = self.ctx.rowbuf_open(box[0][0], self.total)
                except PgtError as e:
                    self.peer_error, ok = str(e), 0
        flag = torch.tensor([ok], dtype=torch.int32, device=self.coll_device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        if int(flag.item()) == 1 and self._peer_selftest():
            self.my_out = self.buf.view(self.offsets[self.rank], self.my_bytes)
            return True
        self._close_peer()
        return False

    def _peer_selftest(self) -> bool:
        """Every rank stores a pattern through its slice of the mapped buffer with a kernel, dst reads the whole
        buffer back and compares.  Collective; all ranks get the same answer."""
        dist, torch = self.dist, self.torch
        ok = 1
        try:
            if self.my_bytes >= 8:
                self.ctx.rowbuf_fill(self.buf.view(self.offsets[self.rank], self.my_bytes // 8 * 8), seed=1000 + self.rank)
            if self.device.type == "cuda":
                torch.cuda.synchronize(self.device)
        except PgtError as e:
            self.peer_error, ok = "self-test store: " + str(e), 0
        self._barrier()  # every rank's stores have completed on its side
        if self.rank == self.dst and ok:
            try:  # a failing read-back must not keep this rank out of the all_reduce below (every other rank waits there)
                raw = self.ctx.rowbuf_read(self.buf, self.total)
                for r in range(self.world):
                    nw = self.counts[r] * self.row_bytes // 8
                    got = raw[self.offsets[r]: self.offsets[r] + 8 * nw].view(np.uint64)
                    if not np.array_equal(got, self.ctx.pattern_words(nw, 1000 + r)):
                        self.peer_error, ok = f"self-test: the pattern rank {r} stored did not arrive in rank {self.dst}'s buffer", 0
                        break
            except PgtError as e:
                self.peer_error, ok = "self-test read-back: " + str(e), 0
        flag = torch.tensor([ok], dtype=torch.int32, device=self.coll_device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return int(flag.item()) == 1

    def _close_peer(self):
        if self.buf is not None:
            try:
                self.ctx.rowbuf_close(self.buf, owner=self.rank == self.dst)
            except PgtError:
                pass
            self.buf = None

    # ---- per scan --------------------------------------------------------------------------
    def begin(self):
        if self.mode == "peer":
            return self.my_out
        w = self.pending[self.k]
        if w is not None:  # the gather that last read this buffer must be done
            # It was issued two scans ago: almost always it HAS completed, and then nothing needs to be put into the launch
            # stream.  A stream-level wait costs more than it looks — a barrier packet in front of the build kernel keeps the
            # build from being dispatched behind the previous query's tail: together with the event the gather itself records
            # that was 17-19 us of idle GPU per scan (profiles/r06/exchange_trace_summary.md).
            done = False
            if self.coll_device.type == "cuda":  # (gloo: wait() is what reports a failed transfer, and costs nothing on the GPU)
                try:
                    done = bool(w.is_completed())
                except Exception:  # noqa: BLE001 — a backend without completion queries: wait as before
                    done = False
            if not done:
                w.wait()
            self.pending[self.k] = None
        return self.outs[self.k]

    def end(self):
        if self.mode == "gather":
            rows = self.bufs[self.k]
            if self.coll_device != self.device and self.coll_device.type == "cpu":
                rows = rows.cpu()  # rehearsal over gloo: CPU staging (synchronises)
            self.staged[self.k] = rows  # keep a staged CPU copy alive until its gather has read it (one per slot)
            self.pending[self.k] = self.gatherer.start_padded(rows, slot=self.k)
            self.last = self.k
            self.k ^= 1
        elif self.mode == "local":
            self.last = 0

    def flush(self):
        """Every transfer issued so far is complete on this rank's side (not a barrier)."""
        if self.mode == "gather":
            for k in range(len(self.pending)):
                if self.pending[k] is not None:
                    self.pending[k].wait()
                    self.pending[k] = None
        if self.device.type == "cuda":
            self.torch.cuda.synchronize()

    def finish(self):
        """Collective: the table of the LAST scan, assembled on dst in window order (uint8 numpy)."""
        torch, dist = self.torch, self.dist
        self.flush()
        if self.world > 1:
            self._barrier()  # peer: every rank's row stores have landed in dst's buffer
        if self.rank != self.dst:
            return None
        if self.mode == "peer":
            raw = self.ctx.rowbuf_read(self.buf, self.total)
            parts = [raw[self.offsets[r]: self.offsets[r] + self.counts[r] * self.row_bytes] for r in range(self.world)]
        elif self.mode == "gather":
            parts = [p.cpu().numpy() for p in self.gatherer.parts(self.last)]
        else:
            parts = [self.outs[0].cpu().numpy()]
        if self.tables == 1:
            return np.concatenate(parts) if len(parts) > 1 else np.array(parts[0], copy=True)
        blocks = [p.reshape(self.tables, -1) for p in parts]  # [tables][rows of rank r * row_bytes]
        return np.concatenate(blocks, axis=1).reshape(-1)

    def close(self):
        if self.world > 1:
            self.flush()
            self._barrier()  # nobody unmaps / frees while a peer may still write
        self._close_peer()


def sharded_scan(win: np.ndarray, row_dtype: np.dtype, load_columns, reduce_rows, device, dst: int = 0, group=None,
                 tables: int = 1, ctx=None, mode: str = "gather", coll_device=None):
    """The whole multi-GPU path for one input, to be called by every rank of the process group.

      win           the full window table (identical on every rank; built on the host in O(#windows))
      load_columns  callable(site_lo, site_hi) -> whatever reduce_rows needs for sites [site_lo, site_hi):
                    a rank only ever touches its own shard (plus a halo of at most one window)
      reduce_rows   callable(columns, local_win, out) -> None; writes tables * len(local_win) packed rows
                    (table-major) into `out`, e.g. lambda c, w, out: ctx.fst_reduce_dev(*c, windows_to_device(w, dev), out=out)
      tables        rows per window: 1, or the number of population pairs of fst_reduce_pairs_dev /
                    fst_af_reduce_dev (BASELINE config 5)
      mode          "gather" (one RCCL/gloo gather; = "auto") or "peer" / "peer_or_gather" (needs ctx: rows stored
                    straight into dst's buffer over xGMI after a self-test of the mapping)
      ctx           when given, its hints (longest window, typical step) are set from the WHOLE table for the
                    duration of the call, so that every rank — and a single-GPU run that does the same — picks the
                    same query strategy and tree levels: rows are bitwise independent of the number of ranks only
                    under identical hints (the strategies sum in different orders)
    Returns on `dst` (a rank of `group`) the assembled rows as a numpy structured array, table-major over
    all windows; None elsewhere.  A rank whose shard holds no window (more ranks than windows) loads
    nothing, reduces nothing and still takes part in the exchange.
    """
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    shard, local_win, shards = shard_windows(win, rank, world)
    counts = (shards["win_end"] - shards["win_begin"]).astype(np.int64) * int(tables)
    ex = RowExchange(ctx, counts, row_dtype.itemsize, device, dst=dst, group=group, mode=mode, tables=tables,
                     coll_device=coll_device)
    out = ex.begin()
    if local_win.size:
        columns = load_columns(int(shard["site_lo"]), int(shard["site_hi"]))
        if ctx is not None:
            with ctx.hints(*table_hints(win)):
                reduce_rows(columns, local_win, out)
        else:
            reduce_rows(columns, local_win, out)
    ex.end()
    packed = ex.finish()
    ex.close()
    if rank != dst:
        return None
    return np.frombuffer(packed.tobytes(), dtype=row_dtype)


def table_hints(win: np.ndarray):
    """(longest window, typical step, typical window) of a whole window table: what every rank that reduces a slice
    of it passes to ctx.hints — the library's own derivation (pgt_table_hints: what the host-buffer entry points
    use while the hints are unset), so a sharded scan takes the strategy and tree levels of the single call."""
    from .window_scan import table_hints as derive
    longest, typical, step = derive(win)
    return longest, step, typical


TOTAL_BLOCK = 1 << 16  # sites per block of the genome-wide dxy total (= the smallest shard alignment)


def sharded_dxy_scan(win: np.ndarray, n_sites: int, load_columns, ctx, minind: int, device, dst: int = 0, group=None,
                     mode: str = "gather", coll_device=None):
    """dxyWindow over several GPUs: the window rows AND the genome-wide line (dxyWindow.cpp:382-385,429-433).

    The total has to see every site once, also sites no window covers (dropped tails), so the site
    axis is cut at the shard starts into one owned range per rank; a rank adds its owned range to its window
    block as windows of TOTAL_BLOCK sites, the rows of those blocks travel with the window rows, and `dst` adds
    them up in block order.  The total is therefore the same bits for every number of ranks (one rank included):
    block starts are multiples of 2^16 on every shard, so a block's sum is always taken over the same tree
    nodes.  It equals pgt_dxy_reduce's own total (since round 5 the sum of the build waves' partial sums in wave
    order — fixed by the static build grid, i.e. by the size of the input) in neff and nskip exactly, and in the
    sum to rounding (different association of the same additions; both are held to 1e-13 of the exact sum by
    tests/test_gpu_parity.py: test_genome_wide_dxy_line_is_pinned_to_the_exact_sum).

      load_columns(site_lo, site_hi) -> (pos, p1, p2, n1, n2) device columns of those sites
    Returns on dst (rows [DXY_ROW_DTYPE, one per window], total [DXY_TOTAL_DTYPE scalar]); (None, None) elsewhere.
    """
    import torch.distributed as dist
    from ._lib import DXY_ROW_DTYPE, DXY_TOTAL_DTYPE, PGT_WIN_COORDS
    from .window_scan import windows_to_device

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    shards = plan_shards(win, world)
    n_sites = int(n_sites)
    # owned site ranges: cut[r] = where rank r's columns start (the next shard start for a rank without windows)
    cut = [0] * (world + 1)
    cut[world] = n_sites
    for r in range(world - 1, 0, -1):
        has = int(shards[r]["win_end"]) > int(shards[r]["win_begin"])
        cut[r] = min(int(shards[r]["site_lo"]), cut[r + 1]) if has else cut[r + 1]
        assert cut[r] % TOTAL_BLOCK == 0 or cut[r] == n_sites, "shard starts are multiples of 2^16 (pgt_plan_shards)"

    def blocks_of(r):
        return range(cut[r] // TOTAL_BLOCK, (cut[r + 1] + TOTAL_BLOCK - 1) // TOTAL_BLOCK) if cut[r + 1] > cut[r] else range(0)
    n_own = [int(shards[r]["win_end"]) - int(shards[r]["win_begin"]) for r in range(world)]
    counts = [n_own[r] + len(blocks_of(r)) for r in range(world)]

    s = shards[rank]
    lo = cut[rank]
    hi = max(int(s["site_hi"]) if n_own[rank] else lo, cut[rank + 1])
    local = np.zeros(counts[rank], dtype=WIN_DTYPE)
    local[: n_own[rank]] = win[int(s["win_begin"]): int(s["win_end"])]
    blk = local[n_own[rank]:]
    first = np.array(list(blocks_of(rank)), dtype=np.uint64) * np.uint64(TOTAL_BLOCK)
    blk["lo"] = first
    blk["hi"] = np.minimum(first + np.uint64(TOTAL_BLOCK), np.uint64(cut[rank + 1]))
    blk["flags"] = PGT_WIN_COORDS  # no coordinates to look up
    local["lo"] -= np.uint64(lo)
    local["hi"] -= np.uint64(lo)

    ex = RowExchange(ctx, counts, DXY_ROW_DTYPE.itemsize, device, dst=dst, group=group, mode=mode, coll_device=coll_device)
    out = ex.begin()
    if counts[rank]:
        pos, p1, p2, n1, n2 = load_columns(lo, hi)
        with ctx.hints(*table_hints(win)):  # strategy and levels follow the whole table, not this rank's slice
            # tot=False: the genome-wide line comes from the 2^16-site block rows below, not from a whole-shard query
            ctx.dxy_reduce_dev(pos, p1, p2, n1, n2, minind, windows_to_device(local, device), out=out, tot=False)
    ex.end()
    packed = ex.finish()
    ex.close()
    if rank != dst:
        return None, None
    allrows = np.frombuffer(packed.tobytes(), dtype=DXY_ROW_DTYPE)
    rows, blocks, at = [], [], 0
    for r in range(world):
        rows.append(allrows[at: at + n_own[r]])
        blocks.append(allrows[at + n_own[r]: at + counts[r]])
        at += counts[r]
    rows, blocks = np.concatenate(rows), np.concatenate(blocks)
    total = np.zeros((), dtype=DXY_TOTAL_DTYPE)
    acc = 0.0
    for v in blocks["sum"].tolist():  # in block order, one addition per block: the same on every rank count
        acc += v
    total["sum"] = acc
    total["neff"] = int(blocks["neff"].astype(np.uint64).sum())
    total["nskip"] = int(blocks["nskip"].astype(np.uint64).sum())
    return rows, total
