"""Host-side mirror of the reference's three window tools over in-memory columns.

Reference interface mirrored (file:line in the reference tree):
  fstWindow  <file> [W] [S]            fstWindow.cpp:37-67,109-155   -> fst_window(chr, pos, a, b, W, S)
  hetWindow  <file> [W] [S]            hetWindow.cpp:34-64,107-153   -> het_window(chr, pos, g, W, S)
  dxyWindow  -winsize -stepsize -minind -fixedsite -sizefile -skip_missing <maf1> <maf2>
                                       dxyWindow.cpp:63-139,253-436  -> dxy_window(...)
Argument meaning, defaults (W=S=1 for fst/het; W=S=0, minind=1, fixedsite=0, skip_missing=0 for
dxy) and error behaviour (an exception where the tool exits 255) follow those lines.

All arithmetic happens in libpgtwin.so on the GPU.  This file only shapes buffers: numpy for
host columns, torch tensors for device-resident columns (torch is plumbing here: allocation,
streams, torch.distributed; it never computes a statistic).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import (DXY_ROW_DTYPE, DXY_TOTAL_DTYPE, EXT_ROW_DTYPE, FST_ROW_DTYPE, HET_ROW_DTYPE, PGT_EXT_IHS,
                   PGT_EXT_XP_MAX, PGT_EXT_XP_MIN, PGT_STAT_DXY, PGT_STAT_EXT, PGT_STAT_FST, PGT_STAT_HET,
                   SHARD_DTYPE, WIN_DTYPE, PgtError, check)


# ---------------------------------------------------------------------------------------------
# window tables (host, no GPU)
# ---------------------------------------------------------------------------------------------
def run_lengths(chr_ids) -> np.ndarray:
    """Lengths of the runs of equal adjacent chromosome ids (the reference compares adjacent
    names only, fstWindow.cpp:132)."""
    c = np.asarray(chr_ids)
    if c.size == 0:
        return np.zeros(0, dtype=np.uint64)
    cuts = np.flatnonzero(c[1:] != c[:-1]) + 1
    edges = np.concatenate(([0], cuts, [c.size]))
    return np.diff(edges).astype(np.uint64)


def build_windows_sites(run_len, W: int, S: int) -> np.ndarray:
    lib = _lib.load()
    rl = np.ascontiguousarray(run_len, dtype=np.uint64)
    n_out = C.c_size_t(0)
    check(lib.pgt_build_windows_sites(rl.ctypes.data, rl.size, W, S, None, 0, C.byref(n_out)))
    out = np.zeros(n_out.value, dtype=WIN_DTYPE)
    check(lib.pgt_build_windows_sites(rl.ctypes.data, rl.size, W, S, out.ctypes.data, out.size, C.byref(n_out)))
    return out


def build_windows_bp(pos, run_len, chr_len, W: int, S: int) -> np.ndarray:
    lib = _lib.load()
    p = np.ascontiguousarray(pos, dtype=np.uint32)
    rl = np.ascontiguousarray(run_len, dtype=np.uint64)
    cl = np.ascontiguousarray(chr_len, dtype=np.uint32)
    if cl.size != rl.size or int(rl.sum()) != p.size:
        raise PgtError(_lib.PGT_EARG, "build_windows_bp: run_len / chr_len / pos sizes disagree")
    n_out = C.c_size_t(0)
    check(lib.pgt_build_windows_bp(p.ctypes.data, rl.ctypes.data, cl.ctypes.data, rl.size, W, S, None, 0, C.byref(n_out)))
    out = np.zeros(n_out.value, dtype=WIN_DTYPE)
    check(lib.pgt_build_windows_bp(p.ctypes.data, rl.ctypes.data, cl.ctypes.data, rl.size, W, S,
                                   out.ctypes.data, out.size, C.byref(n_out)))
    return out


def build_windows_extreme(pos, run_len, chr_len, W: int) -> np.ndarray:
    """ihsWindow / xpehhWindow window table (ihsWindow.cpp:147-218); chr_len[r] = 0 or None where the
    chromosome length is not given."""
    lib = _lib.load()
    p = np.ascontiguousarray(pos, dtype=np.uint32)
    rl = np.ascontiguousarray(run_len, dtype=np.uint64)
    cl = None if chr_len is None else np.ascontiguousarray(chr_len, dtype=np.uint32)
    if int(rl.sum()) != p.size or (cl is not None and cl.size != rl.size):
        raise PgtError(_lib.PGT_EARG, "build_windows_extreme: run_len / chr_len / pos sizes disagree")
    n_out = C.c_size_t(0)
    clp = cl.ctypes.data if cl is not None else None
    check(lib.pgt_build_windows_extreme(p.ctypes.data, rl.ctypes.data, clp, rl.size, W, None, 0, C.byref(n_out)))
    out = np.zeros(n_out.value, dtype=WIN_DTYPE)
    check(lib.pgt_build_windows_extreme(p.ctypes.data, rl.ctypes.data, clp, rl.size, W, out.ctypes.data, out.size,
                                        C.byref(n_out)))
    return out


def plan_shards(win: np.ndarray, n_ranks: int) -> np.ndarray:
    lib = _lib.load()
    w = np.ascontiguousarray(win, dtype=WIN_DTYPE)
    out = np.zeros(n_ranks, dtype=SHARD_DTYPE)
    check(lib.pgt_plan_shards(w.ctypes.data if w.size else None, w.size, n_ranks, out.ctypes.data))
    return out


def table_hints(win: np.ndarray):
    """(longest window, typical window, typical step) of a whole host table, as the host-buffer entry points derive them
    while the hints are unset (pgt_table_hints); the step is 2^64-1 where the table has no typical positive step."""
    import ctypes as C
    lib = _lib.load()
    w = np.ascontiguousarray(win, dtype=WIN_DTYPE)
    m, t, s = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    check(lib.pgt_table_hints(w.ctypes.data if w.size else None, w.size, C.byref(m), C.byref(t), C.byref(s)))
    return m.value, t.value, s.value


# ---------------------------------------------------------------------------------------------
# raw device memory handed out by the library (pgt_rowbuf_*): quacks like the uint8 tensors the
# *_dev wrappers take (data_ptr / numel), so it can be passed as `out=`
# ---------------------------------------------------------------------------------------------
class RowBuffer:
    def __init__(self, ptr: int, nbytes: int, owner=None):
        self._ptr, self._nbytes, self._owner = int(ptr), int(nbytes), owner  # owner keeps the mapping alive

    def data_ptr(self) -> int:
        return self._ptr

    def numel(self) -> int:
        return self._nbytes

    def view(self, offset: int, nbytes: int) -> "RowBuffer":
        if offset < 0 or nbytes < 0 or offset + nbytes > self._nbytes:
            raise PgtError(_lib.PGT_EARG, "RowBuffer.view outside the buffer")
        return RowBuffer(self._ptr + offset, nbytes, self._owner or self)


class DeviceColumn:
    """A typed column in device memory that the library owns (pgt_ingest_column): accepted by the *_dev
    wrappers wherever a CUDA tensor of that dtype is."""

    def __init__(self, ptr: int, numel: int, dtype, owner):
        self._ptr, self._numel, self.dtype, self._owner = int(ptr), int(numel), dtype, owner

    def data_ptr(self) -> int:
        return self._ptr

    def numel(self) -> int:
        return self._numel


class Ingest:
    """Result of Context.ingest_text: parsed columns on the GPU + chromosome runs on the host."""

    def __init__(self, ctx, handle, text: bytes, tokens):
        self._ctx, self._h, self._text, self.tokens = ctx, handle, text, list(tokens)
        lib = ctx._lib
        self.rows = int(lib.pgt_ingest_rows(handle))
        self.bad_line = int(lib.pgt_ingest_bad_line(handle))
        rl, off, ln = C.c_void_p(0), C.c_void_p(0), C.c_void_p(0)
        n = int(lib.pgt_ingest_runs(handle, C.byref(rl), C.byref(off), C.byref(ln)))
        arr = lambda p, t: np.ctypeslib.as_array(C.cast(p, C.POINTER(t)), shape=(n,)).copy() if n else np.zeros(0, dtype=t)  # noqa: E731
        self.run_len = arr(rl, C.c_uint64)
        offs, lens = arr(off, C.c_uint64), arr(ln, C.c_uint32)
        self.run_names = [text[int(o): int(o) + int(k)].decode("latin-1") for o, k in zip(offs, lens)]

    def column(self, token: int):
        """Device column of token `token` (DeviceColumn), as the *_dev wrappers take it."""
        import torch
        dt = {_lib.PGT_TOK_U32: torch.int32, _lib.PGT_TOK_I32: torch.int32, _lib.PGT_TOK_F64: torch.float64,
              _lib.PGT_TOK_FREQ: torch.float64, _lib.PGT_TOK_I8: torch.int8}[self.tokens[token]]
        return DeviceColumn(self._ctx._lib.pgt_ingest_column(self._h, token), self.rows, dt, self)

    def column_np(self, token: int) -> np.ndarray:
        """Host copy of a column (u32 / i32 / f64 / i8 by token kind)."""
        dt = {_lib.PGT_TOK_U32: np.uint32, _lib.PGT_TOK_I32: np.int32, _lib.PGT_TOK_F64: np.float64,
              _lib.PGT_TOK_FREQ: np.float64, _lib.PGT_TOK_I8: np.int8}[self.tokens[token]]
        ptr = self._ctx._lib.pgt_ingest_column(self._h, token)
        raw = self._ctx.rowbuf_read(RowBuffer(ptr, self.rows * np.dtype(dt).itemsize)) if self.rows else np.zeros(0, np.uint8)
        return raw.view(dt)

    def free(self):
        if self._h:
            self._ctx._lib.pgt_ingest_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class WindowTable:
    """A site-window table built on the GPU (pgt_wintab_sites): `n_win` windows, `first[r]` = index of run r's
    first window (n_runs + 1 values), `device()` = the table as a uint8 view usable as `win` of the *_dev calls."""

    def __init__(self, ctx, handle, n_runs):
        self._ctx, self._h = ctx, handle
        lib = ctx._lib
        self.n_win = int(lib.pgt_wintab_size(handle))
        self.first = np.ctypeslib.as_array(C.cast(lib.pgt_wintab_first(handle), C.POINTER(C.c_uint64)), shape=(n_runs + 1,)).copy()

    def device(self):
        return RowBuffer(self._ctx._lib.pgt_wintab_device(self._h) or 0, self.n_win * WIN_DTYPE.itemsize, owner=self)

    def to_host(self) -> np.ndarray:
        raw = self._ctx.rowbuf_read(self.device()) if self.n_win else np.zeros(0, np.uint8)
        return raw.view(WIN_DTYPE)

    def labels(self) -> np.ndarray:
        """label_run of every window, from `first` alone"""
        return (np.searchsorted(self.first, np.arange(self.n_win, dtype=np.uint64), side="right") - 1).astype(np.uint32)

    def free(self):
        if self._h:
            self._ctx._lib.pgt_wintab_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# ---------------------------------------------------------------------------------------------
# context
# ---------------------------------------------------------------------------------------------
class Context:
    """One GPU, one thread (include/pgtwin.h conventions).  Raises PgtError if no gfx950 device
    is usable: there is no CPU path."""

    def __init__(self, device: int = -1):
        self._lib = _lib.load()
        self._ctx = self._lib.pgt_open(device)
        if not self._ctx:
            raise PgtError(_lib.PGT_EDEVICE, _lib.last_error(None))
        if device < 0:  # "the current device": ask the runtime which one that was
            import torch
            device = torch.cuda.current_device()
        self.device = int(device)

    def close(self):
        if self._ctx:
            self._lib.pgt_close(self._ctx)
            self._ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        check(rc, self._ctx)

    # ---- host buffers ------------------------------------------------------------------
    def prepare_host_io(self, expected_column_bytes: int = 0):
        """pgt_prepare_host_io: allocate the pinned staging ring of the host-buffer calls below (unless the expected upload is
        under 32 MiB; 0 = unknown) and make the runtime set up its first copies NOW (30 … 90 ms once per process) — worth
        calling from a thread that opens the device beside a long parse."""
        self._check(self._lib.pgt_prepare_host_io(self._ctx, int(expected_column_bytes)))

    def fst_reduce(self, pos, a, b, win) -> np.ndarray:
        pos = np.ascontiguousarray(pos, dtype=np.uint32)
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        win = np.ascontiguousarray(win, dtype=WIN_DTYPE)
        if not (pos.size == a.size == b.size):
            raise PgtError(_lib.PGT_EARG, "fst_reduce: column lengths differ")
        out = np.zeros(win.size, dtype=FST_ROW_DTYPE)
        self._check(self._lib.pgt_fst_reduce(self._ctx, pos.ctypes.data, a.ctypes.data, b.ctypes.data, pos.size,
                                             win.ctypes.data, win.size, out.ctypes.data))
        return out

    def het_reduce(self, pos, g, win) -> np.ndarray:
        pos = np.ascontiguousarray(pos, dtype=np.uint32)
        g = np.ascontiguousarray(g, dtype=np.int8)
        win = np.ascontiguousarray(win, dtype=WIN_DTYPE)
        if pos.size != g.size:
            raise PgtError(_lib.PGT_EARG, "het_reduce: column lengths differ")
        out = np.zeros(win.size, dtype=HET_ROW_DTYPE)
        self._check(self._lib.pgt_het_reduce(self._ctx, pos.ctypes.data, g.ctypes.data, pos.size,
                                             win.ctypes.data, win.size, out.ctypes.data))
        return out

    def extreme_reduce(self, pos, score, mode, cutoff, win) -> np.ndarray:
        pos = np.ascontiguousarray(pos, dtype=np.uint32)
        score = np.ascontiguousarray(score, dtype=np.float64)
        win = np.ascontiguousarray(win, dtype=WIN_DTYPE)
        if pos.size != score.size:
            raise PgtError(_lib.PGT_EARG, "extreme_reduce: column lengths differ")
        out = np.zeros(win.size, dtype=EXT_ROW_DTYPE)
        self._check(self._lib.pgt_extreme_reduce(self._ctx, pos.ctypes.data, score.ctypes.data, pos.size, int(mode),
                                                 float(cutoff), win.ctypes.data, win.size, out.ctypes.data))
        return out

    def dxy_reduce(self, pos, p1, p2, n1, n2, minind, win):
        pos = np.ascontiguousarray(pos, dtype=np.uint32)
        p1 = np.ascontiguousarray(p1, dtype=np.float64)
        p2 = np.ascontiguousarray(p2, dtype=np.float64)
        n1 = np.ascontiguousarray(n1, dtype=np.int32)
        n2 = np.ascontiguousarray(n2, dtype=np.int32)
        win = np.ascontiguousarray(win, dtype=WIN_DTYPE)
        if not (pos.size == p1.size == p2.size == n1.size == n2.size):
            raise PgtError(_lib.PGT_EARG, "dxy_reduce: column lengths differ")
        out = np.zeros(win.size, dtype=DXY_ROW_DTYPE)
        tot = np.zeros(1, dtype=DXY_TOTAL_DTYPE)
        self._check(self._lib.pgt_dxy_reduce(self._ctx, pos.ctypes.data, p1.ctypes.data, p2.ctypes.data,
                                             n1.ctypes.data, n2.ctypes.data, pos.size, int(minind),
                                             win.ctypes.data, win.size, out.ctypes.data, tot.ctypes.data))
        return out, tot[0]

    # ---- device-resident columns (torch CUDA tensors) -------------------------------------
    @staticmethod
    def _stream(stream):
        import torch
        s = stream if stream is not None else torch.cuda.current_stream()
        return C.c_void_p(s.cuda_stream)

    @staticmethod
    def tree_bytes(stat: int, n_sites: int) -> int:
        return int(_lib.load().pgt_tree_bytes(stat, n_sites))

    def _dev(self, t, dtype, name):
        import torch
        if isinstance(t, RowBuffer) and dtype == torch.uint8:
            return t.data_ptr()
        if isinstance(t, DeviceColumn) and t.dtype == dtype:
            return t.data_ptr() if t.numel() else None
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.is_contiguous() and t.dtype == dtype):
            raise PgtError(_lib.PGT_EARG, f"{name}: expected a contiguous CUDA tensor of {dtype}")
        if t.device.index != self.device:  # the kernels run on the context's GPU: memory of another one may not even be mapped there
            raise PgtError(_lib.PGT_EARG, f"{name}: tensor lives on cuda:{t.device.index}, the context on cuda:{self.device}")
        return t.data_ptr() if t.numel() else None  # an empty shard passes NULL columns (n == 0)

    def _col(self, t, dtype, name):
        """A COLUMN the build kernels stream: as _dev, plus the alignment the C ABI demands (16 bytes: every column is read
        by 16-byte loads; ABI 5 extended that to the i32 count columns).  A view that starts at an odd site offset is
        refused HERE, by name, with the offsets that work — the library's own message cannot name the tensor."""
        ptr = self._dev(t, dtype, name)
        if ptr is not None and ptr % 16:
            import torch
            per = 16 // torch.empty(0, dtype=dtype).element_size()
            raise PgtError(_lib.PGT_EARG, f"{name}: column starts {ptr % 16} bytes past a 16-byte boundary; slice columns at site offsets "
                                          f"that are multiples of {per} for {dtype} (or .clone() the view)")
        return ptr

    @staticmethod
    def _same_len(name, n, *cols):
        """The C ABI takes one n for all columns: a short column would be read out of bounds."""
        for c in cols:
            if c.numel() != n:
                raise PgtError(_lib.PGT_EARG, f"{name}: column lengths differ ({c.numel()} vs {n})")

    @staticmethod
    def _room(name, buf, need_bytes):
        """Early, readable refusal; the C ABI (version 4) checks the same capacities again before any launch."""
        if buf.numel() < need_bytes:
            raise PgtError(_lib.PGT_EARG, f"{name}: buffer holds {buf.numel()} bytes, {need_bytes} needed")

    # ---- device-side text ingest (pgt_ingest_*) ---------------------------------------------
    def ingest_text(self, text: bytes, tokens) -> Ingest:
        """Parse whitespace-separated text lines on the GPU.  tokens: PGT_TOK_* per column, the first being
        PGT_TOK_CHR (e.g. fstWindow: [CHR, U32, F64, F64]).  Raises PgtError(PGT_EDOMAIN) when the input
        has too many irregular lines for the device path."""
        toks = (C.c_uint8 * len(tokens))(*tokens)
        h = C.c_void_p(0)
        text = bytes(text)
        self._check(self._lib.pgt_ingest_text(self._ctx, text if text else None, len(text), toks, len(tokens), C.byref(h)))
        return Ingest(self, h, text, tokens)

    # ---- window tables built on the device (pgt_wintab_*) -----------------------------------
    def window_table_sites(self, run_len, W: int, S: int) -> WindowTable:
        run_len = np.ascontiguousarray(run_len, dtype=np.uint64)
        h = C.c_void_p(0)
        self._check(self._lib.pgt_wintab_sites(self._ctx, run_len.ctypes.data, run_len.size, int(W), int(S), C.byref(h)))
        return WindowTable(self, h, run_len.size)

    def fst_reduce_tab(self, pos, a, b, tab: WindowTable) -> np.ndarray:
        """Host columns + a device window table -> host rows (pgt_fst_reduce_tab)."""
        pos = np.ascontiguousarray(pos, dtype=np.uint32)
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        if not (pos.size == a.size == b.size):
            raise PgtError(_lib.PGT_EARG, "fst_reduce_tab: column lengths differ")
        out = np.zeros(tab.n_win, dtype=FST_ROW_DTYPE)
        self._check(self._lib.pgt_fst_reduce_tab(self._ctx, pos.ctypes.data, a.ctypes.data, b.ctypes.data, pos.size, 0, tab._h,
                                                 out.ctypes.data, out.nbytes))
        return out

    def het_reduce_tab(self, pos, g, tab: WindowTable) -> np.ndarray:
        pos = np.ascontiguousarray(pos, dtype=np.uint32)
        g = np.ascontiguousarray(g, dtype=np.int8)
        if pos.size != g.size:
            raise PgtError(_lib.PGT_EARG, "het_reduce_tab: column lengths differ")
        out = np.zeros(tab.n_win, dtype=HET_ROW_DTYPE)
        self._check(self._lib.pgt_het_reduce_tab(self._ctx, pos.ctypes.data, g.ctypes.data, pos.size, 0, tab._h, out.ctypes.data,
                                                 out.nbytes))
        return out

    def dxy_reduce_tab(self, pos, p1, p2, n1, n2, minind, tab: WindowTable):
        pos = np.ascontiguousarray(pos, dtype=np.uint32)
        p1 = np.ascontiguousarray(p1, dtype=np.float64)
        p2 = np.ascontiguousarray(p2, dtype=np.float64)
        n1 = np.ascontiguousarray(n1, dtype=np.int32)
        n2 = np.ascontiguousarray(n2, dtype=np.int32)
        if not (pos.size == p1.size == p2.size == n1.size == n2.size):
            raise PgtError(_lib.PGT_EARG, "dxy_reduce_tab: column lengths differ")
        out = np.zeros(tab.n_win, dtype=DXY_ROW_DTYPE)
        tot = np.zeros(1, dtype=DXY_TOTAL_DTYPE)
        self._check(self._lib.pgt_dxy_reduce_tab(self._ctx, pos.ctypes.data, p1.ctypes.data, p2.ctypes.data, n1.ctypes.data,
                                                 n2.ctypes.data, pos.size, int(minind), 0, tab._h, out.ctypes.data, out.nbytes,
                                                 tot.ctypes.data))
        return out, tot[0]

    # ---- multi-GPU row buffer (pgt_rowbuf_*) ------------------------------------------------
    def rowbuf_create(self, nbytes: int):
        """-> (RowBuffer on this GPU, 64-byte IPC handle to ship to the other ranks)."""
        ptr = C.c_void_p(0)
        handle = (C.c_ubyte * 64)()
        self._check(self._lib.pgt_rowbuf_create(self._ctx, int(nbytes), C.byref(ptr), handle))
        return RowBuffer(ptr.value, nbytes), bytes(handle)

    def peer_access(self, peer_device: int):
        """Raises PgtError unless this context's GPU can address memory of HIP device `peer_device`."""
        self._check(self._lib.pgt_peer_access(self._ctx, int(peer_device)))

    def rowbuf_open(self, handle: bytes, nbytes: int) -> RowBuffer:
        ptr = C.c_void_p(0)
        h = (C.c_ubyte * 64).from_buffer_copy(handle)
        self._check(self._lib.pgt_rowbuf_open(self._ctx, h, C.byref(ptr)))
        return RowBuffer(ptr.value, nbytes)

    def rowbuf_close(self, buf: RowBuffer, owner: bool):
        self._check(self._lib.pgt_rowbuf_close(self._ctx, C.c_void_p(buf.data_ptr()), int(owner)))

    def rowbuf_fill(self, buf: RowBuffer, seed: int, stream=None):
        """Store the test pattern of pgt_rowbuf_fill through `buf` (asynchronous); pattern_words() is what must read back."""
        self._check(self._lib.pgt_rowbuf_fill(self._ctx, C.c_void_p(buf.data_ptr()), buf.numel() // 8 * 8, int(seed),
                                              self._stream(stream)))

    @staticmethod
    def pattern_words(n_words: int, seed: int) -> np.ndarray:
        """The words pgt_rowbuf_fill writes: splitmix64(seed + i)."""
        with np.errstate(over="ignore"):
            z = np.arange(n_words, dtype=np.uint64) + np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            return z ^ (z >> np.uint64(31))

    def rowbuf_read(self, buf: RowBuffer, nbytes: int | None = None, stream=None) -> np.ndarray:
        nbytes = buf.numel() if nbytes is None else int(nbytes)
        host = np.empty(nbytes, dtype=np.uint8)
        self._check(self._lib.pgt_rowbuf_read(self._ctx, host.ctypes.data, C.c_void_p(buf.data_ptr()), nbytes,
                                              self._stream(stream)))
        return host

    def fst_reduce_dev(self, pos, a, b, win, out=None, tree=None, stream=None):
        """pos u32-as-int32 [n], a/b float64 [n], win uint8 [n_win*32] (WIN_DTYPE bytes) on the GPU.
        Returns (out uint8 [n_win*40], tree).  Asynchronous on `stream`."""
        import torch
        n = a.numel()
        n_win = win.numel() // WIN_DTYPE.itemsize
        tb = self.tree_bytes(PGT_STAT_FST, n)
        if tree is None:
            tree = torch.empty(tb, dtype=torch.uint8, device=a.device)
        if out is None:
            out = torch.empty(n_win * FST_ROW_DTYPE.itemsize, dtype=torch.uint8, device=a.device)
        self._same_len("fst_reduce_dev", n, pos, a, b)
        self._room("fst_reduce_dev: out", out, n_win * FST_ROW_DTYPE.itemsize)
        self._room("fst_reduce_dev: tree", tree, tb)
        self._check(self._lib.pgt_fst_reduce_dev(
            self._ctx, self._dev(pos, torch.int32, "pos"), self._col(a, torch.float64, "a"),
            self._col(b, torch.float64, "b"), n, self._dev(win, torch.uint8, "win"), n_win,
            self._dev(out, torch.uint8, "out"), out.numel(), self._dev(tree, torch.uint8, "tree"), tree.numel(),
            self._stream(stream)))
        return out, tree

    def fst_reduce_pairs_dev(self, pos, a_list, b_list, win, out=None, tree=None, stream=None):
        import torch
        n = a_list[0].numel()
        n_pairs = len(a_list)
        n_win = win.numel() // WIN_DTYPE.itemsize
        tb = self.tree_bytes(PGT_STAT_FST, n) * n_pairs
        if tree is None:
            tree = torch.empty(tb, dtype=torch.uint8, device=pos.device)
        if out is None:
            out = torch.empty(n_pairs * n_win * FST_ROW_DTYPE.itemsize, dtype=torch.uint8, device=pos.device)
        self._same_len("fst_reduce_pairs_dev", n, pos, *a_list, *b_list)
        self._room("fst_reduce_pairs_dev: out", out, n_pairs * n_win * FST_ROW_DTYPE.itemsize)
        self._room("fst_reduce_pairs_dev: tree", tree, tb)
        pa = (C.c_void_p * n_pairs)(*[self._col(t, torch.float64, "a") for t in a_list])
        pb = (C.c_void_p * n_pairs)(*[self._col(t, torch.float64, "b") for t in b_list])
        self._check(self._lib.pgt_fst_reduce_pairs_dev(
            self._ctx, self._dev(pos, torch.int32, "pos"), pa, pb, n_pairs, n,
            self._dev(win, torch.uint8, "win"), n_win, self._dev(out, torch.uint8, "out"), out.numel(),
            self._dev(tree, torch.uint8, "tree"), tree.numel(), self._stream(stream)))
        return out, tree

    def het_reduce_dev(self, pos, g, win, out=None, tree=None, stream=None):
        import torch
        n = g.numel()
        n_win = win.numel() // WIN_DTYPE.itemsize
        if tree is None:
            tree = torch.empty(self.tree_bytes(PGT_STAT_HET, n), dtype=torch.uint8, device=g.device)
        if out is None:
            out = torch.empty(n_win * HET_ROW_DTYPE.itemsize, dtype=torch.uint8, device=g.device)
        self._same_len("het_reduce_dev", n, pos, g)
        self._room("het_reduce_dev: out", out, n_win * HET_ROW_DTYPE.itemsize)
        self._room("het_reduce_dev: tree", tree, self.tree_bytes(PGT_STAT_HET, n))
        self._check(self._lib.pgt_het_reduce_dev(
            self._ctx, self._dev(pos, torch.int32, "pos"), self._col(g, torch.int8, "g"), n,
            self._dev(win, torch.uint8, "win"), n_win, self._dev(out, torch.uint8, "out"), out.numel(),
            self._dev(tree, torch.uint8, "tree"), tree.numel(), self._stream(stream)))
        return out, tree

    def dxy_reduce_dev(self, pos, p1, p2, n1, n2, minind, win, out=None, tot=None, tree=None, stream=None):
        """tot: None = a fresh total buffer is allocated and filled; False = no genome-wide total is wanted (the C
        ABI's tot == NULL: only the tree levels the windows need are built, no whole-input query runs)."""
        import torch
        n = p1.numel()
        n_win = win.numel() // WIN_DTYPE.itemsize
        if tree is None:
            tree = torch.empty(self.tree_bytes(PGT_STAT_DXY, n), dtype=torch.uint8, device=p1.device)
        if out is None:
            out = torch.empty(n_win * DXY_ROW_DTYPE.itemsize, dtype=torch.uint8, device=p1.device)
        if tot is None:
            tot = torch.empty(DXY_TOTAL_DTYPE.itemsize, dtype=torch.uint8, device=p1.device)
        elif tot is False:
            tot = None
        self._same_len("dxy_reduce_dev", n, pos, p1, p2, n1, n2)
        self._room("dxy_reduce_dev: out", out, n_win * DXY_ROW_DTYPE.itemsize)
        self._room("dxy_reduce_dev: tree", tree, self.tree_bytes(PGT_STAT_DXY, n))
        self._check(self._lib.pgt_dxy_reduce_dev(
            self._ctx, self._dev(pos, torch.int32, "pos"), self._col(p1, torch.float64, "p1"),
            self._col(p2, torch.float64, "p2"), self._col(n1, torch.int32, "n1"), self._col(n2, torch.int32, "n2"),
            n, int(minind), self._dev(win, torch.uint8, "win") if n_win else None, n_win,
            self._dev(out, torch.uint8, "out") if n_win else None, out.numel(),
            self._dev(tot, torch.uint8, "tot") if tot is not None else None,
            self._dev(tree, torch.uint8, "tree"), tree.numel(), self._stream(stream)))
        return out, tot, tree

    def dxy_het_reduce_dev(self, pos, p1, p2, n1, n2, g1, g2, minind, win, tree=None, stream=None):
        """BASELINE config 3: dxy + het(g1) + het(g2) over one pos column / window table, two launches."""
        import torch
        n = p1.numel()
        n_win = win.numel() // WIN_DTYPE.itemsize
        dev = p1.device
        if tree is None:
            tree = torch.empty(self.tree_bytes(PGT_STAT_DXY, n) + 2 * self.tree_bytes(PGT_STAT_HET, n),
                               dtype=torch.uint8, device=dev)
        dxy_out = torch.empty(n_win * DXY_ROW_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        tot = torch.empty(DXY_TOTAL_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        h1 = torch.empty(n_win * HET_ROW_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        h2 = torch.empty(n_win * HET_ROW_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        self._same_len("dxy_het_reduce_dev", n, pos, p1, p2, n1, n2, g1, g2)
        self._room("dxy_het_reduce_dev: tree", tree, self.tree_bytes(PGT_STAT_DXY, n) + 2 * self.tree_bytes(PGT_STAT_HET, n))
        self._check(self._lib.pgt_dxy_het_reduce_dev(
            self._ctx, self._dev(pos, torch.int32, "pos"), self._col(p1, torch.float64, "p1"),
            self._col(p2, torch.float64, "p2"), self._col(n1, torch.int32, "n1"), self._col(n2, torch.int32, "n2"),
            self._col(g1, torch.int8, "g1"), self._col(g2, torch.int8, "g2"), n, int(minind),
            self._dev(win, torch.uint8, "win"), n_win, self._dev(dxy_out, torch.uint8, "dxy_out"), dxy_out.numel(),
            self._dev(tot, torch.uint8, "tot"), self._dev(h1, torch.uint8, "het_out1"),
            self._dev(h2, torch.uint8, "het_out2"), min(h1.numel(), h2.numel()), self._dev(tree, torch.uint8, "tree"),
            tree.numel(), self._stream(stream)))
        return dxy_out, tot, h1, h2, tree

    def fst_af_reduce_dev(self, pos, freqs, nsamp, win, out=None, tree=None, stream=None):
        """Allele frequencies of len(freqs) populations -> FST rows of all pairs i<j (pair-major),
        WCFst() of betaAFOutlier.R:400-418 + fstWindow's Σa/Σ(a+b).  freqs: float64 CUDA tensors."""
        import torch
        n_pops = len(freqs)
        n = freqs[0].numel()
        n_pairs = n_pops * (n_pops - 1) // 2
        n_win = win.numel() // WIN_DTYPE.itemsize
        tb = int(self._lib.pgt_af_tree_bytes(n_pops, n))
        if tree is None:
            tree = torch.empty(tb, dtype=torch.uint8, device=pos.device)
        if out is None:
            out = torch.empty(n_pairs * n_win * FST_ROW_DTYPE.itemsize, dtype=torch.uint8, device=pos.device)
        self._same_len("fst_af_reduce_dev", n, pos, *freqs)
        self._room("fst_af_reduce_dev: out", out, n_pairs * n_win * FST_ROW_DTYPE.itemsize)
        self._room("fst_af_reduce_dev: tree", tree, tb)
        pf = (C.c_void_p * n_pops)(*[self._col(t, torch.float64, "freq") for t in freqs])
        ns = (C.c_double * n_pops)(*[float(x) for x in nsamp])
        self._check(self._lib.pgt_fst_af_reduce_dev(
            self._ctx, self._dev(pos, torch.int32, "pos"), pf, ns, n_pops, n, self._dev(win, torch.uint8, "win"),
            n_win, self._dev(out, torch.uint8, "out"), out.numel(), self._dev(tree, torch.uint8, "tree"), tree.numel(),
            self._stream(stream)))
        return out, tree

    def extreme_reduce_dev(self, pos, score, mode, cutoff, win, out=None, tree=None, stream=None):
        """ihsWindow / xpehhWindow rows from a device-resident score column (mode: PGT_EXT_*)."""
        import torch
        n = score.numel()
        n_win = win.numel() // WIN_DTYPE.itemsize
        tb = self.tree_bytes(PGT_STAT_EXT, n)
        if tree is None:
            tree = torch.empty(tb, dtype=torch.uint8, device=score.device)
        if out is None:
            out = torch.empty(n_win * EXT_ROW_DTYPE.itemsize, dtype=torch.uint8, device=score.device)
        self._same_len("extreme_reduce_dev", n, pos, score)
        self._room("extreme_reduce_dev: out", out, n_win * EXT_ROW_DTYPE.itemsize)
        self._room("extreme_reduce_dev: tree", tree, tb)
        self._check(self._lib.pgt_extreme_reduce_dev(
            self._ctx, self._dev(pos, torch.int32, "pos"), self._col(score, torch.float64, "score"), n, int(mode),
            float(cutoff), self._dev(win, torch.uint8, "win"), n_win, self._dev(out, torch.uint8, "out"), out.numel(),
            self._dev(tree, torch.uint8, "tree"), tree.numel(), self._stream(stream)))
        return out, tree

    def set_max_window(self, sites: int):
        """Performance hint for the *_dev calls: no window is longer than `sites` (0 = unknown)."""
        self._check(self._lib.pgt_set_max_window(self._ctx, int(sites)))
        self._hints = (int(sites), getattr(self, "_hints", (0, 0))[1])
        self._typical = 0  # the library resets it: the explicit longest window stands for the typical one

    def set_window_step(self, sites: int):
        """Performance hint for the *_dev calls: consecutive windows start `sites` apart (0 = unknown);
        small steps select the sliding query (include/pgtwin.h)."""
        self._check(self._lib.pgt_set_window_step(self._ctx, int(sites)))
        self._hints = (getattr(self, "_hints", (0, 0))[0], int(sites))

    def set_typical_window(self, sites: int):
        """Performance hint for tables whose windows vary in length (base-pair windows): the typical length decides
        between the query strategies instead of the longest.  After set_max_window, which resets it (0 = the same)."""
        self._check(self._lib.pgt_set_typical_window(self._ctx, int(sites)))
        self._typical = int(sites)

    def hints(self, max_window: int, window_step: int, typical_window: int = 0):
        """with ctx.hints(max_window, step[, typical]): ... — the hints for the duration of the block, then the previous ones."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            saved = getattr(self, "_hints", (0, 0)) + (getattr(self, "_typical", 0),)
            self.set_max_window(max_window)
            self.set_window_step(window_step)
            self.set_typical_window(typical_window)
            try:
                yield self
            finally:
                self.set_max_window(saved[0])
                self.set_window_step(saved[1])
                self.set_typical_window(saved[2])
        return scope()

    # ---- per-kernel timing ------------------------------------------------------------------
    def set_profiling(self, enabled: bool):
        self._check(self._lib.pgt_set_profiling(self._ctx, int(enabled)))

    def last_kernel_ms(self):
        b, q = C.c_float(0), C.c_float(0)
        self._check(self._lib.pgt_last_kernel_ms(self._ctx, C.byref(b), C.byref(q)))
        return b.value, q.value


def rows_from_device(t, dtype: np.dtype) -> np.ndarray:
    """uint8 CUDA tensor of packed rows -> numpy structured array (synchronises)."""
    return np.frombuffer(t.cpu().numpy().tobytes(), dtype=dtype)


def windows_to_device(win: np.ndarray, device):
    import torch
    raw = np.ascontiguousarray(win, dtype=WIN_DTYPE).view(np.uint8)
    return torch.from_numpy(raw.copy()).to(device)


# ---------------------------------------------------------------------------------------------
# the three tools over in-memory columns
# ---------------------------------------------------------------------------------------------
@dataclass
class WindowResult:
    win: np.ndarray   # WIN_DTYPE rows (label_run indexes the chromosome runs)
    rows: np.ndarray  # per-tool row dtype
    total: object = None  # dxy only: DXY_TOTAL_DTYPE scalar


def _own_ctx(ctx):
    return (Context(), True) if ctx is None else (ctx, False)


def fst_window(chr_ids, pos, a, b, W: int = 1, S: int = 1, ctx: Context | None = None) -> WindowResult:
    """fstWindow.cpp:109-155 over columns: rows are (start, end, mid, n, fst)."""
    win = build_windows_sites(run_lengths(chr_ids), W, S)
    ctx, own = _own_ctx(ctx)
    try:
        return WindowResult(win, ctx.fst_reduce(pos, a, b, win))
    finally:
        if own:
            ctx.close()


def het_window(chr_ids, pos, g, W: int = 1, S: int = 1, ctx: Context | None = None) -> WindowResult:
    """hetWindow.cpp:107-153 over columns; genotypes are clipped to int8 (only `>= 0` and `== 1`
    matter, hetWindow.cpp:78-80)."""
    win = build_windows_sites(run_lengths(chr_ids), W, S)
    g8 = np.clip(np.asarray(g), -128, 127).astype(np.int8)
    ctx, own = _own_ctx(ctx)
    try:
        return WindowResult(win, ctx.het_reduce(pos, g8, win))
    finally:
        if own:
            ctx.close()


def dxy_window(chr_ids, pos, p1, p2, n1, n2, W: int = 0, S: int = 0, minind: int = 1, fixedsite: int = 0,
               chr_len=None, skip_missing: int = 0, ctx: Context | None = None) -> WindowResult:
    """dxyWindow.cpp:253-436 over two already synchronised populations.  chr_len[r] is the -sizefile
    length of run r (required unless fixedsite).  Rows suppressed by -skip_missing are dropped, as
    dxyWindow.cpp:189 does."""
    if minind <= 0:
        raise PgtError(_lib.PGT_EARG, "-minind must be at least 1")  # dxyWindow.cpp:105-108
    if W > 0 and S < 1:
        raise PgtError(_lib.PGT_EARG, "Must specify a -stepsize > 0 when -winsize is > 0")  # :128-131
    if not fixedsite and chr_len is None:
        raise PgtError(_lib.PGT_EARG, "Must supply size file unless -fixedsite 1")  # :133-136
    if W == 0 and not fixedsite:
        raise PgtError(_lib.PGT_EDOMAIN, "-winsize 0 needs -fixedsite 1 (the reference crashes here, SURVEY Q10)")
    rl = run_lengths(chr_ids)
    if W == 0:
        win = np.zeros(0, dtype=WIN_DTYPE)
    elif fixedsite:
        win = build_windows_sites(rl, W, S)
    else:
        win = build_windows_bp(pos, rl, chr_len, W, S)
    ctx, own = _own_ctx(ctx)
    try:
        rows, tot = ctx.dxy_reduce(pos, p1, p2, n1, n2, minind, win)
    finally:
        if own:
            ctx.close()
    if skip_missing:
        keep = rows["neff"] > 0
        win, rows = win[keep], rows[keep]
    return WindowResult(win, rows, tot)


def ihs_window(chr_ids, pos, score, W: int = 100000, cutoff: float = 2.0, chr_len=None,
               ctx: Context | None = None) -> WindowResult:
    """ihsWindow.cpp:123-221 over columns (defaults of ihsWindow.cpp:225-226): rows are
    (start, end, nsites, nbig, position, value); proportion = nbig / nsites."""
    if cutoff < 0:
        raise PgtError(_lib.PGT_EARG, "|iHS| cutoff must be >= zero")  # ihsWindow.cpp:60-63
    win = build_windows_extreme(pos, run_lengths(chr_ids), chr_len, W)
    ctx, own = _own_ctx(ctx)
    try:
        return WindowResult(win, ctx.extreme_reduce(pos, score, PGT_EXT_IHS, cutoff, win))
    finally:
        if own:
            ctx.close()


def xpehh_window(chr_ids, pos, score, cutoff: float, W: int = 100000, chr_len=None,
                 ctx: Context | None = None) -> WindowResult:
    """xpehhWindow.cpp:126-232 over columns: a negative cutoff looks for the minimum and counts
    scores below it, otherwise the maximum / above (xpehhWindow.cpp:210-216)."""
    win = build_windows_extreme(pos, run_lengths(chr_ids), chr_len, W)
    mode = PGT_EXT_XP_MIN if cutoff < 0 else PGT_EXT_XP_MAX
    ctx, own = _own_ctx(ctx)
    try:
        return WindowResult(win, ctx.extreme_reduce(pos, score, mode, cutoff, win))
    finally:
        if own:
            ctx.close()
