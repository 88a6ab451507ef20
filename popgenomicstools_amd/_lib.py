"""ctypes binding of include/pgtwin.h (the C-ABI of libpgtwin.so).

This module never computes anything itself: if the shared library is missing it raises, it does
not fall back to numpy, torch or the oracle.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libpgtwin.so")

PGT_ABI_VERSION = 6
PGT_OK, PGT_EARG, PGT_ECAP, PGT_EDEVICE, PGT_EDOMAIN, PGT_ENOMEM = range(6)
PGT_WIN_COORDS = 1
PGT_STAT_FST, PGT_STAT_HET, PGT_STAT_DXY, PGT_STAT_EXT = 0, 1, 2, 3
PGT_EXT_IHS, PGT_EXT_XP_MAX, PGT_EXT_XP_MIN = 0, 1, 2

# numpy views of the C structs (layout asserted against ctypes below)
WIN_DTYPE = np.dtype([("lo", "<u8"), ("hi", "<u8"), ("label_run", "<u4"), ("flags", "<u4"),
                      ("start", "<u4"), ("end", "<u4")])
FST_ROW_DTYPE = np.dtype([("start", "<u4"), ("end", "<u4"), ("mid", "<u4"), ("n", "<u4"),
                          ("fst", "<f8"), ("asum", "<f8"), ("bsum", "<f8")])
HET_ROW_DTYPE = np.dtype([("start", "<u4"), ("end", "<u4"), ("mid", "<u4"), ("nonmissing", "<u4"),
                          ("nhet", "<u4"), ("pad_", "<u4"), ("h", "<f8")])
DXY_ROW_DTYPE = np.dtype([("start", "<u4"), ("end", "<u4"), ("neff", "<u4"), ("nskip", "<u4"),
                          ("sum", "<f8")])
DXY_TOTAL_DTYPE = np.dtype([("sum", "<f8"), ("neff", "<u8"), ("nskip", "<u8")])
EXT_ROW_DTYPE = np.dtype([("start", "<u4"), ("end", "<u4"), ("nsites", "<u4"), ("nbig", "<u4"),
                          ("position", "<u4"), ("pad_", "<u4"), ("value", "<f8")])
SHARD_DTYPE = np.dtype([("win_begin", "<u8"), ("win_end", "<u8"), ("site_lo", "<u8"), ("site_hi", "<u8")])

assert WIN_DTYPE.itemsize == 32 and FST_ROW_DTYPE.itemsize == 40 and HET_ROW_DTYPE.itemsize == 32
assert EXT_ROW_DTYPE.itemsize == 32
assert DXY_ROW_DTYPE.itemsize == 24 and DXY_TOTAL_DTYPE.itemsize == 24 and SHARD_DTYPE.itemsize == 32

# every symbol include/pgtwin.h declares (tests/test_abi.py checks the list against the header)
SYMBOLS = [
    "pgt_open", "pgt_close", "pgt_last_error", "pgt_abi_version", "pgt_prepare_host_io",
    "pgt_build_windows_sites", "pgt_build_windows_bp", "pgt_build_windows_extreme",
    "pgt_extreme_reduce", "pgt_extreme_reduce_dev",
    "pgt_fst_reduce", "pgt_het_reduce", "pgt_dxy_reduce",
    "pgt_tree_bytes", "pgt_fst_reduce_dev", "pgt_het_reduce_dev", "pgt_dxy_reduce_dev",
    "pgt_fst_reduce_pairs_dev", "pgt_dxy_het_reduce_dev", "pgt_af_tree_bytes", "pgt_fst_af_reduce_dev", "pgt_set_max_window", "pgt_set_window_step", "pgt_set_profiling", "pgt_last_kernel_ms", "pgt_plan_shards",
    "pgt_dev_alloc", "pgt_dev_free", "pgt_dev_copy", "pgt_dev_upload", "pgt_dev_memory", "pgt_ingest_text_behind", "pgt_ingest_column_base", "pgt_set_typical_window", "pgt_table_hints", "pgt_ingest_blank_before_end",
    "pgt_peer_access", "pgt_rowbuf_create", "pgt_rowbuf_open", "pgt_rowbuf_close", "pgt_rowbuf_read", "pgt_rowbuf_fill",
    "pgt_fst_reduce_cols", "pgt_het_reduce_cols", "pgt_dxy_reduce_cols", "pgt_extreme_reduce_cols", "pgt_ingest_download", "pgt_ingest_text", "pgt_ingest_rows", "pgt_ingest_bad_line", "pgt_ingest_column", "pgt_ingest_runs", "pgt_ingest_free",
    "pgt_wintab_sites", "pgt_wintab_size", "pgt_wintab_first", "pgt_wintab_device", "pgt_wintab_free",
    "pgt_fst_reduce_tab", "pgt_het_reduce_tab", "pgt_dxy_reduce_tab",
]
PGT_TOK_CHR, PGT_TOK_SKIP, PGT_TOK_U32, PGT_TOK_F64, PGT_TOK_I8, PGT_TOK_I32, PGT_TOK_FREQ, PGT_TOK_CHR_PREFIX = range(8)


class PgtError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libpgtwin error {code}: {msg}")
        self.code = code


_lib = None


def load() -> C.CDLL:
    """Load libpgtwin.so; build.build_lib() rebuilds it first when a source, a header or the compile
    flags changed since it was built (mtime / flag-stamp check, cheap).  Raises if impossible."""
    global _lib
    if _lib is not None:
        return _lib
    from . import build
    build.build_lib()
    # One HIP runtime per process: torch ships its own libamdhip64.so.7 and /opt/rocm has another
    # with the same SONAME; whichever loads first serves both.  If torch loads second it finds no
    # device through the other copy, so let torch's copy load first whenever torch is installed
    # (device tensors, streams and torch.distributed come from it anyway).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    vp, u64, u32, sz, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_size_t, C.c_int
    # fail at load time, not at first use, and say what is wrong — BEFORE any symbol is touched (a library too old to export
    # pgt_abi_version, or some other .so under that name, must not surface as a bare AttributeError): the version first
    # (argument lists changed between ABI versions: never call across them), then any symbol the binding declares and the
    # library lacks
    rebuild = " (rebuild: python -m popgenomicstools_amd.build --force)"
    if not hasattr(lib, "pgt_abi_version"):
        raise RuntimeError(f"{LIB_PATH} does not export pgt_abi_version: not a libpgtwin this binding (ABI {PGT_ABI_VERSION}) can use" + rebuild)
    lib.pgt_abi_version.restype = i32
    if lib.pgt_abi_version() != PGT_ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} has ABI version {lib.pgt_abi_version()}, this binding is written for {PGT_ABI_VERSION}" + rebuild)
    missing = [name for name in SYMBOLS if not hasattr(lib, name)]
    if missing:
        raise RuntimeError(f"{LIB_PATH} is older than this binding although it reports ABI version {PGT_ABI_VERSION}: it lacks "
                           + ", ".join(missing) + rebuild)
    lib.pgt_open.restype = vp
    lib.pgt_open.argtypes = [i32]
    lib.pgt_close.restype = None
    lib.pgt_close.argtypes = [vp]
    lib.pgt_last_error.restype = C.c_char_p
    lib.pgt_last_error.argtypes = [vp]
    lib.pgt_abi_version.restype = i32
    lib.pgt_build_windows_sites.argtypes = [vp, sz, u32, u32, vp, sz, C.POINTER(sz)]
    lib.pgt_build_windows_bp.argtypes = [vp, vp, vp, sz, u32, u32, vp, sz, C.POINTER(sz)]
    lib.pgt_build_windows_extreme.argtypes = [vp, vp, vp, sz, u32, vp, sz, C.POINTER(sz)]
    lib.pgt_extreme_reduce.argtypes = [vp, vp, vp, u64, i32, C.c_double, vp, u64, vp]
    lib.pgt_extreme_reduce_dev.argtypes = [vp, vp, vp, u64, i32, C.c_double, vp, u64, vp, sz, vp, sz, vp]
    lib.pgt_fst_reduce.argtypes = [vp, vp, vp, vp, u64, vp, u64, vp]
    lib.pgt_het_reduce.argtypes = [vp, vp, vp, u64, vp, u64, vp]
    lib.pgt_dxy_reduce.argtypes = [vp, vp, vp, vp, vp, vp, u64, i32, vp, u64, vp, vp]
    lib.pgt_tree_bytes.restype = sz
    lib.pgt_tree_bytes.argtypes = [i32, u64]
    lib.pgt_fst_reduce_dev.argtypes = [vp, vp, vp, vp, u64, vp, u64, vp, sz, vp, sz, vp]
    lib.pgt_het_reduce_dev.argtypes = [vp, vp, vp, u64, vp, u64, vp, sz, vp, sz, vp]
    lib.pgt_dxy_reduce_dev.argtypes = [vp, vp, vp, vp, vp, vp, u64, i32, vp, u64, vp, sz, vp, vp, sz, vp]
    lib.pgt_fst_reduce_pairs_dev.argtypes = [vp, vp, vp, vp, u32, u64, vp, u64, vp, sz, vp, sz, vp]
    lib.pgt_dxy_het_reduce_dev.argtypes = [vp] * 8 + [u64, i32, vp, u64, vp, sz, vp, vp, vp, sz, vp, sz, vp]
    lib.pgt_af_tree_bytes.restype = sz
    lib.pgt_af_tree_bytes.argtypes = [u32, u64]
    lib.pgt_fst_af_reduce_dev.argtypes = [vp, vp, vp, vp, u32, u64, vp, u64, vp, sz, vp, sz, vp]
    lib.pgt_set_max_window.argtypes = [vp, u64]
    lib.pgt_set_window_step.argtypes = [vp, u64]
    lib.pgt_set_profiling.argtypes = [vp, i32]
    lib.pgt_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.pgt_plan_shards.argtypes = [vp, u64, u32, vp]
    lib.pgt_peer_access.argtypes = [vp, i32]
    lib.pgt_rowbuf_create.argtypes = [vp, sz, C.POINTER(vp), vp]
    lib.pgt_rowbuf_open.argtypes = [vp, vp, C.POINTER(vp)]
    lib.pgt_rowbuf_close.argtypes = [vp, vp, i32]
    lib.pgt_rowbuf_read.argtypes = [vp, vp, vp, sz, vp]
    lib.pgt_rowbuf_fill.argtypes = [vp, vp, sz, u64, vp]
    lib.pgt_dev_alloc.argtypes = [vp, sz, C.POINTER(vp)]
    lib.pgt_dev_free.argtypes = [vp, vp]
    lib.pgt_dev_memory.argtypes = [vp, C.POINTER(sz), C.POINTER(sz)]
    lib.pgt_dev_upload.argtypes = [vp, vp, vp, sz]
    lib.pgt_ingest_text_behind.argtypes = [vp, vp, sz, vp, i32, u64, C.POINTER(vp)]
    lib.pgt_ingest_column_base.restype = vp
    lib.pgt_ingest_column_base.argtypes = [vp, i32]
    lib.pgt_set_typical_window.argtypes = [vp, u64]
    lib.pgt_table_hints.argtypes = [vp, u64, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    lib.pgt_dev_copy.argtypes = [vp, vp, vp, vp, sz]
    lib.pgt_ingest_blank_before_end.argtypes = [vp]
    lib.pgt_fst_reduce_cols.argtypes = [vp, vp, vp, vp, u64, vp, u64, vp, sz]
    lib.pgt_het_reduce_cols.argtypes = [vp, vp, vp, u64, vp, u64, vp, sz]
    if hasattr(lib, "pgt_extreme_reduce_cols"):  # absent from the older builds tools/lib_ab.py loads side by side
        lib.pgt_extreme_reduce_cols.argtypes = [vp, vp, vp, u64, i32, C.c_double, vp, u64, vp, sz]
    lib.pgt_dxy_reduce_cols.argtypes = [vp, vp, vp, vp, vp, vp, u64, i32, vp, u64, vp, sz, vp]
    lib.pgt_ingest_download.argtypes = [vp, vp, i32, vp, sz]
    lib.pgt_ingest_text.argtypes = [vp, vp, sz, vp, i32, C.POINTER(vp)]
    lib.pgt_ingest_rows.restype = u64
    lib.pgt_ingest_rows.argtypes = [vp]
    lib.pgt_ingest_bad_line.restype = C.c_int64
    lib.pgt_ingest_bad_line.argtypes = [vp]
    lib.pgt_ingest_column.restype = vp
    lib.pgt_ingest_column.argtypes = [vp, i32]
    lib.pgt_ingest_runs.restype = sz
    lib.pgt_ingest_runs.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    lib.pgt_ingest_free.restype = None
    lib.pgt_ingest_free.argtypes = [vp]
    lib.pgt_wintab_sites.argtypes = [vp, vp, sz, u32, u32, C.POINTER(vp)]
    lib.pgt_wintab_size.restype = u64
    lib.pgt_wintab_size.argtypes = [vp]
    lib.pgt_wintab_first.restype = vp
    lib.pgt_wintab_first.argtypes = [vp]
    lib.pgt_wintab_device.restype = vp
    lib.pgt_wintab_device.argtypes = [vp]
    lib.pgt_wintab_free.restype = None
    lib.pgt_wintab_free.argtypes = [vp]
    lib.pgt_fst_reduce_tab.argtypes = [vp, vp, vp, vp, u64, i32, vp, vp, sz]
    lib.pgt_het_reduce_tab.argtypes = [vp, vp, vp, u64, i32, vp, vp, sz]
    lib.pgt_dxy_reduce_tab.argtypes = [vp, vp, vp, vp, vp, vp, u64, i32, i32, vp, vp, sz, vp]
    if "pgt_prepare_host_io" in SYMBOLS:  # (tools/lib_ab.py drops it from the list to load a round-5 library beside the tree's)
        lib.pgt_prepare_host_io.argtypes = [vp, u64]
    _lib = lib
    return lib


def last_error(ctx=None) -> str:
    msg = load().pgt_last_error(ctx)
    return msg.decode() if msg else ""


def check(rc: int, ctx=None) -> None:
    if rc != PGT_OK:
        raise PgtError(rc, last_error(ctx))


def np_ptr(arr: np.ndarray) -> int:
    return arr.ctypes.data
