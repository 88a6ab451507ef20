"""ctypes binding of oracle/liboracle.so (the CPU restatement).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")

ROW = np.dtype([("label", "<u4"), ("start", "<u4"), ("end", "<u4"), ("mid", "<u4"), ("n", "<u4"),
                ("nskip", "<u4"), ("printed", "<u4"), ("pad_", "<u4"), ("lo", "<u8"), ("hi", "<u8"),
                ("value", "<f8"), ("num", "<f8"), ("den", "<f8")])
TOTAL = np.dtype([("sum", "<f8"), ("neff", "<u4"), ("nskip", "<u4")])
assert ROW.itemsize == 72 and TOTAL.itemsize == 16

_lib = None


def load():
    global _lib
    if _lib is None:
        src = [os.path.join(ORACLE_DIR, f) for f in ("window_oracle.c", "window_oracle.h")]
        if not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in src):
            subprocess.run(["make", "-C", ORACLE_DIR, LIB], check=True, capture_output=True)
        _lib = Oracle(C.CDLL(LIB))
    return _lib


def _u32(x):
    return np.ascontiguousarray(x, dtype=np.uint32)


class Oracle:
    def __init__(self, lib):
        self.lib = lib

    def _rows(self, call, guess):
        cap = max(16, int(guess))
        while True:
            out = np.zeros(cap, dtype=ROW)
            n_out = C.c_size_t(0)
            rc = call(out.ctypes.data, cap, C.byref(n_out))
            if rc == 2:
                cap = n_out.value
                continue
            if rc != 0:
                raise RuntimeError(f"oracle error {rc}")
            return out[: n_out.value].copy()

    def fst_scan(self, chr_ids, pos, a, b, W, S):
        c, p = _u32(chr_ids), _u32(pos)
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        f = self.lib.orc_fst_scan
        f.argtypes = [C.c_void_p] * 4 + [C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p]
        return self._rows(lambda o, cap, n: f(c.ctypes.data, p.ctypes.data, a.ctypes.data, b.ctypes.data, p.size, W, S, o, cap, n),
                          p.size // max(S, 1) + 64)

    def het_scan(self, chr_ids, pos, g, W, S):
        c, p = _u32(chr_ids), _u32(pos)
        g = np.ascontiguousarray(g, dtype=np.int32)
        f = self.lib.orc_het_scan
        f.argtypes = [C.c_void_p] * 3 + [C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p]
        return self._rows(lambda o, cap, n: f(c.ctypes.data, p.ctypes.data, g.ctypes.data, p.size, W, S, o, cap, n),
                          p.size // max(S, 1) + 64)

    def dxy_scan(self, chr_ids, pos, p1, p2, n1, n2, W, S, minind, fixedsite, skip_missing, run_chr_len=None):
        c, p = _u32(chr_ids), _u32(pos)
        p1 = np.ascontiguousarray(p1, dtype=np.float64)
        p2 = np.ascontiguousarray(p2, dtype=np.float64)
        n1 = np.ascontiguousarray(n1, dtype=np.int32)
        n2 = np.ascontiguousarray(n2, dtype=np.int32)
        rl = _u32(run_chr_len) if run_chr_len is not None else None
        tot = np.zeros(1, dtype=TOTAL)
        f = self.lib.orc_dxy_scan
        f.argtypes = [C.c_void_p] * 6 + [C.c_size_t, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                         C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        slots = int(rl.sum()) if rl is not None else 0
        rows = self._rows(
            lambda o, cap, n: f(c.ctypes.data, p.ctypes.data, p1.ctypes.data, p2.ctypes.data, n1.ctypes.data,
                                n2.ctypes.data, p.size, W, S, minind, fixedsite, skip_missing,
                                rl.ctypes.data if rl is not None else None, rl.size if rl is not None else 0,
                                o, cap, n, tot.ctypes.data),
            (p.size + slots) // max(S, 1) + 64)
        return rows, tot[0]

    def wcfst_columns(self, f1, f2, n1, n2):
        """betaAFOutlier.R:400-418 per site -> (a, a+b) columns (parity unpinned, see window_oracle.h)."""
        f1 = np.ascontiguousarray(f1, dtype=np.float64)
        f2 = np.ascontiguousarray(f2, dtype=np.float64)
        a, ab = np.empty_like(f1), np.empty_like(f1)
        f = self.lib.orc_wcfst_columns
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
        f.restype = None
        f(f1.ctypes.data, f2.ctypes.data, f1.size, float(n1), float(n2), a.ctypes.data, ab.ctypes.data)
        return a, ab

    # text front ends ----------------------------------------------------------------------
    def fst_text(self, path, W, S, out_path):
        f = self.lib.orc_fst_text_path
        f.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_char_p]
        return f(path.encode(), W, S, out_path.encode())

    def het_text(self, path, W, S, out_path):
        f = self.lib.orc_het_text_path
        f.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_char_p]
        return f(path.encode(), W, S, out_path.encode())

    def dxy_text(self, maf1, maf2, sizefile, W, S, minind, fixedsite, skip_missing, out_path, err_path):
        f = self.lib.orc_dxy_text_path
        f.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int,
                      C.c_char_p, C.c_char_p]
        return f(maf1.encode(), maf2.encode(), (sizefile or "").encode(), W, S, minind, fixedsite, skip_missing,
                 out_path.encode(), err_path.encode())

    def write_fst_text(self, path, chr_ids, pos, a, b):
        c, p = _u32(chr_ids), _u32(pos)
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        f = self.lib.orc_write_fst_text
        f.argtypes = [C.c_char_p] + [C.c_void_p] * 4 + [C.c_size_t]
        rc = f(path.encode(), c.ctypes.data, p.ctypes.data, a.ctypes.data, b.ctypes.data, p.size)
        if rc:
            raise RuntimeError(f"orc_write_fst_text: {rc}")

    def write_maf_text(self, path, chr_ids, pos, freq, nind):
        c, p = _u32(chr_ids), _u32(pos)
        fr = np.ascontiguousarray(freq, dtype=np.float64)
        ni = np.ascontiguousarray(nind, dtype=np.int32)
        f = self.lib.orc_write_maf_text
        f.argtypes = [C.c_char_p] + [C.c_void_p] * 4 + [C.c_size_t]
        rc = f(path.encode(), c.ctypes.data, p.ctypes.data, fr.ctypes.data, ni.ctypes.data, p.size)
        if rc:
            raise RuntimeError(f"orc_write_maf_text: {rc}")

    def write_het_text(self, path, chr_ids, pos, g):
        c, p = _u32(chr_ids), _u32(pos)
        g = np.ascontiguousarray(g, dtype=np.int32)
        f = self.lib.orc_write_het_text
        f.argtypes = [C.c_char_p] + [C.c_void_p] * 3 + [C.c_size_t]
        rc = f(path.encode(), c.ctypes.data, p.ctypes.data, g.ctypes.data, p.size)
        if rc:
            raise RuntimeError(f"orc_write_het_text: {rc}")


def ref_binary(tool):
    """Path of the compiled UNMODIFIED reference tool (oracle/_ref), or None if absent."""
    p = os.path.join(REF_DIR, tool)
    return p if os.path.exists(p) and os.access(p, os.X_OK) else None


EXT_ROW = np.dtype([("label", "<u4"), ("start", "<u4"), ("end", "<u4"), ("nsites", "<u4"), ("nbig", "<u4"),
                    ("position", "<u4"), ("lo", "<u8"), ("hi", "<u8"), ("value", "<f8")])
assert EXT_ROW.itemsize == 48


def _extreme_scan(self, chr_ids, pos, score, W, mode, cutoff, run_chr_len=None):
    c, p = _u32(chr_ids), _u32(pos)
    s = np.ascontiguousarray(score, dtype=np.float64)
    rl = _u32(run_chr_len) if run_chr_len is not None else None
    f = self.lib.orc_extreme_scan
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_int, C.c_double, C.c_void_p,
                  C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    cap = 64
    while True:
        out = np.zeros(cap, dtype=EXT_ROW)
        n_out = C.c_size_t(0)
        rc = f(c.ctypes.data, p.ctypes.data, s.ctypes.data, p.size, W, mode, float(cutoff),
               rl.ctypes.data if rl is not None else None, rl.size if rl is not None else 0, out.ctypes.data, cap,
               C.byref(n_out))
        if rc == 2:
            cap = n_out.value
            continue
        if rc != 0:
            raise RuntimeError(f"oracle error {rc}")
        return out[: n_out.value].copy()


def _ihs_text(self, path, W, cutoff, chrlen_path, out_path):
    f = self.lib.orc_ihs_text_path
    f.argtypes = [C.c_char_p, C.c_uint32, C.c_double, C.c_char_p, C.c_char_p]
    return f(path.encode(), W, float(cutoff), (chrlen_path or "").encode(), out_path.encode())


def _xpehh_text(self, path, cutoff, W, chrlen_path, out_path):
    f = self.lib.orc_xpehh_text_path
    f.argtypes = [C.c_char_p, C.c_double, C.c_uint32, C.c_char_p, C.c_char_p]
    return f(path.encode(), float(cutoff), W, (chrlen_path or "").encode(), out_path.encode())


Oracle.extreme_scan = _extreme_scan
Oracle.ihs_text = _ihs_text
Oracle.xpehh_text = _xpehh_text
