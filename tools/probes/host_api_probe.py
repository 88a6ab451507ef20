#!/usr/bin/env python3
"""First and second call of a host-buffer entry point in a fresh process, as a command-line host makes it (round 6, VERDICT
item 4): freshly written pageable numpy columns of the CLI's 10^8-site case -> pgt_{fst,het,dxy}_reduce.  PGT_TRACE_API=1
prints the library's own phases; PGT_UPLOAD=plain takes the old path (hipMemcpy from the caller's pages).

    PGT_TRACE_API=1 python tools/probes/host_api_probe.py het 1e8 [prepare]
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import popgenomicstools_amd as pgt  # noqa: E402
from popgenomicstools_amd import _lib  # noqa: E402


def main():
    stat, n = sys.argv[1], int(float(sys.argv[2]))
    prepare = len(sys.argv) > 3 and sys.argv[3] == "prepare"
    lib = _lib.load()
    t0 = time.perf_counter()
    ctx = lib.pgt_open(0)
    assert ctx, _lib.last_error()
    if prepare:
        _lib.check(lib.pgt_prepare_host_io(ctx, 0), ctx)
    t_open = time.perf_counter() - t0
    rng = np.random.default_rng(3)
    pos = np.arange(1, n + 1, dtype=np.uint32)
    win = pgt.build_windows_sites(np.array([n], dtype=np.uint64), 50_000, 10_000)
    nw = win.size
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    if stat == "fst":
        a, b = rng.random(n), rng.random(n) + 0.5
        out = np.zeros(nw, dtype=_lib.FST_ROW_DTYPE)
        call = lambda: lib.pgt_fst_reduce(ctx, p(pos), p(a), p(b), n, p(win), nw, p(out))  # noqa: E731
        nbytes = n * 20
    elif stat == "het":
        g = rng.integers(-1, 3, size=n, dtype=np.int8)
        out = np.zeros(nw, dtype=_lib.HET_ROW_DTYPE)
        call = lambda: lib.pgt_het_reduce(ctx, p(pos), p(g), n, p(win), nw, p(out))  # noqa: E731
        nbytes = n * 5
    else:
        p1, p2 = rng.random(n), rng.random(n)
        n1 = rng.integers(0, 12, size=n, dtype=np.int32)
        n2 = rng.integers(0, 12, size=n, dtype=np.int32)
        out = np.zeros(nw, dtype=_lib.DXY_ROW_DTYPE)
        tot = np.zeros(1, dtype=_lib.DXY_TOTAL_DTYPE)
        call = lambda: lib.pgt_dxy_reduce(ctx, p(pos), p(p1), p(p2), p(n1), p(n2), n, 5, p(win), nw, p(out), p(tot))  # noqa: E731
        nbytes = n * 28
    res = []
    for k in range(3):
        t = time.perf_counter()
        _lib.check(call(), ctx)
        res.append(time.perf_counter() - t)
        print(f"[probe] {stat} n={n:.0e} call {k}: {res[-1] * 1e3:8.2f} ms  ({nbytes / res[-1] / 1e9:5.1f} GB/s over the link)", file=sys.stderr, flush=True)
    chk = out.view(np.uint8).sum()
    print(f"{stat} n={n} upload={os.environ.get('PGT_UPLOAD', 'ring')} prepare={prepare} open_ms={t_open * 1e3:.1f} "
          f"calls_ms={[round(x * 1e3, 2) for x in res]} rows_checksum={int(chk)}")
    lib.pgt_close(ctx)


if __name__ == "__main__":
    main()
