#!/usr/bin/env python3
"""Interleaved A/B of two BUILDS of libpgtwin.so in one process (markdown on stdout).

Box-to-box and process-to-process differences on this pool are +-3-5 % (placement), more than most changes to a
kernel are worth, so two builds are only comparable when their launches alternate in one process on one box.  Both
libraries are loaded side by side (ctypes handles are per path), each gets its own context on GPU 0, and every
configuration is measured A B B A ... on the same columns, tables and workspace.

  python tools/lib_ab.py tools/_ab/libpgtwin_before.so [sites=1e8] [rounds=10] [calls per measurement=1]

A = the library named on the command line (e.g. the previous commit's, built in a scratch worktree:
`git worktree add gpurun_out/wt HEAD~1`, build there, copy the .so to tools/_ab/ — *.so files are not committed but
travel with gpurun), B = the tree's own.  Reported: median build-phase and whole-step times of each and the median of
the paired differences.  With one call per measurement every launch starts on an idle GPU (a synchronise after each
call): the absolute rates then read 2-5 points below those of back-to-back launches (bench.py, measure_configs.py) and only
the difference between A and B is the result; with several calls per measurement a library's launches follow each other
as in bench.py, the libraries still alternate."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import popgenomicstools_amd as pgt  # noqa: E402
from popgenomicstools_amd import _lib  # noqa: E402
from popgenomicstools_amd._lib import PGT_STAT_FST  # noqa: E402
from popgenomicstools_amd.window_scan import windows_to_device  # noqa: E402
from synth_genome import SynthGenome  # noqa: E402


def one(ctx, fn):
    """-> (build phase ms by the library's events, whole step ms by events on the launch stream)"""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn(ctx)
    e1.record()
    torch.cuda.synchronize()
    return ctx.last_kernel_ms()[0], e0.elapsed_time(e1)


BURST = 1  # calls per measurement; > 1 (fourth argument): back-to-back launches of one library, as bench.py issues them


def burst(ctx, fn):
    """BURST calls without a synchronise between them -> (build phase ms of the last call, whole step ms averaged)"""
    if BURST == 1:
        return one(ctx, fn)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(ctx)  # the first call starts on an idle GPU: not timed
    e0.record()
    for _ in range(BURST):
        fn(ctx)
    e1.record()
    torch.cuda.synchronize()
    return ctx.last_kernel_ms()[0], e0.elapsed_time(e1) / BURST


class _Report:
    """AB_REPORT=<path>: the markdown also goes to that file — which must NOT exist yet (round 5 lost two A/B records to a
    later run writing over them; a record that a kernel comment cites must stay what it was)."""

    def __init__(self, path):
        if os.path.exists(path):
            sys.exit(f"lib_ab.py: {path} exists already — reports are never overwritten; choose another name")
        self.fh, self.out = open(path, "x"), sys.stdout

    def write(self, text):
        self.out.write(text)
        self.fh.write(text)

    def flush(self):
        self.out.flush()
        self.fh.flush()


def main():
    global BURST
    if os.environ.get("AB_REPORT"):
        sys.stdout = _Report(os.environ["AB_REPORT"])
    old = os.path.abspath(sys.argv[1])
    n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000_000
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    BURST = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    dev = torch.device("cuda", 0)
    b_name = "the tree's libpgtwin.so"
    if os.environ.get("AB_B_LIB"):  # B = another prebuilt library (a variant cross-compiled into tools/_ab/) instead of the tree's
        _lib.load()  # the tree's own is built / checked once, then set aside
        _lib._lib, _lib.LIB_PATH = None, os.path.abspath(os.environ["AB_B_LIB"])
        b_name = os.path.relpath(_lib.LIB_PATH, ROOT)
    ctx_b = pgt.Context(0)
    _lib._lib, _lib.LIB_PATH = None, old
    _lib.SYMBOLS = [x for x in _lib.SYMBOLS if x not in ("pgt_extreme_reduce_cols", "pgt_prepare_host_io")]  # added in rounds 4 / 6: an older build lacks them, no config here calls them
    import ctypes
    a_abi = ctypes.CDLL(old).pgt_abi_version()
    assert a_abi in (4, 5, 6), a_abi  # 4 -> 5 -> 6 changed no argument list (5: alignment of the i32 columns, the dxy workspace size; 6: pgt_prepare_host_io added): safe to call for this tool
    _lib.PGT_ABI_VERSION = a_abi
    ctx_a = pgt.Context(0)
    assert ctx_a._lib is not ctx_b._lib
    for c in (ctx_a, ctx_b):
        c.set_profiling(True)
        c.set_max_window(50_000)
    chroms, W, S = (40 if n > 200_000_000 else 20), 50_000, 10_000
    genome = SynthGenome(12345, n, chroms)
    pos, a, b = genome.fst_columns_t(0, n, dev)
    win = windows_to_device(pgt.build_windows_sites(genome.run_len, W, S), dev)
    tree = torch.empty(2 * ctx_b.tree_bytes(PGT_STAT_FST, n), dtype=torch.uint8, device=dev)
    out = torch.empty(4 * 40 * (win.numel() // 32), dtype=torch.uint8, device=dev)
    p1, p2, n1, n2 = genome.dxy_columns_t(0, n, dev)
    g1, g2 = genome.genotype_t(0, 0, n, dev), genome.genotype_t(1, 0, n, dev)
    from popgenomicstools_amd._lib import PGT_EXT_IHS
    score = a * 40.0 - 2.0
    ewin_h = pgt.build_windows_extreme(pos.cpu().numpy().view(np.uint32), genome.run_len, None, 100_000)
    ewin = windows_to_device(ewin_h, dev)
    fr = [genome.freq_t(k, 0, n, dev) for k in range(8)] if n <= 300_000_000 else None
    nsamp = [10.0 + k for k in range(8)]
    af_tree = torch.empty(int(_lib.load().pgt_af_tree_bytes(8, n)), dtype=torch.uint8, device=dev) if fr else None
    af_out = torch.empty(28 * 40 * (win.numel() // 32), dtype=torch.uint8, device=dev) if fr else None
    configs = [
        ("fstWindow", 16, lambda c: c.fst_reduce_dev(pos, a, b, win, out=out, tree=tree)),
        ("dxyWindow (with the genome-wide line)", 24, lambda c: c.dxy_reduce_dev(pos, p1, p2, n1, n2, 5, win, out=out, tree=tree)),
        ("hetWindow", 1, lambda c: c.het_reduce_dev(pos, g1, win, out=out, tree=tree)),
        ("dxy + het x2 fused", 26, lambda c: c.dxy_het_reduce_dev(pos, p1, p2, n1, n2, g1, g2, 5, win, tree=tree)),
        ("extreme score (100 kb windows)", 8, lambda c: c.extreme_reduce_dev(pos, score, PGT_EXT_IHS, 2.0, ewin, out=out, tree=tree)),
    ]
    if fr:
        configs.append(("AF front end, 8 populations", 64, lambda c: c.fst_af_reduce_dev(pos, fr, nsamp, win, out=af_out, tree=af_tree)))
        configs.append(("AF front end, 2 populations", 16, lambda c: c.fst_af_reduce_dev(pos, fr[:2], nsamp[:2], win, out=af_out, tree=af_tree)))
    print(f"A = {os.path.relpath(old, ROOT)}, B = {b_name}; {n:.0e} sites, W = {W}, S = {S}, {rounds} rounds of A B B A, {BURST} call(s) per measurement\n")
    print("| config | A build ms | B build ms | B - A paired (median) | A % of 8 TB/s | B % | A step ms | B step ms | step B - A paired | rows A = B |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    if os.environ.get("AB_ONLY"):  # e.g. AB_ONLY=AF: only the configurations whose name contains the string
        configs = [c for c in configs if os.environ["AB_ONLY"] in c[0]]
    emax = int((ewin_h["hi"] - ewin_h["lo"]).max())
    for name, bps, fn in configs:
        for c in (ctx_a, ctx_b):
            c.set_max_window(emax if name.startswith("extreme") else 50_000)
        for c in (ctx_a, ctx_b, ctx_a, ctx_b):
            one(c, fn)
        # rows of the two builds, bit for bit (every tensor a call returns except the tree workspace, which is its last)
        def rows(c):
            r = fn(c)
            torch.cuda.synchronize()
            return [t.clone() for t in r[:-1] if t is not None]
        ra_, rb_ = rows(ctx_a), rows(ctx_b)
        same = all(torch.equal(x, y) for x, y in zip(ra_, rb_))
        if not same:  # say WHICH tensor differs and by how much (a 24-byte tensor is the genome-wide dxy line: f64 sum, two u64 counts)
            for k, (x, y) in enumerate(zip(ra_, rb_)):
                if not torch.equal(x, y):
                    if x.numel() == 24:
                        fa, fb = x.cpu().numpy().view(np.float64)[0], y.cpu().numpy().view(np.float64)[0]
                        ia, ib = x.cpu().numpy().view(np.uint64)[1:], y.cpu().numpy().view(np.uint64)[1:]
                        print(f"  {name}: returned tensor {k} (the genome-wide line) differs: sum {fa!r} vs {fb!r} (relative {abs(fa - fb) / abs(fa):.2e}), counts {ia.tolist()} vs {ib.tolist()}")
                    else:
                        print(f"  {name}: returned tensor {k} ({x.numel()} bytes) differs in {int((x != y).sum())} bytes")
        ra, rb, diff, sdiff = [], [], [], []
        for _ in range(rounds):
            a1, b1, b2, a2 = burst(ctx_a, fn), burst(ctx_b, fn), burst(ctx_b, fn), burst(ctx_a, fn)
            ra += [a1, a2]
            rb += [b1, b2]
            diff += [b1[0] - a1[0], b2[0] - a2[0]]
            sdiff += [b1[1] - a1[1], b2[1] - a2[1]]
        ma, mb = np.median([x[0] for x in ra]), np.median([x[0] for x in rb])
        print(f"| {name} | {ma:.4f} | {mb:.4f} | {np.median(diff) * 1e3:+.1f} us | {bps * n / ma / 8e7:.1f} | {bps * n / mb / 8e7:.1f} | "
              f"{np.median([x[1] for x in ra]):.4f} | {np.median([x[1] for x in rb]):.4f} | {np.median(sdiff) * 1e3:+.1f} us | {'yes' if same else 'NO'} |")


if __name__ == "__main__":
    main()
