#!/usr/bin/env python3
"""Randomised GPU-vs-oracle fuzzer over every statistic (fst, het, dxy fixed-site and bp, extreme
scores, AF front end), random sizes 1..2.5e6 and window geometries (including windows longer than
level-3 tree nodes).  usage: python tests/gpu_fuzz.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))  # this script lives in tests/: it uses the oracle
import oracle_bind  # noqa: E402
import synth  # noqa: E402
import popgenomicstools_amd as pgt  # noqa: E402

REL, ABS = 1e-9, 1e-12


def close(x, y, scale=None):
    x, y = np.asarray(x, float), np.asarray(y, float)
    s = np.abs(y) if scale is None else np.abs(scale)
    return bool(np.all(np.abs(x - y) <= REL * s + ABS))


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    orc = oracle_bind.load()
    ctx = pgt.Context(0)
    t0 = time.time()
    counts = {}
    trial = 0
    last_note = t0
    while time.time() - t0 < budget:
        trial += 1
        if time.time() - last_note > 30:  # keep a long run visibly alive
            print(f"  ... {trial} trials, {time.time() - t0:.0f} s", flush=True)
            last_note = time.time()
        kind = rng.choice(["fst", "het", "dxy_fixed", "dxy_bp", "ext", "af"])
        n = int(rng.choice([rng.integers(1, 300), rng.integers(300, 20_000), rng.integers(20_000, 400_000),
                            rng.integers(400_000, 2_500_000)], p=[0.25, 0.3, 0.3, 0.15]))
        n_chr = int(rng.integers(1, min(n, 6) + 1))
        chr_ids, pos = synth.chromosomes(rng, n, n_chr, equal=bool(rng.random() < 0.3))
        W = int(rng.integers(1, max(2, min(n + 5, 3_000_000))))
        if rng.random() < 0.5:
            W = int(rng.integers(1, max(2, min(n, 5000))))
        S = int(rng.integers(1, W + 1))
        if rng.random() < 0.3:  # the sliding query's regime (step <= 32 sites)
            S = int(rng.integers(1, min(W, 32) + 1))
        if n * (W / S) > 4e8:  # keep the oracle's O(N*W/S) affordable
            S = max(S, W // 50 + 1)
        if kind in ("fst", "het", "dxy_fixed", "af") and rng.random() < 0.35:
            # the group query's regime: steps 1 .. 1024 with windows of at least two level-2 tiles (16384 sites for the f64
            # trees, 131072 for the genotype tree), window lengths on and beside the 128 / 8192-site grids, n sized so that
            # the oracle's O(N*W/S) stays affordable
            W = int(rng.choice([rng.integers(16_384, 60_000), rng.integers(131_072, 300_000), 16_384, 16_383, 24_576, 24_577, 50_000]))
            S = int(rng.choice([rng.integers(1, 65), rng.integers(65, 1025), 1, 64, 65, 128, 1024]))
            n = int(min(rng.integers(W // 2, 2_500_000), max(W // 2, 4e8 * S / W)))
            n_chr = int(rng.integers(1, 6))
            chr_ids, pos = synth.chromosomes(rng, n, n_chr, equal=bool(rng.random() < 0.3))
            counts["(group-regime geometries)"] = counts.get("(group-regime geometries)", 0) + 1
        try:
            if kind == "fst":
                a, b = synth.fst_columns(rng, n)
                ref = orc.fst_scan(chr_ids, pos, a, b, W, S)
                r = pgt.fst_window(chr_ids, pos, a, b, W, S, ctx=ctx).rows
                ok = (r.size == ref.size and all(np.array_equal(r[f], ref[f]) for f in ("start", "end", "mid", "n"))
                      and close(r["asum"], ref["num"], ref["den"]) and close(r["bsum"], ref["den"])
                      and close(r["fst"], ref["value"], np.maximum(np.abs(ref["value"]), 1e-3)))
            elif kind == "het":
                g = synth.het_column(rng, n)
                ref = orc.het_scan(chr_ids, pos, g, W, S)
                r = pgt.het_window(chr_ids, pos, g, W, S, ctx=ctx).rows
                ok = (r.size == ref.size and np.array_equal(r["nonmissing"], ref["n"]) and np.array_equal(r["h"], ref["value"])
                      and np.array_equal(r["start"], ref["start"]) and np.array_equal(r["mid"], ref["mid"]))
            elif kind in ("dxy_fixed", "dxy_bp"):
                p1, p2, n1, n2 = synth.dxy_columns(rng, n)
                fixed = int(kind == "dxy_fixed")
                runs = pgt.run_lengths(chr_ids)
                ends = np.cumsum(runs).astype(np.int64) - 1
                chr_len = (pos[ends].astype(np.int64) + rng.integers(0, 50, size=runs.size)).astype(np.uint32)
                if not fixed and int(chr_len.sum()) > 3_000_000:  # the oracle walks every bp slot
                    continue
                skip = int(rng.integers(0, 2))
                ref, rt = orc.dxy_scan(chr_ids, pos, p1, p2, n1, n2, W, S, 5, fixed, skip, chr_len)
                ref = ref[ref["printed"] == 1]
                res = pgt.dxy_window(chr_ids, pos, p1, p2, n1, n2, W, S, 5, fixed, chr_len, skip, ctx=ctx)
                r = res.rows
                ok = (r.size == ref.size and np.array_equal(r["start"], ref["start"]) and np.array_equal(r["end"], ref["end"])
                      and np.array_equal(r["neff"], ref["n"]) and np.array_equal(r["nskip"], ref["nskip"])
                      and close(r["sum"], ref["value"]) and int(res.total["neff"]) == int(rt["neff"]) and close([res.total["sum"]], [rt["sum"]]))
            elif kind == "ext":
                score = np.round(rng.normal(0, 1.5, n), 4)
                mode = int(rng.integers(0, 3))
                cutoff = float(rng.choice([2.0, 1.0, 0.0]))
                if mode == 2:  # the tool selects "minimum" by cutoff < 0 (xpehhWindow.cpp:210): keep it strictly negative
                    cutoff = -cutoff if cutoff > 0 else -0.5
                Wb = int(rng.choice([10, 100, 5000, 100_000, 5_000_000]))
                runs = pgt.run_lengths(chr_ids)
                ends = np.cumsum(runs).astype(np.int64) - 1
                chr_len = (pos[ends].astype(np.int64) + rng.integers(0, 2 * Wb, size=runs.size)).astype(np.uint32) if rng.random() < 0.6 else None
                if chr_len is not None and int(chr_len.astype(np.int64).sum()) // Wb > 2_000_000:
                    continue
                ref = orc.extreme_scan(chr_ids, pos, score, Wb, mode, cutoff, chr_len)
                r = (pgt.ihs_window(chr_ids, pos, score, Wb, cutoff, chr_len, ctx=ctx) if mode == 0
                     else pgt.xpehh_window(chr_ids, pos, score, cutoff, Wb, chr_len, ctx=ctx)).rows
                ok = r.size == ref.size and all(np.array_equal(r[f], ref[f]) for f in ("start", "end", "nsites", "nbig", "position", "value"))
                if not ok:
                    bad = [f for f in ("start", "end", "nsites", "nbig", "position", "value") if r.size != ref.size or not np.array_equal(r[f], ref[f])]
                    print(f"  ext detail: mode {mode} cutoff {cutoff} Wb {Wb} chr_len {'given' if chr_len is not None else 'none'} rows {r.size}/{ref.size} differing {bad}")
            else:
                import torch
                from popgenomicstools_amd._lib import FST_ROW_DTYPE
                from popgenomicstools_amd.window_scan import rows_from_device, windows_to_device
                npop = int(rng.integers(2, 9))
                base = rng.uniform(0.02, 0.98, n)
                freqs = [np.clip(np.round(base + rng.normal(0, 0.1, n), 6), 0, 1) for _ in range(npop)]
                ns = [float(x) for x in rng.integers(4, 50, npop)]
                win = pgt.build_windows_sites(pgt.run_lengths(chr_ids), W, S)
                dev = torch.device("cuda:0")
                out, _ = ctx.fst_af_reduce_dev(torch.from_numpy(pos.view(np.int32)).to(dev), [torch.from_numpy(f).to(dev) for f in freqs],
                                               ns, windows_to_device(win, dev))
                torch.cuda.synchronize()
                rows = rows_from_device(out, FST_ROW_DTYPE).reshape(npop * (npop - 1) // 2, win.size)
                i, j = sorted(rng.choice(npop, 2, replace=False))
                p = sum(npop - 1 - k for k in range(i)) + (j - i - 1)
                a, ab = orc.wcfst_columns(freqs[i], freqs[j], ns[i], ns[j])
                ref = orc.fst_scan(chr_ids, pos, a, ab, W, S)
                r = rows[p]
                ok = (r.size == ref.size and np.array_equal(r["n"], ref["n"]) and close(r["bsum"], ref["den"])
                      and close(r["asum"], ref["num"], ref["den"]) and bool(np.all(np.abs(r["fst"] - ref["value"]) <= 1e-9)))
        except Exception as e:  # noqa: BLE001
            print(f"EXCEPTION trial {trial} kind {kind} n {n} W {W} S {S}: {e!r}")
            raise
        if not ok:
            print(f"MISMATCH trial {trial} kind {kind} n {n} n_chr {n_chr} W {W} S {S}")
            sys.exit(1)
        counts[kind] = counts.get(kind, 0) + 1
    print(f"gpu_fuzz: {trial} trials in {time.time() - t0:.0f} s, all equal: {counts}")


if __name__ == "__main__":
    main()
