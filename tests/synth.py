"""Seeded synthetic inputs shaped as BASELINE.md defines them (numpy, host side)."""
import numpy as np


def chromosomes(rng, n_sites, n_chr, equal=True):
    """-> (chr_ids u32[n], pos u32[n]); pos = running sum of U{1..59} gaps per chromosome."""
    if equal:
        base = n_sites // n_chr
        lens = np.full(n_chr, base, dtype=np.int64)
        lens[: n_sites - base * n_chr] += 1
    else:
        cuts = np.sort(rng.choice(np.arange(1, n_sites), size=n_chr - 1, replace=False)) if n_chr > 1 else np.array([], dtype=np.int64)
        lens = np.diff(np.concatenate(([0], cuts, [n_sites])))
    lens = lens[lens > 0]
    chr_ids = np.repeat(np.arange(lens.size, dtype=np.uint32), lens)
    gaps = rng.integers(1, 60, size=n_sites, dtype=np.int64)
    csum = np.cumsum(gaps)
    starts = np.concatenate(([0], np.cumsum(lens)[:-1]))
    offset = np.repeat(np.concatenate(([0], csum[np.cumsum(lens)[:-1] - 1])), lens)
    pos = (csum - offset).astype(np.uint32)
    return chr_ids, pos


def fst_columns(rng, n):
    b = np.round(rng.uniform(0.0, 0.3, n), 6)
    a = np.round(b * rng.uniform(-0.1, 0.6, n), 6)
    return a, b


def het_column(rng, n):
    return rng.choice(np.array([0, 1, 2, -1], dtype=np.int32), size=n, p=[0.5, 0.3, 0.15, 0.05])


def dxy_columns(rng, n):
    p1 = np.round(rng.uniform(0, 1, n), 6)
    p2 = np.round(rng.uniform(0, 1, n), 6)
    n1 = rng.integers(0, 21, n, dtype=np.int32)
    n2 = rng.integers(0, 21, n, dtype=np.int32)
    return p1, p2, n1, n2
