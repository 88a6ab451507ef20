"""CPU: the counter-based synthetic genome of bench.py — numpy and torch produce the same bits, and any
site range equals the corresponding slice of the whole genome (what lets a rank build only its shard)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synth_genome import SynthGenome  # noqa: E402


def test_numpy_and_torch_agree_and_ranges_are_slices():
    g = SynthGenome(12345, 300_007, 7)
    g._CHUNK = 1 << 14  # several chunks per chromosome
    cpu = torch.device("cpu")
    p, a, b = g.fst_columns_np(0, g.n)
    tp, ta, tb = g.fst_columns_t(0, g.n, cpu)
    assert np.array_equal(p, tp.numpy().view(np.uint32)) and np.array_equal(a, ta.numpy()) and np.array_equal(b, tb.numpy())
    for lo, hi in ((0, 1), (1, 2), (42_858, 42_859), (100_000, 250_001), (299_000, g.n)):
        sp, sa, sb = g.fst_columns_t(lo, hi, cpu)
        assert np.array_equal(p[lo:hi], sp.numpy().view(np.uint32)) and np.array_equal(a[lo:hi], sa.numpy())
        assert np.array_equal(b[lo:hi], sb.numpy())
        np_p, np_a, _ = g.fst_columns_np(lo, hi)
        assert np.array_equal(p[lo:hi], np_p) and np.array_equal(a[lo:hi], np_a)
    # the BASELINE.md recipe: b in [0, 0.3], a = b * U(-0.1, 0.6), 6 decimals, gaps in 1..59, positions restart per chromosome
    assert 0.0 <= b.min() and b.max() <= 0.3 and np.all(np.abs(a) <= 0.6 * b + 1e-6)
    assert np.array_equal(np.round(a * 1e6), a * 1e6) or np.allclose(np.round(a * 1e6), a * 1e6, atol=1e-6)
    ends = np.cumsum(g.run_len).astype(np.int64)
    starts = np.concatenate(([0], ends[:-1]))
    for s, e in zip(starts, ends):
        d = np.diff(p[s:e].astype(np.int64))
        assert 1 <= p[s] <= 59 and d.min() >= 1 and d.max() <= 59
    assert np.array_equal(g.chr_ids_np(0, g.n), np.repeat(np.arange(g.run_len.size), g.run_len.astype(np.int64)))


def test_other_columns_are_range_addressable():
    g = SynthGenome(7, 50_000, 3)
    g._CHUNK = 1 << 12
    cpu = torch.device("cpu")
    full = g.dxy_columns_t(0, g.n, cpu) + (g.genotype_t(0, 0, g.n, cpu), g.genotype_t(1, 0, g.n, cpu), g.freq_t(3, 0, g.n, cpu)) \
        + g.pair_columns_t(5, 0, g.n, cpu)
    lo, hi = 12_345, 43_210
    part = g.dxy_columns_t(lo, hi, cpu) + (g.genotype_t(0, lo, hi, cpu), g.genotype_t(1, lo, hi, cpu), g.freq_t(3, lo, hi, cpu)) \
        + g.pair_columns_t(5, lo, hi, cpu)
    for f, q in zip(full, part):
        assert torch.equal(f[lo:hi], q)
    n1 = full[2].numpy()
    assert n1.min() >= 0 and n1.max() <= 20
    geno = full[4].numpy()
    assert set(np.unique(geno)) <= {-1, 0, 1, 2} and abs((geno == 0).mean() - 0.5) < 0.02 and abs((geno == -1).mean() - 0.05) < 0.01
    assert not torch.equal(full[4], full[5])  # the two genotype columns are different streams


import pytest  # noqa: E402


@pytest.mark.gpu
def test_gpu_torch_equals_numpy():
    """The same bits on the GPU as on the host — including the division by 1e6, which torch would turn into a
    multiplication by the reciprocal for a Python-scalar divisor (one ulp off for a third of the values)."""
    g = SynthGenome(12345, 2_000_003, 5)
    dev = torch.device("cuda", 0)
    p, a, b = g.fst_columns_np(0, g.n)
    tp, ta, tb = g.fst_columns_t(0, g.n, dev)
    assert np.array_equal(p, tp.cpu().numpy().view(np.uint32))
    assert a.tobytes() == ta.cpu().numpy().tobytes() and b.tobytes() == tb.cpu().numpy().tobytes()
    lo, hi = 777_777, 1_500_001
    sp, sa, sb = g.fst_columns_t(lo, hi, dev)
    assert np.array_equal(p[lo:hi], sp.cpu().numpy().view(np.uint32)) and a[lo:hi].tobytes() == sa.cpu().numpy().tobytes()
