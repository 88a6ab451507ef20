"""Counter-based synthetic genome (BASELINE.md / SURVEY.md §8d workload), addressable by GLOBAL site index.

Every value of every column is a pure function of (seed, column, global site index), so a rank of a
multi-GPU run can materialise exactly its own shard [site_lo, site_hi) of ONE genome, and the host
(numpy) and the device (torch integer ops) produce the same bits:

    h(col, i)  = splitmix64(seed * 0x9E3779B97F4A7C15 + col * 0xD1B54A32D192ED03 + i)
    u(col, i)  = (h >> 11) * 2^-53                                  in [0, 1)
    b[i]       = round(u(0,i) * 0.3e6) / 1e6                        b ~ U(0, 0.3), 6 decimals
    a[i]       = round(b[i] * (u(1,i) * 0.7 - 0.1) * 1e6) / 1e6     a = b * U(-0.1, 0.6), 6 decimals
    gap[i]     = 1 + (h(2,i) >> 33) % 59                            U{1..59}
    pos[i]     = sum of gap over the sites of i's chromosome up to and including i   (< 2^31)
    p1,p2      = round(u(3|4, i) * 1e6) / 1e6 ;  n1,n2 = (h(5|6,i) >> 33) % 21
    g1,g2      = genotype in {0,1,2,-1} with probabilities {.5,.3,.15,.05} from u(7|8, i)

Chromosomes are n_chr runs of (almost) equal length.  All floating-point steps are single IEEE
operations (multiply, round-half-even, divide), identical in numpy and torch on CPU or GPU — the
division is written with a TENSOR divisor on purpose: torch turns `x / 1e6` with a Python scalar into
`x * (1 / 1e6)` on the GPU, which is one ulp off the correctly rounded quotient for about a third of the
values (found by bench.py's ingest check: the text "0.022537" parses to the quotient, not to the product).

This is bench/test plumbing (torch as a device RNG), not part of the product: nothing under
popgenomicstools_amd/ imports it.
"""
from __future__ import annotations

import numpy as np

_GOLD = 0x9E3779B97F4A7C15
_COLMUL = 0xD1B54A32D192ED03
_M1 = 0xBF58476D1CE4E5B9
_M2 = 0x94D049BB133111EB
_MASK = (1 << 64) - 1

COL_B, COL_A, COL_GAP, COL_P1, COL_P2, COL_N1, COL_N2, COL_G1, COL_G2 = range(9)
COL_FREQ0 = 16  # allele-frequency columns of population k: COL_FREQ0 + k


def _s64(x: int) -> int:
    """uint64 constant as the int64 with the same bits (torch has no uint64 arithmetic)."""
    x &= _MASK
    return x - (1 << 64) if x >= (1 << 63) else x


def _div(x, d: float):
    """x / d as a true (correctly rounded) division on every device."""
    import torch
    return torch.div(x, torch.tensor(d, dtype=x.dtype, device=x.device))


class SynthGenome:
    def __init__(self, seed: int, n_sites: int, n_chr: int):
        self.seed, self.n, self.n_chr = int(seed), int(n_sites), int(n_chr)
        base = self.n // self.n_chr
        extra = self.n - base * self.n_chr
        self.run_len = np.array([base + (1 if c < extra else 0) for c in range(self.n_chr)], dtype=np.uint64)
        self.run_len = self.run_len[self.run_len > 0]
        self.chr_start = np.concatenate(([0], np.cumsum(self.run_len)[:-1])).astype(np.int64)
        self.chr_end = np.cumsum(self.run_len).astype(np.int64)

    def _key(self, col: int) -> int:
        return (self.seed * _GOLD + col * _COLMUL) & _MASK

    # ---------------------------------------------------------------- numpy (host) --------
    def _h_np(self, col: int, lo: int, hi: int) -> np.ndarray:
        with np.errstate(over="ignore"):
            z = np.arange(lo, hi, dtype=np.uint64) + np.uint64(self._key(col)) + np.uint64(_GOLD)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(_M1)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(_M2)
            return z ^ (z >> np.uint64(31))

    def _u_np(self, col, lo, hi):
        return (self._h_np(col, lo, hi) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)

    def _pos_np(self, lo, hi):
        pos = np.empty(hi - lo, dtype=np.uint32)
        for c in range(self.run_len.size):
            s, e = int(self.chr_start[c]), int(self.chr_end[c])
            if e <= lo or s >= hi:
                continue
            l, h = max(s, lo), min(e, hi)
            carry = 0
            if l > s:
                carry = int(((self._h_np(COL_GAP, s, l) >> np.uint64(33)) % np.uint64(59)).sum()) + (l - s)
            gaps = ((self._h_np(COL_GAP, l, h) >> np.uint64(33)) % np.uint64(59)).astype(np.int64) + 1
            pos[l - lo: h - lo] = (np.cumsum(gaps) + carry).astype(np.uint32)
        return pos

    def fst_columns_np(self, lo: int, hi: int):
        """-> (pos u32, a f64, b f64) of sites [lo, hi)."""
        b = np.round(self._u_np(COL_B, lo, hi) * 0.3e6) / 1e6
        a = np.round(b * (self._u_np(COL_A, lo, hi) * 0.7 - 0.1) * 1e6) / 1e6
        return self._pos_np(lo, hi), a, b

    def chr_ids_np(self, lo: int, hi: int):
        return (np.searchsorted(self.chr_end, np.arange(lo, hi, dtype=np.int64), side="right")).astype(np.uint32)

    # ---------------------------------------------------------------- torch (device) ------
    _CHUNK = 1 << 24

    def _h_t(self, col, lo, hi, dev):
        import torch
        z = torch.arange(lo, hi, dtype=torch.int64, device=dev) + _s64(self._key(col) + _GOLD)
        z = (z ^ ((z >> 30) & ((1 << 34) - 1))) * _s64(_M1)
        z = (z ^ ((z >> 27) & ((1 << 37) - 1))) * _s64(_M2)
        return z ^ ((z >> 31) & ((1 << 33) - 1))

    def _u_t(self, col, lo, hi, dev):
        import torch
        return ((self._h_t(col, lo, hi, dev) >> 11) & ((1 << 53) - 1)).to(torch.float64) * (1.0 / 9007199254740992.0)

    def _small_t(self, col, lo, hi, dev, mod):
        return ((self._h_t(col, lo, hi, dev) >> 33) & ((1 << 31) - 1)) % mod

    def _gap_sum_t(self, lo, hi, dev) -> int:
        tot = 0
        for c0 in range(lo, hi, self._CHUNK):
            c1 = min(hi, c0 + self._CHUNK)
            tot += int(self._small_t(COL_GAP, c0, c1, dev, 59).sum().item()) + (c1 - c0)
        return tot

    def pos_t(self, lo, hi, dev):
        import torch
        pos = torch.empty(hi - lo, dtype=torch.int32, device=dev)
        for c in range(self.run_len.size):
            s, e = int(self.chr_start[c]), int(self.chr_end[c])
            if e <= lo or s >= hi:
                continue
            l, h = max(s, lo), min(e, hi)
            carry = self._gap_sum_t(s, l, dev) if l > s else 0
            for c0 in range(l, h, self._CHUNK):
                c1 = min(h, c0 + self._CHUNK)
                cs = (self._small_t(COL_GAP, c0, c1, dev, 59) + 1).cumsum(0) + carry
                pos[c0 - lo: c1 - lo] = cs.to(torch.int32)
                carry = int(cs[-1].item())
        return pos

    def _fill(self, out, lo, hi, fn):
        for c0 in range(lo, hi, self._CHUNK):
            c1 = min(hi, c0 + self._CHUNK)
            out[c0 - lo: c1 - lo] = fn(c0, c1)
        return out

    def fst_columns_t(self, lo: int, hi: int, dev):
        """-> (pos int32 [u32 bits], a f64, b f64) torch tensors on `dev` for sites [lo, hi)."""
        import torch
        a = torch.empty(hi - lo, dtype=torch.float64, device=dev)
        b = torch.empty(hi - lo, dtype=torch.float64, device=dev)
        for c0 in range(lo, hi, self._CHUNK):
            c1 = min(hi, c0 + self._CHUNK)
            bb = _div(torch.round(self._u_t(COL_B, c0, c1, dev) * 0.3e6), 1e6)
            b[c0 - lo: c1 - lo] = bb
            a[c0 - lo: c1 - lo] = _div(torch.round(bb * (self._u_t(COL_A, c0, c1, dev) * 0.7 - 0.1) * 1e6), 1e6)
        return self.pos_t(lo, hi, dev), a, b

    def dxy_columns_t(self, lo, hi, dev):
        """-> (p1, p2 f64; n1, n2 int32) for sites [lo, hi) (minind = 5 in the BASELINE workload)."""
        import torch
        e = lambda dt: torch.empty(hi - lo, dtype=dt, device=dev)  # noqa: E731
        p1 = self._fill(e(torch.float64), lo, hi, lambda x, y: _div(torch.round(self._u_t(COL_P1, x, y, dev) * 1e6), 1e6))
        p2 = self._fill(e(torch.float64), lo, hi, lambda x, y: _div(torch.round(self._u_t(COL_P2, x, y, dev) * 1e6), 1e6))
        n1 = self._fill(e(torch.int32), lo, hi, lambda x, y: self._small_t(COL_N1, x, y, dev, 21).to(torch.int32))
        n2 = self._fill(e(torch.int32), lo, hi, lambda x, y: self._small_t(COL_N2, x, y, dev, 21).to(torch.int32))
        return p1, p2, n1, n2

    def genotype_t(self, which, lo, hi, dev):
        """int8 genotypes in {0,1,2,-1} w.p. {.5,.3,.15,.05}."""
        import torch

        def f(x, y):
            u = self._u_t(COL_G1 + which, x, y, dev)
            g = (u >= 0.5).to(torch.int8) + (u >= 0.8).to(torch.int8)
            return torch.where(u >= 0.95, torch.full_like(g, -1), g)
        return self._fill(torch.empty(hi - lo, dtype=torch.int8, device=dev), lo, hi, f)

    def freq_t(self, k, lo, hi, dev):
        """allele-frequency column of population k (U(0,1), 6 decimals)."""
        import torch
        return self._fill(torch.empty(hi - lo, dtype=torch.float64, device=dev), lo, hi,
                          lambda x, y: _div(torch.round(self._u_t(COL_FREQ0 + k, x, y, dev) * 1e6), 1e6))

    def pair_columns_t(self, pair, lo, hi, dev):
        """(a, b) component columns of population pair `pair` (config 5): the fst recipe on its own columns."""
        import torch
        a = torch.empty(hi - lo, dtype=torch.float64, device=dev)
        b = torch.empty(hi - lo, dtype=torch.float64, device=dev)
        ca, cb = 64 + 2 * pair, 65 + 2 * pair
        for c0 in range(lo, hi, self._CHUNK):
            c1 = min(hi, c0 + self._CHUNK)
            bb = _div(torch.round(self._u_t(cb, c0, c1, dev) * 0.3e6), 1e6)
            b[c0 - lo: c1 - lo] = bb
            a[c0 - lo: c1 - lo] = _div(torch.round(bb * (self._u_t(ca, c0, c1, dev) * 0.7 - 0.1) * 1e6), 1e6)
        return a, b
