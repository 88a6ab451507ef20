#!/usr/bin/env python3
"""tests/golden/wcfst_exact.json: WCFst() and genomeFst() of betaAFOutlier.R:400-418,440-446 evaluated in EXACT rational
arithmetic (Python fractions) on a dozen hand-picked sites and three sample-size pairs.

A third, independent derivation beside the oracle's C restatement (oracle/window_oracle.c: orc_wcfst_site) and the device
kernel's algebraically rearranged form (csrc/pgt_af_kernels.hip): no R interpreter exists in this image and the reference
holds no vector for these functions, so the parity of the AF front end stays 'unpinned'; this fixture only removes the
possibility that the two floating-point implementations share a transcription error.  Frequencies are given as decimal
STRINGS (exact rationals); `a` and `a_plus_b` are stored as exact fractions ("num/den") and as the correctly rounded double.

    python tests/golden/make_wcfst_exact.py     (rewrites the fixture; needs nothing but the standard library)"""
import json
import os
from fractions import Fraction as F

SITES = [("0.5", "0.5"), ("0.25", "0.75"), ("0", "1"), ("1", "0"), ("0", "0"), ("1", "1"), ("0.1", "0.9"), ("0.05", "0.06"),
         ("0.333333", "0.666667"), ("0.999999", "0.000001"), ("0.125", "0.125"), ("0.7", "0.2")]
SIZES = [(10, 10), (10, 17), (3, 40)]


def reynolds_var(f1, f2, n1, n2):
    """betaAFOutlier.R:405-413, line by line, in rationals"""
    npool = n1 + n2                                        # :406
    fpool = F(n1, 1) / npool * f1 + F(n2, 1) / npool * f2  # :407
    alpha1 = 2 * f1 * (1 - f1)                             # :408
    alpha2 = 2 * f2 * (1 - f2)                             # :409
    b = (n1 * alpha1 + n2 * alpha2) / (npool - 1)          # :410
    a = (4 * n1 * (f1 - fpool) ** 2 + 4 * n2 * (f2 - fpool) ** 2 - b) / (F(4 * n1 * n2, 1) / npool)  # :411
    return a, b


def main():
    cases = []
    for n1, n2 in SIZES:
        rows, sa, sab = [], F(0), F(0)
        for s1, s2 in SITES:
            a, b = reynolds_var(F(s1), F(s2), n1, n2)
            ab = a + b                                      # :416  varcomp[,2] = varcomp[,2] + varcomp[,1]
            sa += a
            sab += ab
            rows.append({"f1": s1, "f2": s2, "a": f"{a.numerator}/{a.denominator}", "a_plus_b": f"{ab.numerator}/{ab.denominator}",
                         "a_f64": float(a), "a_plus_b_f64": float(ab)})
        fst = sa / sab                                      # :444-445 genomeFst = sum(a) / sum(a+b)
        cases.append({"n1": n1, "n2": n2, "sites": rows, "sum_a": f"{sa.numerator}/{sa.denominator}",
                      "sum_a_plus_b": f"{sab.numerator}/{sab.denominator}", "genome_fst": f"{fst.numerator}/{fst.denominator}",
                      "genome_fst_f64": float(fst)})
    doc = {"source": "exact rational evaluation of betaAFOutlier.R:400-418 (WCFst) and :440-446 (genomeFst) by "
                     "tests/golden/make_wcfst_exact.py; NOT an output of the reference (no R in this image)", "cases": cases}
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "wcfst_exact.json"), "w") as fh:
        json.dump(doc, fh, indent=1)
    print(len(cases), "sample-size pairs x", len(SITES), "sites")


if __name__ == "__main__":
    main()
