"""A second, independent restatement of dxyWindow's streaming loop — pure Python, test infrastructure only.

The oracle (oracle/window_oracle.c) restates dxyWindow.cpp in C over arrays and rings; this model restates the same lines
as the reference runs them: two line readers that can run dry, the catch-up loops of :315-331, the base-pair slot padding
of :334-378 and :407-421, calcWindow with its function-static `nnew` (:172-209), the genome-wide line (:429-433).  Same
author, same source text, different implementation: a transcription slip in either shows as a difference.  It pins nothing
(the reference itself cannot be built here); tests/test_oracle_golden.py fuzzes the oracle against it and
checks it against the hand-walked cases of tests/golden/dxy_hand_walked.json.

Rows are (chr, pos, freq, nind) tuples; small inputs only (a Python loop per site and per base-pair slot)."""


def fmt_g(x):
    """std::cout << double with the default precision: printf('%g')"""
    return "%g" % x


class _Reader:
    """getline over a list of parsed data lines: next() -> row or None (the C++ getline failing at the end of the file)"""

    def __init__(self, rows):
        self.rows, self.i = rows, 0

    def next(self):
        if self.i >= len(self.rows):
            return None
        self.i += 1
        return self.rows[self.i - 1]


def maf2dxy(rows1, rows2, winsize, stepsize, minind, fixed_site, chrsize, skip_missing):
    """-> (rc, stdout text, stderr text).  rows1 / rows2: the data lines of the two MAF files (at least one each: the
    reference reads the first data line of both before it looks at anything, dxyWindow.cpp:282-292)."""
    out, err = [], []
    nnew = [0]  # calcWindow's function-static (:194)

    def calc_window(buf, chrom, nsites):  # :172-209; returns the new nsites
        dxy, neff, nskip = 0.0, 0, 0
        for i in range(nsites):
            v = buf[i][1]
            if v >= 0:
                dxy += v
                neff += 1
            elif v == -9:
                nskip += 1
        if neff > 0 or not skip_missing:
            out.append(f"{chrom}\t{buf[0][0]}\t{buf[nsites - 1][0]}\t{fmt_g(dxy)}\t{neff}\t{nskip}\n")
        if nsites == winsize:  # same chromosome: keep the last winsize - stepsize entries
            nnew[0] = winsize - stepsize
            for i in range(nnew[0]):
                buf[i] = buf[stepsize + i]
        else:
            nnew[0] = 0
        return nnew[0]

    r1, r2 = _Reader(rows1), _Reader(rows2)
    m1, m2 = r1.next(), r2.next()
    chrom = prevchr = m1[0]
    if m2[0] != chrom:
        return 255, "", "Chromosomes in MAF files differ\n"
    nsites, positer = 0, 1
    buf = [None] * max(winsize, 1)
    dxy_global, neff_global, nskip_global = 0.0, 0, 0

    def put(pos, val):
        nonlocal nsites
        buf[nsites] = (pos, val)
        nsites += 1

    while True:  # `while (!maf1line.empty())`: no blank lines in the model's inputs
        if m1[1] != m2[1] or m1[0] != m2[0]:  # :315-331
            if (m1[0] == m2[0] and m1[1] < m2[1]) or (m1[0] != m2[0] and m2[0] != chrom):
                while m1[1] != m2[1]:
                    nxt = r1.next()
                    if nxt is None:
                        break
                    m1 = nxt
                if m1[1] != m2[1]:
                    break
            else:
                while m2[1] < m1[1]:
                    nxt = r2.next()
                    if nxt is None:
                        break
                    m2 = nxt
                if m1[1] != m2[1]:
                    break
        chrom = m1[0]
        if winsize > 0 and chrom != prevchr:  # :334-360
            if not fixed_site:
                if prevchr not in chrsize:
                    return 255, "".join(out), f"Unable to determine size for {prevchr}\n"
                lastpos = chrsize[prevchr]
                while positer <= lastpos:
                    if nsites == winsize:
                        nsites = calc_window(buf, prevchr, nsites)
                    put(positer, -7)
                    positer += 1
                if nsites > winsize - stepsize:
                    nsites = calc_window(buf, prevchr, nsites)
            elif nsites > 0:
                nsites = calc_window(buf, prevchr, nsites)
            positer = 1
        if winsize > 0 and not fixed_site:  # :361-370
            while positer < m1[1]:
                if nsites == winsize:
                    nsites = calc_window(buf, chrom, nsites)
                put(positer, -7)
                positer += 1
        if winsize > 0 and nsites == winsize:  # :372-374
            nsites = calc_window(buf, chrom, nsites)
        d = m1[2] * (1.0 - m2[2]) + m2[2] * (1.0 - m1[2]) if (m1[3] >= minind and m2[3] >= minind) else -9  # :381
        if d != -9:
            dxy_global += d
            neff_global += 1
        else:
            nskip_global += 1
        if winsize > 0:  # :388-393
            put(m1[1], d)
            positer += 1
        prevchr = chrom
        nxt = r1.next()  # :398-403
        if nxt is None:
            break
        m1 = nxt
        nxt = r2.next()
        if nxt is None:
            break
        m2 = nxt
    if not fixed_site:  # :407-421 (also with -winsize 0: the size file is looked up all the same)
        if chrom not in chrsize:
            return 255, "".join(out), f"Unable to determine size for {chrom}\n"
        lastpos = chrsize[chrom]
        while positer <= lastpos:
            if nsites == winsize:
                nsites = calc_window(buf, chrom, nsites)
            put(positer, -7)
            positer += 1
    if nsites > winsize - stepsize and nsites <= winsize:  # :424 (unsigned arithmetic: winsize >= stepsize always holds here)
        nsites = calc_window(buf, chrom, nsites)
    line = f"{fmt_g(dxy_global)}\t{neff_global}\t{nskip_global}\n"  # :429-433
    if winsize == 0:
        out.append(line)
    else:
        err.append(line)
    return 0, "".join(out), "".join(err)
