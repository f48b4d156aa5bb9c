"""CPU oracle of candidate extraction -- TEST INFRASTRUCTURE ONLY; the product never imports it.

Restates /root/reference/dataPrepScripts/ExtractVariantCandidates.py: the per-read pileup with its
position sweep (MakeCandidates :118-246) and the per-position decision (OutputCandidate :22-42).
Pinned: tests/test_pileup_oracle.py checks its rows against tests/golden/pileup/*.evc.gz, which the
reference itself wrote (tests/golden/make_golden_evc.py).

Ties between equal counts keep the order A,C,G,T,I,D,N (dict insertion order: CPython >= 3.7 and PyPy,
the interpreter the reference recommends for this script; CPython 2.7 would order A,C,D,G,I,N,T).
--gen4Training draws from an unseeded RNG in the reference (:203-205): not reproducible, not modelled.
"""
import re

EXPAND = 1000000
_CIGAR = re.compile(r"(\d+)([MIDNSHP=X])")
SYMBOLS = "ACGTIDN"


def decide(counts, ref_base, min_coverage, threshold):
    """OutputCandidate :22-42 -> (total, [(symbol, count) sorted]) or None; counts in SYMBOLS order"""
    total = sum(counts)
    if total < min_coverage:
        return None
    den = total if total else 1
    order = sorted(zip(SYMBOLS, counts), key=lambda x: -x[1])       # stable
    p0 = float(order[0][1]) / den
    p1 = float(order[1][1]) / den
    if (p0 <= 1.0 - threshold and p1 >= threshold) or order[0][0] != ref_base:
        return total, order
    return None


def bed_intervals(rows, ctg):
    """:90-103 -> list of half-open [begin, end) of this contig, or None when it has no interval"""
    out = []
    seen = False
    for row in rows:
        f = row.split()
        if not f:
            continue
        if f[0] != ctg:
            continue
        seen = True
        b, e = int(f[1]), int(f[2]) - 1
        if e == b:
            e += 1
        out.append((b, e))
    return out if seen else None


def candidates(ctg, ref_contig, sam_lines, ctgStart=None, ctgEnd=None, minMQ=0, minCoverage=4, threshold=0.125,
               bed=None):
    """whole run on in-memory inputs -> candidate rows in the reference's output order"""
    if ctgStart is not None and ctgEnd is not None:
        cs = ctgStart + 1
        ce = ctgEnd
        rs = max(cs - EXPAND, 1)
        ref_seq = ref_contig[rs - 1:ce + EXPAND]
        shift = rs - 1
    else:
        cs = ce = None
        ref_seq = ref_contig
        shift = 0

    def wanted(p):
        if cs is not None and not (cs <= p <= ce):
            return False
        if bed is not None and not any(b <= p < e for b, e in bed):
            return False
        return True

    rows = []

    def flush(p, cnt):
        if wanted(p):
            d = decide(cnt, ref_seq[p - shift], minCoverage, threshold)
            if d is not None:
                total, order = d
                rows.append(" ".join([ctg, str(p + 1), ref_seq[p - shift], str(total)] + ["%s %d" % x for x in order]))

    pile = {}
    sweep = 0
    for line in sam_lines:
        f = line.split()
        if not f or f[0][0] == "@" or f[2] != ctg:
            continue
        if cs is not None:                     # what `samtools view CTG:S-E` keeps
            p1 = int(f[3])
            span = sum(int(n) for n, op in _CIGAR.findall(f[5]) if op in "MDN=X")
            if p1 + max(span, 1) - 1 < cs or p1 > ce:
                continue
        if int(f[1]) & 2308:
            continue
        pos0 = int(f[3]) - 1
        if int(f[4]) < minMQ:
            continue
        ops = [(int(n), op) for n, op in _CIGAR.findall(f[5])]
        skip = sum(n for n, op in ops if op == "S")
        if 1.0 - float(skip) / (sum(n for n, _ in ops) + 1) < 0.55:
            continue
        r, q, seq = pos0, 0, f[9]
        for n, op in ops:
            if op == "S":
                q += n
            elif op in "M=X":
                for _ in range(n):
                    pile.setdefault(r, [0] * 7)[SYMBOLS.index(seq[q])] += 1
                    r += 1
                    q += 1
            elif op == "I":
                pile.setdefault(r - 1, [0] * 7)[4] += 1
                q += n
            elif op == "D":
                pile.setdefault(r - 1, [0] * 7)[5] += 1
                r += n
        while sweep < pos0:
            if sweep in pile:
                flush(sweep, pile.pop(sweep))
            sweep += 1
    for p in sorted(pile):
        flush(p, pile[p])
    return rows
