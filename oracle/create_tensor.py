"""CPU oracle of the pileup front end -- TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(),
bench.py's cpu_baseline leg); the product never imports it.

Restates /root/reference/dataPrepScripts/CreateTensor.py: the per-read CIGAR walk with its
activation state machine (OutputAlnTensor :140-246) and the per-candidate accumulation
(GenerateTensor :23-54).  Pinned: tests/test_pileup_oracle.py checks it row for row against
tests/golden/pileup/*.tensor.gz, which the reference itself wrote (tests/golden/make_golden_pileup.py).

Differences from the reference, all outside what its outputs can show on valid input:
  * the 10 000 000-entry `availableSlots` cap (:96,182,194,209,218) is not modelled -- it silently drops
    alignment columns once that many are buffered; nothing here buffers columns;
  * rows come out in ascending candidate order (the reference flushes in dict order, :232-246);
  * a reference index outside the loaded sequence is treated as 'not ACGT' (the reference would raise
    IndexError past the end, or wrap around for a negative index).
"""
import re

FLANK = 16                     # param.flankingBaseNum
WIDTH = 2 * FLANK + 1          # 33
EXPAND = 1000000               # param.expandReferenceRegion
_CIGAR = re.compile(r"(\d+)([MIDNSHP=X])")
_IDX = {"A": 0, "C": 1, "G": 2, "T": 3}


def region_bounds(ctg_start, ctg_end):
    """CreateTensor.py:99-109 -> (ctgStart, ctgEnd, refStart, refEnd), all None without a region.
    ctgStart is moved from 0-based to 1-based, the reference window is widened by 1 Mb each side."""
    if ctg_start is None or ctg_end is None:
        return None, None, None, None
    cs = ctg_start + 1
    return cs, ctg_end, max(cs - EXPAND, 1), ctg_end + EXPAND


def select_candidates(rows, ctg, ctg_start, ctg_end):
    """CreateTensor.py:56-62: positions (1-based) of the rows of this contig inside [ctgStart, ctgEnd]"""
    out = []
    for row in rows:
        f = row.split()
        if not f or f[0] != ctg:
            continue
        p = int(f[1])
        if ctg_start is not None and p < ctg_start:
            continue
        if ctg_end is not None and p > ctg_end:
            continue
        out.append(p)
    return out


class _Acc(object):
    __slots__ = ("counts", "depth")

    def __init__(self):
        self.counts = [0] * (WIDTH * 16)
        self.depth = [0] * WIDTH


def _add(acc, center, ref_pos, query_adv, ref_base, query_base):
    """GenerateTensor's loop body (:27-48) for one alignment column"""
    if ref_base not in "ACGT-" or query_base not in "ACGT-":
        return
    d = ref_pos - center
    if d < -(FLANK + 1) or d >= FLANK:
        return
    off = d + FLANK + 1
    c = acc.counts
    if query_base != "-":
        if ref_base != "-":
            acc.depth[off] += 1
            c[16 * off + 4 * _IDX[ref_base] + 0] += 1
            c[16 * off + 4 * _IDX[query_base] + 1] += 1
            c[16 * off + 4 * _IDX[ref_base] + 2] += 1
            c[16 * off + 4 * _IDX[query_base] + 3] += 1
        else:
            idx = min(off + query_adv, WIDTH - 1)
            c[16 * idx + 4 * _IDX[query_base] + 1] += 1
    elif ref_base != "-":
        c[16 * off + 4 * _IDX[ref_base] + 2] += 1


def pileup(ref_seq, ref_start, sam_lines, candidates, min_mq=0, dcov=250, consider_left_edge=True):
    """-> {center: _Acc} for every candidate some read was activated for.
    ref_seq[0] is 1-based position `ref_start` (None = 1); candidates are 1-based positions."""
    shift = 0 if ref_start is None else ref_start - 1

    def ref_at(p):
        i = p - shift
        return ref_seq[i] if 0 <= i < len(ref_seq) else "?"

    # window starts -> (end, center)   (GetCandidate :63-71)
    begin = {}
    for pos in candidates:
        if consider_left_edge:
            for i in range(pos - (FLANK + 1), pos + (FLANK + 1)):
                begin.setdefault(i, []).append((pos + FLANK + 1, pos))
        else:
            begin[pos - (FLANK + 1)] = [(pos + FLANK + 1, pos)]
    out = {}
    prev_pos = 0
    cap = 0
    for line in sam_lines:
        f = line.split()
        if not f or f[0][0] == "@":
            continue
        pos0 = int(f[3]) - 1
        if int(f[4]) < min_mq:
            continue
        if prev_pos != pos0:
            prev_pos = pos0
            cap = 0
        else:
            cap += 1
            if cap >= dcov:
                continue
        cigar, seq = f[5], f[9]
        active = set()
        end_of = {}
        r = pos0
        q = 0

        def wake(rp):
            for r_end, center in begin.get(rp, ()):
                if center not in active:
                    end_of[r_end] = center
                    active.add(center)
                    out.setdefault(center, _Acc())

        for m in _CIGAR.finditer(cigar):
            n, op = int(m.group(1)), m.group(2)
            if op == "S":
                q += n
            elif op in "M=X":
                for _ in range(n):
                    wake(r)
                    for center in active:
                        _add(out[center], center, r, 0, ref_at(r), seq[q] if q < len(seq) else "?")
                    if r in end_of:
                        active.discard(end_of[r])
                    r += 1
                    q += 1
            elif op == "I":
                for k in range(n):
                    for center in active:
                        _add(out[center], center, r, k, "-", seq[q] if q < len(seq) else "?")
                    q += 1
            elif op == "D":
                for _ in range(n):
                    for center in active:
                        _add(out[center], center, r, 0, ref_at(r), "-")
                    wake(r)
                    if r in end_of:
                        active.discard(end_of[r])
                    r += 1
            # N, H, P: nothing (the reference does not advance on N either)
    return out


def tensor_rows(ctg, ref_seq, ref_start, acc_by_center, min_coverage=0):
    """GenerateTensor's output step (:50-54), ascending by candidate"""
    shift = 0 if ref_start is None else ref_start - 1
    rows = []
    for center in sorted(acc_by_center):
        acc = acc_by_center[center]
        new_pos = center - shift
        if new_pos - (FLANK + 1) >= 0 and acc.depth[FLANK] >= min_coverage:
            rows.append("%s %d %s %s" % (ctg, center, ref_seq[new_pos - (FLANK + 1):new_pos + FLANK],
                                          " ".join("%0.1f" % x for x in acc.counts)))
    return rows


def create_tensor(ctg, ref_contig, sam_lines, candidate_rows, ctgStart=None, ctgEnd=None, minMQ=0, dcov=250,
                  minCoverage=0, considerleftedge=True):
    """whole CreateTensor.py run on in-memory inputs; `ref_contig` is the full contig sequence (the
    faidx slice :101-116 and the `view` region filter :128-130 are applied here)"""
    cs, ce, rs, re_ = region_bounds(ctgStart, ctgEnd)
    if rs is None:
        ref_seq = ref_contig
    else:
        ref_seq = ref_contig[rs - 1:re_]
    cands = select_candidates(candidate_rows, ctg, cs, ce)
    lines = []
    for line in sam_lines:
        f = line.split("\t")
        if line.startswith("@") or f[2] != ctg or (int(f[1]) & 2308):
            continue
        if cs is not None:
            p = int(f[3])
            span = sum(int(n) for n, op in _CIGAR.findall(f[5]) if op in "MDN=X")
            if p + max(span, 1) - 1 < cs or p > ce:
                continue
        lines.append(line)
    acc = pileup(ref_seq, rs, lines, cands, minMQ, dcov, considerleftedge)
    return tensor_rows(ctg, ref_seq, rs, acc, minCoverage)
