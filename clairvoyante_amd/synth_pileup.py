"""Deterministic synthetic alignments for the pileup front end (tests, golden vectors, bench):
a reference contig, a position-sorted SAM text and a candidate list in the formats
/root/reference/dataPrepScripts/CreateTensor.py reads (SAM columns :143-153, candidate rows
`ctg pos ...` :56-62).  NumPy + stdlib only, so the golden-vector generator can load it under the
interpreter that runs the reference.
"""
import numpy as np

BASES = "ACGT"


def make_reference(rng, length, lower_frac=0.02, n_frac=0.01):
    """random contig with a few soft-masked (lower-case) and N stretches: both are 'not ACGT' for the
    pileup (CreateTensor.py:28-31)"""
    seq = np.array(list(BASES))[rng.randint(0, 4, length)]
    for frac, fn in ((lower_frac, lambda a: np.char.lower(a)), (n_frac, lambda a: np.full(a.shape, "N"))):
        k = int(length * frac / 8)
        for s in rng.randint(0, max(length - 8, 1), k):
            seq[s:s + 8] = fn(seq[s:s + 8])
    return "".join(seq)


def _mutate(rng, base):
    return BASES[(BASES.index(base.upper()) + rng.randint(1, 4)) % 4] if base.upper() in BASES else "A"


def make_read(rng, ref, pos0, target_len, profile):
    """One read starting at 0-based pos0: returns (cigar, seq, reference span).  `profile`:
    dict(sub, ins, dele, clip, weird) event probabilities per base / per read."""
    ops = []      # (op, length)
    seq = []
    r = pos0

    def push(op, n):
        if n <= 0:
            return
        if ops and ops[-1][0] == op:
            ops[-1] = (op, ops[-1][1] + n)
        else:
            ops.append((op, n))
    if rng.rand() < profile["clip"]:
        if rng.rand() < 0.3:
            push("H", rng.randint(1, 20))
        else:
            n = rng.randint(1, 12)
            push("S", n); seq.extend(BASES[i] for i in rng.randint(0, 4, n))
    if rng.rand() < profile["weird"]:      # a read that opens with an insertion or a deletion
        if rng.rand() < 0.5:
            n = rng.randint(1, 4)
            push("I", n); seq.extend(BASES[i] for i in rng.randint(0, 4, n))
        else:
            n = rng.randint(1, 4)
            push("D", n); r += n
    produced = 0
    while produced < target_len and r < len(ref):
        u = rng.rand()
        if u < profile["ins"]:
            n = 1 + rng.geometric(0.5) if rng.rand() < 0.9 else rng.randint(20, 45)
            push("I", n); seq.extend(BASES[i] for i in rng.randint(0, 4, n)); produced += n
        elif u < profile["ins"] + profile["dele"]:
            n = min(1 + rng.geometric(0.5) if rng.rand() < 0.9 else rng.randint(20, 45), len(ref) - r)
            push("D", n); r += n
        elif u < profile["ins"] + profile["dele"] + profile.get("skip", 0.0):
            push("N", rng.randint(1, 30))                                # the pileup ignores N entirely
        elif u < profile["ins"] + profile["dele"] + profile.get("skip", 0.0) + profile.get("pad", 0.0):
            push("P", rng.randint(1, 3))
        else:
            b = ref[r]
            v = rng.rand()
            if v < profile["sub"]:
                q = _mutate(rng, b); op = "X" if profile.get("eqx") else "M"
            elif v < profile["sub"] + 0.004:
                q = "N"; op = "M"
            else:
                q = b.upper() if b.upper() in BASES else BASES[rng.randint(0, 4)]
                op = "=" if profile.get("eqx") else "M"
            push(op, 1); seq.append(q); r += 1; produced += 1
    if rng.rand() < profile["clip"]:
        n = rng.randint(1, 12)
        push("S", n); seq.extend(BASES[i] for i in rng.randint(0, 4, n))
    if not any(o in "M=XD" for o, _ in ops):
        push("M", 1); seq.append("A"); r += 1
    return "".join("%d%s" % (n, o) for o, n in ops), "".join(seq), r - pos0


DEFAULT_PROFILE = dict(sub=0.02, ins=0.01, dele=0.01, clip=0.15, weird=0.03)
NOISY_PROFILE = dict(sub=0.06, ins=0.05, dele=0.05, clip=0.3, weird=0.1, skip=0.003, pad=0.002)


def make_alignments(seed, ref_len=4000, n_reads=600, read_len=(40, 151), profile=None, ctg="ctgA", stack=0,
                    start_lo=0, start_hi=None):
    """-> (ref sequence, list of SAM lines sorted by POS).  `stack` > 0 adds runs of reads sharing one
    POS (exercises --dcov, CreateTensor.py:165-172)."""
    rng = np.random.RandomState(seed)
    profile = dict(DEFAULT_PROFILE if profile is None else profile)
    ref = make_reference(rng, ref_len)
    hi = ref_len - 5 if start_hi is None else start_hi
    starts = np.sort(rng.randint(start_lo, hi, n_reads))
    if stack:
        for s in rng.randint(0, n_reads - stack - 1, 4):
            starts[s:s + stack] = starts[s]
        starts = np.sort(starts)
    lines = []
    for i, p in enumerate(starts):
        cigar, seq, _span = make_read(rng, ref, int(p), rng.randint(*read_len), profile)
        mq = int(rng.choice([0, 3, 20, 40, 60], p=[0.05, 0.05, 0.1, 0.2, 0.6]))
        flag = int(rng.choice([0, 16, 99, 147]))
        lines.append("\t".join(["r%05d" % i, str(flag), ctg, str(int(p) + 1), str(mq), cigar, "*", "0", "0", seq,
                                "*" if rng.rand() < 0.5 else "I" * len(seq)]))
    return ref, lines


def make_candidate_positions(seed, ref_len, n, lo=1, hi=None, clusters=True):
    """sorted unique 1-based candidate positions, some in dense clusters, some at the contig edges"""
    rng = np.random.RandomState(seed + 77)
    hi = ref_len if hi is None else hi
    pos = list(rng.randint(lo, hi + 1, n))
    if clusters:
        for c in rng.randint(lo, hi + 1, max(n // 20, 1)):
            pos.extend(int(c) + d for d in range(0, 12, rng.randint(1, 4)))
        pos.extend([lo, lo + 1, min(lo + 16, hi), min(lo + 17, hi), hi, max(hi - 15, lo)])
    return sorted(set(int(p) for p in pos if lo <= p <= hi))


def candidate_rows(ctg, positions, other_ctg=None):
    """rows in the candidate-file shape; CreateTensor.py only reads columns 0 and 1"""
    rows = []
    for k, p in enumerate(positions):
        if other_ctg is not None and k % 17 == 5:
            rows.append("%s %d X 9 A 5 C 4 G 0 T 0" % (other_ctg, p))      # must be ignored (:59)
        rows.append("%s %d X 9 A 5 C 4 G 0 T 0" % (ctg, p))
    return rows


def fast_alignments(n_reads, ref_len, read_len=150, seed=3, sub=0.01, ctg="ctgA"):
    """bench-scale workload built with array ops (~1 s per million reads): position-sorted reads of one
    shape (<read_len>M) over a random contig, substitution rate `sub`.  -> (reference bytes, SAM text bytes)"""
    rng = np.random.RandomState(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    ref = acgt[rng.randint(0, 4, ref_len)]
    pos = np.sort(rng.randint(0, ref_len - read_len, n_reads))
    seq = ref[pos[:, None] + np.arange(read_len)[None, :]]
    mut = rng.rand(n_reads, read_len) < sub
    seq = np.where(mut, acgt[rng.randint(0, 4, (n_reads, read_len))], seq)
    head = np.frombuffer(("r\t0\t%s\t" % ctg).encode(), dtype=np.uint8)
    mid = np.frombuffer(("\t60\t%dM\t*\t0\t0\t" % read_len).encode(), dtype=np.uint8)
    tail = np.frombuffer(b"\t*\n", dtype=np.uint8)
    digits = 10
    p1 = pos + 1
    pd = np.empty((n_reads, digits), dtype=np.uint8)
    for k in range(digits):
        pd[:, digits - 1 - k] = 48 + (p1 // 10 ** k) % 10
    line = np.concatenate([np.broadcast_to(head, (n_reads, len(head))), pd, np.broadcast_to(mid, (n_reads, len(mid))),
                           seq, np.broadcast_to(tail, (n_reads, len(tail)))], axis=1)
    return ref.tobytes(), np.ascontiguousarray(line).tobytes()
