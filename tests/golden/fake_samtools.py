#!/usr/bin/env python3
"""Test stand-in for the two samtools invocations the pileup front end makes
(CreateTensor.py:101-104 `faidx REF CTG[:A-B]`, :128-130 `view -F 2308 BAM CTG[:A-B]`): the
"BAM" is a plain SAM text file, the FASTA a plain single- or multi-record file.  It only feeds
prepared inputs to the code under test -- the reference when the golden vectors are generated
(make_golden_pileup.py), this repository's CreateTensor in the tests.  `view` keeps the records
of CTG whose reference span [POS, POS + reflen) overlaps the 1-based inclusive region, which is
what samtools does for a sorted, indexed BAM (the -F filter is applied to FLAG as well)."""
import re
import sys


def parse_region(s):
    if ":" in s:
        name, rng = s.rsplit(":", 1)
        a, b = rng.split("-")
        return name, int(a), int(b)
    return s, None, None


def faidx(fa, region):
    name, a, b = parse_region(region)
    seq = []
    on = False
    found = False
    for line in open(fa):
        if line.startswith(">"):
            on = line[1:].split()[0] == name
            found = found or on
        elif on:
            seq.append(line.strip())
    if not found:
        sys.stderr.write("[faidx] region %s not found\n" % region)
        sys.exit(1)
    seq = "".join(seq)
    if a is not None:
        seq = seq[max(a, 1) - 1:b]
    sys.stdout.write(">%s\n" % region)
    for i in range(0, len(seq), 60):
        sys.stdout.write(seq[i:i + 60] + "\n")


def view(args):
    flt = 0
    rest = []
    i = 0
    while i < len(args):
        if args[i] == "-F":
            flt = int(args[i + 1]); i += 2
        else:
            rest.append(args[i]); i += 1
    sam, region = rest[0], rest[1]
    name, a, b = parse_region(region)
    for line in open(sam):
        if line.startswith("@"):
            continue
        f = line.split("\t")
        if f[2] != name or (int(f[1]) & flt):
            continue
        if a is not None:
            pos = int(f[3])
            span = sum(int(n) for n, op in re.findall(r"(\d+)([MIDNSHP=X])", f[5]) if op in "MDN=X")
            if pos + max(span, 1) - 1 < a or pos > b:
                continue
        sys.stdout.write(line)


if __name__ == "__main__":
    if sys.argv[1] == "faidx":
        faidx(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "view":
        view(sys.argv[2:])
    else:
        sys.exit("unsupported: %r" % sys.argv[1:])
