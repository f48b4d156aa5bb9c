"""Variant-candidate extraction from a sorted BAM: command line, inputs and output rows of
/root/reference/dataPrepScripts/ExtractVariantCandidates.py (OutputCandidate :22-42, MakeCandidates
:53-246, main :249-308), computed on the GPU (clairvoyante_amd/pileup.py, csrc/cv_pileup.hip).

    python -m clairvoyante_amd.ExtractVariantCandidates --bam_fn IN.bam --ref_fn REF.fa --ctgName chr21 \
           [--ctgStart S --ctgEnd E] [--bed_fn REGIONS.bed] [--threshold 0.125] [--minCoverage 4] > CANDIDATES

One row per selected position: "<ctg> <pos 1-based> <ref base> <total> <symbol count> x 7", symbols sorted
by descending count, ties in the order A,C,G,T,I,D,N (the dict insertion order of PyPy, which the
reference recommends for this script, and of CPython >= 3.7).  Rows come in the reference's order:
positions before the last read's POS ascending, then its final loop (:215-241).  --gen4Training
subsamples with an unseeded RNG exactly like the reference (:203-205) and is therefore not reproducible.
"""
import os
import random
import shlex
import subprocess
import sys

import numpy as np

if __package__ in (None, ""):      # run as `python <dir>/ExtractVariantCandidates.py` (the reference's way): make the package importable
    import os as _os, sys as _sys
    _sys.path[0] = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    import clairvoyante_amd  # noqa: F401
    __package__ = "clairvoyante_amd"
from . import param
from .CreateTensor import READ_CHUNK, load_reference, region_of
from .pileup import Pileup

SYMBOLS = "ACGTIDN"


def read_bed(bed_fn, ctgName):
    """:90-108 -> list of half-open (begin, end) of this contig"""
    f = subprocess.Popen(shlex.split("gzip -fdc %s" % bed_fn), stdout=subprocess.PIPE, bufsize=8388608)
    out = []
    seen = False
    name = ctgName.encode()
    for row in f.stdout:
        row = row.strip().split()
        if not row or row[0] != name:
            continue
        seen = True
        begin, end = int(row[1]), int(row[2]) - 1
        if end == begin:
            end += 1
        out.append((begin, end))
    f.stdout.close()
    f.wait()
    if not seen:
        sys.stderr.write("ctgName is not in the bed file, are you using the correct bed file (%s)?\n" % bed_fn)
        sys.exit(1)
    return out


def candidate_rows(ctgName, res, ref_seq, shift):
    """text rows in the reference's order from Pileup.extract_candidates()"""
    pos0, late, counts = res["pos0"], res["late"], res["counts"]
    last = res["last_pos"]
    tail = (late != 0) | (pos0 >= last)              # reported by the final loop (:215-241), sorted by position
    order = np.concatenate([np.nonzero(~tail)[0], np.nonzero(tail)[0]])
    rows = []
    for i in order:
        c = counts[i]
        srt = sorted(zip(SYMBOLS, (int(v) for v in c)), key=lambda x: -x[1])
        rb = chr(ref_seq[int(pos0[i]) - shift])
        rows.append(" ".join([ctgName, str(int(pos0[i]) + 1), rb, str(int(c.sum()))] + ["%s %d" % x for x in srt]))
    return rows


def view_chunks(args, ctgStart, ctgEnd):
    """the text of `samtools view -F 2308 BAM CTG[:S-E]` (CreateTensor.py:128-130) in READ_CHUNK pieces"""
    if args.samtools == "native":                       # no external process: clairvoyante_amd/bam.py
        from .bam import BamFile
        bf = BamFile(args.bam_fn)
        for chunk in bf.view(args.ctgName, ctgStart, ctgEnd, chunk=READ_CHUNK):
            yield chunk
        bf.close()
        return
    where = args.ctgName if ctgStart is None else "%s:%d-%d" % (args.ctgName, ctgStart, ctgEnd)
    p2 = subprocess.Popen(shlex.split("%s view -F 2308 %s %s" % (args.samtools, args.bam_fn, where)),
                          stdout=subprocess.PIPE, bufsize=8388608)
    while True:
        chunk = p2.stdout.read(READ_CHUNK)
        if not chunk:
            break
        yield chunk
    p2.stdout.close()
    p2.wait()


def stream_alignments(args, pl, ctgStart, ctgEnd, source=None):
    """feed the alignments of the region to the pileup; `source`: already fetched text pieces (a prefetching
    caller), default: spawn samtools now"""
    if source is None and args.samtools == "native":    # BAM records go to the pileup as they are: no SAM text
        from .bam import BamFile
        bf = BamFile(args.bam_fn)
        try:
            pl.add_bam(bf, args.ctgName, ctgStart, ctgEnd)
        finally:
            bf.close()
        return
    for chunk in (view_chunks(args, ctgStart, ctgEnd) if source is None else source):
        pl.add_sam(chunk)


def MakeCandidates(args):
    if args.gen4Training:
        args.minCoverage = 0
        args.threshold = 0
        args.outputProb = (args.candidates * 2.) / args.genomeSize
    if not os.path.isfile("%s.fai" % args.ref_fn):
        sys.stderr.write("Fasta index %s.fai doesn't exist.\n" % args.ref_fn)
        sys.exit(1)
    ctgStart, ctgEnd, refStart, refEnd = region_of(args)
    ref_seq = load_reference(args, refStart, refEnd)
    shift = 0 if refStart is None else refStart - 1
    bed = read_bed(args.bed_fn, args.ctgName) if args.bed_fn is not None else None
    pl = Pileup(evc=True, evc_minMQ=args.minMQ, contig=args.ctgName, minMQ=1 << 30)   # tensor pass switched off
    pl.set_reference(ref_seq, shift)
    stream_alignments(args, pl, ctgStart, ctgEnd)
    res = pl.extract_candidates(args.threshold, args.minCoverage, (ctgStart, ctgEnd) if ctgStart is not None else None, bed)
    pl.close()
    rows = candidate_rows(args.ctgName, res, ref_seq, shift)
    if args.gen4Training:
        rows = [r for r in rows if not random.uniform(0, 1) > args.outputProb]
    if args.can_fn != "PIPE":
        fpo = open(args.can_fn, "wb")
        fp = subprocess.Popen(shlex.split("gzip -c"), stdin=subprocess.PIPE, stdout=fpo, stderr=sys.stderr, bufsize=8388608)
        out = fp.stdin
    else:
        fpo = fp = None
        out = sys.stdout.buffer
    for r in rows:
        out.write(r.encode())
        out.write(b"\n")
    if fp is not None:
        fp.stdin.close()
        fp.wait()
        fpo.close()
    else:
        out.flush()
    if res["reads"] == 0:
        sys.stderr.write("No read has been process, either the genome region you specified has no read cover, or please "
                         "check the correctness of your BAM input (%s).\n" % args.bam_fn)
        sys.exit(0)
    return rows


_CLI = (
    ("--bam_fn", str, "input.bam", "Sorted bam file input, default: %(default)s"),
    ("--ref_fn", str, "ref.fa", "Reference fasta file input, default: %(default)s"),
    ("--bed_fn", str, None, "Call variant only in these regions, works in intersection with ctgName, ctgStart and "
                            "ctgEnd, optional, default: as defined by ctgName, ctgStart and ctgEnd"),
    ("--can_fn", str, "PIPE", "Pile-up count output, use PIPE for standard output, default: %(default)s"),
    ("--threshold", float, 0.125, "Minimum allele frequence of the 1st non-reference allele for a site to be considered "
                                  "as a condidate site, default: %(default)f"),
    ("--minCoverage", float, 4, "Minimum coverage required to call a variant, default: %(default)f"),
    ("--minMQ", int, 0, "Minimum Mapping Quality. Mapping quality lower than the setting will be filtered, "
                        "default: %(default)d"),
    ("--candidates", int, 7000000, "Use with gen4Training, number of variant candidates to be generated, "
                                   "default: %(default)s"),
    ("--genomeSize", int, 3000000000, "Use with gen4Training, default: %(default)s"),
    ("--ctgName", str, "chr17", "The name of sequence to be processed, default: %(default)s"),
    ("--ctgStart", int, None, "The 1-bsae starting position of the sequence to be processed"),
    ("--ctgEnd", int, None, "The inclusive ending position of the sequence to be processed"),
    ("--samtools", str, "samtools", "Path to the 'samtools', default: %(default)s"),
)


def build_parser():
    import argparse
    parser = argparse.ArgumentParser(description="Generate variant candidates using alignments")
    for flag, typ, default, text in _CLI:
        parser.add_argument(flag, type=typ, default=default, help=text)
    parser.add_argument("--gen4Training", type=param.str2bool, nargs="?", const=True, default=False,
                        help="Output all genome positions as candidate for model training (Set --threshold to 0, "
                             "--minCoverage to 0), default: %(default)s")
    return parser


def main():
    parser = build_parser()
    args = parser.parse_args()
    if not sys.argv[1:]:
        parser.print_help()
        sys.exit(1)
    MakeCandidates(args)


if __name__ == "__main__":
    main()
