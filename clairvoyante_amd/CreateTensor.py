"""Tensor generation from a sorted BAM and a candidate list: command line, inputs and output rows of
/root/reference/dataPrepScripts/CreateTensor.py (GetCandidate :56-77, OutputAlnTensor :93-258,
main :261-308), computed on the GPU (clairvoyante_amd/pileup.py, csrc/cv_pileup.hip).

    python -m clairvoyante_amd.CreateTensor --bam_fn IN.bam --ref_fn REF.fa --can_fn CANDIDATES \
           --tensor_fn OUT.gz --ctgName chr21 [--ctgStart S --ctgEnd E]

Like the reference it shells out to `samtools faidx` for the reference window (ctgStart-1Mb ..
ctgEnd+1Mb, :99-104) and to `samtools view -F 2308` for the alignments (:128-130), reads candidates
from --can_fn (through `gzip -fdc`) or standard input, and writes one row per candidate that some
read reached: "<ctg> <pos> <33 reference bases> <528 counts %0.1f>" (:52), to `gzip -c > tensor_fn`
or standard output.  Differences: candidates are read completely before the alignments are streamed
(the reference interleaves the two, :160-162) and rows come out in ascending position order (the
reference: dict order, :232-246).  `pileup_region()` returns the tensors in HBM instead of text --
callVarBam feeds them to the network without the text round trip.
"""
import argparse
import shlex
import subprocess
import sys

import numpy as np

if __package__ in (None, ""):      # run as `python <dir>/CreateTensor.py` (the reference's way): make the package importable
    import os as _os, sys as _sys
    _sys.path[0] = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    import clairvoyante_amd  # noqa: F401
    __package__ = "clairvoyante_amd"
from . import param
from .pileup import FLANK, Pileup, format_rows

EXPAND = 1000000          # dataPrepScripts/param.py:3 expandReferenceRegion
READ_CHUNK = 8 << 20


def region_of(args):
    """CreateTensor.py:99-109: (ctgStart 1-based, ctgEnd, refStart, refEnd) or four None"""
    if args.ctgStart is not None and args.ctgEnd is not None:
        cs = args.ctgStart + 1
        return cs, args.ctgEnd, max(cs - EXPAND, 1), args.ctgEnd + EXPAND
    return None, None, None, None


def load_reference(args, refStart, refEnd):
    if args.samtools == "native":                       # REF.fai + a seek instead of `samtools faidx`
        from .bam import faidx
        seq = faidx(args.ref_fn, args.ctgName, refStart, refEnd)
        if len(seq) == 0:
            sys.stderr.write("Failed to load reference seqeunce. Please check if the provided reference fasta %s and the "
                             "ctgName %s are correct.\n" % (args.ref_fn, args.ctgName))
            sys.exit(1)
        return seq
    where = args.ctgName if refStart is None else "%s:%d-%d" % (args.ctgName, refStart, refEnd)
    p = subprocess.Popen(shlex.split("%s faidx %s %s" % (args.samtools, args.ref_fn, where)), stdout=subprocess.PIPE,
                         bufsize=8388608)
    out = p.stdout.read()
    p.stdout.close()
    p.wait()
    lines = out.split(b"\n")
    seq = b"".join(l.rstrip() for l in lines[1:])
    if p.returncode != 0 or len(seq) == 0:
        sys.stderr.write("Failed to load reference seqeunce. Please check if the provided reference fasta %s and the "
                         "ctgName %s are correct.\n" % (args.ref_fn, args.ctgName))
        sys.exit(1)
    return seq


def read_candidates(args, ctgStart, ctgEnd):
    """GetCandidate :56-62: 1-based positions of this contig inside [ctgStart, ctgEnd]"""
    if args.can_fn != "PIPE":
        f = subprocess.Popen(shlex.split("gzip -fdc %s" % args.can_fn), stdout=subprocess.PIPE, bufsize=8388608)
        fo = f.stdout
    else:
        f = None
        fo = sys.stdin.buffer
    name = args.ctgName.encode()
    pos = []
    for row in fo:
        row = row.split()
        if not row or row[0] != name:
            continue
        p = int(row[1])
        if ctgStart is not None and p < ctgStart:
            continue
        if ctgEnd is not None and p > ctgEnd:
            continue
        pos.append(p)
    if f is not None:
        fo.close()
        f.wait()
    return np.unique(np.asarray(pos, dtype=np.int64))


def pileup_region(args, subtract=False, device=None):
    """-> dict(centers [k] int64, tensors [k,33,4,4] fp32 on the device, ref_seq bytes, shift, stats) for the
    candidates that get a row (reached by a read, window inside the loaded reference, centre depth >=
    minCoverage; :50-51)."""
    import torch
    ctgStart, ctgEnd, refStart, refEnd = region_of(args)
    ref_seq = load_reference(args, refStart, refEnd)
    shift = 0 if refStart is None else refStart - 1
    centers = read_candidates(args, ctgStart, ctgEnd)
    pl = Pileup(device=device, minMQ=args.minMQ, dcov=args.dcov, considerleftedge=args.considerleftedge)
    pl.set_reference(ref_seq, shift)
    pl.set_candidates(centers)
    from .ExtractVariantCandidates import stream_alignments
    stream_alignments(args, pl, ctgStart, ctgEnd)
    t, depth, touched = pl.finish(subtract=subtract)
    inside = torch.from_numpy((pl.centers - shift - (FLANK + 1)) >= 0).to(t.device)
    keep = touched & inside & (depth >= args.minCoverage)
    idx = torch.nonzero(keep).squeeze(1)
    out = {"centers": pl.centers[idx.cpu().numpy()], "tensors": t.index_select(0, idx), "ref_seq": ref_seq,
           "shift": shift, "stats": pl.stats(), "reads": pl.reads_kept}
    pl.close()
    return out


class TensorStdout(object):
    def __init__(self, handle):
        self.stdin = handle


def OutputAlnTensor(args):
    res = pileup_region(args, subtract=False)
    if args.tensor_fn != "PIPE":
        fpo = open(args.tensor_fn, "wb")
        fp = subprocess.Popen(shlex.split("gzip -c"), stdin=subprocess.PIPE, stdout=fpo, stderr=sys.stderr,
                              bufsize=8388608)
    else:
        fpo = None
        fp = TensorStdout(sys.stdout.buffer)
    step = 32768
    centers, tensors = res["centers"], res["tensors"]
    for s in range(0, len(centers), step):
        host = tensors[s:s + step].cpu().numpy()
        for row in format_rows(args.ctgName, centers[s:s + step], res["ref_seq"], res["shift"], host):
            fp.stdin.write(row)
            fp.stdin.write(b"\n")
    if fpo is not None:
        fp.stdin.close()
        fp.wait()
        fpo.close()
    else:
        fp.stdin.flush()
    return res


_CLI = (
    ("--bam_fn", str, "input.bam", "Sorted bam file input, default: %(default)s"),
    ("--ref_fn", str, "ref.fa", "Reference fasta file input, default: %(default)s"),
    ("--can_fn", str, "PIPE", "Variant candidate list generated by ExtractVariantCandidates.py or true variant list "
                              "generated by GetTruth.py, use PIPE for standard input, default: %(default)s"),
    ("--tensor_fn", str, "PIPE", "Tensor output, use PIPE for standard output, default: %(default)s"),
    ("--minMQ", int, 0, "Minimum Mapping Quality. Mapping quality lower than the setting will be filtered, "
                        "default: %(default)d"),
    ("--ctgName", str, "chr17", "The name of sequence to be processed, default: %(default)s"),
    ("--ctgStart", int, None, "The 1-bsae starting position of the sequence to be processed"),
    ("--ctgEnd", int, None, "The inclusive ending position of the sequence to be processed"),
    ("--samtools", str, "samtools", "Path to the 'samtools', default: %(default)s"),
    ("--dcov", int, 250, "Cap depth per position at %(default)d"),
    ("--minCoverage", int, 0, "Minimum coverage required to generate a tensor, default: %(default)d"),
)


def build_parser():
    parser = argparse.ArgumentParser(description="Generate tensors summarizing local alignments from a BAM file and "
                                                 "a list of candidate locations")
    for flag, typ, default, text in _CLI:
        parser.add_argument(flag, type=typ, default=default, help=text)
    parser.add_argument("--considerleftedge", type=param.str2bool, nargs="?", const=True, default=True,
                        help="Count the left-most base-pairs of a read for coverage even if the starting position of "
                             "a read is after the starting position of a tensor, default: %(default)s")
    return parser


def main():
    parser = build_parser()
    args = parser.parse_args()
    if not sys.argv[1:]:
        parser.print_help()
        sys.exit(1)
    OutputAlnTensor(args)


if __name__ == "__main__":
    main()
