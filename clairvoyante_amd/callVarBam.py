"""BAM -> VCF in one process: command line and result of /root/reference/clairvoyante/callVarBam.py
(Run :58-142, main :145-223), which chains three processes through text pipes --
ExtractVariantCandidates.py (or GetTruth.py with --vcf_fn) | CreateTensor.py | callVar.py (:116-131).

    python -m clairvoyante_amd.callVarBam --chkpnt_fn MODEL --bam_fn IN.bam --ref_fn REF.fa \
           --ctgName chr21 [--ctgStart S --ctgEnd E] --call_fn OUT.vcf

Here the alignments are streamed from `samtools view` ONCE, parsed into segments that stay in HBM; the
candidate pass, the tensor pass, the network and the per-candidate reductions all run on the GPU and
only the records of non-reference calls come back to the host (clairvoyante_amd/pileup.py,
csrc/cv_pileup.hip).  Each stage equals its stand-alone drop-in (ExtractVariantCandidates, CreateTensor,
callVar), so the VCF is the one the reference's pipe produces, with records in ascending position.
--pypy, --threads and --delay are accepted and ignored (no interpreter / TensorFlow start-up to stagger).
"""
import argparse
import logging
import os
import shlex
import subprocess
import sys
import time

import numpy as np

if __package__ in (None, ""):      # run as `python <dir>/callVarBam.py` (the reference's way): make the package importable
    import os as _os, sys as _sys
    _sys.path[0] = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    import clairvoyante_amd  # noqa: F401
    __package__ = "clairvoyante_amd"
from . import param
from .CreateTensor import load_reference, read_candidates, region_of
from .ExtractVariantCandidates import read_bed, stream_alignments
from .pileup import FLANK, Pileup

logging.basicConfig(format='%(message)s', level=logging.INFO)


def CheckFileExist(fn, sfx=""):
    if not os.path.isfile(fn + sfx):
        sys.exit("Error: %s not found" % (fn + sfx))
    return os.path.abspath(fn)


def CheckCmdExist(cmd):
    if cmd == "native":                 # the built-in BAM / FASTA readers (clairvoyante_amd/bam.py)
        return cmd
    try:
        subprocess.check_output("which %s" % shlex.split(cmd)[0], shell=True)
    except Exception:
        sys.exit("Error: %s executable not found" % cmd)
    return cmd


def truth_positions(args, ctgStart, ctgEnd):
    """candidate sites from --vcf_fn: what GetTruth.py prints and CreateTensor.py then filters (:59-61)"""
    from .GetTruth import open_vcf, variant_rows
    vcf = open_vcf(args.vcf_fn, args.ctgName, ctgStart, ctgEnd)
    pos = [int(r.split()[1]) for r in variant_rows(vcf.stdout, args.ctgName, ctgStart, ctgEnd)]
    vcf.stdout.close()
    vcf.wait()
    pos = [p for p in pos if (ctgStart is None or p >= ctgStart) and (ctgEnd is None or p <= ctgEnd)]
    return np.unique(np.asarray(pos, dtype=np.int64))


def region_tensors(args, device=None, source=None):
    """candidates + tensors of one region, on the device.  -> dict(centers, tensors (matrices 1..3 minus
    matrix 0), seqs, stats, reads, candidates)"""
    import torch
    ctgStart, ctgEnd, refStart, refEnd = region_of(args)
    ref_seq = load_reference(args, refStart, refEnd)
    shift = 0 if refStart is None else refStart - 1
    if args.vcf_fn is not None:
        pl = Pileup(device=device, dcov=args.dcov, considerleftedge=args.considerleftedge)
        pl.set_reference(ref_seq, shift)
        pl.set_candidates(truth_positions(args, ctgStart, ctgEnd))
        stream_alignments(args, pl, ctgStart, ctgEnd, source)
        n_candidates = pl.n
    else:
        bed = read_bed(args.bed_fn, args.ctgName) if args.bed_fn is not None else None
        pl = Pileup(device=device, dcov=args.dcov, considerleftedge=args.considerleftedge, evc=True, retain=True,
                    contig=args.ctgName)
        pl.set_reference(ref_seq, shift)
        stream_alignments(args, pl, ctgStart, ctgEnd, source)
        pl.extract_candidates(args.threshold, args.minCoverage, (ctgStart, ctgEnd) if ctgStart is not None else None, bed)
        pl.adopt_candidates(ctgStart, ctgEnd)
        n_candidates = pl.n
    t, depth, touched = pl.finish(subtract=True)
    centers = pl.centers
    # a row exists iff a read reached the candidate and its window starts inside the loaded reference
    # (CreateTensor.py:50-51, --minCoverage 0); the reader drops rows whose centre base is not ACGT after
    # upper-casing (utils_v2.py:38-40)
    new_pos = centers - shift
    seqs = [ref_seq[max(int(p) - (FLANK + 1), 0):int(p) + FLANK].upper() if p - (FLANK + 1) >= 0 else b"" for p in new_pos]
    ok = np.array([len(sq) > FLANK and sq[FLANK:FLANK + 1] in (b"A", b"C", b"G", b"T") for sq in seqs], dtype=bool)
    keep = touched & torch.from_numpy(ok).to(t.device)
    idx = torch.nonzero(keep).squeeze(1)
    idx_h = idx.cpu().numpy()
    out = {"centers": centers[idx_h], "tensors": t.index_select(0, idx), "seqs": [seqs[i] for i in idx_h],
           "stats": pl.stats(), "reads": pl.reads_kept, "candidates": n_candidates}
    pl.close()
    return out


def Run(args, model=None, source=None):
    """`model`: an already restored Clairvoyante object to reuse (callVarBamParallel --run keeps one per rank);
    `source`: the region's `samtools view` text, already fetched (pieces of bytes)"""
    from . import callVar
    chkpnt_fn = CheckFileExist(args.chkpnt_fn, sfx=".meta")
    args.bam_fn = CheckFileExist(args.bam_fn)
    args.ref_fn = CheckFileExist(args.ref_fn)
    CheckCmdExist(args.samtools)
    if args.bed_fn is not None:
        args.bed_fn = CheckFileExist(args.bed_fn)
    if args.vcf_fn is not None:
        args.vcf_fn = CheckFileExist(args.vcf_fn)
    if args.ctgName is None:
        sys.exit("--ctgName must be specified. You can call variants on multiple chromosomes simultaneously.")
    if not (args.ctgStart is not None and args.ctgEnd is not None and int(args.ctgStart) <= int(args.ctgEnd)):
        args.ctgStart = args.ctgEnd = None          # callVarBam.py:94-97
    if args.v2:
        sys.exit("Clairvoyante v2 topologies are not part of this build (v3 / v3 slim only)")
    if args.slim:
        from . import clairvoyante_v3_slim as cv
    else:
        from . import clairvoyante_v3 as cv
    t0 = time.time()
    if model is None:
        m = cv.Clairvoyante()
        m.init()
        m.restoreParameters(chkpnt_fn)
    else:
        m = model
    res = region_tensors(args, device=m.device.index, source=source)
    t1 = time.time()
    cargs = argparse.Namespace(call_fn=args.call_fn, qual=args.qual, sampleName=args.sampleName, ref_fn=args.ref_fn,
                               showRef=False)
    ctg = args.ctgName
    centers, seqs = res["centers"], res["seqs"]
    with open(args.call_fn, "w") as call_fh:
        callVar.PrintVCFHeader(cargs, call_fh)
        from .utils_v2 import PosBatch
        callVar.CallFromDevice(cargs, m, call_fh, res["tensors"], PosBatch.from_columns(ctg, centers, seqs))
    st = res["stats"]
    logging.info("reads %d, candidates %d, tensors %d; pileup %.2f s (GPU: candidates %.1f ms, scatter %.1f ms, "
                 "finalize %.1f ms), calling %.2f s" % (res["reads"], res["candidates"], len(centers), t1 - t0,
                                                        st["candidate_ms"], st["scatter_ms"], st["finalize_ms"],
                                                        time.time() - t1))
    if model is None:
        m.close()
    return res


_CLI = (
    ("--chkpnt_fn", str, None, "Input a Clairvoyante model"),
    ("--ref_fn", str, "ref.fa", "Reference fasta file input, default: %(default)s"),
    ("--bed_fn", str, None, "Call variant only in these regions, works in intersection with ctgName, ctgStart and "
                            "ctgEnd, optional, default: as defined by ctgName, ctgStart and ctgEnd"),
    ("--bam_fn", str, "bam.bam", "BAM file input, default: %(default)s"),
    ("--call_fn", str, None, "Output variant predictions"),
    ("--vcf_fn", str, None, "Candidate sites VCF file input, if provided, variants will only be called at the sites "
                            "in the VCF file,  default: %(default)s"),
    ("--threshold", float, 0.125, "Minimum allele frequence of the 1st non-reference allele for a site to be "
                                  "considered as a condidate site, default: %(default)f"),
    ("--minCoverage", float, 4, "Minimum coverage required to call a variant, default: %(default)d"),
    ("--qual", int, None, "If set, variant with equal or higher quality will be marked PASS, or LowQual otherwise, "
                          "optional"),
    ("--sampleName", str, "SAMPLE", "Define the sample name to be shown in the VCF file"),
    ("--ctgName", str, None, "The name of sequence to be processed, default: %(default)s"),
    ("--ctgStart", int, None, "The 1-bsae starting position of the sequence to be processed"),
    ("--ctgEnd", int, None, "The inclusive ending position of the sequence to be processed"),
    ("--dcov", int, 250, "Cap depth per position at %(default)s"),
    ("--samtools", str, "samtools", "Path to the 'samtools', default: %(default)s"),
    ("--pypy", str, "pypy", "Accepted for compatibility; no interpreter is spawned"),
    ("--threads", int, None, "Accepted for compatibility"),
    ("--delay", int, 10, "Accepted for compatibility; nothing is staggered"),
)
_SWITCHES = (("--considerleftedge", True, "Count the left-most base-pairs of a read for coverage even if the starting "
                                          "position of a read is after the starting position of a tensor, "
                                          "default: %(default)s"),
             ("--v3", True, "Use Clairvoyante version 3"), ("--v2", False, "Use Clairvoyante version 2"),
             ("--slim", False, "Train using the slim version of Clairvoyante, optional"))


def build_parser():
    parser = argparse.ArgumentParser(description="Call variants using a trained Clairvoyante model and a BAM file")
    for flag, typ, default, text in _CLI:
        parser.add_argument(flag, type=typ, default=default, help=text)
    for flag, default, text in _SWITCHES:
        parser.add_argument(flag, type=param.str2bool, nargs="?", const=True, default=default, help=text)
    return parser


def main():
    parser = build_parser()
    args = parser.parse_args()
    if not sys.argv[1:]:
        parser.print_help()
        sys.exit(1)
    Run(args)


if __name__ == "__main__":
    main()
