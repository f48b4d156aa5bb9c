"""Whole-genome driver: command line and printed command list of
/root/reference/clairvoyante/callVarBamParallel.py (Run :21-87, main :90-154) -- one callVarBam command
per reference chunk (default 10 Mbp) of every major contig in REF.fai, skipping chunks without a BED
interval -- plus `--run`, which executes the chunks here instead of printing them: one process per GPU
(`torchrun --nproc-per-node N -m clairvoyante_amd.callVarBamParallel --run ...`), chunk k goes to rank
k mod N, the model is loaded once per rank, no collective is needed (chunks are independent; the
per-chunk VCFs are concatenated afterwards exactly as with the reference, README.md:197).
"""
import argparse
import os
import shlex
import subprocess
import sys

if __package__ in (None, ""):      # run as `python <dir>/callVarBamParallel.py` (the reference's way): make the package importable
    import os as _os, sys as _sys
    _sys.path[0] = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    import clairvoyante_amd  # noqa: F401
    __package__ = "clairvoyante_amd"
from . import param

_nums = [str(a) for a in list(range(0, 23)) + ["X", "Y"]]
majorContigs = set("chr" + a for a in _nums) | set(_nums)          # callVarBamParallel.py:9


def CheckFileExist(fn, sfx=""):
    if not os.path.isfile(fn + sfx):
        sys.exit("Error: %s not found" % (fn + sfx))
    return os.path.abspath(fn)


def bed_by_contig(bed_fn):
    """:49-62 -> {contig: [(begin, end)]} half-open"""
    tree = {}
    fp = subprocess.Popen(shlex.split("gzip -fdc %s" % bed_fn), stdout=subprocess.PIPE, bufsize=8388608)
    for row in fp.stdout:
        row = row.strip().split()
        if not row:
            continue
        begin, end = int(row[1]), int(row[2]) - 1
        if end == begin:
            end += 1
        tree.setdefault(row[0].decode(), []).append((begin, end))
    fp.stdout.close()
    fp.wait()
    return tree


def chunks(args):
    """(contig, start, end, output file) of every chunk the reference would emit (:64-86)"""
    tree = bed_by_contig(args.bed_fn) if args.bed_fn is not None else None
    for line in open(args.ref_fn + ".fai"):
        fields = line.strip().split("\t")
        name = fields[0]
        if not args.includingAllContigs and name not in majorContigs:
            continue
        length = int(fields[1])
        start = 0
        while start < length:
            end = min(start + args.refChunkSize, length)
            if tree is None or any(b < end and start < e for b, e in tree.get(name, ())):   # IntervalTree.search(start, end)
                yield name, start, end, "%s.%s_%d_%d.vcf" % (args.output_prefix, name, start, end)
            start = end


def Run(args):
    here = os.path.dirname(os.path.abspath(__file__))
    chk = CheckFileExist(args.chkpnt_fn, sfx=".meta")
    bam = CheckFileExist(args.bam_fn)
    ref = CheckFileExist(args.ref_fn)
    CheckFileExist(args.ref_fn + ".fai")
    bed = CheckFileExist(args.bed_fn) if args.bed_fn is not None else None
    vcf = "--vcf_fn %s" % CheckFileExist(args.vcf_fn) if args.vcf_fn is not None else ""
    left = "--considerleftedge" if args.considerleftedge else ""
    qual = "--qual %d" % args.qual if args.qual else ""
    todo = list(chunks(args))
    if not args.run:
        for name, start, end, out in todo:
            bedopt = "--bed_fn %s " % bed if bed is not None else ""
            print("python %s --chkpnt_fn %s --ref_fn %s --bam_fn %s %s--ctgName %s --ctgStart %d --ctgEnd %d --call_fn %s "
                  "--threshold %f --minCoverage %f --pypy %s --samtools %s --delay %d --threads %d --sampleName %s %s %s %s"
                  % (os.path.join(here, "callVarBam.py"), chk, ref, bam, bedopt, name, start, end, out, args.threshold,
                     args.minCoverage, args.pypy, args.samtools, args.delay, args.tensorflowThreads, args.sampleName,
                     vcf, left, qual))
        return todo
    # --run: this rank's share of the chunks, in this process
    from . import callVarBam, parallel
    rank, ws, _local = parallel.init_from_env()
    if args.slim:
        from . import clairvoyante_v3_slim as cv
    else:
        from . import clairvoyante_v3 as cv
    m = cv.Clairvoyante()                      # one model per rank, restored once
    m.init()
    m.restoreParameters(chk)
    mine = [t for k, t in enumerate(todo) if k % ws == rank]

    def chunk_args(name, start, end, out):
        return callVarBam.build_parser().parse_args(
            ["--chkpnt_fn", chk, "--ref_fn", ref, "--bam_fn", bam, "--ctgName", name, "--ctgStart", str(start), "--ctgEnd",
             str(end), "--call_fn", out, "--threshold", str(args.threshold), "--minCoverage", str(args.minCoverage),
             "--samtools", args.samtools, "--sampleName", args.sampleName, "--considerleftedge", str(bool(args.considerleftedge))]
            + (["--bed_fn", bed] if bed else []) + (["--vcf_fn", args.vcf_fn] if args.vcf_fn else [])
            + (["--qual", str(args.qual)] if args.qual else []) + (["--slim"] if args.slim else []))

    # `samtools view` (BAM decoding) is the slow producer: fetch the text of the next chunks in background
    # threads while the GPU works on the current one
    from concurrent.futures import ThreadPoolExecutor
    from .CreateTensor import region_of
    from .ExtractVariantCandidates import view_chunks

    def fetch(a):
        cs, ce, _rs, _re = region_of(a)
        return list(view_chunks(a, cs, ce))

    jobs = [chunk_args(*t) for t in mine]
    for a in jobs:      # callVarBam.Run's own normalisation of the region (callVarBam.py:94-97)
        if not (a.ctgStart is not None and a.ctgEnd is not None and int(a.ctgStart) <= int(a.ctgEnd)):
            a.ctgStart = a.ctgEnd = None
    depth = max(1, int(args.prefetch))
    with ThreadPoolExecutor(max_workers=depth) as pool:
        pending = [pool.submit(fetch, a) for a in jobs[:depth]]
        for i, a in enumerate(jobs):
            text = pending[i].result()
            pending[i] = None
            if i + depth < len(jobs):
                pending.append(pool.submit(fetch, jobs[i + depth]))
            callVarBam.Run(a, model=m, source=text)
    m.close()
    return todo


_CLI = (
    ("--chkpnt_fn", str, None, "Input a Clairvoyante model"),
    ("--ref_fn", str, "ref.fa", "Reference fasta file input, default: %(default)s"),
    ("--bed_fn", str, None, "Call variant only in these regions, optional, default: whole genome"),
    ("--refChunkSize", int, 10000000, "Divide job with smaller genome chunk size for parallelism, default: %(default)s"),
    ("--bam_fn", str, "bam.bam", "BAM file input, default: %(default)s"),
    ("--vcf_fn", str, None, "Candidate sites VCF file input, if provided, variants will only be called at the sites in "
                            "the VCF file,  default: %(default)s"),
    ("--output_prefix", str, None, "Output prefix"),
    ("--tensorflowThreads", int, 4, "Number of threads per tensorflow job, default: %(default)s"),
    ("--threshold", float, 0.2, "Minimum allele frequence of the 1st non-reference allele for a site to be considered "
                                "as a condidate site, default: %(default)f"),
    ("--minCoverage", float, 4, "Minimum coverage required to call a variant, default: %(default)d"),
    ("--qual", int, None, "If set, variant with equal or higher quality will be marked PASS, or LowQual otherwise, optional"),
    ("--sampleName", str, "SAMPLE", "Define the sample name to be shown in the VCF file"),
    ("--samtools", str, "samtools", "Path to the 'samtools', default: %(default)s"),
    ("--pypy", str, "pypy", "Path to the 'pypy', default: %(default)s"),
    ("--delay", int, 10, "Wait a short while for no more than %(default)s to start the job."),
    ("--prefetch", int, 3, "(--run) chunks whose `samtools view` text is fetched ahead, default: %(default)s"),
)
_SWITCHES = (("--includingAllContigs", False, "Call variants on all contigs, default: chr{1..22,X,Y,M,MT} and {1..22,X,Y,MT}"),
             ("--considerleftedge", True, "Count the left-most base-pairs of a read for coverage even if the starting "
                                          "position of a read is after the starting position of a tensor, default: %(default)s"),
             ("--slim", False, "(--run) use the slim topology"),
             ("--run", False, "execute the chunks on the GPUs of this node instead of printing the commands"))


def build_parser():
    parser = argparse.ArgumentParser(description="Create commands for calling variants in parallel using a trained "
                                                 "Clairvoyante model and a BAM file")
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
