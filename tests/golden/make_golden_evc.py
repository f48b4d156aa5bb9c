#!/opt/conda/bin/python3.9
"""Golden vectors for candidate extraction: the reference's own
/root/reference/dataPrepScripts/ExtractVariantCandidates.py (2to3-converted copy in a temp dir, pipes
in text mode, intervaltree 3 `at` standing in for the 2.x `search`) run on the synthetic alignments of
make_golden_pileup.py (same .fa / .sam files) through fake_samtools.py.

  pileup/<case>.evc.args.json    options of the run
  pileup/<case>.evc.gz           the candidate rows it wrote
  pileup/<case>.bed              (region_bed only) the BED file given to --bed_fn

Ties between equal counts are ordered by the interpreter's dict order: this recording (CPython 3.9) and
PyPy -- which the reference recommends for this script, README.md:101 -- keep insertion order
A,C,G,T,I,D,N; CPython 2.7 would give A,C,D,G,I,N,T.  The build follows the insertion order.
Run:  /opt/conda/bin/python3.9 tests/golden/make_golden_evc.py
"""
import gzip
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "pileup")
REF = "/root/reference/dataPrepScripts"

CASES = {
    # name: (alignment case of make_golden_pileup.py, options)
    "plain": ("plain", {}),
    "region_bed": ("region", {"ctgStart": 600, "ctgEnd": 2900, "minMQ": 10, "minCoverage": 3, "threshold": 0.2,
                              "bed_fn": "BED"}),
    "noisy": ("noisy", {"ctgStart": 0, "ctgEnd": 2000, "minCoverage": 1, "threshold": 0.05}),
    "lowcov": ("eqx", {"minCoverage": 0, "threshold": 0.3}),
}

SHIM = '''
import intervaltree as _it
if not hasattr(_it.IntervalTree, "search"):
    _it.IntervalTree.search = lambda self, p: self.at(p)
'''


def prepare():
    tmp = tempfile.mkdtemp(prefix="cv_refevc_")
    for f in ("ExtractVariantCandidates.py", "param.py"):
        shutil.copy(os.path.join(REF, f), tmp)
    subprocess.check_call(["/opt/conda/bin/2to3", "-nw", os.path.join(tmp, "ExtractVariantCandidates.py"),
                           os.path.join(tmp, "param.py")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    p = os.path.join(tmp, "ExtractVariantCandidates.py")
    src = open(p).read()
    n = src.count("bufsize=8388608)")
    src = src.replace("bufsize=8388608)", "bufsize=8388608, universal_newlines=True)")
    assert n == 6, n
    src = src.replace("import intervaltree\n", "import intervaltree\n" + SHIM, 1)
    open(p, "w").write(src)
    return tmp


def main():
    tmp = prepare()
    fake = "%s %s" % (sys.executable, os.path.join(HERE, "fake_samtools.py"))
    try:
        for name, (aln, opts) in CASES.items():
            base = os.path.join(OUT, aln)
            if not os.path.exists(base + ".fa.fai"):
                open(base + ".fa.fai", "w").write("ctgA\t0\t0\t0\t0\n")       # only its existence is tested (:62)
            opts = dict(opts)
            if opts.get("bed_fn") == "BED":
                bed = os.path.join(OUT, name + ".bed")
                with open(bed, "w") as fh:
                    fh.write("ctgA\t650\t900\nctgA\t1200\t1201\nctgA\t1500\t2400\nother\t1\t50\nctgA\t2600\t2950\n")
                opts["bed_fn"] = bed
            out = os.path.join(OUT, name + ".evc.gz")
            cmd = [sys.executable, os.path.join(tmp, "ExtractVariantCandidates.py"), "--bam_fn", base + ".sam", "--ref_fn",
                   base + ".fa", "--can_fn", out, "--ctgName", "ctgA", "--samtools", fake]
            for k, v in opts.items():
                cmd += ["--" + k, str(v)]
            subprocess.check_call(cmd, cwd=tmp)
            rows = gzip.open(out, "rt").read()
            with open(out, "wb") as raw:
                with gzip.GzipFile(fileobj=raw, mode="wb", mtime=0) as gz:
                    gz.write(rows.encode())
            rec = dict(opts)
            if "bed_fn" in rec:
                rec["bed_fn"] = os.path.basename(rec["bed_fn"])
            json.dump({"alignments": aln, "options": rec}, open(os.path.join(OUT, name + ".evc.args.json"), "w"))
            print(name, "candidate rows", rows.count("\n"), os.path.getsize(out), "bytes")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
