#!/opt/conda/bin/python3.9
"""Golden command list of the reference's whole-genome driver: /root/reference/clairvoyante/
callVarBamParallel.py (2to3 copy in a temp dir; intervaltree 3 `at/overlap` standing in for 2.x `search`)
on a small .fai and BED.  Absolute paths are replaced by @DIR@ / @REF@.

  parallel/ref.fa.fai, parallel/regions.bed   inputs
  parallel/cmds_plain.txt, cmds_bed.txt       what the reference printed
Run:  /opt/conda/bin/python3.9 tests/golden/make_golden_parallel.py
"""
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "parallel")
REF = "/root/reference/clairvoyante"
SHIM = ("import intervaltree\nif not hasattr(intervaltree.IntervalTree,'search'):\n"
        "    intervaltree.IntervalTree.search = lambda self, a, b=None: self.at(a) if b is None else self.overlap(a, b)\n")


def main():
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "ref.fa.fai"), "w") as fh:
        fh.write("chr1\t25000000\t6\t60\t61\nchr2\t9999999\t100\t60\t61\nchrUn_x\t5000\t200\t60\t61\n"
                 "22\t10000001\t300\t60\t61\nchrM\t16571\t400\t60\t61\nX\t3\t500\t60\t61\n")
    with open(os.path.join(OUT, "regions.bed"), "w") as fh:
        fh.write("chr1\t100\t200\nchr1\t19999999\t20000001\n22\t10000000\t10000001\nchrUn_x\t1\t50\n")
    tmp = tempfile.mkdtemp(prefix="cv_refpar_")
    try:
        for f in ("callVarBamParallel.py", "param.py", "callVarBam.py"):
            shutil.copy(os.path.join(REF, f), tmp)
        subprocess.check_call(["/opt/conda/bin/2to3", "-nw"] + [os.path.join(tmp, f) for f in os.listdir(tmp)],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        p = os.path.join(tmp, "callVarBamParallel.py")
        src = open(p).read().replace("bufsize=8388608)", "bufsize=8388608, universal_newlines=True)")
        src = src.replace("import intervaltree\n", SHIM, 1)
        open(p, "w").write(src)
        work = os.path.join(tmp, "w")
        os.makedirs(work)
        for f in ("model.meta", "in.bam", "ref.fa"):
            open(os.path.join(work, f), "w").write("x")
        shutil.copy(os.path.join(OUT, "ref.fa.fai"), os.path.join(work, "ref.fa.fai"))
        shutil.copy(os.path.join(OUT, "regions.bed"), os.path.join(work, "regions.bed"))
        common = [sys.executable, p, "--chkpnt_fn", os.path.join(work, "model"), "--bam_fn", os.path.join(work, "in.bam"),
                  "--ref_fn", os.path.join(work, "ref.fa"), "--output_prefix", "out/calls", "--pypy", "python3", "--samtools",
                  "gzip", "--sampleName", "NA1"]
        for tag, extra in (("plain", ["--includingAllContigs"]),
                           ("bed", ["--bed_fn", os.path.join(work, "regions.bed"), "--qual", "100", "--threshold", "0.25",
                                    "--refChunkSize", "5000000"])):
            out = subprocess.check_output(common + extra, cwd=tmp).decode()
            out = out.replace(work, "@DIR@").replace(tmp, "@REF@")
            open(os.path.join(OUT, "cmds_%s.txt" % tag), "w").write(out)
            print(tag, out.count("\n"), "commands")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
