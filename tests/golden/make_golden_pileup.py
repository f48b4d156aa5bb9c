#!/opt/conda/bin/python3.9
"""Golden vectors for the pileup front end: the reference's own
/root/reference/dataPrepScripts/CreateTensor.py (2to3-converted copy in a temp dir, pipes switched
to text mode -- python 2 pipes are str) run on synthetic alignments through fake_samtools.py.

  pileup/<case>.fa, .sam, .can      inputs (reference contig, SAM text, candidate rows)
  pileup/<case>.args.json           the command-line options of the case
  pileup/<case>.tensor.gz           what the reference wrote to --tensor_fn

Run:  /opt/conda/bin/python3.9 tests/golden/make_golden_pileup.py
"""
import gzip
import importlib.util
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "pileup")
REF = "/root/reference/dataPrepScripts"
spec = importlib.util.spec_from_file_location("synth_pileup", os.path.join(HERE, "..", "..", "clairvoyante_amd",
                                                                           "synth_pileup.py"))
sp = importlib.util.module_from_spec(spec)
spec.loader.exec_module(sp)

CASES = {
    # name: (generator kwargs, candidate kwargs, CreateTensor options)
    "plain": (dict(seed=11, ref_len=3000, n_reads=420), dict(n=90), {}),
    "region": (dict(seed=12, ref_len=3600, n_reads=520, stack=6), dict(n=110),
               {"ctgStart": 600, "ctgEnd": 2900, "minMQ": 10, "dcov": 3, "minCoverage": 4}),
    "noleftedge": (dict(seed=13, ref_len=3000, n_reads=420, read_len=(25, 90)), dict(n=90),
                   {"considerleftedge": "False"}),
    "noisy": (dict(seed=14, ref_len=2600, n_reads=380, profile=sp.NOISY_PROFILE, stack=4), dict(n=100),
              {"ctgStart": 0, "ctgEnd": 2000, "dcov": 2}),
    "eqx": (dict(seed=15, ref_len=2000, n_reads=260, profile=dict(sp.DEFAULT_PROFILE, eqx=True)), dict(n=60),
            {"minCoverage": 9}),
    "handmade_dcov1": ("handmade", None, {"dcov": 1}),
    "handmade": ("handmade", None, {}),
}


def handmade():
    """a few records that hit the reference's corners: the very first read at POS 1 with --dcov 1 (previousPos
    starts at 0, CreateTensor.py:139), a read that opens with a long insertion, runs of more than 64 columns, a
    clipped-only record, P and H runs"""
    ref = "ACGTTGCA" * 40
    recs = [
        "a\t0\tctgA\t1\t60\t20M\t*\t0\t0\t" + ref[0:20] + "\t*",
        "b\t0\tctgA\t1\t60\t10M5I10M\t*\t0\t0\t" + ref[0:10] + "GGGGG" + ref[10:20] + "\t*",
        "d\t0\tctgA\t5\t60\t70I1M\t*\t0\t0\t" + "A" * 70 + "C\t*",
        "e\t0\tctgA\t9\t60\t5S100M20D50M3H\t*\t0\t0\t" + "T" * 5 + ref[8:108] + ref[128:178] + "\t*",
        "f\t0\tctgA\t9\t60\t8S\t*\t0\t0\tACGTACGT\t*",
        "g\t0\tctgA\t40\t60\t3M2P4M\t*\t0\t0\t" + ref[39:46] + "\t*",
        "h\t0\tctgA\t300\t60\t12M\t*\t0\t0\t" + ref[299:311] + "\t*",
    ]
    return ref, recs, [1, 2, 17, 18, 19, 25, 40, 60, 100, 129, 150, 300, 310, len(ref)]


def prepare():
    tmp = tempfile.mkdtemp(prefix="cv_refct_")
    for f in ("CreateTensor.py", "param.py"):
        shutil.copy(os.path.join(REF, f), tmp)
    subprocess.check_call(["/opt/conda/bin/2to3", "-nw", os.path.join(tmp, "CreateTensor.py"), os.path.join(tmp, "param.py")],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    p = os.path.join(tmp, "CreateTensor.py")
    src = open(p).read()
    n = src.count("bufsize=8388608)")
    src = src.replace("bufsize=8388608)", "bufsize=8388608, universal_newlines=True)")
    assert n == 6, n
    open(p, "w").write(src)
    return tmp


def main():
    os.makedirs(OUT, exist_ok=True)
    tmp = prepare()
    fake = "%s %s" % (sys.executable, os.path.join(HERE, "fake_samtools.py"))
    try:
        for name, (gkw, ckw, opts) in CASES.items():
            ctg = "ctgA"
            if gkw == "handmade":
                ref, lines, pos = handmade()
            else:
                ref, lines = sp.make_alignments(ctg=ctg, **gkw)
                pos = sp.make_candidate_positions(gkw["seed"], len(ref), **ckw)
            base = os.path.join(OUT, name)
            with open(base + ".fa", "w") as fh:
                fh.write(">%s synthetic\n" % ctg)
                for i in range(0, len(ref), 70):
                    fh.write(ref[i:i + 70] + "\n")
                fh.write(">other\nACGTACGTAC\n")
            with open(base + ".sam", "w") as fh:
                fh.write("@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:%s\tLN:%d\n" % (ctg, len(ref)))
                fh.write("\n".join(lines) + "\n")
            with open(base + ".can", "w") as fh:
                fh.write("\n".join(sp.candidate_rows(ctg, pos, other_ctg="other")) + "\n")
            json.dump(opts, open(base + ".args.json", "w"))
            cmd = [sys.executable, os.path.join(tmp, "CreateTensor.py"), "--bam_fn", base + ".sam", "--ref_fn", base + ".fa",
                   "--can_fn", base + ".can", "--tensor_fn", base + ".tensor.gz", "--ctgName", ctg, "--samtools", fake]
            for k, v in opts.items():
                cmd += ["--" + k, str(v)]
            subprocess.check_call(cmd, cwd=tmp)
            # rewrite deterministically (no gzip timestamp)
            rows = gzip.open(base + ".tensor.gz", "rt").read()
            with open(base + ".tensor.gz", "wb") as raw:
                with gzip.GzipFile(fileobj=raw, mode="wb", mtime=0) as gz:
                    gz.write(rows.encode())
            print(name, "reads", len(lines), "candidates", len(pos), "tensor rows", rows.count("\n"),
                  os.path.getsize(base + ".tensor.gz"), "bytes")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
