"""Development probe: callVar.py end to end on text tensors (BASELINE.json configs[0] path at a larger size):
rows/s for a plain and a gzip-compressed tensor file, with the share of each stage."""
import cProfile
import gzip
import os
import pstats
import subprocess
import sys
import tempfile
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import common
    from oracle import cv_oracle as O
    from clairvoyante_amd import callVar, clairvoyante_v3, synth
    from clairvoyante_amd.pileup import format_rows
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000         # plain text; the .gz leg takes the first ngz rows
    ngz = min(n, 200000)
    tmp = tempfile.mkdtemp(prefix="cv_cvtext_")
    x = synth.make_candidates(ngz, seed=9, device="cuda").cpu().numpy()
    x[..., 1:] += x[..., 0:1]                       # back to raw counts (the file holds them; the reader subtracts)
    x = np.maximum(x, 0)
    t0 = time.time()
    txt = os.path.join(tmp, "t.txt")
    small = os.path.join(tmp, "s.txt")
    with open(txt, "wb") as fh:                     # the same ngz tensors at advancing coordinates until n rows are written
        for s in range(0, n, ngz):
            k = min(ngz, n - s)
            rows = format_rows("chr1", np.arange(100 + s, 100 + s + k), b"N" * 83 + b"ACGT" * ((n + 200) // 4 + 8), 0, x[:k])
            blob = b"\n".join(rows) + b"\n"
            fh.write(blob)
            if s == 0:
                open(small, "wb").write(blob)
    print("wrote %d rows, %.0f MB in %.1f s" % (n, os.path.getsize(txt) / 1e6, time.time() - t0))
    subprocess.check_call("gzip -1 -c %s > %s.gz" % (small, small), shell=True)
    m = clairvoyante_v3.Clairvoyante(); m.init(); m.setParameters(common.bench_params(O, "full", seed=11))
    chk = os.path.join(tmp, "model-000001"); m.saveParameters(chk); m.close()
    # a list of compressed files (one per chunk of the genome, the form that scales): 8 copies of the small .gz, read ahead
    # several at a time by one process (utils_v2.GetTensorFiles)
    copies = []
    for i in range(8):
        c = os.path.join(tmp, "c%d.txt.gz" % i)
        os.link(small + ".gz", c)
        copies.append(c)
    gzlist = ",".join(copies)
    want = {txt: n, small: ngz, small + ".gz": ngz, gzlist: 8 * ngz}
    for fn in (txt, txt, small, small + ".gz", gzlist, gzlist):          # the large file twice: the second pass finds it in the page cache
        n = want[fn]
        a = types.SimpleNamespace(tensor_fn=fn, chkpnt_fn=chk, call_fn=os.path.join(tmp, "out.vcf"), qual=None,
                                  sampleName="S", ref_fn=None, threads=None, showRef=False, v3=True, v2=False, slim=False)
        pr = cProfile.Profile()
        t0 = time.time(); pr.enable(); callVar.Run(a); pr.disable(); dt = time.time() - t0
        nrec = sum(1 for l in open(a.call_fn) if not l.startswith("#"))
        print("%s: %.2f s -> %.0f rows/s, %d VCF records" % (os.path.basename(fn) if "," not in fn else "8 x s.txt.gz as a list", dt, n / dt, nrec))
        pstats.Stats(pr).sort_stats("tottime").print_stats(6)
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
