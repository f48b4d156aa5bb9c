#!/opt/conda/bin/python3.9
"""Golden rows of /root/reference/dataPrepScripts/GetTruth.py (2to3 copy, text-mode pipes) on a small VCF.
  truth/sites.vcf                 input
  truth/rows_all.txt, rows_region.txt   what the reference printed (whole contig; --ctgStart 400 --ctgEnd 1500)
Run:  /opt/conda/bin/python3.9 tests/golden/make_golden_truth.py
"""
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "truth")
VCF = """##fileformat=VCFv4.1
##contig=<ID=ctgA,length=3000>
#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS
ctgA\t120\t.\tA\tC\t50\tPASS\t.\tGT:DP\t0/1:30
ctgA\t400\t.\tA\tC\t50\tPASS\t.\tGT:DP\t1|1:30
ctgA\t401\t.\tAT\tA,ATTT\t50\tPASS\t.\tGT:DP\t1/2:30
ctgA\t455\t.\tG\tGAC,GA\t50\tPASS\t.\tGT:DP\t2|1:12
ctgA\t999\t.\tC\tT\t50\tPASS\t.\tGT\t./1
other\t5\t.\tA\tC\t50\tPASS\t.\tGT\t0/1
ctgA\t1500\t.\tT\tTA\t50\tPASS\t.\tGT:AD:DP\t1/0:3,9:12
ctgA\t1501\trs1\tT\tG\t50\tPASS\tDB\tGT\t1|0
ctgA\t2222\t.\tGCC\tG\t50\tPASS\t.\tGT\t1/1
"""


def main():
    os.makedirs(OUT, exist_ok=True)
    open(os.path.join(OUT, "sites.vcf"), "w").write(VCF)
    tmp = tempfile.mkdtemp(prefix="cv_reftruth_")
    try:
        shutil.copy("/root/reference/dataPrepScripts/GetTruth.py", tmp)
        p = os.path.join(tmp, "GetTruth.py")
        subprocess.check_call(["/opt/conda/bin/2to3", "-nw", p], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        src = open(p).read().replace("bufsize=8388608)", "bufsize=8388608, universal_newlines=True)")
        open(p, "w").write(src)
        for tag, extra in (("all", []), ("region", ["--ctgStart", "400", "--ctgEnd", "1500"])):
            out = subprocess.check_output([sys.executable, p, "--vcf_fn", os.path.join(OUT, "sites.vcf"), "--ctgName", "ctgA"]
                                          + extra, cwd=tmp).decode()
            open(os.path.join(OUT, "rows_%s.txt" % tag), "w").write(out)
            print(tag, out.count("\n"), "rows")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
