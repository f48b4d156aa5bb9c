"""Truth variants from a VCF: command line and output rows of
/root/reference/dataPrepScripts/GetTruth.py (OutputVariant :29-87, main :90-115) -- host text only.

    python -m clairvoyante_amd.GetTruth --vcf_fn TRUTH.vcf.gz --ctgName chr21 [--ctgStart S --ctgEnd E] > VARS

Row: "<ctg> <pos> <ref> <alt> <gt1> <gt2>"; the genotype comes from the last column, a 1/2 call with several
ALT alleles is reduced to 0/1 on the shortest ALT (:66-76).  Uses `tabix` for a region when the VCF has an
index and tabix is installed, otherwise streams the file through `gzip -fdc` (:45-51).
"""
import argparse
import os
import shlex
import subprocess
import sys


def variant_rows(stream, ctgName, ctgStart, ctgEnd):
    name = ctgName.encode()
    for row in stream:
        row = row.strip().split()
        if not row or row[0][:1] == b"#" or row[0] != name:
            continue
        if ctgStart is not None and ctgEnd is not None:
            if int(row[1]) < ctgStart or int(row[1]) > ctgEnd:
                continue
        gt = row[-1].split(b":")[0].replace(b"/", b"|").replace(b".", b"0").split(b"|")
        p1, p2 = (int(x) for x in gt)
        if p1 > p2:
            p1, p2 = p2, p1
        alt = row[4]
        if p1 == 1 and p2 == 2 and b"," in alt:
            p1, p2 = 0, 1
            shortest = b""
            best = 99
            for a in alt.split(b","):
                if len(a) < best:
                    best, shortest = len(a), a
            alt = shortest
        yield b" ".join([row[0], row[1], row[3], alt, str(p1).encode(), str(p2).encode()])


def open_vcf(vcf_fn, ctgName, ctgStart, ctgEnd):
    if ctgStart is not None and ctgEnd is not None and os.path.isfile("%s.tbi" % vcf_fn):
        try:
            subprocess.check_output("which tabix", shell=True)
            return subprocess.Popen(shlex.split("tabix -f -p vcf %s %s:%s-%s" % (vcf_fn, ctgName, ctgStart, ctgEnd)),
                                    stdout=subprocess.PIPE, bufsize=8388608)
        except subprocess.CalledProcessError:
            pass
    return subprocess.Popen(shlex.split("gzip -fdc %s" % vcf_fn), stdout=subprocess.PIPE, bufsize=8388608)


def OutputVariant(args):
    ctgStart, ctgEnd = args.ctgStart, args.ctgEnd
    if ctgStart is not None and ctgEnd is not None:
        ctgStart += 1
    if args.var_fn != "PIPE":
        fpo = open(args.var_fn, "wb")
        fp = subprocess.Popen(shlex.split("gzip -c"), stdin=subprocess.PIPE, stdout=fpo, stderr=sys.stderr, bufsize=8388608)
        out = fp.stdin
    else:
        fpo = fp = None
        out = sys.stdout.buffer
    vcf = open_vcf(args.vcf_fn, args.ctgName, ctgStart, ctgEnd)
    for r in variant_rows(vcf.stdout, args.ctgName, ctgStart, ctgEnd):
        out.write(r)
        out.write(b"\n")
    vcf.stdout.close()
    vcf.wait()
    if fp is not None:
        fp.stdin.close()
        fp.wait()
        fpo.close()
    else:
        out.flush()


def main():
    parser = argparse.ArgumentParser(description="Extract variant type and allele from a Truth dataset")
    parser.add_argument("--vcf_fn", type=str, default="input.vcf", help="Truth vcf file input, default: %(default)s")
    parser.add_argument("--var_fn", type=str, default="PIPE",
                        help="Truth variants output, use PIPE for standard output, default: %(default)s")
    parser.add_argument("--ctgName", type=str, default="chr17", help="The name of sequence to be processed, default: %(default)s")
    parser.add_argument("--ctgStart", type=int, default=None, help="The 1-bsae starting position of the sequence to be processed")
    parser.add_argument("--ctgEnd", type=int, default=None, help="The inclusive ending position of the sequence to be processed")
    args = parser.parse_args()
    if not sys.argv[1:]:
        parser.print_help()
        sys.exit(1)
    OutputVariant(args)


if __name__ == "__main__":
    main()
