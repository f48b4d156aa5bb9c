"""Development probe: end-to-end callVarBam on a synthetic 30x / 10 Mbp chunk (SAM text served by the test
stand-in for samtools, i.e. decode cost excluded), with a cProfile of the host side.
    python tools/gpu_e2e_bam.py [n_reads] [contig_len]"""
import cProfile
import os
import pstats
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import common
    from oracle import cv_oracle as O
    from clairvoyante_amd import callVarBam, clairvoyante_v3, synth_pileup
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 10000000
    tmp = tempfile.mkdtemp(prefix="cv_e2e_")
    ref, text = synth_pileup.fast_alignments(n_reads, L, sub=0.01)
    fa = os.path.join(tmp, "ref.fa")
    with open(fa, "w") as fh:
        fh.write(">ctgA\n")
        r = ref.decode()
        fh.write("\n".join(r[i:i + 60] for i in range(0, len(r), 60)) + "\n")
    open(fa + ".fai", "w").write("ctgA\t%d\t6\t60\t61\n" % L)
    sam = os.path.join(tmp, "reads.sam")
    open(sam, "wb").write(text)
    m = clairvoyante_v3.Clairvoyante(); m.init(); m.setParameters(common.bench_params(O, "full", seed=11))
    chk = os.path.join(tmp, "model-000001"); m.saveParameters(chk); m.close()
    fake = "%s %s" % (sys.executable, os.path.join(ROOT, "tests", "golden", "fake_samtools.py"))
    a = callVarBam.build_parser().parse_args(["--chkpnt_fn", chk, "--bam_fn", sam, "--ref_fn", fa, "--ctgName", "ctgA",
                                              "--call_fn", os.path.join(tmp, "out.vcf"), "--samtools", fake,
                                              "--threshold", "0.06"])
    t0 = time.time()
    pr = cProfile.Profile(); pr.enable()
    res = callVarBam.Run(a)
    pr.disable()
    dt = time.time() - t0
    nrec = sum(1 for l in open(a.call_fn) if not l.startswith("#"))
    print("callVarBam: %.2f s for %d reads, %d candidates, %d tensors, %d VCF records" % (dt, n_reads, res["candidates"],
                                                                                      len(res["centers"]), nrec))
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
    # the chunked whole-contig driver, with and without fetching the next chunks' text ahead
    from clairvoyante_amd import callVarBamParallel as par
    for prefetch in (1, 3):
        pa = par.build_parser().parse_args(["--chkpnt_fn", chk, "--bam_fn", sam, "--ref_fn", fa, "--output_prefix",
                                            os.path.join(tmp, "par%d" % prefetch), "--samtools", fake, "--threshold", "0.06",
                                            "--refChunkSize", str(L // 5), "--includingAllContigs", "--run", "--prefetch",
                                            str(prefetch)])
        t0 = time.time()
        todo = par.Run(pa)
        print("callVarBamParallel --run, %d chunks, prefetch %d: %.2f s" % (len(todo), prefetch, time.time() - t0))


if __name__ == "__main__":
    main()
