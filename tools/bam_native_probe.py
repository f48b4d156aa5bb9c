"""Development probe: SAM text rate of the native BAM reader (BGZF inflate + record printing) by thread count, and
callVarBam end to end with it, on a synthetic 30x BAM written by tests/bam_writer.py."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from bam_writer import write_bam
    from clairvoyante_amd import synth_pileup
    from clairvoyante_amd.bam import BamFile
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
    L = n_reads * 5
    tmp = tempfile.mkdtemp(prefix="cv_bamp_")
    ref, text = synth_pileup.fast_alignments(n_reads, L)
    t0 = time.time()
    bam = os.path.join(tmp, "a.bam")
    write_bam(bam, text.decode().splitlines(), [("ctgA", L)])
    print("wrote %d reads: BAM %.0f MB (SAM text %.0f MB) in %.0f s" % (n_reads, os.path.getsize(bam) / 1e6, len(text) / 1e6,
                                                                       time.time() - t0))
    for threads in (1, 4, 16):
        bf = BamFile(bam, threads=threads)
        t0 = time.time()
        nbytes = sum(len(c) for c in bf.view("ctgA"))
        dt = time.time() - t0
        bf.close()
        print("threads %2d: %.2f s -> %.0f MB/s of SAM text, %.0f MB/s of BAM, %.2f M reads/s" % (
            threads, dt, nbytes / dt / 1e6, os.path.getsize(bam) / dt / 1e6, n_reads / dt / 1e6))
    if "--feed" in sys.argv:
        import ctypes
        from clairvoyante_amd import _lib
        for threads in (4, 16):                   # inflate + record walk alone
            bf = BamFile(bam, threads=threads)
            _lib.check(bf.lib.cv_bam_view_begin(bf.h, b"ctgA", 0, 0, 2308, 0))
            base = ctypes.c_void_p(); offs = ctypes.c_void_p(); done = ctypes.c_int(0)
            t0 = time.time(); k = 0
            while not done.value:
                k += bf.lib.cv_bam_view_records(bf.h, 64 << 20, ctypes.byref(base), ctypes.byref(offs), ctypes.byref(done))
            dt = time.time() - t0
            bf.close()
            print("records only threads %2d: %.3f s -> %.2f M reads/s (%d records)" % (threads, dt, n_reads / dt / 1e6, k))
        # host side of the front end: BAM -> segments queued for the GPU, through the SAM text and straight from the records
        from clairvoyante_amd.pileup import Pileup
        for feed in ("text", "records"):
            for threads in (4, 16):
                pl = Pileup(evc=True, retain=True, contig="ctgA", threads=threads)
                pl.set_reference(ref, 0)
                bf = BamFile(bam, threads=threads)
                t0 = time.time()
                if feed == "text":
                    for c in bf.view("ctgA"):
                        pl.add_sam(c)
                else:
                    pl.add_bam(bf, "ctgA")
                import torch
                torch.cuda.synchronize()
                dt = time.time() - t0
                bf.close()
                print("feed %-7s threads %2d: %.2f s -> %.2f M reads/s, %.0f MB/s of BAM (%d reads kept)" % (
                    feed, threads, dt, n_reads / dt / 1e6, os.path.getsize(bam) / dt / 1e6, pl.reads_kept))
                pl.close()
    if "--e2e" in sys.argv:
        import common
        from oracle import cv_oracle as O
        from clairvoyante_amd import callVarBam, clairvoyante_v3
        fa = os.path.join(tmp, "ref.fa")
        r = ref.decode()
        with open(fa, "w") as fh:
            fh.write(">ctgA\n" + "\n".join(r[i:i + 60] for i in range(0, len(r), 60)) + "\n")
        open(fa + ".fai", "w").write("ctgA\t%d\t6\t60\t61\n" % L)
        m = clairvoyante_v3.Clairvoyante(); m.init(); m.setParameters(common.bench_params(O, "full", seed=11))
        chk = os.path.join(tmp, "model-000001"); m.saveParameters(chk); m.close()
        a = callVarBam.build_parser().parse_args(["--chkpnt_fn", chk, "--bam_fn", bam, "--ref_fn", fa, "--ctgName", "ctgA",
                                                  "--call_fn", os.path.join(tmp, "o.vcf"), "--samtools", "native",
                                                  "--threshold", "0.06"])
        t0 = time.time()
        res = callVarBam.Run(a)
        print("callVarBam --samtools native: %.2f s, %d reads -> %d candidates" % (time.time() - t0, res["reads"], res["candidates"]))
        if "--profile" in sys.argv:
            import cProfile
            import pstats
            pr = cProfile.Profile(); pr.enable()
            callVarBam.Run(a)
            pr.disable()
            pstats.Stats(pr).sort_stats("cumulative").print_stats(22)


if __name__ == "__main__":
    main()
