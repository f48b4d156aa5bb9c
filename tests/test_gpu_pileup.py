"""On-GPU pileup (csrc/cv_pileup.hip through the C ABI) against (1) the rows the reference's own
CreateTensor.py wrote (tests/golden/pileup/), (2) the CPU oracle (oracle/create_tensor.py, itself pinned
against those rows) on larger random alignments, (3) the text path: tensors handed to the network
straight from HBM equal the ones that went through the text rows and utils_v2.GetTensor.  Bit-exact:
the tensors are integer counts."""
import gzip
import os
import sys
import types

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from test_pileup_oracle import CASES, G, load_case, norm_opts  # noqa: E402

pytestmark = pytest.mark.gpu
FAKE = "%s %s" % (sys.executable, os.path.join(HERE, "golden", "fake_samtools.py"))


def ct_args(name, tmp_path, **over):
    base = os.path.join(G, name)
    a = dict(bam_fn=base + ".sam", ref_fn=base + ".fa", can_fn=base + ".can", tensor_fn=str(tmp_path / "out.gz"),
             minMQ=0, ctgName="ctgA", ctgStart=None, ctgEnd=None, samtools=FAKE, dcov=250, minCoverage=0,
             considerleftedge=True)
    a.update(over)
    return types.SimpleNamespace(**a)


@pytest.mark.parametrize("name", CASES)
def test_createtensor_rows_equal_reference_rows(name, tmp_path):
    from clairvoyante_amd import CreateTensor
    _, _, _, opts, want = load_case(name)
    args = ct_args(name, tmp_path, **norm_opts(opts))
    res = CreateTensor.OutputAlnTensor(args)
    got = gzip.open(args.tensor_fn, "rt").read().splitlines()
    assert sorted(got) == sorted(want)
    assert res["stats"]["columns"] > (1000 if not name.startswith("handmade") else 100) and res["stats"]["launches"] >= 1


def run_pileup(ref, first0, lines, centers, chunk, **kw):
    from clairvoyante_amd.pileup import Pileup
    pl = Pileup(**kw)
    pl.set_reference(ref, first0)
    pl.set_candidates(centers)
    text = ("\n".join(lines) + "\n").encode()
    for s in range(0, len(text), chunk):
        pl.add_sam(text[s:s + chunk])
    t, d, u = pl.finish()
    st = pl.stats()
    pl.close()
    return pl.centers, t.cpu().numpy(), d.cpu().numpy(), u.cpu().numpy(), st


def oracle_arrays(ref, lines, centers, min_mq, dcov, left):
    from oracle import create_tensor as ct
    acc = ct.pileup(ref, None, lines, list(centers), min_mq, dcov, left)
    T = np.zeros((len(centers), 33, 4, 4), dtype=np.float32)
    D = np.zeros(len(centers), dtype=np.int32)
    U = np.zeros(len(centers), dtype=bool)
    for i, c in enumerate(centers):
        if c in acc:
            T[i] = np.asarray(acc[c].counts, dtype=np.float32).reshape(33, 4, 4)
            D[i] = acc[c].depth[16]
            U[i] = True
    return T, D, U


@pytest.mark.parametrize("seed,left,profile,read_len,chunk", [
    (1, True, "default", (40, 151), 1 << 20),
    (2, False, "default", (30, 120), 977),          # chunk boundaries inside lines
    (3, True, "noisy", (40, 200), 4099),
    (4, False, "noisy", (20, 90), 1 << 20),
    (5, True, "noisy", (1500, 4000), 65536),         # long reads: thousands of CIGAR runs, > 64-column runs
])
def test_pileup_equals_oracle_on_random_alignments(seed, left, profile, read_len, chunk):
    from clairvoyante_amd import synth_pileup as sp
    prof = sp.NOISY_PROFILE if profile == "noisy" else sp.DEFAULT_PROFILE
    long_reads = read_len[0] > 1000
    ref, lines = sp.make_alignments(seed=500 + seed, ref_len=12000 if long_reads else 6000,
                                    n_reads=120 if long_reads else 900, read_len=read_len, profile=prof, stack=5)
    centers = np.asarray(sp.make_candidate_positions(500 + seed, len(ref), 400), dtype=np.int64)
    c, T, D, U, st = run_pileup(ref, 0, lines, centers, chunk, minMQ=3, dcov=4, considerleftedge=left)
    assert np.array_equal(c, centers)
    To, Do, Uo = oracle_arrays(ref, lines, centers, 3, 4, left)
    assert np.array_equal(U, Uo)
    assert np.array_equal(D[U], Do[U])
    assert np.array_equal(T, To)
    assert T.sum() > 1000 and st["columns"] > 10000


def test_reference_window_offset_and_flush_between_batches():
    """reference slice that does not start at position 0, reads fed in two flushed batches, candidates
    partly outside the slice"""
    from clairvoyante_amd import synth_pileup as sp
    from clairvoyante_amd.pileup import Pileup
    from oracle import create_tensor as ct
    ref, lines = sp.make_alignments(seed=77, ref_len=5000, n_reads=700, start_lo=900, start_hi=3800)
    centers = np.asarray(sp.make_candidate_positions(77, len(ref), 300, lo=950, hi=4000), dtype=np.int64)
    first0 = 1000                    # slice [1000, 4200): reads starting before it see 'not ACGT'
    sl = ref[first0:4200]
    pl = Pileup(minMQ=0, dcov=250, considerleftedge=True)
    pl.set_reference(sl, first0)
    pl.set_candidates(centers)
    half = len(lines) // 2
    pl.add_sam(("\n".join(lines[:half]) + "\n").encode())
    pl.lib.cv_pileup_flush(pl.h, pl._stream())
    pl.add_sam(("\n".join(lines[half:])).encode())          # no trailing newline: taken at finish()
    t, d, u = pl.finish()
    acc = ct.pileup(sl, first0 + 1, lines, list(centers), 0, 250, True)
    T = t.cpu().numpy()
    for i, c in enumerate(centers):
        if c in acc:
            assert bool(u[i])
            assert np.array_equal(T[i].reshape(-1), np.asarray(acc[c].counts, dtype=np.float32)), c
        else:
            assert not bool(u[i]) and T[i].sum() == 0
    # subtract mode = the reader's transform (utils_v2.py:46)
    t2, _, _ = pl.finish(subtract=True)
    ex = T.copy()
    ex[..., 1:] -= ex[..., 0:1]
    assert np.array_equal(t2.cpu().numpy(), ex)
    pl.close()


def test_tensors_from_hbm_equal_the_text_round_trip(tmp_path, oracle):
    """CreateTensor rows -> utils_v2.GetTensor  ==  pileup_region(subtract=True), and so do the calls"""
    import torch
    from clairvoyante_amd import CreateTensor, clairvoyante_v3, utils_v2
    sys.path.insert(0, HERE)
    from common import bench_params
    args = ct_args("plain", tmp_path)
    CreateTensor.OutputAlnTensor(args)
    res = CreateTensor.pileup_region(args, subtract=True)
    Xs, pos = [], []
    for end, n, X, p in utils_v2.GetTensor(args.tensor_fn, 1000):
        Xs.append(X[:n]); pos.extend(p[:n])
        if end:
            break
    Xt = np.concatenate(Xs)
    # GetTensor upper-cases the row's bases and drops rows whose centre base is not ACGT (utils_v2.py:38-40)
    keep = np.array([chr(res["ref_seq"][int(c) - res["shift"] - 1]).upper() in "ACGT" for c in res["centers"]])
    assert keep.sum() == len(Xt) and keep.sum() > 50
    Xd = res["tensors"][torch.from_numpy(keep).to(res["tensors"].device)]
    assert np.array_equal(Xd.cpu().numpy(), Xt)
    m = clairvoyante_v3.Clairvoyante()
    m.init()
    m.setParameters(bench_params(oracle, "full", seed=3))
    a = np.concatenate(m.predict(Xt), axis=1)
    b = m.predict_device(Xd.contiguous()).cpu().numpy()
    assert np.array_equal(a, b)
    want = oracle.predict("full", bench_params(oracle, "full", seed=3), Xt)
    assert np.array_equal(a, want)
    m.close()


def test_bad_inputs_are_reported():
    from clairvoyante_amd import _lib
    from clairvoyante_amd.pileup import Pileup
    pl = Pileup()
    pl.set_reference("ACGT" * 50, 0)
    with pytest.raises(_lib.CvError):
        pl.set_candidates([])            # fine: empty
        pl.add_sam(b"r1\t0\tctgA\t5\t60\n", final=True)     # truncated record
    pl.close()
    pl = Pileup()
    with pytest.raises(_lib.CvError):
        pl.add_sam(b"r1\t0\tctgA\t5\t60\t4M\t*\t0\t0\tACGT\t*\n")
        pl.finish()                      # no reference / candidates yet
    pl.close()
    # damaged CIGARs are refused before they size anything: a run beyond BAM's 2^28 - 1, a digit string that would overflow,
    # more than 2 Gi query bases in one record; a truncated record at the very end of a buffer that is not NUL-terminated
    for cigar in (b"300000000M", b"99999999999999999999999M", b"268435455M" * 9):
        pl = Pileup()
        pl.set_reference("ACGT" * 50, 0)
        pl.set_candidates([20])
        with pytest.raises(_lib.CvError, match="CIGAR"):
            pl.add_sam(b"r1\t0\tctgA\t5\t60\t" + cigar + b"\t*\t0\t0\tACGT\t*\n", final=True)
        pl.close()
    pl = Pileup()
    pl.set_reference("ACGT" * 50, 0)
    pl.set_candidates([20])
    with pytest.raises(_lib.CvError, match="fields"):
        pl.add_sam(bytes(bytearray(b"r1\t0\tctgA\t5")), final=True)
    pl.close()


# ---- candidate extraction (ExtractVariantCandidates.py) -------------------------------------------

from test_pileup_oracle import EVC_CASES, load_evc_case  # noqa: E402


def evc_args(aln, tmp_path, **over):
    base = os.path.join(G, aln)
    a = dict(bam_fn=base + ".sam", ref_fn=base + ".fa", bed_fn=None, can_fn=str(tmp_path / "can.gz"), threshold=0.125,
             minCoverage=4, minMQ=0, gen4Training=False, candidates=7000000, genomeSize=3000000000, ctgName="ctgA",
             ctgStart=None, ctgEnd=None, samtools=FAKE)
    a.update(over)
    return types.SimpleNamespace(**a)


@pytest.mark.parametrize("name", EVC_CASES)
def test_extract_candidates_rows_equal_reference_rows(name, tmp_path):
    from clairvoyante_amd import ExtractVariantCandidates as evc
    aln, _, _, opts, _, want = load_evc_case(name)
    opts = dict(opts)
    meta = __import__("json").load(open(os.path.join(G, name + ".evc.args.json")))["options"]
    if "bed_fn" in meta:
        opts["bed_fn"] = os.path.join(G, meta["bed_fn"])
    args = evc_args(aln, tmp_path, **opts)
    evc.MakeCandidates(args)
    got = gzip.open(args.can_fn, "rt").read().splitlines()
    assert got == want                      # same rows, same order (late entries at the end)


@pytest.mark.parametrize("seed,profile,mincov,thr", [(21, "default", 4, 0.125), (22, "noisy", 1, 0.05),
                                                     (23, "noisy", 0, 0.3), (24, "default", 6, 0.2)])
def test_extract_candidates_equals_oracle_on_random_alignments(seed, profile, mincov, thr):
    from clairvoyante_amd import ExtractVariantCandidates as evc
    from clairvoyante_amd import synth_pileup as sp
    from clairvoyante_amd.pileup import Pileup
    from oracle import extract_candidates as ec
    prof = sp.NOISY_PROFILE if profile == "noisy" else sp.DEFAULT_PROFILE
    ref, lines = sp.make_alignments(seed=700 + seed, ref_len=8000, n_reads=1500, profile=prof, stack=6)
    want = ec.candidates("ctgA", ref, lines, minMQ=5, minCoverage=mincov, threshold=thr)
    pl = Pileup(evc=True, evc_minMQ=5, contig="ctgA", minMQ=1 << 30)
    pl.set_reference(ref, 0)
    text = ("\n".join(lines) + "\n").encode()
    for s in range(0, len(text), 30011):
        pl.add_sam(text[s:s + 30011])
    res = pl.extract_candidates(thr, mincov)
    pl.close()
    got = evc.candidate_rows("ctgA", res, ref.encode(), 0)
    assert len(want) > 20
    assert got == want


def test_fused_extract_then_tensors_equals_the_two_step_pipeline(tmp_path):
    """one parse of the SAM text (evc + retain) == ExtractVariantCandidates rows piped into CreateTensor"""
    from clairvoyante_amd import CreateTensor
    from clairvoyante_amd import ExtractVariantCandidates as evc
    from clairvoyante_amd.pileup import Pileup
    contigs, sam, _, _, _ = load_case("noisy")
    a = evc_args("noisy", tmp_path, ctgStart=0, ctgEnd=2000, minCoverage=2)
    evc.MakeCandidates(a)
    c = ct_args("noisy", tmp_path, can_fn=a.can_fn, ctgStart=0, ctgEnd=2000, dcov=3)
    two = CreateTensor.pileup_region(c)
    # fused
    cs, ce, rs, re_ = CreateTensor.region_of(types.SimpleNamespace(ctgStart=0, ctgEnd=2000))
    ref_seq = contigs["ctgA"][rs - 1:re_].encode()
    pl = Pileup(evc=True, retain=True, contig="ctgA", dcov=3)
    pl.set_reference(ref_seq, rs - 1)
    import shlex
    import subprocess
    text = subprocess.check_output(shlex.split("%s view -F 2308 %s ctgA:%d-%d" % (FAKE, c.bam_fn, cs, ce)))
    half = len(text) // 2
    pl.add_sam(text[:half]); pl.lib.cv_pileup_flush(pl.h, pl._stream()); pl.add_sam(text[half:])
    pl.extract_candidates(0.125, 2, (cs, ce))
    centers = pl.adopt_candidates(cs, ce)
    t, d, u = pl.finish()
    keep = u.cpu().numpy() & ((centers - (rs - 1) - 17) >= 0)
    assert np.array_equal(centers[keep], two["centers"])
    assert np.array_equal(t.cpu().numpy()[keep], two["tensors"].cpu().numpy())
    assert len(two["centers"]) > 50
    pl.close()


# ---- callVarBam: BAM -> VCF fused on the device ------------------------------------------------------

def _checkpoint(oracle, tmp_path, arch="full"):
    from clairvoyante_amd import clairvoyante_v3, clairvoyante_v3_slim
    from common import bench_params
    m = (clairvoyante_v3 if arch == "full" else clairvoyante_v3_slim).Clairvoyante()
    m.init()
    m.setParameters(bench_params(oracle, arch, seed=11))
    prefix = str(tmp_path / "model-000001")
    m.saveParameters(prefix)
    m.close()
    return prefix


@pytest.mark.parametrize("case,over", [("plain", {}), ("noisy", {"ctgStart": 0, "ctgEnd": 2000, "minCoverage": 2, "dcov": 3}),
                                       ("region", {"ctgStart": 600, "ctgEnd": 2900, "threshold": 0.2, "bed": True,
                                                   "slim": True})])
def test_callvarbam_vcf_equals_the_three_stage_pipeline(case, over, tmp_path, oracle):
    """fused BAM->VCF == ExtractVariantCandidates | CreateTensor | callVar run as separate text stages"""
    sys.path.insert(0, HERE)
    from clairvoyante_amd import CreateTensor, callVar, callVarBam
    from clairvoyante_amd import ExtractVariantCandidates as evc
    over = dict(over)
    slim = over.pop("slim", False)
    bed = os.path.join(G, "region_bed.bed") if over.pop("bed", False) else None
    chk = _checkpoint(oracle, tmp_path, "slim" if slim else "full")
    base = os.path.join(G, case)
    region = {k: over[k] for k in ("ctgStart", "ctgEnd") if k in over}
    thr = over.get("threshold", 0.125); mincov = over.get("minCoverage", 4); dcov = over.get("dcov", 250)
    # three stages through files
    a = evc_args(case, tmp_path, threshold=thr, minCoverage=mincov, bed_fn=bed, **region)
    evc.MakeCandidates(a)
    c = ct_args(case, tmp_path, can_fn=a.can_fn, dcov=dcov, **region)
    CreateTensor.OutputAlnTensor(c)
    v = types.SimpleNamespace(tensor_fn=c.tensor_fn, chkpnt_fn=chk, call_fn=str(tmp_path / "staged.vcf"), qual=None,
                              sampleName="SAMPLE", ref_fn=base + ".fa", threads=None, showRef=False, v3=True, v2=False,
                              slim=slim)
    callVar.Run(v)
    # fused
    f = callVarBam.build_parser().parse_args(
        ["--chkpnt_fn", chk, "--bam_fn", base + ".sam", "--ref_fn", base + ".fa", "--ctgName", "ctgA", "--call_fn",
         str(tmp_path / "fused.vcf"), "--samtools", FAKE, "--threshold", str(thr), "--minCoverage", str(mincov),
         "--dcov", str(dcov)] + (["--bed_fn", bed] if bed else []) + (["--slim"] if slim else []) +
        sum([["--" + k, str(x)] for k, x in region.items()], []))
    res = callVarBam.Run(f)
    staged = open(v.call_fn).read().splitlines()
    fused = open(f.call_fn).read().splitlines()
    body = lambda L: [l for l in L if not l.startswith("#")]
    assert [l for l in staged if l.startswith("#")] == [l for l in fused if l.startswith("#")]
    assert body(fused) == body(staged)
    assert len(body(fused)) > 10 and res["candidates"] >= len(res["centers"]) > 20


def test_callvarbam_with_candidate_sites_from_a_vcf(tmp_path, oracle):
    """--vcf_fn: sites come from the VCF (GetTruth rows) instead of the candidate pass (callVarBam.py:121-125)"""
    sys.path.insert(0, HERE)
    from clairvoyante_amd import CreateTensor, callVar, callVarBam, GetTruth
    chk = _checkpoint(oracle, tmp_path)
    base = os.path.join(G, "plain")
    sites = tmp_path / "sites.vcf"
    want_pos = [120, 455, 456, 457, 1010, 1500, 2222, 2950]
    with open(sites, "w") as fh:
        fh.write("##fileformat=VCFv4.1\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS\n")
        for k, p in enumerate(want_pos):
            gt = ["0/1", "1|1", "1/2", "./1"][k % 4]
            fh.write("ctgA\t%d\t.\tA\t%s\t50\tPASS\t.\tGT:DP\t%s:30\n" % (p, "C,CT" if gt == "1/2" else "C", gt))
        fh.write("other\t5\t.\tA\tC\t50\tPASS\t.\tGT\t0/1\n")
    g = types.SimpleNamespace(vcf_fn=str(sites), var_fn=str(tmp_path / "var.gz"), ctgName="ctgA", ctgStart=None, ctgEnd=None)
    GetTruth.OutputVariant(g)
    rows = gzip.open(g.var_fn, "rt").read().splitlines()
    assert [int(r.split()[1]) for r in rows] == want_pos
    assert rows[2].split()[2:] == ["A", "C", "0", "1"] and rows[1].split()[4:] == ["1", "1"] and rows[3].split()[4:] == ["0", "1"]
    c = ct_args("plain", tmp_path, can_fn=g.var_fn)
    CreateTensor.OutputAlnTensor(c)
    v = types.SimpleNamespace(tensor_fn=c.tensor_fn, chkpnt_fn=chk, call_fn=str(tmp_path / "staged.vcf"), qual=30,
                              sampleName="S1", ref_fn=base + ".fa", threads=None, showRef=False, v3=True, v2=False, slim=False)
    callVar.Run(v)
    f = callVarBam.build_parser().parse_args(
        ["--chkpnt_fn", chk, "--bam_fn", base + ".sam", "--ref_fn", base + ".fa", "--ctgName", "ctgA", "--call_fn",
         str(tmp_path / "fused.vcf"), "--samtools", FAKE, "--vcf_fn", str(sites), "--qual", "30", "--sampleName", "S1"])
    callVarBam.Run(f)
    assert open(f.call_fn).read() == open(v.call_fn).read()


def test_parser_thread_count_does_not_change_the_result():
    """2 MiB chunks parsed by 1 thread and by 7 threads: identical candidates, tensors, depths and read counts
    (per-POS depth cap and late-entry state cross the slice boundaries)"""
    from clairvoyante_amd import synth_pileup as sp
    from clairvoyante_amd.pileup import Pileup
    ref, lines = sp.make_alignments(seed=901, ref_len=40000, n_reads=24000, profile=sp.NOISY_PROFILE, stack=9, read_len=(40, 120))
    text = ("\n".join(lines) + "\n").encode()
    assert len(text) > (2 << 20)
    outs = []
    for threads in (1, 7):
        pl = Pileup(evc=True, retain=True, contig="ctgA", dcov=3, minMQ=3, evc_minMQ=5, threads=threads)
        pl.set_reference(ref, 0)
        for s0 in range(0, len(text), 2 << 20):
            pl.add_sam(text[s0:s0 + (2 << 20)])
        res = pl.extract_candidates(0.1, 3)
        centers = pl.adopt_candidates()
        t, d, u = pl.finish()
        outs.append((res["pos0"], res["late"], res["counts"], res["reads"], centers, t.cpu().numpy(), d.cpu().numpy(),
                     u.cpu().numpy(), pl.reads_kept))
        pl.close()
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(np.asarray(a), np.asarray(b))
    assert outs[0][1].sum() > 0 and len(outs[0][4]) > 1000          # late entries exist, many candidates


def test_callvarbamparallel_run_executes_the_chunks(tmp_path, oracle):
    """--run: the chunks of the command list are executed in this process (rank 0 of 1), one VCF per chunk,
    each equal to a stand-alone callVarBam of that region"""
    sys.path.insert(0, HERE)
    from clairvoyante_amd import callVarBam, callVarBamParallel
    chk = _checkpoint(oracle, tmp_path)
    base = os.path.join(G, "plain")
    import shutil
    fa = str(tmp_path / "ref.fa")
    shutil.copy(base + ".fa", fa)
    with open(fa + ".fai", "w") as fh:
        fh.write("ctgA\t3000\t16\t70\t71\nother\t10\t3100\t10\t11\n")
    prefix = str(tmp_path / "calls")
    a = callVarBamParallel.build_parser().parse_args(
        ["--chkpnt_fn", chk, "--bam_fn", base + ".sam", "--ref_fn", fa, "--output_prefix", prefix, "--samtools", FAKE,
         "--refChunkSize", "1600", "--includingAllContigs", "--threshold", "0.125", "--run"])
    todo = callVarBamParallel.Run(a)
    assert [(c, s, e) for c, s, e, _ in todo] == [("ctgA", 0, 1600), ("ctgA", 1600, 3000), ("other", 0, 10)]
    total = 0
    for ctg, s0, e0, out in todo:
        f = callVarBam.build_parser().parse_args(
            ["--chkpnt_fn", chk, "--bam_fn", base + ".sam", "--ref_fn", fa, "--ctgName", ctg, "--ctgStart", str(s0),
             "--ctgEnd", str(e0), "--call_fn", str(tmp_path / "single.vcf"), "--samtools", FAKE, "--threshold", "0.125"])
        callVarBam.Run(f)
        assert open(out).read() == open(f.call_fn).read()
        total += sum(1 for l in open(out) if not l.startswith("#"))
    assert total > 10


def test_pileup_corner_cases_equal_oracle():
    """hand-made records: first read at POS 1 under --dcov 1 (the reference's previousPos starts at 0, so that read
    is already 'the second at its POS' and is dropped, CreateTensor.py:139,165-172), SEQ '*', a read that is one
    long insertion, runs longer than 64 columns, clipped-only and padded CIGARs, no reads / no candidates"""
    from clairvoyante_amd.pileup import Pileup
    from oracle import create_tensor as ct
    ref = ("ACGTTGCA" * 40)
    L = len(ref)
    recs = [
        "a\t0\tctgA\t1\t60\t20M\t*\t0\t0\t" + ref[0:20] + "\t*",
        "b\t0\tctgA\t1\t60\t10M5I10M\t*\t0\t0\t" + ref[0:10] + "GGGGG" + ref[10:20] + "\t*",
        "c\t0\tctgA\t3\t60\t30M\t*\t0\t0\t*\t*",                                   # no SEQ
        "d\t0\tctgA\t5\t60\t70I1M\t*\t0\t0\t" + "A" * 70 + "C\t*",                 # opens with a long insertion
        "e\t0\tctgA\t9\t60\t5S100M20D50M3H\t*\t0\t0\t" + "T" * 5 + ref[8:108] + ref[128:178] + "\t*",
        "f\t0\tctgA\t9\t60\t8S\t*\t0\t0\tACGTACGT\t*",                             # nothing aligned
        "g\t0\tctgA\t40\t60\t3M2P4M\t*\t0\t0\t" + ref[39:46] + "\t*",
        "h\t0\tctgA\t300\t60\t12M\t*\t0\t0\t" + ref[299:311] + "\t*",
    ]
    centers = np.asarray([1, 2, 17, 18, 19, 25, 40, 60, 100, 129, 150, 300, 310, L], dtype=np.int64)
    for dcov, left in ((1, True), (250, True), (250, False)):
        pl = Pileup(dcov=dcov, considerleftedge=left)
        pl.set_reference(ref, 0)
        pl.set_candidates(centers)
        pl.add_sam("\n".join(recs).encode(), final=True)
        t, d, u = pl.finish()
        acc = ct.pileup(ref, None, recs, list(centers), 0, dcov, left)
        T = t.cpu().numpy(); U = u.cpu().numpy()
        for i, c in enumerate(centers):
            assert bool(U[i]) == (int(c) in acc), (dcov, left, c)
            exp = np.asarray(acc[int(c)].counts, dtype=np.float32) if int(c) in acc else np.zeros(528, np.float32)
            assert np.array_equal(T[i].reshape(-1), exp), (dcov, left, c)
        pl.close()
    # no reads at all / no candidates at all
    pl = Pileup()
    pl.set_reference(ref, 0)
    pl.set_candidates(centers)
    t, d, u = pl.finish()
    assert float(t.abs().sum()) == 0.0 and not bool(u.any())
    pl.set_candidates([])
    pl.add_sam(("\n".join(recs) + "\n").encode())
    t, d, u = pl.finish()
    assert t.shape[0] == 0
    pl.close()


def test_callvarbam_with_the_native_bam_reader(tmp_path, oracle):
    """--samtools native: BAM (+.bai) and FASTA (+.fai) read in-process; same VCF as through the samtools stand-in"""
    sys.path.insert(0, HERE)
    import shutil
    from bam_writer import write_bam
    from clairvoyante_amd import callVarBam
    chk = _checkpoint(oracle, tmp_path)
    base = os.path.join(G, "noisy")
    recs = [l.rstrip("\n") for l in open(base + ".sam") if not l.startswith("@")]
    bam = str(tmp_path / "noisy.bam")
    write_bam(bam, recs, [("ctgA", 2600), ("other", 10)], block_payload=9001)
    fa = str(tmp_path / "ref.fa")
    shutil.copy(base + ".fa", fa)
    # .fai of the 70-column FASTA the generator writes: ">ctgA synthetic\n" is 16 bytes
    with open(fa + ".fai", "w") as fh:
        fh.write("ctgA\t2600\t16\t70\t71\n")
    common_args = ["--chkpnt_fn", chk, "--ref_fn", fa, "--ctgName", "ctgA", "--threshold", "0.125", "--minCoverage", "2"]
    for region in ([], ["--ctgStart", "300", "--ctgEnd", "1900"]):
        a = callVarBam.build_parser().parse_args(common_args + region + ["--bam_fn", base + ".sam", "--samtools", FAKE,
                                                                        "--call_fn", str(tmp_path / "fake.vcf")])
        callVarBam.Run(a)
        b = callVarBam.build_parser().parse_args(common_args + region + ["--bam_fn", bam, "--samtools", "native",
                                                                        "--call_fn", str(tmp_path / "native.vcf")])
        callVarBam.Run(b)
        assert open(b.call_fn).read() == open(a.call_fn).read()
        assert sum(1 for l in open(b.call_fn) if not l.startswith("#")) > 10


def test_pileup_counts_are_additive_over_the_reads_at_scale():
    """size-independent property at a size the Python oracle cannot reach (300 000 reads, 45 M alignment columns): the
    tensors are counts, so the run over all reads equals the sum of the runs over the even and the odd reads (the
    depth cap never triggers at 30x); matrix 3 at the centre sums to the reported depth; and the candidate pass finds the
    same candidates when the text arrives in one piece or in 1 MiB pieces"""
    from clairvoyante_amd import synth_pileup as sp
    from clairvoyante_amd.pileup import Pileup
    ref, text = sp.fast_alignments(300000, 1500000, sub=0.02)
    lines = text.split(b"\n")[:-1]
    centers = np.arange(50, 1500000 - 50, 97, dtype=np.int64)

    def run(chunks):
        pl = Pileup()
        pl.set_reference(ref, 0)
        pl.set_candidates(centers)
        for c in chunks:
            pl.add_sam(c)
        t, d, u = pl.finish()
        out = (t.cpu().numpy(), d.cpu().numpy(), u.cpu().numpy())
        pl.close()
        return out
    full = run([text])
    even = run([b"\n".join(lines[0::2]) + b"\n"])
    odd = run([b"\n".join(lines[1::2]) + b"\n"])
    assert np.array_equal(full[0], even[0] + odd[0])
    assert np.array_equal(full[1], even[1] + odd[1]) and full[2].all()
    assert np.array_equal(full[0][:, 16, :, 3].sum(axis=1), full[1].astype(np.float32))
    assert 25 < full[1].mean() < 35
    cands = []
    for step in (len(text), 1 << 20):
        pl = Pileup(evc=True, contig="ctgA", minMQ=1 << 30)
        pl.set_reference(ref, 0)
        for s0 in range(0, len(text), step):
            pl.add_sam(text[s0:s0 + step])
        res = pl.extract_candidates(0.1, 4)
        cands.append((res["pos0"].copy(), res["counts"].copy(), res["reads"]))
        pl.close()
    assert np.array_equal(cands[0][0], cands[1][0]) and np.array_equal(cands[0][1], cands[1][1])
    assert cands[0][2] == 300000 and len(cands[0][0]) > 1000


def _both_feeds(tmp_path, ref, recs, centers, region=(None, None), payload=60000, **kw):
    """the same alignments through the SAM text and straight from BAM records -> two tuples of results"""
    from bam_writer import write_bam
    from clairvoyante_amd.bam import BamFile
    from clairvoyante_amd.pileup import Pileup
    bam = str(tmp_path / ("f%d.bam" % payload))
    write_bam(bam, recs, [("ctgA", len(ref)), ("zzz", 50)], block_payload=payload, index=True)
    outs = []
    for feed in ("text", "bam"):
        pl = Pileup(contig="ctgA", **kw)
        pl.set_reference(ref, 0)
        if centers is not None:
            pl.set_candidates(centers)
        bf = BamFile(bam, threads=3)
        if feed == "text":
            for chunk in bf.view("ctgA", region[0], region[1], chunk=1 << 20):
                pl.add_sam(chunk)
        else:
            pl.add_bam(bf, "ctgA", region[0], region[1], window=1 << 20)
        bf.close()
        res = None
        if kw.get("evc"):
            res = pl.extract_candidates(0.1, 3)
            if kw.get("retain"):
                pl.adopt_candidates()
        t, d, u = pl.finish()
        outs.append((pl.centers.copy(), t.cpu().numpy(), d.cpu().numpy(), u.cpu().numpy(), pl.reads_kept,
                     None if res is None else (res["pos0"], res["late"], res["counts"], res["reads"])))
        pl.close()
    return outs


def _same(a, b):
    for x, y in zip(a[:5], b[:5]):
        assert np.array_equal(np.asarray(x), np.asarray(y))
    if a[5] is not None:
        for x, y in zip(a[5], b[5]):
            assert np.array_equal(np.asarray(x), np.asarray(y))


def test_bam_records_feed_equals_the_sam_text_feed_on_corner_cases(tmp_path):
    """cv_pileup_add_bam takes every record as the line `samtools view` prints for it: SEQ '*', CIGAR '*', '='/'X',
    padding, clips, an insertion-only read, runs longer than 64 columns, the per-POS depth cap"""
    ref = ("ACGTTGCA" * 40)
    recs = [
        "a\t0\tctgA\t1\t60\t20M\t*\t0\t0\t" + ref[0:20] + "\t*",
        "b\t0\tctgA\t1\t60\t10M5I10M\t*\t0\t0\t" + ref[0:10] + "GGGGG" + ref[10:20] + "\t*",
        "c\t0\tctgA\t3\t60\t30M\t*\t0\t0\t*\t*",
        "c2\t0\tctgA\t4\t60\t*\t*\t0\t0\tACGT\t*",
        "d\t0\tctgA\t5\t60\t70I1M\t*\t0\t0\t" + "A" * 70 + "C\t*",
        "d2\t0\tctgA\t5\t7\t2I6=1X3D4M\t*\t0\t0\tTT" + ref[4:10] + "G" + ref[14:18] + "\t*",
        "e\t0\tctgA\t9\t60\t5S100M20D50M3H\t*\t0\t0\t" + "T" * 5 + ref[8:108] + ref[128:178] + "\t*",
        "f\t0\tctgA\t9\t60\t8S\t*\t0\t0\tACGTACGT\t*",
        "g\t0\tctgA\t40\t60\t3M2P4M\t*\t0\t0\t" + ref[39:46] + "\t*",
        "g2\t0\tctgA\t40\t60\t3M5N4M\t*\t0\t0\t" + ref[39:42] + ref[47:51] + "\t*",
        "g3\t1024\tctgA\t41\t60\t9M\t*\t0\t0\t" + ref[40:49] + "\t*",            # duplicate: -F 2308 drops it
        "h\t0\tctgA\t300\t60\t12M\t*\t0\t0\t" + ref[299:311].lower() + "\t*",
    ]
    centers = np.asarray([1, 2, 5, 6, 17, 18, 19, 25, 40, 44, 60, 100, 129, 150, 300, 310, len(ref)], dtype=np.int64)
    for kw in (dict(dcov=1), dict(dcov=250, considerleftedge=False), dict(dcov=250, minMQ=10)):
        a, b = _both_feeds(tmp_path, ref, recs, centers, **kw)
        _same(a, b)
        assert a[4] > 0 and a[3].any()
    a, b = _both_feeds(tmp_path, ref, recs, None, evc=True, retain=True, evc_minMQ=5, dcov=2)
    _same(a, b)
    a, b = _both_feeds(tmp_path, ref, recs, centers, region=(10, 45), payload=311)
    _same(a, b)


@pytest.mark.parametrize("threads,region", [(1, (None, None)), (7, (None, None)), (5, (9000, 21000))])
def test_bam_records_feed_equals_the_sam_text_feed_on_random_alignments(tmp_path, threads, region):
    """24 000 noisy reads, candidate extraction + retained tensor pass, depth cap 3: identical candidates, late marks,
    counters, tensors, depths and read counts whether the alignments arrive as text or as BAM records"""
    from clairvoyante_amd import synth_pileup as sp
    ref, lines = sp.make_alignments(seed=902, ref_len=40000, n_reads=24000, profile=sp.NOISY_PROFILE, stack=9, read_len=(40, 120))
    a, b = _both_feeds(tmp_path, ref, lines, None, region=region, evc=True, retain=True, dcov=3, minMQ=3, evc_minMQ=5,
                       threads=threads)
    _same(a, b)
    assert len(a[0]) > 300 and a[5][1].sum() > 0


def test_bam_records_feed_takes_long_cigars_from_the_cg_tag(tmp_path):
    """a read with 70 001 CIGAR operations (placeholder inline, real CIGAR in CG:B,I): same counts from the text view
    (which prints the real CIGAR) and from the raw records"""
    from test_bam_native import long_cigar_records
    recs = long_cigar_records()
    rng = np.random.RandomState(8)
    ref = "".join("ACGT"[i] for i in rng.randint(0, 4, 200000))
    centers = np.asarray([90, 100, 101, 117, 5000, 50000, 52600, 52700, 60000, 199990], dtype=np.int64)
    a, b = _both_feeds(tmp_path, ref, recs, centers, dcov=250)
    _same(a, b)
    assert a[3][:7].all() and a[1].sum() > 0
    a, b = _both_feeds(tmp_path, ref, recs, None, evc=True, retain=True, evc_minMQ=5)
    _same(a, b)
    assert len(a[0]) > 5          # minCoverage 3: only where the three reads overlap
