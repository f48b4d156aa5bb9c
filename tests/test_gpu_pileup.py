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
    assert res["stats"]["columns"] > 1000 and res["stats"]["launches"] >= 1


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
