"""GPU tests of the drivers around the network: checkpoint round trip, the callVar command
line end to end (BASELINE.json configs[0]: text tensors + checkpoint -> VCF), training step
parity with the oracle, and the train command line on a small .bin."""
import ctypes
import gzip
import io
import os
import pickle
import subprocess
import sys
import types

import numpy as np
import pytest

import common

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(arch):
    from clairvoyante_amd import clairvoyante_v3, clairvoyante_v3_slim
    return clairvoyante_v3.Clairvoyante() if arch == "full" else clairvoyante_v3_slim.Clairvoyante()


def _write_text_tensors(path, x_sub, seed=3, bad_every=97):
    """CreateTensor format: raw counts (matrix 0 added back), '%0.1f' (CreateTensor.py:24,56)"""
    rng = np.random.RandomState(seed)
    raw = x_sub.copy()
    for i in range(1, 4):
        raw[:, :, :, i] += raw[:, :, :, 0]
    with gzip.open(path, "wt") as f:
        for j in range(raw.shape[0]):
            seq = "".join(rng.choice(list("ACGT"), 33))
            if j % bad_every == 5:
                seq = seq[:16] + "N" + seq[17:]
            f.write("%s %d %s %s\n" % ("chr%d" % (1 + j % 4), 10000 + 7 * j, seq,
                                       " ".join("%0.1f" % v for v in raw[j].reshape(-1))))


@pytest.mark.parametrize("arch", ["full", "slim"])
def test_checkpoint_round_trip(oracle, arch, tmp_path):
    P = common.bench_params(oracle, arch)
    a = _model(arch); a.setParameters(P); a._adam_t = 7
    prefix = str(tmp_path / "ckpt" / "model-000012")
    a.saveParameters(prefix)
    for sfx in (".index", ".data-00000-of-00001", ".meta"):
        assert os.path.exists(prefix + sfx)
    b = _model(arch); b.init(); b.restoreParameters(prefix)
    for k, v in P.items():
        assert np.array_equal(b.getParameter(k), v), k
    assert b._adam_t == 7
    x = common.inputs(64)
    assert all(np.array_equal(u, w) for u, w in zip(a.predict(x), b.predict(x)))
    a.close(); b.close()


@pytest.mark.parametrize("arch,flags", [("full", []), ("slim", ["--slim"])])
def test_callvar_command_line_end_to_end(oracle, arch, flags, tmp_path):
    from clairvoyante_amd import callVar, utils_v2
    P = common.bench_params(oracle, arch)
    m = _model(arch); m.setParameters(P)
    prefix = str(tmp_path / "model")
    m.saveParameters(prefix); m.close()
    x = common.inputs(3000, seed=17)
    tfn = str(tmp_path / "tensors.gz")
    _write_text_tensors(tfn, x)
    out = str(tmp_path / "calls.vcf")
    env = dict(os.environ, PYTHONPATH=ROOT)
    subprocess.check_call([sys.executable, "-m", "clairvoyante_amd.callVar", "--chkpnt_fn", prefix, "--tensor_fn", tfn,
                           "--call_fn", out, "--sampleName", "NA12878", "--qual", "30"] + flags, env=env, cwd=ROOT)
    # expected: reference-validated host formatter on ORACLE predictions of the parsed tensors
    args = types.SimpleNamespace(v2=False, v3=True, showRef=False, qual=30, ref_fn=None, sampleName="NA12878")
    fh = io.StringIO()
    callVar.PrintVCFHeader(args, fh)
    nrec = 0
    for end, c, xb, pos in utils_v2.GetTensor(tfn, 1000, log=False):
        o = oracle.predict(arch, P, xb)
        callVar.Output(args, fh, c, xb, pos, o[:, 0:4], o[:, 4:6], o[:, 6:10], o[:, 10:16])
        nrec += c
    assert nrec == 3000 - len([j for j in range(3000) if j % 97 == 5])
    got = open(out).read().splitlines()
    want = fh.getvalue().splitlines()
    assert len(want) > 100 and got == want


@pytest.mark.parametrize("arch,n", [("full", 80), ("slim", 80), ("full", 17), ("slim", 1), ("full", 83)])
def test_gradients_and_adam_match_oracle(oracle, arch, n):
    """n not a multiple of 16: the last group of 16 candidates is ragged (padded lanes carry no gradient)"""
    import torch
    from clairvoyante_amd import _lib, synth
    xt, cls, rf, alt, il = synth.make_candidates(n, seed=9, return_class=True)
    y = synth.make_labels(cls, rf, alt, il).numpy(); x = xt.numpy()
    P = common.bench_params(oracle, arch)
    m = _model(arch); m.setParameters(P)
    lam = 0.01
    l_or, parts, g_or = oracle.loss_grad(arch, P, x, y, lam=lam)
    assert abs(float(m.getLoss(x, y)) - oracle.loss_grad(arch, P, x, y, lam=0.0, want_grads=False)[0]) <= 1e-4 * l_or
    m.dropoutRateFC4Val = 0.0; m.setL2RegularizationLambda(lam); m.setLearningRate(1e-3)
    loss, summ = m.train(x, y)
    assert abs(loss - l_or) <= 1e-5 * abs(l_or)
    for k, ref in zip(("loss1", "loss2", "loss3", "loss4", "lossL2"), parts):
        assert abs(summ[k] - ref) <= 1e-4 * max(1.0, abs(ref))
    gb = torch.empty(m.numParameters, device="cuda")
    _lib.check(m._lib.cv_flat_copy(m._h, 1, ctypes.c_void_p(gb.data_ptr()), 0, None))
    gb = gb.cpu().numpy()
    off = 0
    for name in oracle.PARAM_NAMES:
        sz = g_or[name].size
        g = gb[off:off + sz].reshape(g_or[name].shape); off += sz
        gref = g_or[name] - (lam * P[name] if "bias" not in name else 0)     # data terms only
        assert np.abs(g - gref).max() <= 2e-5 * np.abs(gref).max() + 1e-7, name
        w = P[name].copy().ravel(); mm = np.zeros_like(w); vv = np.zeros_like(w)
        # the optimizer kernel on ITS OWN gradient (+ lambda*w for kernels): TF1 Adam, step 1
        gfull = (g + (lam * P[name] if "bias" not in name else 0)).astype(np.float32)
        oracle.adam_step(w, mm, vv, np.ascontiguousarray(gfull.ravel()), 1e-3, 1)
        assert np.abs(m.getParameter(name).ravel() - w).max() <= 1e-6, name
    m.close()


def test_training_reduces_the_loss_and_dropout_runs(oracle):
    from clairvoyante_amd import synth
    n = 2000
    xt, cls, rf, alt, il = synth.make_candidates(n, seed=12, return_class=True)
    y = synth.make_labels(cls, rf, alt, il).numpy(); x = xt.numpy()
    m = _model("slim"); m.init()
    m.setLearningRate(1e-3); m.setL2RegularizationLambda(1e-3)
    first = float(m.getLoss(x, y))
    for _ in range(40):
        loss, _s = m.train(x, y)
        assert np.isfinite(loss)
    last = float(m.getLoss(x, y))
    assert last < 0.7 * first
    m.close()


def test_train_command_line_on_a_small_bin(oracle, tmp_path, monkeypatch):
    """train.Run on a .bin built by GetTrainingArray-compatible blocks; epochs capped through param"""
    from clairvoyante_amd import param, synth, train, utils_v2
    n = 2600
    xt, cls, rf, alt, il = synth.make_candidates(n, seed=21, return_class=True)
    y = synth.make_labels(cls, rf, alt, il).numpy().astype(np.float64); x = xt.numpy()
    XC = [utils_v2.pack_array(x[s:s + 500]) for s in range(0, n + 1, 500)]
    YC = [utils_v2.pack_array(y[s:s + 500]) for s in range(0, n + 1, 500)]
    binfn = str(tmp_path / "mini.bin")
    with open(binfn, "wb") as fh:
        pickle.dump(n, fh); pickle.dump(XC, fh); pickle.dump(YC, fh); pickle.dump([], fh)
    monkeypatch.setattr(param, "maxEpoch", 4)
    prefix = str(tmp_path / "out" / "model")
    args = types.SimpleNamespace(bin_fn=binfn, tensor_fn=None, var_fn=None, bed_fn=None, chkpnt_fn=None,
                                 learning_rate=1e-3, lambd=1e-3, ochk_prefix=prefix, olog_dir=str(tmp_path / "log"),
                                 v2=False, v3=True, slim=True)
    train.Run(args)
    for e in (1, 2, 3):
        assert os.path.exists("%s-%06d.index" % (prefix, e))
    assert os.path.getsize(str(tmp_path / "log" / "scalars.tsv")) > 0
    # resume from the epoch-3 checkpoint: the epoch counter continues (train.py:82)
    monkeypatch.setattr(param, "maxEpoch", 6)
    args.chkpnt_fn = "%s-%06d" % (prefix, 3)
    train.Run(args)
    assert os.path.exists("%s-%06d.index" % (prefix, 5)) and not os.path.exists("%s-%06d.index" % (prefix, 6))


def test_evaluate_report_on_a_small_bin(oracle, tmp_path, caplog):
    """evaluate.Run: predictions over a .bin and the confusion-matrix report (evaluate.py:37-107)"""
    import logging
    from clairvoyante_amd import evaluate, synth, utils_v2
    n = 1700
    xt, cls, rf, alt, il = synth.make_candidates(n, seed=31, return_class=True)
    y = synth.make_labels(cls, rf, alt, il).numpy().astype(np.float64); x = xt.numpy()
    XC = [utils_v2.pack_array(x[s:s + 500]) for s in range(0, n + 1, 500)]
    YC = [utils_v2.pack_array(y[s:s + 500]) for s in range(0, n + 1, 500)]
    binfn = str(tmp_path / "e.bin")
    with open(binfn, "wb") as fh:
        pickle.dump(n, fh); pickle.dump(XC, fh); pickle.dump(YC, fh); pickle.dump([], fh)
    P = common.bench_params(oracle, "slim")
    m = _model("slim"); m.setParameters(P)
    prefix = str(tmp_path / "model"); m.saveParameters(prefix); m.close()
    args = types.SimpleNamespace(bin_fn=binfn, tensor_fn=None, var_fn=None, bed_fn=None, chkpnt_fn=prefix,
                                 v2=False, v3=True, slim=True)
    with caplog.at_level(logging.INFO):
        evaluate.Run(args)
    text = caplog.text
    o = oracle.predict("slim", P, x)
    order = np.argsort(o[:, 0:4], axis=1, kind="stable")[:, ::-1]
    truth = np.argmax(y[:, 0:4], axis=1)
    top1 = int(np.sum(order[:, 0] == truth))
    assert "Dataset size: %d" % n in text
    assert "all/top1/top2/top1p/top2p: %d/%d/" % (n, top1) in text


@pytest.mark.gpu
def test_training_reduces_the_loss_and_keeps_predictions_finite():
    """functional check of the whole step (forward, backward, Adam, dropout): 100 steps on one labelled batch bring
    the loss far down and the network then predicts the labels of that batch (at 60 steps the type head is still on
    its way -- 0.90 to 0.99 depending on the dropout stream; the stream is seeded here)"""
    import torch
    from clairvoyante_amd import clairvoyante_v3, synth
    m = clairvoyante_v3.Clairvoyante()
    m._seed_rng.seed(7)
    m.init()
    m.setLearningRate(1e-3)
    m._dropout_seed = 77
    xt, cls, rf, alt, il = synth.make_candidates(2000, seed=11, device="cuda", return_class=True)
    y = synth.make_labels(cls, rf, alt, il)
    first = m.getLoss(xt, y)
    for _ in range(100):
        loss, _ = m.train(xt, y)
    last = m.getLoss(xt, y)
    assert np.isfinite(first) and np.isfinite(last) and last < 0.25 * first
    out = m.predict_device(xt).cpu().numpy()
    assert np.isfinite(out).all()
    yh = y.cpu().numpy()
    zyg = (out[:, 4:6].argmax(1) == yh[:, 4:6].argmax(1)).mean()
    typ = (out[:, 6:10].argmax(1) == yh[:, 6:10].argmax(1)).mean()
    assert zyg > 0.9 and typ > 0.9
    m.close()


@pytest.mark.gpu
def test_loss_and_gradient_are_additive_over_the_batch_at_scale(oracle):
    """size-independent property: the loss is a SUM over candidates (v3.py:140-151), so loss and data gradients of
    40 000 candidates equal those of two halves added up.  (The comparison of steps of this size with the ORACLE is
    tests/test_gpu_train_parity.py; this one only checks additivity, which any per-candidate error would pass.)"""
    import torch
    from clairvoyante_amd import _lib, synth
    m = _model("full")
    m.setParameters(common.bench_params(oracle, "full"))
    m.dropoutRateFC4Val = 0.0
    m.setLearningRate(0.0)              # the Adam update must not move the weights between the calls
    m.setL2RegularizationLambda(0.0)
    n = 40000
    xt, cls, rf, alt, il = synth.make_candidates(n, seed=21, device="cuda", return_class=True)
    y = synth.make_labels(cls, rf, alt, il)

    def grad_of(lo, hi):
        loss, _ = m.train(xt[lo:hi].contiguous(), y[lo:hi].contiguous())
        g = torch.empty(m.numParameters, device="cuda")
        _lib.check(m._lib.cv_flat_copy(m._h, 1, ctypes.c_void_p(g.data_ptr()), 0, None))
        return float(loss), g.double()
    before = m.getParameter("fc4/kernel").copy()
    l_all, g_all = grad_of(0, n)
    l_a, g_a = grad_of(0, n // 2)
    l_b, g_b = grad_of(n // 2, n)
    assert np.array_equal(before, m.getParameter("fc4/kernel"))
    assert abs(l_all - (l_a + l_b)) <= 1e-5 * abs(l_all)
    err = (g_all - (g_a + g_b)).abs().max().item()
    assert err <= 2e-5 * g_all.abs().max().item()
    assert g_all.abs().max().item() > 0
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("arch", ["full", "slim"])
def test_training_is_reproducible_bit_for_bit(arch):
    """every weight gradient is reduced in a fixed order (per-range tiles summed by a second pass, no float
    atomics), and dropout is a counter-based hash of (seed, step, candidate, unit): two runs from the same seed
    end in the same weights, bit for bit -- on train.py's batch of 10 000 (several candidate ranges per kernel)"""
    import torch
    from clairvoyante_amd import _lib, synth
    xt, cls, rf, alt, il = synth.make_candidates(10000, seed=31, device="cuda", return_class=True)
    y = synth.make_labels(cls, rf, alt, il)

    def run():
        m = _model(arch)
        m._seed_rng.seed(5)
        m._dropout_seed = 12345          # the reference's dropout is unseeded; the mirror draws its seed from os.urandom
        m.init()
        m.setLearningRate(1e-3)
        losses = [float(m.train(xt, y)[0]) for _ in range(4)]
        w = torch.empty(m.numParameters, device="cuda")
        _lib.check(m._lib.cv_flat_copy(m._h, 0, ctypes.c_void_p(w.data_ptr()), 0, None))
        w = w.cpu().numpy().copy()
        m.close()
        return losses, w
    l1, w1 = run()
    l2, w2 = run()
    assert np.isfinite(w1).all() and l1[-1] < l1[0]
    assert np.array_equal(w1.view(np.uint32), w2.view(np.uint32))
    assert l1 == l2      # the loss sums too: block rows added in a fixed order (heads_train_tm, t_loss_header), no atomics


def test_epoch_sums_do_not_depend_on_the_validation_pass_size(tmp_path):
    """run_epoch hands a real model validation passes of 16 000 candidates (device batches) where the reference asks
    for 1 000 at a time (train.py:95-102); the loss is a sum over candidates, so both ways of walking the same
    epoch give the same training and validation sums (learning rate 0: the weights do not move in between)"""
    from clairvoyante_amd import param, synth, train, utils_v2
    n = 40000
    xt, cls, rf, alt, il = synth.make_candidates(n, seed=23, device="cuda", return_class=True)
    y = synth.make_labels(cls, rf, alt, il).cpu().numpy().astype(np.float64); x = xt.cpu().numpy()
    XC = [utils_v2.pack_array(x[s:s + 500]) for s in range(0, n + 1, 500)]
    YC = [utils_v2.pack_array(y[s:s + 500]) for s in range(0, n + 1, 500)]
    vstart = int(n * param.trainingDatasetPercentage) + 1
    m = _model("full")
    m._seed_rng.seed(3)
    m.init()
    m.setLearningRate(0.0)
    m.dropoutRateFC4Val = 0.0

    class HostOnly(object):              # the same model without the device-batch capability: the reference's sequence
        accepts_device_batches = False

        def __init__(self, inner):
            self.inner, self.sizes = inner, []

        def trainNoRT(self, X, Y):
            self.inner.trainNoRT(X, Y); self.trainLossRTVal = self.inner.trainLossRTVal
            self.trainSummaryRTVal = self.inner.trainSummaryRTVal

        def getLossNoRT(self, X, Y):
            self.sizes.append(len(X))
            self.inner.getLossNoRT(X, Y); self.getLossLossRTVal = self.inner.getLossLossRTVal

        def getLoss(self, X, Y):
            self.sizes.append(len(X))
            return self.inner.getLoss(X, Y)

    stream = train._BatchStream(utils_v2, XC, YC, n, vstart)
    t_dev, v_dev = train.run_epoch(stream, m, 0, 1, None, 1, vstart)
    host = HostOnly(m)
    t_ref, v_ref = train.run_epoch(stream, host, 0, 1, None, 1, vstart)
    # the clipped batch that ends exactly at validationStart is evaluated, not trained (train.py:104); then 1 000s
    assert max(host.sizes[1:]) <= param.predictBatchSize and len(host.sizes) >= 4
    assert abs(t_dev - t_ref) <= 1e-6 * abs(t_ref) and abs(v_dev - v_ref) <= 1e-6 * abs(v_ref)
    assert v_ref > 0 and t_ref > 0
    m.close()
