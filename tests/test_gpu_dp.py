"""GPU tests of the training step as the data-parallel host drives it:

* alpha-dropout (selu.py:34-69 applied at clairvoyante_v3.py:111): the keep mask the device drew is
  exported (cv_get_activation layer 6) and fed to the oracle -- forward values, loss parts and all 18 gradients of
  the DEFAULT training configuration (rate 0.5) are compared, full + slim;
* init() really resets the optimizer (v3.py:177);
* the step on two streams (weight gradients on the side stream) gives the same bits as on one;
* deferred losses (no host round trip per step) sum to the per-step values;
* the real data-parallel step with TWO ranks on the one GPU of the box (backend gloo: RCCL refuses two ranks on
  one device): broadcast from deliberately different weights, model.train on each rank's shard_range of a global
  batch through parallel.exchange_bucket, compared with one process on the whole batch; then train.run_epoch.
"""
import ctypes
import os
import pickle
import sys

import numpy as np
import pytest

import common

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(arch):
    from clairvoyante_amd import clairvoyante_v3, clairvoyante_v3_slim
    return clairvoyante_v3.Clairvoyante() if arch == "full" else clairvoyante_v3_slim.Clairvoyante()


def _flat(m, which):
    import torch
    from clairvoyante_amd import _lib
    t = torch.empty(m.numParameters, device="cuda")
    _lib.check(m._lib.cv_flat_copy(m._h, which, ctypes.c_void_p(t.data_ptr()), 0, None))
    torch.cuda.synchronize()
    return t.cpu().numpy().copy()


def _data(n, seed):
    from clairvoyante_amd import synth
    xt, cls, rf, alt, il = synth.make_candidates(n, seed=seed, return_class=True)
    return xt.numpy(), synth.make_labels(cls, rf, alt, il).numpy()


ALPHA_P = np.float32(-1.7580993408473766)


def _affine(rate):
    q = np.float32(1.0) - np.float32(rate)
    a = np.float32(np.sqrt(np.float32(1.0) / (q * ((np.float32(1.0) - q) * (ALPHA_P * ALPHA_P) + np.float32(1.0)))))
    return a


@pytest.mark.parametrize("arch", ["full", "slim"])
@pytest.mark.parametrize("n,ksplit", [(17, 0), (80, 0), (1000, 0), (1000, 1)])
def test_alpha_dropout_forward_and_backward_match_oracle(oracle, arch, n, ksplit):
    """ksplit 0 (default): fc4 of the training pass is the oracle's single ascending-k chain (tight bounds); 1 (option
    train_ksplit): eight partial sums added in order -- the same values up to fp32 summation order"""
    rate, lam = 0.5, 0.01
    x, y = _data(n, seed=9)
    P = common.bench_params(oracle, arch)
    m = _model(arch); m.setParameters(P)
    m.setOption("train_ksplit", ksplit)
    tol_d4, tol_g = (1e-6, 2e-5) if ksplit == 0 else (5e-5, 1e-4)
    m.dropoutRateFC4Val = rate; m.setL2RegularizationLambda(lam); m.setLearningRate(1e-3)
    m._dropout_seed = 777
    loss, summ = m.train(x, y)
    amask = m.getActivation(6, n).cpu().numpy()
    d4 = m.getActivation(7, n).cpu().numpy()
    a = _affine(rate)
    assert set(np.unique(amask)).issubset({np.float32(0.0), a}), np.unique(amask)[:5]
    keep = (amask != 0).astype(np.float32)
    # the mask is a fair coin per (candidate, unit)
    N = keep.size
    assert abs(keep.mean() - 0.5) <= 4 * 0.5 / np.sqrt(N)
    if n > 1:
        assert len({r.tobytes() for r in keep}) == n                # no two candidates share a mask
    # forward: dropout4 of the oracle under the device's mask
    fa = oracle.forward_all(arch, P, x, mask4=keep, rate4=rate)
    assert np.abs(d4 - fa["d4"]).max() <= tol_d4
    # loss parts and gradients
    l_or, parts, g_or = oracle.loss_grad(arch, P, x, y, lam=lam, mask4=keep, rate4=rate)
    assert abs(loss - l_or) <= 1e-5 * abs(l_or)
    for k, ref in zip(("loss1", "loss2", "loss3", "loss4", "lossL2"), parts):
        assert abs(summ[k] - ref) <= 1e-5 * max(1.0, abs(ref)), k
    gb = _flat(m, 1)
    off = 0
    for name in oracle.PARAM_NAMES:
        sz = g_or[name].size
        g = gb[off:off + sz].reshape(g_or[name].shape); off += sz
        gref = g_or[name] - (lam * P[name] if "bias" not in name else 0)     # data terms only
        assert np.abs(g - gref).max() <= tol_g * np.abs(gref).max() + 1e-7, name
    # the mask changes with the step and with the seed (each rank draws its own seed from os.urandom)
    m.train(x, y)
    keep2 = (m.getActivation(6, n).cpu().numpy() != 0)
    assert 0.35 < (keep2 == (keep != 0)).mean() < 0.65
    m2 = _model(arch); m2.setParameters(P); m2.dropoutRateFC4Val = rate; m2._dropout_seed = 778
    m2.train(x, y)
    keep3 = (m2.getActivation(6, n).cpu().numpy() != 0)
    assert 0.35 < (keep3 == (keep != 0)).mean() < 0.65
    f1 = _model(arch); f2 = _model(arch)
    assert f1._dropout_seed != f2._dropout_seed                        # fresh models (ranks): fresh seeds
    f1.close(); f2.close()
    # rate 0: identity, mask of ones
    m2.dropoutRateFC4Val = 0.0
    m2.train(x, y)
    assert np.array_equal(m2.getActivation(6, n).cpu().numpy(), np.ones((n, amask.shape[1]), np.float32))
    m.close(); m2.close()


@pytest.mark.parametrize("arch", ["full", "slim"])
def test_init_after_training_resets_the_optimizer(oracle, arch):
    """tf.global_variables_initializer also re-initialises "<var>/Adam", "<var>/Adam_1" and the beta powers
    (v3.py:177): a model that has trained and is initialised again takes the same first step as a fresh one"""
    x, y = _data(300, seed=3)
    P = common.bench_params(oracle, arch)
    P2 = common.bench_params(oracle, arch, seed=2)

    def first_step(m):
        m.setParameters(P); m._dropout_seed = 5; m._train_step = 0
        m.setLearningRate(1e-3)
        loss, _ = m.train(x, y)
        return float(loss), _flat(m, 0), _flat(m, 2), _flat(m, 3)
    fresh = _model(arch)
    want = first_step(fresh)
    used = _model(arch); used.setParameters(P2); used.setLearningRate(1e-3)
    for _ in range(3):
        used.train(x, y)
    assert used._adam_t == 3 and np.abs(_flat(used, 2)).max() > 0
    used.init()
    assert used._adam_t == 0 and not _flat(used, 2).any() and not _flat(used, 3).any()
    got = first_step(used)
    assert got[0] == want[0]
    for u, v in zip(got[1:], want[1:]):
        assert np.array_equal(u.view(np.uint32), v.view(np.uint32))
    fresh.close(); used.close()


@pytest.mark.parametrize("arch,n", [("full", 10000), ("slim", 10000), ("full", 1250), ("full", 70000)])
def test_side_stream_weight_gradients_give_the_same_bits(oracle, arch, n):
    """option train_overlap: the weight-gradient kernels on the side stream next to the data-gradient chain, or
    everything in stream order -- same kernels, same operands, same order among the weight gradients"""
    import torch
    from clairvoyante_amd import synth
    xt, cls, rf, alt, il = synth.make_candidates(n, seed=41, device="cuda", return_class=True)
    y = synth.make_labels(cls, rf, alt, il)
    P = common.bench_params(oracle, arch)

    def run(overlap):
        m = _model(arch); m.setParameters(P); m.setOption("train_overlap", overlap)
        m._dropout_seed = 99; m.setLearningRate(1e-3)
        losses = [float(m.train(xt, y)[0]) for _ in range(3)]
        out = (losses, _flat(m, 0), _flat(m, 1))
        m.close()
        return out
    a = run(1); b = run(0)
    assert a[0] == b[0]
    assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
    assert np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))
    assert np.isfinite(a[1]).all()


@pytest.mark.parametrize("arch,n", [("full", 1250), ("slim", 1250), ("full", 83), ("full", 2560), ("slim", 2560), ("full", 17),
                                    ("full", 5000), ("slim", 5000), ("full", 6400)])
def test_tiny_batch_kernel_variants_give_the_same_bits(oracle, arch, n):
    """batches of few groups (a rank's share of train.py's batch) split the serial loops of the training step over
    more waves: position ranges in the convolutions (pooled layers recompute the window overlap), one thread per
    row in the pooling backward pass, all weight packing in one launch -- row for row the same arithmetic, so with
    the k-split of fc4 switched off the step must equal the regular kernels bit for bit; with it on, the fc4
    pre-activations are eight partial sums added in order and the step stays within rounding of it"""
    import torch
    from clairvoyante_amd import synth
    # only sizes where the switch changes the path: above 400 groups both settings run the regular kernels (those sizes
    # are compared with the oracle in tests/test_gpu_train_parity.py).  Up to 80 groups the whole small-batch set runs;
    # from 161 to 400 everything but the position parts of the convolutions and the chained join.
    assert (n + 15) // 16 <= 400
    xt, cls, rf, alt, il = synth.make_candidates(n, seed=43, device="cuda", return_class=True)
    y = synth.make_labels(cls, rf, alt, il)
    P = common.bench_params(oracle, arch)

    def run(tiny_groups, ksplit, **opts):
        m = _model(arch); m.setParameters(P)
        m.setOption("train_tiny_groups", tiny_groups); m.setOption("train_ksplit", ksplit)
        for k, v in opts.items():
            m.setOption(k, v)
        m._dropout_seed = 99; m.setLearningRate(1e-3); m.setL2RegularizationLambda(1e-3)
        losses = [float(m.train(xt, y)[0]) for _ in range(2)]
        out = (losses, _flat(m, 0), _flat(m, 1), float(m.getLoss(xt, y)))
        m.close()
        return out
    regular = run(0, 0)
    tiny = run(400, 0)
    assert regular[0] == tiny[0] and regular[3] == tiny[3]
    assert np.array_equal(regular[1].view(np.uint32), tiny[1].view(np.uint32))
    assert np.array_equal(regular[2].view(np.uint32), tiny[2].view(np.uint32))
    ks = run(400, 1)
    assert np.allclose(regular[0], ks[0], rtol=1e-6, atol=0)
    gmax = np.abs(regular[2]).max()
    assert np.abs(regular[2] - ks[2]).max() <= 1e-4 * gmax
    assert np.isfinite(ks[1]).all()
    # full topology: everything behind fc4's k ranges as ONE kernel (train_tail_tm) or as three (dbg2 = 5): the same
    # arithmetic per value -- weights and gradients bit for bit; the loss sums leave as one row per group instead of one
    # per four groups, so the reported loss may differ in its last bits
    three = run(400, 1, dbg2=5)
    assert np.array_equal(ks[1].view(np.uint32), three[1].view(np.uint32))
    assert np.array_equal(ks[2].view(np.uint32), three[2].view(np.uint32))
    assert np.allclose(ks[0], three[0], rtol=1e-12, atol=0) and abs(ks[3] - three[3]) <= 1e-12 * abs(three[3])


@pytest.mark.parametrize("arch,n", [("full", 1250), ("full", 83), ("full", 640), ("full", 10000), ("slim", 1250), ("slim", 10000)])
def test_backward_kernel_variants_give_the_same_bits(oracle, arch, n):
    """development switches of the backward pass (full topology): fc4's data gradient fused with conv3's unpool or as
    two kernels (dbg3), one or three side streams for the weight gradients, weight packing on
    the side stream or in stream order (dbg5); slim: the selu' factor of a layer without pooling on the data-gradient
    kernel's store or as its own pass (dbg4 = 3); the convolution kernels over position parts of whole groups (dbg0 = dbg1 = 9)
    fc4's alpha-dropout on the store of its forward kernel or as its own pass (dbg2 = 3);
    or over equal ranges of the flat (group, row) sequence (dbg0 = dbg1 = 7: also at small batches, where the parts are the
    default): the same arithmetic in the same order -- same losses, weights, gradients"""
    import torch
    from clairvoyante_amd import synth
    xt, cls, rf, alt, il = synth.make_candidates(n, seed=47, device="cuda", return_class=True)
    y = synth.make_labels(cls, rf, alt, il)
    P = common.bench_params(oracle, arch)

    def run(opts):
        m = _model(arch); m.setParameters(P)
        for k, v in opts.items():
            m.setOption(k, v)
        m._dropout_seed = 99; m.setLearningRate(1e-3); m.setL2RegularizationLambda(1e-3)
        losses = [float(m.train(xt, y)[0]) for _ in range(2)]
        out = (losses, _flat(m, 0), _flat(m, 1))
        m.close()
        return out
    ref = run({})
    # (dbg5 = 8 / 4: fc4's weight-gradient kernel with one / two input fragments per wave over the same candidate ranges)
    # (round 5: dbg2 = 4 the thread-per-row unpool at tiny batches instead of row segments; option train_sched: the bits of
    # the re-cut schedule -- early loss header, conv1's weight gradient on the main stream, one fork marker, shared launch-site
    # markers, per-layout packing -- switched off in groups)
    # (round 6: dbg6 = row parts of fc4's data gradient + unpool, + 100 = 4-wave workgroups; the default picks them by the
    # number of groups -- 83 candidates: 4 waves x 4 parts, 640: 4 waves x 2, 1 250: 8 waves x 2, larger: 8 waves x 1)
    variants = ({"dbg3": 1}, {"train_side_streams": 1}, {"dbg5": 1}, {"dbg6": 3}, {"dbg6": 1}, {"dbg6": 104}, {"dbg6": 102}, {"dbg7": 1},
                {"dbg0": 9, "dbg1": 9}, {"dbg0": 7, "dbg1": 7}, {"dbg1": 8}, {"dbg2": 3}, {"dbg3": 1, "train_overlap": 0},
                {"dbg2": 4}, {"train_sched": 0}, {"train_sched": 254}, {"train_sched": 21}, {"train_sched": 42}, {"train_sched": 223}, {"train_sched": 191}, {"train_sched": 127}, {"train_sched": 511}, {"train_sched": 255}, {"train_sched": 1023}, {"train_sched": 1791}, {"dbg4": 4}, {"dbg2": 1}, {"dbg2": 2}, {"dbg5": 8}, {"dbg5": 4}) if arch == "full" else \
               ({"dbg4": 3}, {"dbg4": 9}, {"train_side_streams": 1}, {"dbg0": 9, "dbg1": 9}, {"dbg0": 7, "dbg1": 7},
                {"dbg5": 1, "dbg4": 3, "train_overlap": 0}, {"train_sched": 0}, {"train_sched": 21}, {"train_sched": 42}, {"train_sched": 223}, {"train_sched": 191}, {"train_sched": 127}, {"train_sched": 255}, {"train_sched": 1791})
    # (batches above the tiny range: fc5 + heads + losses + head gradients are one kernel behind fc4's by default, train_sched
    # bit 10; its loss sums leave as one row per group instead of one per four, so a variant that switches it off -- every
    # explicit train_sched value here -- may differ from the default in the last bits of the reported loss)
    big = arch == "full" and n > 2560
    if big:
        variants = variants + ({"train_sched": 767},)
    for opts in variants:
        got = run(opts)
        if big and "train_sched" in opts and not (opts["train_sched"] & 1024):
            assert np.allclose(ref[0], got[0], rtol=1e-12, atol=0), opts
        else:
            assert ref[0] == got[0], opts
        assert np.array_equal(ref[1].view(np.uint32), got[1].view(np.uint32)), opts
        assert np.array_equal(ref[2].view(np.uint32), got[2].view(np.uint32)), opts


@pytest.mark.parametrize("arch", ["full", "slim"])
def test_packed_layouts_follow_the_weights_through_mixed_passes(oracle, arch):
    """A pass packs only the MFMA layouts of the weights that ITS kernels read (a training step of 79 groups reads two of
    the four fc4 layouts; an inference pass of 65 536 candidates another one) and leaves the rest stale until a pass
    that reads them.  Through a sequence that mixes training steps, predict() and getLoss() at sizes that switch
    kernels, every pass of the long-lived model must give the bits of a FRESH model that was handed its current
    weights -- a stale layout would show as different outputs / gradients (or as the library's own "layout is stale"
    error).  Dropout off, so a step's gradient is a function of weights and batch alone."""
    import torch
    from clairvoyante_amd import synth
    P = common.bench_params(oracle, arch)
    used = _model(arch); used.setParameters(P)
    used.dropoutRateFC4Val = 0.0; used.setLearningRate(1e-3); used.setL2RegularizationLambda(1e-3)
    plan = [("train", 1250), ("predict", 1000), ("train", 10000), ("predict", 40000), ("train", 40010), ("loss", 3000),
            ("train", 83), ("predict", 300), ("train", 2561), ("predict", 65536), ("train", 1250)]
    data = {}
    for op, n in plan:
        if n not in data:
            xt, cls, rf, alt, il = synth.make_candidates(n, seed=100 + n, device="cuda", return_class=True)
            data[n] = (xt, synth.make_labels(cls, rf, alt, il))
        x, y = data[n]
        fresh = _model(arch); fresh.setParameters(used.getParameters())
        fresh.dropoutRateFC4Val = 0.0; fresh.setLearningRate(1e-3); fresh.setL2RegularizationLambda(1e-3)
        try:
            if op == "predict":
                a = used.predict_device(x).cpu().numpy(); b = fresh.predict_device(x).cpu().numpy()
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (op, n)
            elif op == "loss":
                assert float(used.getLoss(x, y)) == float(fresh.getLoss(x, y)), (op, n)
            else:
                la, lb = float(used.train(x, y)[0]), float(fresh.train(x, y)[0])
                assert la == lb, (op, n)
                assert np.array_equal(_flat(used, 1).view(np.uint32), _flat(fresh, 1).view(np.uint32)), (op, n)
        finally:
            fresh.close()
    used.close()


@pytest.mark.parametrize("arch,n", [("full", 1250), ("full", 10000), ("slim", 3000), ("full", 70001)])
def test_a_step_writes_every_gradient_element(oracle, arch, n):
    """the step does not zero its gradient bucket (the second passes of the weight gradients STORE for the first slice of
    a step): with the bucket full of NaN beforehand every element must come out finite and equal to the step that starts
    from a zeroed bucket (train_sched bit 6 off) -- one slice, and two (the second one adds)"""
    import torch
    from clairvoyante_amd import synth
    xt, cls, rf, alt, il = synth.make_candidates(n, seed=49, device="cuda", return_class=True)
    y = synth.make_labels(cls, rf, alt, il)
    P = common.bench_params(oracle, arch)
    out = []
    for sched in (3839, 3775):
        m = _model(arch); m.setParameters(P); m.setOption("train_sched", sched)
        m._dropout_seed = 7; m.setLearningRate(1e-3); m.setL2RegularizationLambda(1e-3)
        m._ensure_bucket().fill_(float("nan"))
        loss = float(m.train(xt, y)[0])
        g = _flat(m, 1)
        assert np.isfinite(g).all() and np.isfinite(loss)
        out.append((loss, g, _flat(m, 0)))
        m.close()
    assert out[0][0] == out[1][0]
    assert np.array_equal(out[0][1].view(np.uint32), out[1][1].view(np.uint32))
    assert np.array_equal(out[0][2].view(np.uint32), out[1][2].view(np.uint32))


def test_deferred_losses_sum_to_the_per_step_losses(oracle):
    x, y = _data(2000, seed=13)
    P = common.bench_params(oracle, "slim")
    a = _model("slim"); b = _model("slim")
    for m in (a, b):
        m.setParameters(P); m._dropout_seed = 4; m.setLearningRate(1e-3); m.setL2RegularizationLambda(1e-3)
    per_step = [a.train(x, y)[1] for _ in range(5)]
    for _ in range(5):
        b.trainDeferred(x, y)
    l, steps = b.readLosses()
    assert steps == 5
    for i, k in enumerate(("loss1", "loss2", "loss3", "loss4", "lossL2", "loss")):
        want = sum(s[k] for s in per_step)
        assert abs(l[i] - want) <= 1e-9 * abs(want), k
    assert b.readLosses() == ([0.0] * 6, 0)
    # mixing: a synchronous step in between does not lose what the deferred ones accumulated
    b.trainDeferred(x, y)
    one = b.train(x, y)[1]["loss"]
    l2, s2 = b.readLosses()
    assert s2 == 1 and abs(l2[5] - a.train(x, y)[1]["loss"]) <= 1e-9 * abs(l2[5])
    assert abs(one - a.train(x, y)[1]["loss"]) <= 1e-9 * abs(one)
    assert np.array_equal(_flat(a, 0).view(np.uint32), _flat(b, 0).view(np.uint32))
    a.close(); b.close()


# ---- two ranks on one GPU ---------------------------------------------------------------------------------

def _dp_worker(rank, ws, port, tmp, arch, n, steps, rate):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(ws),
                      LOCAL_RANK="0")
    import torch
    from clairvoyante_amd import parallel
    from oracle import cv_oracle as O
    import common as C
    torch.cuda.set_device(0)
    r, w, _ = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, ws)
    m = _model(arch)
    # deliberately different weights and optimizer state per rank: the broadcast must replace them
    m.setParameters(C.bench_params(O, arch, seed=1 + rank))
    m.train(*_data(64, seed=50))                # a (collective) step: non-zero Adam slots everywhere
    m.setParameters(C.bench_params(O, arch, seed=1 + rank))
    if rank == 0:
        m._zero_adam()                          # rank 0 holds the state every rank must end up with
    else:
        assert _flat(m, 2).any()
    parallel.broadcast_parameters(m)
    assert np.array_equal(_flat(m, 0), np.concatenate([C.bench_params(O, arch, seed=1)[k].ravel() for k in O.PARAM_NAMES]))
    assert not _flat(m, 2).any() and not _flat(m, 3).any()
    m._adam_t = 0
    m.dropoutRateFC4Val = rate; m.setLearningRate(1e-3); m.setL2RegularizationLambda(0.01)
    x, y = _data(n, seed=60)
    lo, hi = parallel.shard_range(n, rank, ws)
    losses = []
    g1 = None
    for s in range(steps):
        loss, summ = m.train(x[lo:hi], y[lo:hi])
        losses.append([summ[k] for k in ("loss1", "loss2", "loss3", "loss4", "lossL2", "loss")])
        if s == 0:
            g1 = _flat(m, 1)          # the exchanged gradient of the FIRST step: same weights as the single-process run
    g = _flat(m, 1)
    np.savez(os.path.join(tmp, "rank%d.npz" % rank), w=_flat(m, 0), am=_flat(m, 2), av=_flat(m, 3), g=g, g1=g1,
             losses=np.asarray(losses))
    torch.distributed.barrier()
    m.close()
    torch.distributed.destroy_process_group()


def _spawn(fn, args, nprocs=2):
    import torch.multiprocessing as mp
    port = 29900 + os.getpid() % 500
    mp.spawn(fn, args=(nprocs, port) + tuple(args), nprocs=nprocs, join=True)


@pytest.mark.parametrize("arch,n,ws", [("full", 10000, 2), ("slim", 4001, 2), ("full", 10000, 8), ("slim", 4001, 3)])
def test_data_parallel_step_equals_the_single_process_step(oracle, arch, n, ws, tmp_path):
    """("full", 10000, 8) is BASELINE config 4 as it runs: train.py's batch of 10 000 split over 8 ranks = 1 250
    candidates (79 groups, ragged) per rank, here as 8 processes on the one GPU (backend gloo)"""
    steps = 3
    _spawn(_dp_worker, (str(tmp_path), arch, n, steps, 0.0), nprocs=ws)
    r0 = np.load(str(tmp_path / "rank0.npz"))
    # identical replicas: every rank applied the same update to the same weights
    for r in range(1, ws):
        r1 = np.load(str(tmp_path / ("rank%d.npz" % r)))
        for k in ("w", "am", "av", "g"):
            assert np.array_equal(r0[k].view(np.uint32), r1[k].view(np.uint32)), (k, r)
        assert np.array_equal(r0["losses"], r1["losses"])
    # one process on the whole batch
    m = _model(arch); m.setParameters(common.bench_params(oracle, arch, seed=1))
    m.dropoutRateFC4Val = 0.0; m.setLearningRate(1e-3); m.setL2RegularizationLambda(0.01)
    x, y = _data(n, seed=60)
    losses = []
    g1 = None
    for s in range(steps):
        loss, summ = m.train(x, y)
        losses.append([summ[k] for k in ("loss1", "loss2", "loss3", "loss4", "lossL2", "loss")])
        if s == 0:
            g1 = _flat(m, 1)
    assert np.allclose(r0["losses"][0], np.asarray(losses)[0], rtol=2e-6)       # first step: identical weights
    assert np.allclose(r0["losses"], np.asarray(losses), rtol=1e-4)             # later steps: weights equal to rounding
    g = _flat(m, 1); w = _flat(m, 0); am = _flat(m, 2); av = _flat(m, 3)
    off = 0
    P = common.bench_params(oracle, arch, seed=1)
    for name in oracle.PARAM_NAMES:
        sz = P[name].size
        sl = slice(off, off + sz); off += sz
        # the sum of the shard gradients against the full-batch gradient at the SAME weights: summation order only
        # (shards of up to 400 groups -- 1 250 on 8 ranks, 5 000 on 2 -- run the small-batch step, whose fc4 forward is
        # eight k ranges added in order, against ONE ascending-k chain for the whole batch: 1e-4, the bound
        # test_alpha_dropout_* uses for that variant; measured 2.9e-5 on fc4/kernel with two shards of 5 000)
        ksplit_shard = arch == "full" and ((n + ws - 1) // ws + 15) // 16 <= 400 < (n + 15) // 16
        assert np.abs(r0["g1"][sl] - g1[sl]).max() <= (1e-4 if ksplit_shard else 2e-5) * np.abs(g1[sl]).max() + 1e-7, name
        # after three updates the weights differ in their last bits (Adam divides by sqrt(v)): looser
        assert np.abs(r0["g"][sl] - g[sl]).max() <= 1e-3 * np.abs(g[sl]).max() + 1e-6, name
        assert np.abs(r0["am"][sl] - am[sl]).max() <= 1e-3 * np.abs(am[sl]).max() + 1e-9, name
        assert np.abs(r0["av"][sl] - av[sl]).max() <= 2e-3 * np.abs(av[sl]).max() + 1e-12, name
    # weights: Adam divides by sqrt(v), so an element whose gradient is pure rounding noise may move differently;
    # all but a vanishing fraction agree to 1e-6, none differs by more than the three steps can move it
    dw = np.abs(r0["w"] - w)
    assert (dw <= 1e-6).mean() >= 0.999, (dw <= 1e-6).mean()
    assert dw.max() <= 2 * steps * 1e-3
    assert np.abs(w - np.concatenate([P[k].ravel() for k in oracle.PARAM_NAMES])).max() > 1e-4     # it did train
    m.close()


@pytest.mark.parametrize("ws,n", [(2, 1), (8, 3)])
def test_step_with_empty_shards(oracle, tmp_path, ws, n):
    """a global batch smaller than the rank count leaves ranks without candidates: they still take part in the
    exchange (zero gradient, zero losses) and every rank applies the same update"""
    _spawn(_dp_worker, (str(tmp_path), "slim", n, 2, 0.0), nprocs=ws)
    r0 = np.load(str(tmp_path / "rank0.npz"))
    for r in range(1, ws):
        r1 = np.load(str(tmp_path / ("rank%d.npz" % r)))
        for k in ("w", "am", "av", "g"):
            assert np.array_equal(r0[k].view(np.uint32), r1[k].view(np.uint32)), (k, r)
    m = _model("slim"); m.setParameters(common.bench_params(oracle, "slim", seed=1))
    m.dropoutRateFC4Val = 0.0; m.setLearningRate(1e-3); m.setL2RegularizationLambda(0.01)
    x, y = _data(n, seed=60)
    loss, summ = m.train(x, y)
    assert abs(r0["losses"][0][5] - summ["loss"]) <= 1e-6 * abs(summ["loss"])
    assert np.abs(r0["g1"] - _flat(m, 1)).max() <= 1e-6 * np.abs(r0["g1"]).max()
    m.close()


def test_gradient_bucket_binding_is_checked(oracle):
    import ctypes
    import torch
    from clairvoyante_amd import _lib
    m = _model("slim")
    cnt = ctypes.c_int64(); hdr = ctypes.c_int64(); dense = ctypes.c_int64()
    _lib.check(m._lib.cv_grad_bucket_info(m._h, ctypes.byref(cnt), ctypes.byref(hdr), ctypes.byref(dense)))
    assert cnt.value == m.numParameters + hdr.value and hdr.value == 16 and hdr.value < dense.value < cnt.value
    t = torch.zeros(cnt.value + 4, device="cuda")
    with pytest.raises(_lib.CvError, match="floats"):
        _lib.check(m._lib.cv_bind_grad_bucket(m._h, ctypes.c_void_p(t.data_ptr()), cnt.value - 1))
    with pytest.raises(_lib.CvError, match="aligned"):
        _lib.check(m._lib.cv_bind_grad_bucket(m._h, ctypes.c_void_p(t.data_ptr() + 4), cnt.value))
    _lib.check(m._lib.cv_bind_grad_bucket(m._h, ctypes.c_void_p(t.data_ptr()), cnt.value))
    x, y = _data(40, seed=2)
    m.setParameters(common.bench_params(oracle, "slim"))
    _lib.check(m._lib.cv_bind_grad_bucket(m._h, None, 0))           # back to the library's own bucket
    loss, _ = m.train(x, y)                                          # (the model binds its own tensor on first use)
    assert np.isfinite(loss) and float(m.gradients().abs().max()) > 0
    assert float(t.abs().max()) == 0.0                               # the unbound tensor was never written
    m.close()


def test_two_rank_replicas_stay_identical_with_dropout(oracle, tmp_path):
    """each rank draws its own dropout stream; the exchanged gradient is what both apply"""
    _spawn(_dp_worker, (str(tmp_path), "slim", 3000, 4, 0.5))
    r0 = np.load(str(tmp_path / "rank0.npz")); r1 = np.load(str(tmp_path / "rank1.npz"))
    for k in ("w", "am", "av", "g"):
        assert np.array_equal(r0[k].view(np.uint32), r1[k].view(np.uint32)), k
    assert np.isfinite(r0["w"]).all() and np.array_equal(r0["losses"], r1["losses"])


def _epoch_worker(rank, ws, port, tmp, binfn, n):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(ws),
                      LOCAL_RANK="0")
    import torch
    from clairvoyante_amd import param, parallel, train, utils_v2
    from oracle import cv_oracle as O
    import common as C
    torch.cuda.set_device(0)
    parallel.init_from_env(backend="gloo")
    with open(binfn, "rb") as fh:
        total = pickle.load(fh); XC = pickle.load(fh); YC = pickle.load(fh)
    m = _model("slim"); m.setParameters(C.bench_params(O, "slim", seed=1 + rank))
    parallel.broadcast_parameters(m)
    m.dropoutRateFC4Val = 0.0; m.setLearningRate(1e-3); m.setL2RegularizationLambda(1e-3)
    vstart = int(total * param.trainingDatasetPercentage) + 1
    stream = train._BatchStream(utils_v2, XC, YC, total, vstart, rank, ws)
    sums = [train.run_epoch(stream, m, rank, ws, None, e, vstart) for e in (1, 2)]
    np.savez(os.path.join(tmp, "epoch%d.npz" % rank), sums=np.asarray(sums, dtype=np.float64), w=_flat(m, 0))
    torch.distributed.barrier()
    m.close()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("ws", [2, 8])
def test_data_parallel_epoch_equals_the_single_process_epoch(oracle, tmp_path, monkeypatch, ws):
    """train.run_epoch under two / eight ranks: every rank walks the same schedule on its slice of each batch; training sums
    come out of the exchanged loss header, validation sums through parallel.allreduce_scalar (train.py:113-123)"""
    from clairvoyante_amd import param, train, utils_v2
    n = 26000
    x, y = _data(n, seed=71)
    y = y.astype(np.float64)
    XC = [utils_v2.pack_array(x[s:s + 500]) for s in range(0, n + 1, 500)]
    YC = [utils_v2.pack_array(y[s:s + 500]) for s in range(0, n + 1, 500)]
    binfn = str(tmp_path / "dp.bin")
    with open(binfn, "wb") as fh:
        pickle.dump(n, fh); pickle.dump(XC, fh); pickle.dump(YC, fh); pickle.dump([], fh)
    _spawn(_epoch_worker, (str(tmp_path), binfn, n), nprocs=ws)
    e0 = np.load(str(tmp_path / "epoch0.npz"))
    for r in range(1, ws):
        e1 = np.load(str(tmp_path / ("epoch%d.npz" % r)))
        assert np.array_equal(e0["sums"], e1["sums"]) and np.array_equal(e0["w"].view(np.uint32), e1["w"].view(np.uint32))
    m = _model("slim"); m.setParameters(common.bench_params(oracle, "slim", seed=1))
    m.dropoutRateFC4Val = 0.0; m.setLearningRate(1e-3); m.setL2RegularizationLambda(1e-3)
    vstart = int(n * param.trainingDatasetPercentage) + 1
    stream = train._BatchStream(utils_v2, XC, YC, n, vstart)
    sums = np.asarray([train.run_epoch(stream, m, 0, 1, None, e, vstart) for e in (1, 2)], dtype=np.float64)
    assert np.allclose(e0["sums"], sums, rtol=1e-4), (e0["sums"], sums)
    assert sums[1, 0] < sums[0, 0]                     # the second epoch starts from trained weights
    m.close()


@pytest.mark.parametrize("arch,flags", [("full", []), ("slim", ["--slim"])])
def test_callvar_command_line_under_two_ranks_writes_the_single_rank_vcf(oracle, arch, flags, tmp_path):
    """BASELINE configs[2] through the CLI: `torchrun ... -m clairvoyante_amd.callVar` shards the input lines over
    the ranks (here: two processes on the one GPU, backend gloo) and rank 0 joins the record fragments; the VCF must
    be byte-identical to the single-process run of the same command"""
    import subprocess
    import test_gpu_pipeline as tp
    P = common.bench_params(oracle, arch)
    m = _model(arch); m.setParameters(P)
    prefix = str(tmp_path / "model")
    m.saveParameters(prefix); m.close()
    x = common.inputs(3000, seed=17)
    tfn = str(tmp_path / "tensors.gz")
    tp._write_text_tensors(tfn, x)
    base = [sys.executable, "-m", "clairvoyante_amd.callVar", "--chkpnt_fn", prefix, "--tensor_fn", tfn,
            "--sampleName", "NA12878", "--qual", "30"] + flags
    env = dict(os.environ, PYTHONPATH=ROOT)
    one = str(tmp_path / "one.vcf")
    subprocess.check_call(base + ["--call_fn", one], env=env, cwd=ROOT)
    two = str(tmp_path / "two.vcf")
    port = 29400 + os.getpid() % 500
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                 CV_DIST_BACKEND="gloo", CV_SHARD_BLOCK_LINES="256")
        procs.append(subprocess.Popen(base + ["--call_fn", two], env=e, cwd=ROOT))
    assert [p.wait(timeout=600) for p in procs] == [0, 0]
    a, b = open(one).read(), open(two).read()
    assert a == b and a.count("\n") > 200
    assert not [f for f in os.listdir(str(tmp_path)) if ".rank" in f]


@pytest.mark.parametrize("ws,sizes", [(2, (1500, 20000, 700)), (8, (1500, 20000, 700, 90, 0, 2500, 33, 1200, 640, 17, 3100))])
def test_callvar_command_line_under_several_ranks_over_a_list_of_files(oracle, tmp_path, ws, sizes):
    """--tensor_fn a.gz,b.gz,c.gz,... under two / eight ranks: file k belongs to rank k % ws -- the form that scales (no
    rank inflates input it does not call; the reference's own recipe is one job per chunk, README.md:184-202); the VCF
    equals the concatenation of the single-process runs (bodies in list order).  Eight ranks over 11 files: ranks with one
    and with two files, an empty file."""
    import subprocess
    import test_gpu_pipeline as tp
    P = common.bench_params(oracle, "full")
    m = _model("full"); m.setParameters(P)
    prefix = str(tmp_path / "model")
    m.saveParameters(prefix); m.close()
    files = []
    for k, n in enumerate(sizes):           # the second file spans two reader batches of 16 384 rows
        fn = str(tmp_path / ("t%d.gz" % k))
        tp._write_text_tensors(fn, common.inputs(max(n, 1), seed=50 + k)[:n])
        files.append(fn)
    env = dict(os.environ, PYTHONPATH=ROOT)
    base = [sys.executable, "-m", "clairvoyante_amd.callVar", "--chkpnt_fn", prefix, "--sampleName", "NA12878"]
    bodies, header = [], None
    for k, fn in enumerate(files):
        out = str(tmp_path / ("one%d.vcf" % k))
        subprocess.check_call(base + ["--tensor_fn", fn, "--call_fn", out], env=env, cwd=ROOT)
        lines = open(out).read().splitlines(True)
        header = [l for l in lines if l.startswith("#")]
        bodies += [l for l in lines if not l.startswith("#")]
    two = str(tmp_path / "two.vcf")
    port = 29300 + os.getpid() % 500
    procs = []
    for r in range(ws):
        e = dict(env, RANK=str(r), WORLD_SIZE=str(ws), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                 CV_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen(base + ["--tensor_fn", ",".join(files), "--call_fn", two], env=e, cwd=ROOT))
    assert [p.wait(timeout=600) for p in procs] == [0] * ws
    assert open(two).read() == "".join(header + bodies) and len(bodies) > 500


_RCCL_ONE_RANK = r"""
import os, sys, ctypes
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["CV_ROOT"]); sys.path.insert(0, os.path.join(os.environ["CV_ROOT"], "tests"))
import common
from oracle import cv_oracle as O
from clairvoyante_amd import parallel, clairvoyante_v3, synth, _lib
rank, ws, local = parallel.init_from_env()
assert dist.is_initialized() and dist.get_backend() == "nccl" and parallel._active()
m = clairvoyante_v3.Clairvoyante(); m.setParameters(common.bench_params(O, "full"))
parallel.broadcast_parameters(m)
assert parallel.comm_stream(m) is not None
m._dropout_seed = 99; m.setLearningRate(1e-3); m.setL2RegularizationLambda(1e-3)
xt, cls, rf, alt, il = synth.make_candidates(3000, seed=41, device="cuda", return_class=True)
y = synth.make_labels(cls, rf, alt, il)
losses = [float(m.train(xt, y)[0]) for _ in range(3)]
for _ in range(2):
    m.trainDeferred(xt, y)
acc, steps = m.readLosses()
t = torch.empty(m.numParameters, device="cuda")
_lib.check(m._lib.cv_flat_copy(m._h, 0, ctypes.c_void_p(t.data_ptr()), 0, None)); torch.cuda.synchronize()
np.savez(os.environ["CV_OUT"], w=t.cpu().numpy(), losses=np.asarray(losses), acc=np.asarray(acc), steps=steps,
         one=parallel.allreduce_scalar(1.5, m))
dist.barrier(); dist.destroy_process_group()
"""


def test_rccl_code_path_with_one_rank_equals_the_plain_step(oracle, tmp_path):
    """The RCCL branch of the exchange -- communication stream behind the 'dense gradients final' event, two in-place
    asynchronous all-reduces of the bucket, Adam behind both, the loss header divided by the rank count -- run for
    real through backend nccl with ONE rank (CV_FORCE_DIST=1; the box has one GPU and RCCL refuses two ranks per
    device): a sum over one rank is the identity, so weights and losses must equal the non-distributed step bit for
    bit."""
    import subprocess
    import torch
    from clairvoyante_amd import synth
    out = str(tmp_path / "rccl.npz")
    env = dict(os.environ, CV_FORCE_DIST="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(29300 + os.getpid() % 500), CV_ROOT=ROOT, CV_OUT=out, PYTHONPATH=ROOT)
    env.pop("CV_DIST_BACKEND", None)
    subprocess.check_call([sys.executable, "-c", _RCCL_ONE_RANK], env=env, cwd=ROOT)
    got = np.load(out)
    m = _model("full"); m.setParameters(common.bench_params(oracle, "full"))
    m._dropout_seed = 99; m.setLearningRate(1e-3); m.setL2RegularizationLambda(1e-3)
    xt, cls, rf, alt, il = synth.make_candidates(3000, seed=41, device="cuda", return_class=True)
    y = synth.make_labels(cls, rf, alt, il)
    losses = [float(m.train(xt, y)[0]) for _ in range(3)]
    for _ in range(2):
        m.trainDeferred(xt, y)
    acc, steps = m.readLosses()
    assert np.array_equal(got["w"].view(np.uint32), _flat(m, 0).view(np.uint32))
    assert list(got["losses"]) == losses
    assert int(got["steps"]) == steps == 2 and np.allclose(got["acc"], acc, rtol=1e-9, atol=0)
    assert float(got["one"]) == 1.5
    m.close()


def test_bench_line_under_two_ranks_sharing_the_gpu(tmp_path):
    """`python bench.py --gpus 2` end to end with two real ranks (backend gloo, both on the one GPU: CV_SHARE_DEVICES): it
    starts its own ranks, rank 0 prints one line with n_gpus = rccl_ranks = 2, the slim leg and the training legs (global
    batch 10 000 split over the ranks, and 10 000 per rank) -- the flow the driver runs at N > 1, timings aside"""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(CV_SHARE_DEVICES="1", CV_DIST_BACKEND="gloo", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--no-cpu"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    b = lines[0]
    assert b["n_gpus"] == 2 and b["rccl_ranks"] == 2 and b["backend"] == "gloo" and b["value"] > 1e6
    assert b["slim"]["value"] > 1e6
    assert set(b["train"]) == {"10000", "10000_per_rank", "slim_10000"}
    assert b["train"]["10000"]["per_rank_batch"] == 5000 and b["train"]["10000_per_rank"]["per_rank_batch"] == 10000
    assert all(np.isfinite(v["final_loss"]) for v in b["train"].values())


@pytest.mark.parametrize("mode", ["infer", "train", "exchange"])
def test_bench_line_under_eight_ranks_sharing_the_gpu(mode):
    """the driver's 8-GPU command, functionally: `python bench.py --gpus 8` with eight real ranks on the one GPU (backend
    gloo, CV_SHARE_DEVICES) -- one line, n_gpus = rccl_ranks = 8; in train mode the line separates the exchange from the
    compute (exchange_ms, compute_ms_per_step, exchange_hidden_frac)"""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(CV_SHARE_DEVICES="1", CV_DIST_BACKEND="gloo", PYTHONPATH=ROOT)
    extra = ["--mode", mode] if mode != "infer" else ["--no-extras"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--no-cpu"] + extra,
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    b = lines[0]
    assert b["n_gpus"] == 8 and b["rccl_ranks"] == 8 and b["backend"] == "gloo"
    if mode == "exchange":       # only the bucket all-reduce: whole bucket, its two pieces, a fifth; time + bandwidths per piece
        assert b["unit"] == "GB/s" and b["value"] > 0 and len(b["pieces"]) == 4
        assert b["pieces"][0]["bytes"] == 4 * (1631496 + 16) and b["pieces"][1]["bytes"] + b["pieces"][2]["bytes"] == b["pieces"][0]["bytes"]
        assert all(p["ms"] > 0 and abs(p["busbw_GBps"] - p["algbw_GBps"] * 2 * 7 / 8) < 1e-9 * max(1.0, p["busbw_GBps"]) for p in b["pieces"])
        return
    assert b["value"] > 1e3     # functional: 8 ranks share one GPU, gloo through the host
    if mode == "train":
        assert b["config"]["global_batch"] == 10000 and b["scaling"] == "strong"
        assert b["exchange_ms"] > 0 and b["compute_ms_per_step"] > 0 and 0.0 <= b["exchange_hidden_frac"] <= 1.0
        assert b["exchange_bytes"] == 4 * (1631496 + 16) and np.isfinite(b["final_loss"])
        # the step's own exchange as bandwidths (what the first run on real links is read by), and the rank check that ran
        # before anything was timed: eight ranks counted, here on ONE device, flagged as shared
        assert abs(b["exchange_algbw_GBps"] - b["exchange_bytes"] / (b["exchange_ms"] * 1e-3) / 1e9) <= 1e-6 * b["exchange_algbw_GBps"]
        assert abs(b["exchange_busbw_GBps"] - b["exchange_algbw_GBps"] * 2 * 7 / 8) <= 1e-9 * max(1.0, b["exchange_busbw_GBps"])
        assert b["counted_ranks"] == 8 and b["distinct_devices"] == 1 and b["shared_devices"] is True
        assert b["exchange_plan"] == "one"          # 1 250 candidates per rank: one collective per step
        assert b["other_plan"]["plan"] == "split" and b["other_plan"]["ms_per_step"] > 0      # ... and the same steps under the other plan
    else:
        assert b["scaling"] == "weak" and b["config"]["batch"] == 65536
        pr = b["per_rank_ms"]                       # a straggler would show: slowest / fastest rank, ms per step
        assert 0 < pr["min"] <= pr["max"] and 0 <= pr["argmax_rank"] < 8 and abs(pr["max"] - b["ms_per_step"]) < 1e-6 * pr["max"]
