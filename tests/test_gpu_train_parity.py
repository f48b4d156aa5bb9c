"""The training step against the ORACLE at the sizes it is used and benchmarked at (SURVEY 8 a9, a10, a12;
clairvoyante_v3.py:140-152, 174, 183-227 at param.trainBatchSize = 10 000, train.py:95-102).

Above 160 (and again above 400) groups of 16 candidates the library chooses kernels and schedules by SIZE that the small oracle tests
(tests/test_gpu_dp.py n <= 1 000, tests/test_gpu_pipeline.py n <= 83) never reach: flat (group, row) ranges, the
two-group fc4 data gradient fused with conv3's unpool, the three-slab fc4 forward with the dropout on its store,
the tile unpool kernels, the second side stream, the two-groups-per-wave fc4 forward above 2 048 groups, several
slices above 65 536 candidates.  Here every one of them runs with DEFAULT options and is compared with
oracle/cv_oracle.c (cvo_loss_grad: OpenMP over candidates, double accumulators) under the keep mask the device
drew: the five loss parts, the dropout output, all 18 gradients, and weights + both Adam slots after the update;
getLoss (phase False: no dropout, lambda 0) on the same batch.
"""
import ctypes

import numpy as np
import pytest

import common

pytestmark = pytest.mark.gpu

LOSS_KEYS = ("loss1", "loss2", "loss3", "loss4", "lossL2")


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


def _split(flat, oracle, shapes):
    out, off = {}, 0
    for name in oracle.PARAM_NAMES:
        sz = int(np.prod(shapes[name]))
        out[name] = flat[off:off + sz].reshape(shapes[name]); off += sz
    assert off == flat.size
    return out


def compare_step(oracle, arch, n, rate, lam, lr=1e-3, seed=9, options=None, ksplit=None):
    """One m.train(x, y) with default options (or `options`) vs the oracle; returns the measured distances (for the
    logs).  ksplit: whether this step's fc4 forward runs as eight k ranges (default: what the library does at this size)."""
    x, y = _data(n, seed=seed)
    P = common.bench_params(oracle, arch)
    m = _model(arch); m.setParameters(P)
    for k, v in (options or {}).items():
        m.setOption(k, v)
    if ksplit is None:          # full: only in the small-batch regime (<= 400 groups); slim: at every size
        ksplit = arch == "slim" or (n + 15) // 16 <= 400
    m.dropoutRateFC4Val = rate; m.setL2RegularizationLambda(lam); m.setLearningRate(lr)
    m._dropout_seed = 4242
    # getLoss first (v3.py:207-216: phase False, dropout 0, lambda 0); it must not disturb the step that follows
    l_eval = float(m.getLoss(x, y))
    want_eval = oracle.loss_grad(arch, P, x, y, lam=0.0, want_grads=False)[0]
    assert abs(l_eval - want_eval) <= 1e-5 * abs(want_eval), (l_eval, want_eval)

    loss, summ = m.train(x, y)
    keep = None
    if rate > 0.0:
        amask = m.getActivation(6, n).cpu().numpy()          # (several slices: needs option keep_activations)
        keep = (amask != 0).astype(np.float32)
        assert abs(keep.mean() - (1.0 - rate)) <= 4 * 0.5 / np.sqrt(keep.size)
        d4 = m.getActivation(7, n).cpu().numpy()
        fa = oracle.forward_all(arch, P, x, mask4=keep, rate4=rate)
        d4_err = float(np.abs(d4 - fa["d4"]).max())
        # full: fc4 of a training pass above the tiny-batch range is the oracle's single ascending-k chain.  slim: its
        # fc4 forward runs as EIGHT k ranges added in order at every batch size (reproducible, never used by
        # cv_forward) -- the same sum in another fp32 order, 396 terms: a few 1e-6 of the largest value
        tol = 5e-6 if ksplit else 1e-6
        assert d4_err <= tol * max(1.0, float(np.abs(fa["d4"]).max())), d4_err
        del fa
    l_or, parts, g_or = oracle.loss_grad(arch, P, x, y, lam=lam, mask4=keep, rate4=rate)
    rel = {"loss": abs(float(loss) - l_or) / abs(l_or)}
    assert rel["loss"] <= 1e-5, (loss, l_or)
    for k, ref in zip(LOSS_KEYS, parts):
        rel[k] = abs(summ[k] - ref) / max(1.0, abs(ref))
        assert rel[k] <= 1e-5, (k, summ[k], ref)
    shapes = m.paramShapes()
    g_dev = _split(_flat(m, 1), oracle, shapes)
    w_dev, m_dev, v_dev = (_split(_flat(m, w), oracle, shapes) for w in (0, 2, 3))
    worst = 0.0
    for name in oracle.PARAM_NAMES:
        l2 = lam * P[name] if "bias" not in name else 0
        gref = g_or[name] - l2                                              # the bucket holds the data terms
        err = float(np.abs(g_dev[name] - gref).max() / (np.abs(gref).max() + 1e-30))
        worst = max(worst, err)
        # full: 2e-5 of the largest entry at train.py's batch; a gradient is an fp32 sum over all candidates of the batch,
        # whose rounding grows like the square root of the number of terms (measured worst: 3.7e-6 at 10 000, 6.6e-6 at
        # 40 010).  slim: 1e-4, the bound tests/test_gpu_dp.py uses for a k-split fc4 forward -- selu' jumps from 1.05 to
        # 1.76 at 0, so a pre-activation within rounding of 0 whose eight partial sums come out on the other side than the
        # oracle's single chain changes one candidate's contribution to a whole column of dW (seen at 20 000: one column
        # of fc4/kernel off by 4e-5 of the largest entry, every other entry at 1e-7)
        # The k-split applies wherever the library uses it: slim at every size, full in the tiny-batch regime; with it
        # switched off (option train_ksplit 0) slim is held to the full topology's bound -- the 1e-4 is the ORDER of the
        # fc4 sum, not the kernels (test_slim_as_a_single_chain_meets_the_tight_bound).
        tol_g = 1e-4 if ksplit else 2e-5 * max(1.0, np.sqrt(n / 10000.0))
        assert np.abs(g_dev[name] - gref).max() <= tol_g * np.abs(gref).max() + 1e-7, (name, err)
        # the optimizer on the DEVICE's gradient (+ lambda w): TF1 Adam, step 1 (v3.py:174)
        w = P[name].copy().ravel(); mm = np.zeros_like(w); vv = np.zeros_like(w)
        gfull = np.ascontiguousarray((g_dev[name] + l2).astype(np.float32).ravel())
        oracle.adam_step(w, mm, vv, gfull, lr, 1)
        assert np.abs(w_dev[name].ravel() - w).max() <= 1e-6, name
        assert np.abs(m_dev[name].ravel() - mm).max() <= 1e-6 * max(1.0, float(np.abs(mm).max())), name
        assert np.abs(v_dev[name].ravel() - vv).max() <= 1e-6 * max(1.0, float(np.abs(vv).max())), name
    m.close()
    rel["grad_worst_rel_to_max"] = worst
    return rel


@pytest.mark.parametrize("arch", ["full", "slim"])
@pytest.mark.parametrize("n", [320, 480, 1250, 2561, 5000, 6401, 10000, 20000, 40010])
def test_training_step_matches_oracle_at_training_sizes(oracle, arch, n):
    """320 / 480 = 20 / 30 groups: the two smallest workgroup shapes of fc4's data gradient (4 waves, 4 / 2 row parts;
    640 with this seed is a case of the k-split note in compare_step: one selu' flip, conv1/kernel off by 3e-4 of its
    largest entry in the eight-range order, 2.4e-7 as a single chain -- it runs in the single-chain test below);
    1 250 = a rank's share of train.py's batch on 8 GPUs (BASELINE config 4 as it runs: 79 groups, the tiny-batch
    kernel set); 2 561 = the first size past the position parts of the convolutions (161 groups, ragged last group: the
    rest of the small-batch kernel set on flat convolution ranges); 5 000 = a rank's share on 2 GPUs; 6 401 = the first size
    past the small-batch regime (401 groups, ragged); 10 000 = train.py's
    batch (the benchmarked step); 20 000 = two ranks' worth; 40 010 = 2 501 groups (> 2 048: the fc4 forward changes
    kernel), ragged.  The reference's training defaults: dropout 0.5 on fc4, lambda from param.py."""
    from clairvoyante_amd import param
    r = compare_step(oracle, arch, n, rate=param.dropoutRateFC4, lam=param.l2RegularizationLambda)
    print("train parity %s n=%d: %s" % (arch, n, {k: "%.2e" % v for k, v in r.items()}))


@pytest.mark.parametrize("arch,n", [("slim", 10000), ("slim", 1250), ("full", 1250), ("full", 640)])
def test_slim_as_a_single_chain_meets_the_tight_bound(oracle, arch, n):
    """Where the fc4 forward of a training pass runs as eight k ranges (slim always, full at tiny batches) the gradients
    are held to 1e-4 of the tensor maximum instead of 2e-5.  With option train_ksplit 0 the same kernels run the
    oracle's single ascending-k chain and meet the full topology's bound: the looser one is the summation ORDER (selu'
    jumps at 0 and a reordered sum can land on the other side), not the kernels."""
    from clairvoyante_amd import param
    r = compare_step(oracle, arch, n, rate=param.dropoutRateFC4, lam=param.l2RegularizationLambda,
                     options={"train_ksplit": 0}, ksplit=False)
    print("train parity %s n=%d, single chain: %s" % (arch, n, {k: "%.2e" % v for k, v in r.items()}))


@pytest.mark.parametrize("arch", ["full", "slim"])
@pytest.mark.parametrize("rate", [0.0, 0.5])
def test_training_step_over_several_slices_matches_oracle(oracle, arch, rate):
    """above 65 536 candidates a step runs as equal slices whose gradients and losses accumulate on the device; with
    dropout 0.5 the keep masks of BOTH slices are kept (option keep_activations) and handed to the oracle"""
    n = 70001
    r = compare_step(oracle, arch, n, rate=rate, lam=1e-3, options={"keep_activations": 1})
    print("train parity %s n=%d (2 slices, dropout %.1f): %s" % (arch, n, rate, {k: "%.2e" % v for k, v in r.items()}))


@pytest.mark.parametrize("arch", ["full", "slim"])
def test_random_batch_sizes_against_the_plain_kernels(oracle, arch):
    """The tile path picks its kernels and launch shapes by the batch size (position parts, flat ranges, k ranges of fc4,
    row parts of the data gradients, one- or several-kernel tails, side streams).  Sixteen seeded batch sizes, log-uniform
    over 1 .. 12 000 candidates: one optimizer step with the reference's dropout rate on the tile path and on the plain
    one-thread-per-output kernels (option impl 0: cv_kernels_ref.hip, the same dropout stream) -- loss within 1e-5,
    every gradient within 1e-4 of the bucket's largest entry (the two paths sum over the candidates in different orders);
    the tile path both with its default k-range fc4 forward (looser: the summation order) and as a single chain."""
    import torch
    from clairvoyante_amd import param, synth
    P = common.bench_params(oracle, arch)
    rng = np.random.RandomState(606)
    sizes = sorted(set(int(v) for v in np.exp(rng.uniform(0.0, np.log(12000.0), 16)).astype(np.int64)) | {1, 12000})
    worst = 0.0
    for n in sizes:
        xt, cls, rf, alt, il = synth.make_candidates(n, seed=1000 + n, device="cuda", return_class=True)
        y = synth.make_labels(cls, rf, alt, il)
        res = []
        for impl, ksplit in ((1, 1), (1, 0), (0, 1)):
            m = _model(arch); m.setParameters(P); m.setOption("impl", impl); m.setOption("train_ksplit", ksplit)
            m.dropoutRateFC4Val = param.dropoutRateFC4; m.setL2RegularizationLambda(param.l2RegularizationLambda)
            m.setLearningRate(1e-3); m._dropout_seed = 31
            loss, _ = m.train(xt, y)
            res.append((float(loss), _flat(m, 1)))
            m.close()
        (lk, gk), (l1, g1), (l0, g0) = res
        assert abs(l1 - l0) <= 1e-5 * abs(l0) and abs(lk - l0) <= 1e-5 * abs(l0), (n, lk, l1, l0)
        # fc4 as a single chain: the plain kernels' order of that sum -- 1e-4; as eight k ranges (the default of slim at every
        # size and of full up to 400 groups) an fc4 pre-activation within rounding of 0 may take the other selu' branch
        # (compare_step's note): 1e-3.  (This test found two one-in-1e7 events that are now fixed: an activation that rounds
        # to zero from below read as the x >= 0 branch by the tile path's selu'-from-output -- slim, 145 candidates, 7.5e-4 --
        # and the plain kernels' pooling backward comparing pre-activations where two of them an ulp apart share one
        # activation -- full, 1 250 candidates, 1.5e-4.)
        err = float(np.abs(g1 - g0).max() / (np.abs(g0).max() + 1e-30))
        errk = float(np.abs(gk - g0).max() / (np.abs(g0).max() + 1e-30))
        worst = max(worst, err)
        assert err <= 1e-4 and errk <= 1e-3, (n, err, errk)
    print("tile path vs plain kernels, %s, %d sizes: worst gradient distance %.2e of the largest entry" % (arch, len(sizes), worst))


@pytest.mark.parametrize("arch,n,cands", [("slim", 145, (45,)), ("full", 1250, (170, 303))])
def test_one_in_ten_million_elements_take_the_reference_branch(oracle, arch, n, cands):
    """Single candidates that hold one of two rare elements, found by the random-size test above in round 6:
    slim, candidate 45 of synth seed 1145: a conv2 pre-activation of -2.6e-8 -- exp rounds to 1, the activation to zero;
    selu' is scale*alpha there (selu.py:21-25: x >= 0 is false), which the tile path reads off the output's SIGN (-0.0);
    full, candidates 170 / 303 of seed 2250: two conv2 pre-activations one ulp apart inside a pooling window whose
    activations are the same float; the window's gradient goes to the FIRST of them (the pooling sees activations).
    Tile and plain kernels against the oracle, every gradient within 2e-5 of the bucket's largest entry."""
    from clairvoyante_amd import synth
    P = common.bench_params(oracle, arch)
    xt, cls, rf, alt, il = synth.make_candidates(n, seed=1000 + n, device="cuda", return_class=True)
    y = synth.make_labels(cls, rf, alt, il)
    for i in cands:
        x1, y1 = xt[i:i + 1].contiguous(), y[i:i + 1].contiguous()
        _, _, g_or = oracle.loss_grad(arch, P, x1.cpu().numpy(), y1.cpu().numpy(), lam=0.0)
        want = np.concatenate([g_or[name].ravel() for name in oracle.PARAM_NAMES])
        for impl in (1, 0):
            m = _model(arch); m.setParameters(P); m.setOption("impl", impl); m.setOption("train_ksplit", 0)
            m.dropoutRateFC4Val = 0.0; m.setL2RegularizationLambda(0.0); m.setLearningRate(1e-3)
            m.train(x1, y1)
            got = _flat(m, 1); m.close()
            off = 0
            for name in oracle.PARAM_NAMES:
                sz = g_or[name].size
                err = np.abs(got[off:off + sz] - want[off:off + sz]).max() / (np.abs(want[off:off + sz]).max() + 1e-30)
                assert err <= 2e-5, (arch, i, impl, name, err)
                off += sz


@pytest.mark.parametrize("arch", ["full", "slim"])
@pytest.mark.parametrize("n", [37, 1250])
def test_zero_biases_and_empty_positions_tie_everywhere(oracle, arch, n):
    """The state train.py starts from (v3.py:54-109: tf.layers' bias_initializer is zeros) on sparse pileups: every bias
    0, whole position ranges of a candidate empty, a tenth of the candidates entirely empty -- pre-activations of exactly
    0 (selu' = scale there: x >= 0) and pooling windows whose entries are all equal (the gradient goes to the first).
    One step with the reference's dropout rate, tile and plain kernels against the oracle under the device's keep mask:
    loss within 1e-5, every gradient within 2e-5 of the bucket's largest entry."""
    from clairvoyante_amd import param, synth
    P = {k: v.copy() for k, v in common.bench_params(oracle, arch).items()}
    for k in P:
        if "bias" in k:
            P[k][...] = 0.0
    rng = np.random.RandomState(n)
    xt, cls, rf, alt, il = synth.make_candidates(n, seed=77 + n, return_class=True)
    x = xt.numpy().copy(); y = synth.make_labels(cls, rf, alt, il).numpy()
    for i in range(n):
        lo = rng.randint(0, 33); hi = rng.randint(lo, 34)
        x[i, lo:hi] = 0.0
    x[rng.rand(n) < 0.1] = 0.0
    lam = param.l2RegularizationLambda
    for impl in (1, 0):
        m = _model(arch); m.setParameters(P); m.setOption("impl", impl); m.setOption("train_ksplit", 0)
        m.dropoutRateFC4Val = param.dropoutRateFC4; m.setL2RegularizationLambda(lam); m.setLearningRate(1e-3)
        m._dropout_seed = 5
        loss, _ = m.train(x, y)
        keep = (m.getActivation(6, n).cpu().numpy() != 0).astype(np.float32)
        got = _flat(m, 1); m.close()
        l_or, _, g_or = oracle.loss_grad(arch, P, x, y, lam=lam, mask4=keep, rate4=param.dropoutRateFC4)
        assert abs(float(loss) - l_or) <= 1e-5 * abs(l_or), (impl, loss, l_or)
        off = 0
        for name in oracle.PARAM_NAMES:
            sz = g_or[name].size
            gref = (g_or[name] - (lam * P[name] if "bias" not in name else 0)).ravel()
            err = np.abs(got[off:off + sz] - gref).max() / (np.abs(gref).max() + 1e-30)
            assert err <= 2e-5, (arch, n, impl, name, err)
            off += sz
