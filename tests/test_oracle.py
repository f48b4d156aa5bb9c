"""Pins the CPU oracle (oracle/cv_oracle.c) -- the reference ships no golden vectors for the
arithmetic, so the restatement is checked against an independently written
torch formulation of the same graph (tests/torch_ref.py) in float64, and against committed
fixtures of that formulation (tests/golden/forward_*.npz, made by make_golden_forward.py)."""
import os

import numpy as np
import pytest
import torch

import common
import torch_ref

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("arch", ["full", "slim"])
def test_forward_matches_torch_float64(oracle, arch):
    P = common.bench_params(oracle, arch)
    x = common.inputs(96, stress=8)
    got = oracle.forward_all(arch, P, x)
    ref = torch_ref.forward(arch, P, x)
    for k in ("pool1", "pool2", "pool3", "fc4", "fc5"):
        b = ref[k].numpy().reshape(got[k].shape)
        assert np.abs(got[k] - b).max() <= 2e-5 * max(1.0, np.abs(b).max()), k
    out = ref["out"].numpy()
    assert np.abs(got["out"] - out).max() <= 1e-5
    # argmax-exact wherever the float64 margin is not a rounding-level tie
    for lo, hi in common.HEADS:
        srt = np.sort(out[:, lo:hi], 1)
        clear = (srt[:, -1] - srt[:, -2]) > 1e-5
        assert np.array_equal(np.argmax(got["out"][clear, lo:hi], 1), np.argmax(out[clear, lo:hi], 1))


@pytest.mark.parametrize("arch", ["full", "slim"])
def test_forward_matches_committed_fixture(oracle, arch):
    d = np.load(os.path.join(G, "forward_%s.npz" % arch))
    P = common.bench_params(oracle, arch, seed=int(d["seed"]))
    got = oracle.forward_all(arch, P, d["x"].astype(np.float32))
    assert np.abs(got["out"] - d["out64"]).max() <= 1e-5
    k = d["pool3_64"].shape[0]
    assert np.abs(got["pool3"][:k] - d["pool3_64"]).max() <= 2e-5 * np.abs(d["pool3_64"]).max()
    assert np.abs(got["fc5"] - d["fc5_64"]).max() <= 2e-5 * max(1.0, np.abs(d["fc5_64"]).max())


def test_unscaled_reference_initialiser_weights(oracle):
    """raw reference initialiser (no input scaling): saturating regime, still within 1e-4"""
    P = oracle.init_params("full", seed=3)
    x = common.inputs(48)
    got = oracle.predict("full", P, x)
    ref = torch_ref.forward("full", P, x)["out"].numpy()
    assert np.abs(got - ref).max() <= 1e-4


def test_expf_selu_sigmoid_accuracy(oracle):
    xs = np.concatenate([np.linspace(-87, 88, 20001), [-100.0, 0.0, -0.0, 1e-8, -1e-8]]).astype(np.float32)
    got = np.array([oracle.expf(float(v)) for v in xs])
    want = np.exp(xs.astype(np.float64))
    nz = want > 1.2e-38
    rel = np.abs(got[nz] - want[nz]) / want[nz]
    assert rel.max() < 2.5e-7                      # <= ~2 ulp
    assert oracle.expf(-100.0) == 0.0 and oracle.expf(0.0) == 1.0
    lib = oracle.lib()
    import ctypes
    for v in (-3.0, -1e-3, 0.0, 2.5):
        s = lib.cvo_selu_scalar(ctypes.c_float(v))
        w = torch_ref.selu(torch.tensor(v, dtype=torch.float64)).item()
        assert abs(s - w) <= 2e-7 * max(1.0, abs(w))


@pytest.mark.parametrize("arch", ["full", "slim"])
def test_loss_and_gradients_match_torch_autograd(oracle, arch):
    from clairvoyante_amd import synth
    n = 24
    xt, cls, rf, alt, il = synth.make_candidates(n, seed=4, return_class=True)
    y = synth.make_labels(cls, rf, alt, il).numpy()
    x = xt.numpy()
    P = common.bench_params(oracle, arch)
    rng = np.random.RandomState(0)
    mask = (rng.uniform(size=(n, P["fc4/bias"].size)) < 0.5).astype(np.float32)
    lam = 0.01
    loss, parts, grads = oracle.loss_grad(arch, P, x, y, lam=lam, mask4=mask, rate4=0.5)
    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in P.items()}
    tl, tparts = torch_ref.loss(arch, tp, x, y, lam, mask4=mask, rate4=0.5)
    tl.backward()
    assert abs(loss - tl.item()) <= 1e-5 * abs(tl.item())
    for a, b in zip(parts, tparts):
        assert abs(a - float(b)) <= 1e-4 * max(1.0, abs(float(b)))
    for k in P:
        g = tp[k].grad.numpy()
        assert np.abs(grads[k] - g).max() <= 2e-4 * max(1e-6, np.abs(g).max()), k


def test_label_generator_encoding():
    """synthetic labels use the reference's 16-vector layout (utils_v2.py:90-117,142-147)"""
    from clairvoyante_amd import synth
    xt, cls, rf, alt, il = synth.make_candidates(500, seed=2, return_class=True)
    y = synth.make_labels(cls, rf, alt, il).numpy()
    assert np.allclose(y[:, 0:4].sum(1)[cls.numpy() < 3], 1.0)
    assert np.allclose(y[:, 4:6].sum(1), 1.0) and np.allclose(y[:, 6:10].sum(1), 1.0)
    assert np.allclose(y[:, 10:16].sum(1), 1.0)
    assert (y[cls.numpy() == 0, 6] == 1).all() and (y[cls.numpy() == 0, 5] == 1).all()


def test_adam_step_matches_tf_formula(oracle):
    rng = np.random.RandomState(1)
    w = rng.standard_normal(1000).astype(np.float32); g = rng.standard_normal(1000).astype(np.float32)
    m = np.zeros_like(w); v = np.zeros_like(w)
    w0 = w.copy()
    for t in (1, 2, 3):
        oracle.adam_step(w, m, v, g, 1e-3, t)
    m64 = np.zeros(1000); v64 = np.zeros(1000); w64 = w0.astype(np.float64)
    for t in (1, 2, 3):
        lr_t = 1e-3 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        m64 = 0.9 * m64 + 0.1 * g; v64 = 0.999 * v64 + 0.001 * g.astype(np.float64) ** 2
        w64 -= lr_t * m64 / (np.sqrt(v64) + 1e-8)
    assert np.abs(w - w64).max() < 1e-6


def test_selu_is_monotone_over_every_negative_float(oracle):
    """exhaustive: no pair of adjacent negative floats where the canonical SELU increases as x decreases
    (x >= 0 is a rounded multiplication by a positive constant).  Max-pooling therefore commutes with the
    activation bit for bit, max_j selu(a_j + b) == selu(max_j a_j + b), which the convolution kernels use to apply
    the activation once per POOLED row (tools/selu_monotone.c is the stand-alone form of this sweep)."""
    viol, chk = oracle.selu_sweep()
    assert viol == 0 and chk != 0
    # the seam between the two branches
    assert oracle.lib().cvo_selu_scalar(np.float32(-1e-45)) <= oracle.lib().cvo_selu_scalar(np.float32(0.0))
    x = -np.abs(np.random.RandomState(0).standard_normal(4000).astype(np.float32) * 20)
    x.sort()
    y = np.array([oracle.lib().cvo_selu_scalar(v) for v in x], dtype=np.float32)
    assert (np.diff(y) >= 0).all()


def test_seeded_weights_of_the_product_are_the_oracle_initialiser(oracle):
    """bench.py and smoke() draw their weights from clairvoyante_amd/synth.py (nothing from oracle/ inside the timed
    function); it is the same seeded stream as the oracle's own initialiser, so fixtures made with either agree"""
    from clairvoyante_amd import synth
    for arch in ("full", "slim"):
        assert synth.param_shapes(arch) == oracle.param_shapes(arch)
        for seed in (0, 1, 7):
            a = oracle.init_params(arch, seed=seed, bias_scale=0.05)
            b = synth.seeded_params(arch, seed=seed, bias_scale=0.05)
            assert list(a) == list(b) == oracle.PARAM_NAMES
            assert all(np.array_equal(a[k], b[k]) for k in a)


def test_loss_grad_does_not_depend_on_the_thread_count(oracle):
    """cvo_loss_grad walks contiguous candidate ranges on OpenMP threads with double accumulators added in thread order:
    the float results are the single-thread ones (tests/test_gpu_train_parity.py leans on it at 10 000+ candidates)"""
    import subprocess
    import sys
    code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import common\nfrom oracle import cv_oracle as O\nfrom clairvoyante_amd import synth\n"
            "xt, c, r, a, l = synth.make_candidates(300, seed=3, return_class=True)\n"
            "y = synth.make_labels(c, r, a, l).numpy(); P = common.bench_params(O, 'slim')\n"
            "L, parts, g = O.loss_grad('slim', P, xt.numpy(), y, lam=0.01)\n"
            "np.save(sys.argv[1], np.concatenate([g[k].ravel() for k in O.PARAM_NAMES] + [np.float32(parts)]))\n") % (
                ROOT, os.path.join(ROOT, "tests"))
    outs = []
    for nt in ("1", "5"):
        fn = os.path.join(os.environ.get("TMPDIR", "/tmp"), "cvo_lg_%s_%d.npy" % (nt, os.getpid()))
        subprocess.check_call([sys.executable, "-c", code, fn], env=dict(os.environ, OMP_NUM_THREADS=nt))
        outs.append(np.load(fn)); os.remove(fn)
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))
