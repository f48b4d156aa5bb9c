"""train.TrainAll against the reference's own loop (golden: tests/golden/train_schedule.json,
recorded by driving /root/reference/clairvoyante/train.py:37-218 with the same mock model):
batch schedule incl. the validationStart quirks, synchronous last batch, LR / lambda zig-zag
decay, stop at the third switch, checkpoint names, resume epoch, log lines, final report."""
import json
import logging
import os
import pickle
import types

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class MockModel(object):
    def __init__(self, script):
        self.calls = []; self.script = list(script); self.lr = None; self.lam = None
        self.trainLossRTVal = None; self.trainSummaryRTVal = None; self.getLossLossRTVal = None

    def _tag(self, X):
        return [int(X[0, 0]) if len(X) else -1, int(len(X))]

    def trainNoRT(self, X, Y):
        self.calls.append(["train"] + self._tag(X)); self.trainLossRTVal = float(len(X)); self.trainSummaryRTVal = None

    def getLossNoRT(self, X, Y):
        self.calls.append(["val"] + self._tag(X)); self.getLossLossRTVal = 0.0

    def getLoss(self, X, Y):
        self.calls.append(["val_sync"] + self._tag(X)); return self.script.pop(0)

    def predict(self, X):
        self.calls.append(["predict"] + self._tag(X))
        i = X[:, 0].astype(np.int64)
        oh = lambda k, v: np.eye(k, dtype=np.float32)[v % k]
        return oh(4, i), oh(2, i // 3), oh(4, i // 5), oh(6, i // 7)

    def setLearningRate(self, v=None):
        self.lr = self.lr * 0.1 if v is None else v; self.calls.append(["lr", self.lr]); return self.lr

    def setL2RegularizationLambda(self, v=None):
        self.lam = self.lam * 0.1 if v is None else v; self.calls.append(["lambda", self.lam]); return self.lam

    def saveParameters(self, fn):
        self.calls.append(["save", os.path.basename(fn)])


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_trainall_replays_reference_schedule(tag, tmp_path):
    from clairvoyante_amd import train, utils_v2
    g = json.load(open(os.path.join(G, "train_schedule.json")))[tag]
    total = g["total"]
    idx = np.arange(total)
    rng = np.random.RandomState(5)
    ylab = np.zeros((total, 16)); ylab[idx, rng.randint(0, 4, total)] = 1; ylab[idx, 4 + rng.randint(0, 2, total)] = 1
    ylab[idx, 6 + rng.randint(0, 4, total)] = 1; ylab[idx, 10 + rng.randint(0, 6, total)] = 1
    XC, YC = [], []
    for s in range(0, total + 1, 500):
        XC.append(utils_v2.pack_array(idx[s:s + 500].reshape(-1, 1).astype(np.float32)))
        YC.append(utils_v2.pack_array(ylab[s:s + 500]))
    binfn = str(tmp_path / "sched.bin")
    with open(binfn, "wb") as fh:
        pickle.dump(total, fh); pickle.dump(XC, fh); pickle.dump(YC, fh); pickle.dump([], fh)
    m = MockModel(g["script"])
    args = types.SimpleNamespace(bin_fn=binfn, tensor_fn=None, var_fn=None, bed_fn=None, chkpnt_fn=g["chkpnt_fn"],
                                 learning_rate=1e-3, lambd=1e-3, ochk_prefix="/tmp/out/model", olog_dir=None,
                                 v2=False, v3=True, slim=False)
    logs = []

    class H(logging.Handler):
        def emit(self, rec):
            msg = rec.getMessage()
            if "time elapsed" not in msg:
                logs.append(msg)
    h = H(); logging.getLogger().addHandler(h); logging.getLogger().setLevel(logging.INFO)
    try:
        train.TrainAll(args, m, utils_v2)
    finally:
        logging.getLogger().removeHandler(h)
    want = [[c[0]] + [float(v) if isinstance(v, float) else v for v in c[1:]] for c in g["calls"]]
    got = [[c[0]] + [float(v) if isinstance(v, float) else v for v in c[1:]] for c in m.calls]
    assert len(got) == len(want)
    for a, b in zip(got, want):
        if a[0] in ("lr", "lambda"):
            assert a[0] == b[0] and abs(a[1] - b[1]) <= 1e-12 * abs(b[1])
        else:
            assert a == b
    assert logs == g["logs"]


def test_zigzag_rule():
    from clairvoyante_amd import train
    mk = lambda xs: [(x, i) for i, x in enumerate(xs)]
    assert train._zigzag(mk([3, 4, 3, 4, 3, 4])) and train._zigzag(mk([4, 3, 4, 3, 4, 3]))
    assert train._zigzag(mk([2, 2, 9, 9, 9, 1]))            # flat first step
    assert not train._zigzag(mk([9, 8, 7, 6, 5, 4])) and not train._zigzag(mk([3, 4, 3, 4, 4, 3]))


def test_sharded_batch_stream_covers_every_batch_exactly(tmp_path):
    """data parallel: each rank's _BatchStream decompresses only its slice; the slices of all ranks, in rank order,
    are the single-process batch -- for every batch of the epoch schedule, with and without the prefetch thread"""
    from clairvoyante_amd import param, train, utils_v2
    total = 23456
    idx = np.arange(total)
    X = idx.reshape(-1, 1).astype(np.float32)
    Y = np.stack([idx * 2.0, idx * 3.0], axis=1)
    XC = [utils_v2.pack_array(X[s:s + 500]) for s in range(0, total + 1, 500)]
    YC = [utils_v2.pack_array(Y[s:s + 500]) for s in range(0, total + 1, 500)]
    vstart = int(total * 0.9) + 1
    for ws in (2, 3):
        for use_prefetch in (False, True):
            one = train._BatchStream(utils_v2, XC, YC, total, vstart)
            ranks = [train._BatchStream(utils_v2, XC, YC, total, vstart, r, ws) for r in range(ws)]
            streams = [one] + ranks
            if use_prefetch:
                for s in streams:
                    s.prefetch(param.trainBatchSize, lambda p: train._next_batch_size(p, vstart))
            size = param.trainBatchSize
            nb = 0
            while True:
                full = one.fetch(size)
                parts = [s.fetch(size) for s in ranks]
                assert all(p[2:] == full[2:] for p in parts)                 # start, count of the whole batch, last
                assert np.array_equal(np.concatenate([p[0] for p in parts]), full[0])
                assert np.array_equal(np.concatenate([p[1] for p in parts]), full[1])
                assert abs(len(parts[0][0]) - len(parts[-1][0])) <= 1
                nb += 1
                if full[4]:
                    break
                size = one.next_size()
                assert all(s.next_size() == size for s in ranks)
            assert nb > 5
