#!/opt/conda/bin/python3.9
"""Golden call/log traces of the reference's auxiliary drivers, recorded by
importing them from /root/reference/clairvoyante (2to3-converted copies in a temp dir, see
make_golden_ref.py) and driving them with the same recording mock model as train_schedule.json:

  aux_schedule.json
    nonstop / nonstop_resume   trainNonstop.TrainAll                     (trainNonstop.py:39-129)
    noval                      trainWithoutValidationNonstop.TrainAll    (:39-111)
    devdiff                    calTrainDevDiff.CalcAll + its stderr      (calTrainDevDiff.py:29-82)
    evallist                   evaluateListOfModels.Run                  (evaluateListOfModels.py:14-107)

param.maxEpoch is lowered to 4 for the recording (the loops are `while i < param.maxEpoch`); the
test lowers it the same way.  Run:  /opt/conda/bin/python3.9 tests/golden/make_golden_aux.py
"""
import contextlib
import importlib
import io
import json
import logging
import os
import pickle
import shutil
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_ref as G  # noqa: E402

MAX_EPOCH = 4


class Mock(G.MockModel):
    def __init__(self):
        G.MockModel.__init__(self, [])
        self.k = 0

    def trainNoRT(self, X, Y):
        G.MockModel.trainNoRT(self, X, Y)
        self.trainLossRTVal = float(len(X)) * 0.5

    def getLossNoRT(self, X, Y):
        self.calls.append(["val"] + self._tag(X))
        self.k += 1
        self.getLossLossRTVal = float(len(X)) + self.k      # distinguishable, order dependent

    def getLoss(self, X, Y):
        self.calls.append(["val_sync"] + self._tag(X))
        return float(len(X)) * 2.0

    def restoreParameters(self, fn):
        self.calls.append(["restore", os.path.basename(fn)])

    def init(self):
        self.calls.append(["init"])


def dataset(total):
    import blosc
    idx = np.arange(total)
    rng = np.random.RandomState(5)
    ylab = np.zeros((total, 16)); ylab[idx, rng.randint(0, 4, total)] = 1; ylab[idx, 4 + rng.randint(0, 2, total)] = 1
    ylab[idx, 6 + rng.randint(0, 4, total)] = 1; ylab[idx, 10 + rng.randint(0, 6, total)] = 1
    XC, YC = [], []
    for s in range(0, total + 1, 500):
        XC.append(blosc.pack_array(idx[s:s + 500].reshape(-1, 1).astype(np.float32), cname="lz4hc"))
        YC.append(blosc.pack_array(ylab[s:s + 500], cname="lz4hc"))
    fn = os.path.join(tempfile.gettempdir(), "cv_aux_%d.bin" % total)
    with open(fn, "wb") as fh:
        pickle.dump(total, fh); pickle.dump(XC, fh); pickle.dump(YC, fh); pickle.dump([], fh)
    return fn


@contextlib.contextmanager
def captured():
    logs = []

    class H(logging.Handler):
        def emit(self, rec):
            msg = rec.getMessage()
            if "time elapsed" not in msg:
                logs.append(msg)
    h = H(); logging.getLogger().addHandler(h); logging.getLogger().setLevel(logging.INFO)
    err = io.StringIO()
    old = sys.stderr
    sys.stderr = err
    try:
        yield logs, err
    finally:
        sys.stderr = old
        logging.getLogger().removeHandler(h)


def main():
    tmp = G.prepare_reference(extra=("trainNonstop.py", "trainWithoutValidationNonstop.py", "calTrainDevDiff.py",
                                     "evaluateListOfModels.py"))
    out = {"max_epoch": MAX_EPOCH}
    try:
        utils = importlib.import_module("utils_v2")
        param = importlib.import_module("param")
        param.maxEpoch = MAX_EPOCH
        ns = lambda fn, **kw: types.SimpleNamespace(bin_fn=fn, tensor_fn=None, var_fn=None, bed_fn=None, learning_rate=1e-3,
                                                    lambd=1e-3, ochk_prefix="/tmp/out/model", olog_dir=None, v2=False,
                                                    v3=True, slim=False, **kw)
        for tag, modname, total, chk in (("nonstop", "trainNonstop", 23456, None),
                                         ("nonstop_resume", "trainNonstop", 11000, "model-000001"),
                                         ("noval", "trainWithoutValidationNonstop", 23456, None),
                                         ("noval_exact", "trainWithoutValidationNonstop", 20000, None),
                                         ("noval_small", "trainWithoutValidationNonstop", 7300, None)):
            mod = importlib.import_module(modname)
            fn = dataset(total)
            m = Mock()
            with captured() as (logs, err):
                mod.TrainAll(ns(fn, chkpnt_fn=chk), m, utils)
            os.remove(fn)
            out[tag] = {"module": modname, "total": total, "chkpnt_fn": chk, "calls": m.calls, "logs": logs}
            print(tag, len(m.calls), "calls", len(logs), "log lines")
        for tag, total in (("devdiff", 23456), ("devdiff_b", 10000), ("devdiff_c", 12010)):
            mod = importlib.import_module("calTrainDevDiff")
            fn = dataset(total)
            m = Mock()
            with captured() as (logs, err):
                mod.CalcAll(ns(fn, chkpnt_fn=["run/model-000003", "run/model-000007"]), m, utils)
            os.remove(fn)
            out[tag] = {"total": total, "calls": m.calls, "stderr": err.getvalue()}
            print(tag, len(m.calls), "calls", repr(err.getvalue()))
        # evaluateListOfModels.Run builds its own model: hand it the mock through a stub module
        mod = importlib.import_module("evaluateListOfModels")
        m = Mock()
        stub = types.ModuleType("clairvoyante_v3"); stub.Clairvoyante = lambda: m
        sys.modules["clairvoyante_v3"] = stub
        utils.SetupEnv = lambda: None
        fn = dataset(3456)
        lst = os.path.join(tempfile.gettempdir(), "cv_aux_models.txt")
        open(lst, "w").write("run/model-000002\nrun/model-000005\n")
        with captured() as (logs, err):
            mod.Run(ns(fn, chkpnt_list=lst))
        os.remove(fn); os.remove(lst)
        out["evallist"] = {"total": 3456, "calls": m.calls, "logs": logs}
        print("evallist", len(m.calls), "calls", len(logs), "log lines")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    json.dump(out, open(os.path.join(HERE, "aux_schedule.json"), "w"))


if __name__ == "__main__":
    main()
