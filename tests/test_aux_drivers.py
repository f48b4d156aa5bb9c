"""Auxiliary drivers against the reference's own loops: tests/golden/aux_schedule.json
was recorded by driving /root/reference/clairvoyante/{trainNonstop,trainWithoutValidationNonstop,
calTrainDevDiff,evaluateListOfModels}.py with this same mock model (tests/golden/make_golden_aux.py)."""
import contextlib
import importlib
import io
import json
import logging
import os
import pickle
import sys
import types

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLD = json.load(open(os.path.join(G, "aux_schedule.json")))


class Mock(object):
    def __init__(self):
        self.calls = []; self.k = 0; self.lr = None; self.lam = None
        self.trainLossRTVal = None; self.trainSummaryRTVal = None; self.getLossLossRTVal = None

    def _tag(self, X):
        return [int(X[0, 0]) if len(X) else -1, int(len(X))]

    def init(self):
        self.calls.append(["init"])

    def trainNoRT(self, X, Y):
        self.calls.append(["train"] + self._tag(X)); self.trainLossRTVal = float(len(X)) * 0.5

    def getLossNoRT(self, X, Y):
        self.calls.append(["val"] + self._tag(X)); self.k += 1; self.getLossLossRTVal = float(len(X)) + self.k

    def getLoss(self, X, Y):
        self.calls.append(["val_sync"] + self._tag(X)); return float(len(X)) * 2.0

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

    def restoreParameters(self, fn):
        self.calls.append(["restore", os.path.basename(fn)])


def dataset(tmp_path, total):
    from clairvoyante_amd import utils_v2
    idx = np.arange(total)
    rng = np.random.RandomState(5)
    ylab = np.zeros((total, 16)); ylab[idx, rng.randint(0, 4, total)] = 1; ylab[idx, 4 + rng.randint(0, 2, total)] = 1
    ylab[idx, 6 + rng.randint(0, 4, total)] = 1; ylab[idx, 10 + rng.randint(0, 6, total)] = 1
    XC, YC = [], []
    for s in range(0, total + 1, 500):
        XC.append(utils_v2.pack_array(idx[s:s + 500].reshape(-1, 1).astype(np.float32)))
        YC.append(utils_v2.pack_array(ylab[s:s + 500]))
    fn = str(tmp_path / ("aux_%d.bin" % total))
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
    err = io.StringIO(); old = sys.stderr; sys.stderr = err
    try:
        yield logs, err
    finally:
        sys.stderr = old
        logging.getLogger().removeHandler(h)


def ns(fn, **kw):
    return types.SimpleNamespace(bin_fn=fn, tensor_fn=None, var_fn=None, bed_fn=None, learning_rate=1e-3, lambd=1e-3,
                                 ochk_prefix="/tmp/out/model", olog_dir=None, v2=False, v3=True, slim=False, **kw)


def same_calls(got, want):
    assert len(got) == len(want)
    for a, b in zip(got, want):
        if a[0] in ("lr", "lambda"):
            assert a[0] == b[0] and abs(a[1] - b[1]) <= 1e-12 * abs(b[1])
        else:
            assert list(a) == list(b)


@pytest.fixture
def short_epochs(monkeypatch):
    from clairvoyante_amd import param
    monkeypatch.setattr(param, "maxEpoch", GOLD["max_epoch"])


@pytest.mark.parametrize("tag", ["nonstop", "nonstop_resume", "noval", "noval_exact", "noval_small"])
def test_nonstop_trainers_replay_reference(tag, tmp_path, short_epochs):
    from clairvoyante_amd import utils_v2
    g = GOLD[tag]
    mod = importlib.import_module("clairvoyante_amd." + g["module"])
    m = Mock()
    with captured() as (logs, _):
        mod.TrainAll(ns(dataset(tmp_path, g["total"]), chkpnt_fn=g["chkpnt_fn"]), m, utils_v2)
    same_calls(m.calls, g["calls"])
    assert logs == g["logs"]


@pytest.mark.parametrize("tag", ["devdiff", "devdiff_b", "devdiff_c"])
def test_caltraindevdiff_replays_reference(tag, tmp_path):
    from clairvoyante_amd import calTrainDevDiff, utils_v2
    g = GOLD[tag]
    m = Mock()
    with captured() as (_, err):
        calTrainDevDiff.CalcAll(ns(dataset(tmp_path, g["total"]), chkpnt_fn=["run/model-000003", "run/model-000007"]),
                                m, utils_v2)
    same_calls(m.calls, g["calls"])
    assert err.getvalue() == g["stderr"]


def test_evaluate_list_of_models_replays_reference(tmp_path, monkeypatch):
    from clairvoyante_amd import clairvoyante_v3, evaluateListOfModels, utils_v2
    g = GOLD["evallist"]
    m = Mock()
    monkeypatch.setattr(clairvoyante_v3, "Clairvoyante", lambda: m)
    monkeypatch.setattr(utils_v2, "SetupEnv", lambda: None)
    lst = tmp_path / "models.txt"
    lst.write_text("run/model-000002\nrun/model-000005\n")
    with captured() as (logs, _):
        evaluateListOfModels.Run(ns(dataset(tmp_path, g["total"]), chkpnt_list=str(lst)))
    same_calls(m.calls, g["calls"])
    assert logs == g["logs"]


def test_nonstop_cli_defaults():
    from clairvoyante_amd import trainNonstop
    parser = trainNonstop.build_parser("x")
    args = parser.parse_args(["--bin_fn", "nope.bin"])
    assert args.ochk_prefix is None and args.learning_rate == 1e-3 and args.v3 is True and args.slim is False


@pytest.mark.parametrize("tag,extra", [("plain", ["--includingAllContigs"]),
                                       ("bed", ["--bed_fn", "regions.bed", "--qual", "100", "--threshold", "0.25",
                                                "--refChunkSize", "5000000"])])
def test_callvarbamparallel_prints_the_reference_command_list(tag, extra, tmp_path, capsys):
    """golden: tests/golden/parallel/cmds_*.txt, printed by the reference's callVarBamParallel.py
    (tests/golden/make_golden_parallel.py)"""
    import shutil
    from clairvoyante_amd import callVarBamParallel as par
    P = os.path.join(G, "parallel")
    work = str(tmp_path)
    for f in ("model.meta", "in.bam", "ref.fa"):
        open(os.path.join(work, f), "w").write("x")
    shutil.copy(os.path.join(P, "ref.fa.fai"), os.path.join(work, "ref.fa.fai"))
    shutil.copy(os.path.join(P, "regions.bed"), os.path.join(work, "regions.bed"))
    extra = [os.path.join(work, e) if e == "regions.bed" else e for e in extra]
    args = par.build_parser().parse_args(
        ["--chkpnt_fn", os.path.join(work, "model"), "--bam_fn", os.path.join(work, "in.bam"), "--ref_fn",
         os.path.join(work, "ref.fa"), "--output_prefix", "out/calls", "--pypy", "python3", "--samtools", "gzip",
         "--sampleName", "NA1"] + extra)
    par.Run(args)
    got = capsys.readouterr().out.replace(work, "@DIR@").replace(os.path.dirname(os.path.abspath(par.__file__)), "@REF@")
    assert got == open(os.path.join(P, "cmds_%s.txt" % tag)).read()


def test_printed_parallel_commands_are_runnable(tmp_path, capsys):
    """the printed `python <pkg>/callVarBam.py ...` lines must start (package-relative imports
    used to fail when the driver was run as a script); executed here from a foreign directory up to the argument
    check, plus every other driver with --help"""
    import shlex
    import shutil
    import subprocess
    from clairvoyante_amd import callVarBamParallel as par
    P = os.path.join(G, "parallel")
    work = str(tmp_path)
    for f in ("model.meta", "in.bam", "ref.fa"):
        open(os.path.join(work, f), "w").write("x")
    shutil.copy(os.path.join(P, "ref.fa.fai"), os.path.join(work, "ref.fa.fai"))
    args = par.build_parser().parse_args(
        ["--chkpnt_fn", os.path.join(work, "model"), "--bam_fn", os.path.join(work, "in.bam"), "--ref_fn",
         os.path.join(work, "ref.fa"), "--output_prefix", "out/calls", "--pypy", "python3", "--samtools", "gzip",
         "--sampleName", "NA1", "--includingAllContigs"])
    par.Run(args)
    line = capsys.readouterr().out.splitlines()[0]
    argv = shlex.split(line)
    assert argv[0] == "python" and argv[1].endswith("callVarBam.py")
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    r = subprocess.run([sys.executable] + argv[1:] + ["--help"], cwd=work, env=env, capture_output=True, text=True)
    assert r.returncode == 0 and "--chkpnt_fn" in r.stdout, r.stderr
    pkg = os.path.dirname(os.path.abspath(par.__file__))
    for drv in ("callVar.py", "train.py", "evaluate.py", "tensor2Bin.py", "CreateTensor.py", "callVarBamParallel.py",
                "trainNonstop.py", "trainWithoutValidationNonstop.py", "calTrainDevDiff.py", "evaluateListOfModels.py",
                "ExtractVariantCandidates.py", "GetTruth.py"):
        r = subprocess.run([sys.executable, os.path.join(pkg, drv), "--help"], cwd=work, env=env, capture_output=True, text=True)
        assert r.returncode == 0 and "usage" in r.stdout.lower(), (drv, r.stderr)


@pytest.mark.parametrize("tag,region", [("all", (None, None)), ("region", (400, 1500))])
def test_gettruth_rows_equal_reference_rows(tag, region, tmp_path, capfd):
    """golden: tests/golden/truth/rows_*.txt, printed by the reference's GetTruth.py (make_golden_truth.py)"""
    from clairvoyante_amd import GetTruth
    T = os.path.join(G, "truth")
    args = types.SimpleNamespace(vcf_fn=os.path.join(T, "sites.vcf"), var_fn=str(tmp_path / "rows.gz"), ctgName="ctgA",
                                 ctgStart=region[0], ctgEnd=region[1])
    GetTruth.OutputVariant(args)
    import gzip
    assert gzip.open(args.var_fn, "rt").read() == open(os.path.join(T, "rows_%s.txt" % tag)).read()


def test_submodule_invocator_lists_and_dispatches(capsys, monkeypatch):
    """python -m clairvoyante_amd SubmoduleName ... (the reference's clairvoyante.py:12-45)"""
    import runpy
    monkeypatch.setattr(sys, "argv", ["clairvoyante_amd"])
    with pytest.raises(SystemExit) as e:
        runpy.run_module("clairvoyante_amd", run_name="__main__")
    assert e.value.code == 0
    out = capsys.readouterr().out
    for name in ("callVarBam", "callVar", "train", "CreateTensor", "ExtractVariantCandidates"):
        assert "- %s\n" % name in out
    monkeypatch.setattr(sys, "argv", ["clairvoyante_amd", "train"])          # no options: the submodule prints its help
    with pytest.raises(SystemExit) as e:
        runpy.run_module("clairvoyante_amd", run_name="__main__")
    assert e.value.code == 1 and "--bin_fn" in capsys.readouterr().out
    monkeypatch.setattr(sys, "argv", ["clairvoyante_amd", "demoRun"])
    with pytest.raises(SystemExit) as e:
        runpy.run_module("clairvoyante_amd", run_name="__main__")
    assert "not part of this build" in str(e.value.code)
