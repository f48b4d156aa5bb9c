#!/opt/conda/bin/python3.9
"""Generates the golden vectors that pin the HOST logic of the hot path against the
reference itself (run in the build container only; /root/reference never travels).

    /opt/conda/bin/python3.9 tests/golden/make_golden_ref.py

The reference is Python 2; its files are copied to a temp dir OUTSIDE the repo, passed
through 2to3, and imported under numpy 1.26 (reference-era type promotion: a float32
scalar plus a Python float is float64 -- this matters for `qual`, callVar.py:72).  Only
inputs and the reference's outputs are stored here (data, not source):

  output_cases.npz + output_*.vcf     callVar.Output / PrintVCFHeader   (callVar.py:50-178)
  gettensor_rows.txt.gz + gettensor.npz   utils_v2.GetTensor            (utils_v2.py:23-59)
  trainarray_*.{txt.gz,npz}           utils_v2.GetTrainingArray labels  (utils_v2.py:62-186)
  train_schedule.json                 train.TrainAll driven with a recording mock model:
                                      batch schedule, LR/lambda decay, checkpoint names, logs
                                      (train.py:37-218)
  decompress.npz + mini.bin           DecompressArray + tensor2Bin layout (utils_v2.py:189-207,
                                      tensor2Bin.py:24-28); blocks packed by the real c-blosc
                                      (/opt/conda/lib/libblosc.so) through a shim that follows
                                      python-blosc's documented pack_array = compress(pickle.dumps(
                                      arr, HIGHEST_PROTOCOL), typesize=itemsize, clevel=9, shuffle,
                                      cname) -- python-blosc itself is not installed: the blosc
                                      envelope is "parity unpinned", the c-blosc stream is real.
"""
import ctypes
import gzip
import importlib
import io
import os
import pickle
import shutil
import subprocess
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/clairvoyante"


def prepare_reference(extra=()):
    tmp = tempfile.mkdtemp(prefix="cv_ref23_")
    for f in ("callVar.py", "utils_v2.py", "param.py", "tensor2Bin.py", "train.py") + tuple(extra):
        shutil.copy(os.path.join(REF, f), tmp)
    subprocess.check_call(["/opt/conda/bin/2to3", "-nw"] + [os.path.join(tmp, f) for f in os.listdir(tmp)],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    # python 2 pipes yield str; keep that meaning under python 3 (text-mode pipes)
    up = os.path.join(tmp, "utils_v2.py")
    src = open(up).read().replace("stdout=subprocess.PIPE, bufsize=8388608)",
                                  "stdout=subprocess.PIPE, bufsize=8388608, universal_newlines=True)")
    open(up, "w").write(src)
    for f in ("train.py",) + tuple(extra):  # np.int was removed from NumPy 1.24 (reference era: alias of int)
        tp = os.path.join(tmp, f)
        tsrc = open(tp).read().replace("dtype=np.int )", "dtype=int)").replace("dtype=np.int)", "dtype=int)")
        open(tp, "w").write(tsrc)
    # shim modules the image lacks (python-blosc) / changed API (intervaltree 3: search -> at)
    blosc = types.ModuleType("blosc")
    lib = ctypes.CDLL("/opt/conda/lib/libblosc.so.1")
    lib.blosc_compress_ctx.restype = ctypes.c_int
    lib.blosc_decompress_ctx.restype = ctypes.c_int

    def compress(data, typesize, clevel=9, shuffle=1, cname="blosclz"):
        out = ctypes.create_string_buffer(len(data) + 16 + 64)
        n = lib.blosc_compress_ctx(ctypes.c_int(clevel), ctypes.c_int(shuffle), ctypes.c_size_t(typesize),
                                   ctypes.c_size_t(len(data)), data, out, ctypes.c_size_t(len(out)),
                                   cname.encode(), ctypes.c_size_t(0), ctypes.c_int(1))
        assert n > 0
        return out.raw[:n]

    def decompress(data):
        nbytes = int.from_bytes(data[4:8], "little")
        out = ctypes.create_string_buffer(nbytes)
        n = lib.blosc_decompress_ctx(data, out, ctypes.c_size_t(nbytes), ctypes.c_int(1))
        assert n == nbytes
        return out.raw

    blosc.pack_array = lambda a, clevel=9, shuffle=1, cname="blosclz": compress(
        pickle.dumps(a, pickle.HIGHEST_PROTOCOL), a.itemsize, clevel, shuffle, cname)
    blosc.unpack_array = lambda b: pickle.loads(decompress(b))
    blosc.set_nthreads = lambda n: None
    sys.modules["blosc"] = blosc
    import intervaltree
    if not hasattr(intervaltree.IntervalTree, "search"):
        intervaltree.IntervalTree.search = lambda self, p: self.at(p)
    sys.path.insert(0, tmp)
    return tmp


def pileups(rng, n):
    """integer pileup-like tensors, matrices 1..3 already minus matrix 0"""
    x = np.zeros((n, 33, 4, 4), dtype=np.float32)
    for i in range(n):
        depth = int(rng.randint(4, 80))
        ref = rng.randint(0, 4, 33)
        m0 = np.zeros((33, 4)); m1 = np.zeros((33, 4)); m2 = np.zeros((33, 4)); m3 = np.zeros((33, 4))
        for p in range(33):
            d = rng.binomial(depth, 0.95)
            e = rng.binomial(d, 0.02)
            m0[p, ref[p]] = d
            m3[p, ref[p]] = d - e
            m3[p, (ref[p] + rng.randint(1, 4)) % 4] += e
        m1[:] = m3
        m2[:] = m0
        kind = rng.randint(0, 5)
        if kind == 1:
            a = rng.binomial(depth, rng.choice([0.5, 1.0]))
            a = min(a, m3[16, ref[16]])
            m3[16, ref[16]] -= a
            m3[16, (ref[16] + 1 + rng.randint(0, 3)) % 4] += a
        elif kind in (2, 3):
            L = int(rng.choice([1, 2, 3, 4, 6, 9, 16]))
            c = rng.binomial(depth, rng.choice([0.2, 0.5, 0.9]))
            for p in range(17, min(33, 17 + L)):
                if kind == 2:
                    m1[p, rng.randint(0, 4)] += c
                else:
                    m2[p, ref[p]] += c
        x[i] = np.stack([m0, m1 - m0, m2 - m0, m3 - m0], axis=-1)
    return x


def probs(rng, n, k, temp):
    z = rng.standard_normal((n, k)) * temp
    e = np.exp(z - z.max(1, keepdims=True))
    return (e / e.sum(1, keepdims=True)).astype(np.float32)


def gen_output(callVar, param):
    rng = np.random.RandomState(20260927)
    n = 480
    X = pileups(rng, n)
    X[5] = 0; X[77] = 0                                   # dp == 0 -> no record (callVar.py:88)
    # long indels so that the length-inference loops reach the SV branch (callVar.py:106-137)
    for j in (11, 12, 13, 14):
        X[j, 17:33, :, 1] = 0; X[j, 17:33, :, 2] = 0
        X[j, 17:33, 1, 1] = 30; X[j, 17:33, 2, 2] = 30
    seqs = ["".join(rng.choice(list("ACGT"), 33)) for _ in range(n)]
    pos = ["chr%d:%d:%s" % (1 + j % 3, 1000 + 37 * j, seqs[j]) for j in range(n)]
    base = (1.0 / (1.0 + np.exp(-rng.standard_normal((n, 4)) * 3))).astype(np.float32)
    z = probs(rng, n, 2, 2.0); t = probs(rng, n, 4, 3.0); l = probs(rng, n, 6, 3.0)
    # force every (type, length) combination and a few exact ties / saturations
    k = 0
    for vt in range(4):
        for vl in range(6):
            for rep in range(3):
                j = 20 + k; k += 1
                t[j] = 0.01; t[j, vt] = 0.97
                l[j] = 0.004; l[j, vl] = 0.98
    for j in (11, 13):
        t[j] = [0.01, 0.01, 0.97, 0.01]; l[j] = [0.004] * 5 + [0.98]
    for j in (12, 14):
        t[j] = [0.01, 0.01, 0.01, 0.97]; l[j] = [0.004] * 5 + [0.98]
    t[200] = [0.25, 0.25, 0.25, 0.25]; z[200] = [0.5, 0.5]; l[200] = [1 / 6.0] * 6
    base[201] = [1.0, 1.0, 1.0, 1.0]; t[201] = [0.0, 1.0, 0.0, 0.0]; l[201] = [1, 0, 0, 0, 0, 0]; z[201] = [0.0, 1.0]
    base[202] = [0.5, 0.9, 0.9, 0.1]; t[202] = [0.0, 1.0, 0.0, 0.0]
    base[203] = [0.9, 0.2, 0.9, 0.9]; t[203] = [0.1, 0.8, 0.05, 0.05]
    t[204] = [0.5, 0.5, 0.0, 0.0]
    t[205] = [1e-30, 1.0, 1e-30, 1e-30]; z[205] = [1e-20, 1.0]; l[205] = [1.0, 1e-25, 0, 0, 0, 0]
    np.savez_compressed(os.path.join(HERE, "output_cases.npz"), X=X.astype(np.int16), pos=np.array(pos),
                        base=base, z=z, t=t, l=l)
    fai = os.path.join(HERE, "mini.fa.fai")
    with open(fai, "w") as f:
        f.write("chr1\t248956422\t6\t60\t61\nchr2\t242193529\t253105708\t60\t61\nchr3\t198295559\t499335803\t60\t61\n")
    for showRef, qual, ref_fn, tag in ((False, None, None, "a"), (True, 20, os.path.join(HERE, "mini.fa"), "b"),
                                       (False, 150, None, "c")):
        args = types.SimpleNamespace(v2=False, v3=True, showRef=showRef, qual=qual, ref_fn=ref_fn,
                                     sampleName="SAMPLE" if tag != "b" else "HG001")
        fh = io.StringIO()
        callVar.PrintVCFHeader(args, fh)
        for s in range(0, n, 100):                     # several batches like Test() does
            e = min(n, s + 100)
            callVar.Output(args, fh, e - s, X[s:e], pos[s:e], base[s:e], z[s:e], t[s:e], l[s:e])
        with open(os.path.join(HERE, "output_%s.vcf" % tag), "w") as f:
            f.write(fh.getvalue())
        print("output_%s.vcf: %d lines" % (tag, fh.getvalue().count("\n")))


def tensor_rows(rng, n, with_bad=True):
    X = pileups(rng, n)
    # undo the subtraction: the text format holds RAW counts (CreateTensor.py:24,56)
    raw = X.copy()
    for i in range(1, 4):
        raw[:, :, :, i] += raw[:, :, :, 0]
    rows = []
    for j in range(n):
        seq = "".join(rng.choice(list("ACGT"), 33))
        if with_bad and j % 11 == 3:
            seq = seq[:16] + "N" + seq[17:]
        if with_bad and j % 7 == 2:
            seq = seq.lower()
        rows.append("%s %d %s %s" % ("chr%d" % (1 + j % 2), 5000 + 13 * j, seq,
                                     " ".join("%0.1f" % v for v in raw[j].reshape(-1))))
    return rows


def gen_gettensor(utils):
    rng = np.random.RandomState(7)
    for tag, n, num in (("a", 47, 10), ("b", 30, 10)):     # b: multiple of the batch size AFTER filtering? see below
        rows = tensor_rows(rng, n, with_bad=(tag == "a"))
        fn = os.path.join(HERE, "gettensor_%s.txt.gz" % tag)
        with gzip.open(fn, "wt") as f:
            f.write("\n".join(rows) + "\n")
        ends, nums, Xs, poss = [], [], [], []
        for end, c, x, pos in utils.GetTensor(fn, num):
            ends.append(end); nums.append(c); Xs.append(np.array(x[:c])); poss += list(pos)
        np.savez_compressed(os.path.join(HERE, "gettensor_%s.npz" % tag), ends=np.array(ends), nums=np.array(nums),
                            X=np.concatenate(Xs).astype(np.float32), pos=np.array(poss), num=num)
        print("gettensor_%s: batches %s" % (tag, nums))


def gen_trainarray(utils, param):
    rng = np.random.RandomState(11)
    n = 1500
    rows = tensor_rows(rng, n, with_bad=True)
    tfn = os.path.join(HERE, "trainarray_tensor.txt.gz")
    with gzip.open(tfn, "wt") as f:
        f.write("\n".join(rows) + "\n")
    # truth variants for a subset of the sites: ctg pos ref alt gt1 gt2  (GetTruth.py:61-78)
    var = []
    for j, r in enumerate(rows):
        ctg, p, seq = r.split()[:3]
        seq = seq.upper()
        if j % 5 == 0:
            kind = (j // 5) % 6
            refb = seq[16] if seq[16] in "ACGT" else "A"
            alt = "ACGT"[("ACGT".index(refb) + 1) % 4]
            if kind == 0: rec = (refb, alt, "0", "1")
            elif kind == 1: rec = (refb, alt, "1", "1")
            elif kind == 2: rec = (refb, refb + "GT", "0", "1")
            elif kind == 3: rec = (refb + "ACGTAC", refb, "1", "1")
            elif kind == 4: rec = (refb + "A", refb, "0", "1")
            else: rec = (refb, refb + "ACGTA", "1", "1")
            var.append("%s %s %s %s %s %s" % ((ctg, p) + rec))
    var.append("chr9 1 A C 0 1")                          # a truth variant without a tensor
    vfn = os.path.join(HERE, "trainarray_var.txt.gz")
    with gzip.open(vfn, "wt") as f:
        f.write("\n".join(var) + "\n")
    bfn = os.path.join(HERE, "trainarray.bed.gz")
    with gzip.open(bfn, "wt") as f:
        f.write("chr1 5000 21000\nchr2 5000 9000\nchr2 9100 9101\nchr2 12000 21000\nchr9 0 10\n")
    import random
    random.seed(1234)                                       # reference shuffles unseeded (utils_v2.py:157)
    total, XC, YC, PC = utils.GetTrainingArray(tfn, vfn, bfn)
    X, _, _ = utils.DecompressArray(XC, 0, total, total)
    Y, _, _ = utils.DecompressArray(YC, 0, total, total)
    Pz, _, _ = utils.DecompressArray(PC, 0, total, total)
    np.savez_compressed(os.path.join(HERE, "trainarray.npz"), total=total, X=X.astype(np.float32), Y=Y,
                        pos=np.array([str(s) for s in Pz]), nblocks=len(XC))
    print("trainarray: total %d, %d blocks" % (total, len(XC)))
    # the .bin layout (tensor2Bin.py:24-28): four back-to-back pickles
    with open(os.path.join(HERE, "mini.bin"), "wb") as fh:
        pickle.dump(total, fh); pickle.dump(XC, fh); pickle.dump(YC, fh); pickle.dump(PC, fh)
    # a python-2 style variant of the same file: protocol-0 outer pickles (py2 default)
    with open(os.path.join(HERE, "mini_py2proto.bin"), "wb") as fh:
        for obj in (total, XC, YC, PC):
            pickle.dump(obj, fh, protocol=0)
    # DecompressArray slices (utils_v2.py:189-207)
    assert total > 1001
    cases = [(0, 10, total), (495, 10, total), (500, 500, total), (499, 502, total), (1000, 100, total),
             (total - 3, 10, total), (total - 500, 500, total), (0, total, total), (250, 1000, total - 20),
             (7, 1, total), (0, 500, total), (500, 1000, total)]
    outs = {}
    for k, (st, nm, mx) in enumerate(cases):
        a, nn, ef = utils.DecompressArray(XC, st, nm, mx)
        outs["x%d" % k] = np.asarray(a, dtype=np.float32); outs["m%d" % k] = np.array([st, nm, mx, nn, ef])
    np.savez_compressed(os.path.join(HERE, "decompress.npz"), **outs)


class MockModel(object):
    """records what train.TrainAll asks of the model object (train.py:37-218)"""

    def __init__(self, script):
        self.calls = []
        self.script = list(script)
        self.lr = None; self.lam = None
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
        n = len(X); i = X[:, 0].astype(np.int64)
        oh = lambda k, v: np.eye(k, dtype=np.float32)[v % k]
        return oh(4, i), oh(2, i // 3), oh(4, i // 5), oh(6, i // 7)

    def setLearningRate(self, v=None):
        self.lr = self.lr * 0.1 if v is None else v; self.calls.append(["lr", self.lr]); return self.lr

    def setL2RegularizationLambda(self, v=None):
        self.lam = self.lam * 0.1 if v is None else v; self.calls.append(["lambda", self.lam]); return self.lam

    def saveParameters(self, fn):
        self.calls.append(["save", os.path.basename(fn)])


def gen_train_schedule(train, utils):
    import json
    import logging
    out = {}
    for tag, total, chk in (("a", 23456, None), ("b", 30000, None), ("c", 1234, "model-000041")):
        idx = np.arange(total)
        rng = np.random.RandomState(5)
        ylab = np.zeros((total, 16)); ylab[idx, rng.randint(0, 4, total)] = 1; ylab[idx, 4 + rng.randint(0, 2, total)] = 1
        ylab[idx, 6 + rng.randint(0, 4, total)] = 1; ylab[idx, 10 + rng.randint(0, 6, total)] = 1
        XC, YC = [], []
        import blosc
        for s in range(0, total + 1, 500):          # trailing (possibly empty) block like utils_v2.py:181
            XC.append(blosc.pack_array(idx[s:s + 500].reshape(-1, 1).astype(np.float32), cname="lz4hc"))
            YC.append(blosc.pack_array(ylab[s:s + 500], cname="lz4hc"))
        script = [10, 9, 8, 7, 6, 5, 4, 3, 4, 3, 4, 3, 4, 2, 2, 9, 9, 9, 1, 2, 1, 2, 1, 2, 1, 2, 1, 2, 1, 2, 1, 2, 1]
        m = MockModel(script)
        binfn = os.path.join(tempfile.gettempdir(), "cv_sched_%s.bin" % tag)
        with open(binfn, "wb") as fh:
            pickle.dump(total, fh); pickle.dump(XC, fh); pickle.dump(YC, fh); pickle.dump([], fh)
        args = types.SimpleNamespace(bin_fn=binfn, tensor_fn=None, var_fn=None, bed_fn=None, chkpnt_fn=chk,
                                     learning_rate=1e-3, lambd=1e-3, ochk_prefix="/tmp/out/model", olog_dir=None,
                                     v2=False, v3=True, slim=False)
        logs = []

        class H(logging.Handler):
            def emit(self, rec):
                msg = rec.getMessage()
                if "time elapsed" not in msg:
                    logs.append(msg)
        h = H(); logging.getLogger().addHandler(h)
        train.TrainAll(args, m, utils)
        logging.getLogger().removeHandler(h)
        os.remove(binfn)
        out[tag] = {"total": total, "chkpnt_fn": chk, "script": script, "calls": m.calls, "logs": logs}
        print("train schedule %s: %d calls, %d log lines" % (tag, len(m.calls), len(logs)))
    json.dump(out, open(os.path.join(HERE, "train_schedule.json"), "w"))


def main():
    tmp = prepare_reference()
    try:
        callVar = importlib.import_module("callVar")
        utils = importlib.import_module("utils_v2")
        param = importlib.import_module("param")
        assert np.__version__.startswith("1."), "needs NumPy 1.x promotion rules (reference era)"
        gen_output(callVar, param)
        gen_gettensor(utils)
        gen_trainarray(utils, param)
        train = importlib.import_module("train")
        gen_train_schedule(train, utils)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
