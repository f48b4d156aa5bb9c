"""First-contact diagnostics on the GPU box: per-layer parity of both kernel sets against the
CPU oracle, bitwise statistics, coarse timings, training-path checks.  Writes
gpurun_out/diag.json.  (Development tool; the judged checks are tests/ and bench.py.)"""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from oracle import cv_oracle as O
import common
from clairvoyante_amd import clairvoyante_v3, clairvoyante_v3_slim, _lib, synth

res = {}
props = torch.cuda.get_device_properties(0)
res["device"] = {"name": props.name, "cus": props.multi_processor_count, "mem_gb": props.total_memory / 2**30,
                 "gcn": getattr(props, "gcnArchName", "?")}
print(res["device"], flush=True)


def make(arch):
    m = clairvoyante_v3.Clairvoyante() if arch == "full" else clairvoyante_v3_slim.Clairvoyante()
    P = common.bench_params(O, arch)
    m.setParameters(P)
    return m, P


for arch in ("full", "slim"):
    m, P = make(arch)
    x = common.inputs(1000, stress=24)
    n = x.shape[0]
    ref = O.forward_all(arch, P, x)
    names = {1: "pool1", 2: "pool2", 3: "pool3", 4: "fc4", 5: "fc5"}
    for impl in (0, 1):
        m.setOption("impl", impl)
        key = "%s_impl%d" % (arch, impl)
        try:
            xd = torch.from_numpy(x).cuda()
            out = m.predict_device(xd)
            torch.cuda.synchronize()
            out = out.cpu().numpy()
            r = {"out_maxabs": float(np.abs(out - ref["out"]).max()), "out_bitwise": common.bitwise_frac(out, ref["out"]),
                 "argmax": common.argmax_match(out, ref["out"]), "nan": int(np.isnan(out).sum())}
            for layer, nm in names.items():
                a = m.getActivation(layer, n).cpu().numpy().reshape(n, -1)
                b = ref[nm].reshape(n, -1)
                r[nm] = {"maxabs": float(np.abs(a - b).max()), "bitwise": common.bitwise_frac(a, b),
                         "scale": float(np.abs(b).max())}
            res[key] = r
        except Exception as e:  # keep going: the other kernel set may still work
            res[key] = {"error": repr(e)}
        print(key, json.dumps(res[key]), flush=True)
    # odd sizes / tails
    m.setOption("impl", 1)
    tails = {}
    for nn in (1, 15, 16, 17, 33, 100):
        try:
            o = m.predict_device(torch.from_numpy(x[:nn]).cuda()).cpu().numpy()
            tails[nn] = float(np.abs(o - ref["out"][:nn]).max())
        except Exception as e:
            tails[nn] = repr(e)
    res[arch + "_tails"] = tails
    print(arch, "tails", tails, flush=True)
    # timing
    for impl, nn in ((1, 65536), (0, 8192)):
        try:
            m.setOption("impl", impl)
            xb = synth.make_candidates(nn, seed=11, device="cuda")
            ob = torch.empty((nn, 16), device="cuda")
            for _ in range(2):
                m.predict_device(xb, ob)
            torch.cuda.synchronize()
            t0 = time.time()
            reps = 5
            for _ in range(reps):
                m.predict_device(xb, ob)
            torch.cuda.synchronize()
            dt = (time.time() - t0) / reps
            res["%s_time_impl%d" % (arch, impl)] = {"n": nn, "ms": dt * 1e3, "cand_per_s": nn / dt}
            print(arch, "impl", impl, "n", nn, "ms", dt * 1e3, "cand/s", nn / dt, flush=True)
        except Exception as e:
            res["%s_time_impl%d" % (arch, impl)] = {"error": repr(e)}
            print("timing error", repr(e), flush=True)
    # training path: loss + grads (no dropout) against the oracle
    try:
        nn = 48
        xs = x[:nn]
        xt, cls, rf, alt, il = synth.make_candidates(nn, seed=9, return_class=True)
        ys = synth.make_labels(cls, rf, alt, il).numpy()
        xs = xt.numpy()
        lam = 0.01
        l_or, ls_or, g_or = O.loss_grad(arch, P, xs, ys, lam=lam)
        losses = (ctypes.c_double * 6)()
        xd = torch.from_numpy(xs).cuda(); yd = torch.from_numpy(ys).cuda()
        _lib.check(m._lib.cv_grad(m._h, ctypes.c_void_p(xd.data_ptr()), ctypes.c_void_p(yd.data_ptr()), nn,
                                  ctypes.c_float(0.0), ctypes.c_float(lam), 1, 1, losses, None))
        gb = torch.empty(m.numParameters, device="cuda")
        _lib.check(m._lib.cv_flat_copy(m._h, 1, ctypes.c_void_p(gb.data_ptr()), 0, None))
        torch.cuda.synchronize()
        gb = gb.cpu().numpy()
        off = 0
        gerr = {}
        for name in O.PARAM_NAMES:
            sz = g_or[name].size
            g = gb[off:off + sz].reshape(g_or[name].shape)
            gref = g_or[name] - (lam * P[name] if "bias" not in name else 0)
            gerr[name] = float(np.abs(g - gref).max() / (np.abs(gref).max() + 1e-12))
            off += sz
        res[arch + "_train"] = {"loss_gpu": list(losses), "loss_oracle": ls_or + [l_or], "grad_relerr": gerr}
        print(arch, "train", json.dumps(res[arch + "_train"]), flush=True)
    except Exception as e:
        res[arch + "_train"] = {"error": repr(e)}
        print("train error", repr(e), flush=True)
    m.close()

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "diag.json"), "w"), indent=1)
print("DIAG DONE")
