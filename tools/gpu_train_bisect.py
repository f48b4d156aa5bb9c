"""Which candidates of synth.make_candidates(N, seed=1000+N) give different gradients on the tile and the plain kernels:
64-candidate ranges first, single candidates inside the ranges that differ.  usage: gpu_train_bisect.py full|slim N"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import common
from oracle import cv_oracle as O
O.build()
from clairvoyante_amd import clairvoyante_v3_slim, clairvoyante_v3, synth, _lib, param
def flat(m, which):
    t = torch.empty(m.numParameters, device="cuda")
    _lib.check(m._lib.cv_flat_copy(m._h, which, ctypes.c_void_p(t.data_ptr()), 0, None))
    torch.cuda.synchronize(); return t.cpu().numpy().copy()
arch = sys.argv[1]; mod = clairvoyante_v3_slim if arch == "slim" else clairvoyante_v3
N = int(sys.argv[2])
P = common.bench_params(O, arch)
xt, cls, rf, alt, il = synth.make_candidates(N, seed=1000 + N, device="cuda", return_class=True)
y = synth.make_labels(cls, rf, alt, il)
M = {}
for name, impl in (("tile", 1), ("plain", 0)):
    m = mod.Clairvoyante(); m.setParameters(P); m.setOption("impl", impl); m.setOption("train_ksplit", 0)
    m.dropoutRateFC4Val = 0.0; m.setL2RegularizationLambda(0.0); m.setLearningRate(1e-3); m._dropout_seed = 31
    M[name] = m
def bad(lo, hi):
    out = {}
    for name, m in M.items():
        m.setParameters(P); m.train(xt[lo:hi].contiguous(), y[lo:hi].contiguous()); out[name] = flat(m, 1)
    off = 0; worst = 0
    for pn in O.PARAM_NAMES:
        sz = int(np.prod(P[pn].shape))
        gt = out["tile"][off:off + sz]; gp = out["plain"][off:off + sz]; off += sz
        worst = max(worst, np.abs(gt - gp).max() / (np.abs(gp).max() + 1e-30))
    return worst
for lo in range(0, N, 64):
    hi = min(N, lo + 64)
    if bad(lo, hi) > 2e-5:
        for i in range(lo, hi):
            w = bad(i, i + 1)
            if w > 2e-5: print("candidate", i, "%.2e" % w, flush=True)
