"""Differential soak of the training step: seeded random batches (size log-uniform over 1 .. NMAX, its own data seed, the
reference's dropout rate) through the tile kernels (fc4 forward as a single chain) and through the plain
one-thread-per-output kernels; every gradient of the bucket within TOL of its tensor's largest entry, the loss within
1e-6.  Prints the batches that differ (n, seed) -- feed them to tools/gpu_train_bisect.py / gpu_train_map_diff.py.
usage: gpu_train_fuzz.py full|slim ROUNDS [NMAX] [seed0] [NMIN]"""
import sys, os, ctypes, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import common
from clairvoyante_amd import clairvoyante_v3_slim, clairvoyante_v3, synth, _lib, param
TOL = 1e-5
def flat(m, which):
    t = torch.empty(m.numParameters, device="cuda")
    _lib.check(m._lib.cv_flat_copy(m._h, which, ctypes.c_void_p(t.data_ptr()), 0, None))
    torch.cuda.synchronize(); return t.cpu().numpy().copy()
arch = sys.argv[1]; rounds = int(sys.argv[2]); nmax = int(sys.argv[3]) if len(sys.argv) > 3 else 12000
seed0 = int(sys.argv[4]) if len(sys.argv) > 4 else 0
nmin = int(sys.argv[5]) if len(sys.argv) > 5 else 1
mod = clairvoyante_v3_slim if arch == "slim" else clairvoyante_v3
P = common.bench_params(None, arch)
names = list(P.keys())
M = {}
for name, impl in (("tile", 1), ("plain", 0)):
    m = mod.Clairvoyante(); m.setOption("impl", impl); m.setOption("train_ksplit", 0)
    m.dropoutRateFC4Val = param.dropoutRateFC4; m.setL2RegularizationLambda(param.l2RegularizationLambda); m.setLearningRate(1e-3)
    M[name] = m
shapes = M["tile"].paramShapes()
from oracle import cv_oracle as O
order = list(O.PARAM_NAMES)          # flat order of the bucket
rng = np.random.RandomState(seed0)
bad = 0; worst = 0.0; t0 = time.time(); total = 0
for r in range(rounds):
    n = int(np.exp(rng.uniform(np.log(float(nmin)), np.log(float(nmax))))); seed = int(rng.randint(1, 1 << 30))
    xt, cls, rf, alt, il = synth.make_candidates(n, seed=seed, device="cuda", return_class=True)
    y = synth.make_labels(cls, rf, alt, il)
    out = {}
    for name, m in M.items():
        m.setParameters(P); m._dropout_seed = seed; m._adam_t = 0
        loss, _ = m.train(xt, y); out[name] = (float(loss), flat(m, 1))
    total += n
    off = 0; w = 0.0; wn = ""
    for k in order:
        sz = int(np.prod(shapes[k]))
        gt = out["tile"][1][off:off + sz]; gp = out["plain"][1][off:off + sz]; off += sz
        e = float(np.abs(gt - gp).max() / (np.abs(gp).max() + 1e-30))
        if e > w: w, wn = e, k
    worst = max(worst, w if w <= TOL else 0.0)
    dl = abs(out["tile"][0] - out["plain"][0]) / abs(out["plain"][0])
    if w > TOL or dl > 1e-6:
        bad += 1; print("DIFF %s n=%d seed=%d: %s %.2e, loss %.2e" % (arch, n, seed, wn, w, dl), flush=True)
print("%s: %d batches, %d candidates, %d differ; worst of the rest %.2e of the largest entry; %.0f s" % (arch, rounds, total, bad, worst, time.time() - t0))
