"""Lays the maps of one training step on the tile kernels beside the plain one-thread-per-output kernels' and the oracle's:
pooled maps bitwise, pre-activation gradients element by element (cv_get_activation 11..13 / 21..23), then both paths' weight
gradients against oracle.loss_grad.  usage: gpu_train_map_diff.py full|slim N lo hi [option=value ...]  (candidates lo..hi of
synth.make_candidates(N, seed=1000+N)).  Found in round 6: selu' read off an output that rounded to +0 from below, and the
plain kernels' pooling backward comparing pre-activations."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import common
from oracle import cv_oracle as O
O.build()
from clairvoyante_amd import clairvoyante_v3_slim, clairvoyante_v3, synth, _lib, param
arch = sys.argv[1]; mod = clairvoyante_v3_slim if arch == "slim" else clairvoyante_v3
N = int(sys.argv[2]); lo, hi = int(sys.argv[3]), int(sys.argv[4])
P = common.bench_params(O, arch)
xt, cls, rf, alt, il = synth.make_candidates(N, seed=1000 + N, device="cuda", return_class=True)
y = synth.make_labels(cls, rf, alt, il)
x1 = xt[lo:hi].contiguous(); y1 = y[lo:hi].contiguous(); n = hi - lo
out = {}
for name, impl in (("tile", 1), ("plain", 0)):
    m = mod.Clairvoyante(); m.setParameters(P); m.setOption("impl", impl); m.setOption("train_ksplit", 0)
    for kv in sys.argv[5:]:
        k, v = kv.split("="); 
        if impl == 1: m.setOption(k, int(v))
    m.dropoutRateFC4Val = 0.0; m.setL2RegularizationLambda(0.0); m.setLearningRate(1e-3); m._dropout_seed = 31
    loss, _ = m.train(x1, y1)
    acts = {}
    for L in (11, 12, 13, 23, 22, 21):
        try: acts[L] = m.getActivation(L, n).cpu().numpy()
        except Exception as e: acts[L] = None; print(name, L, e)
    out[name] = acts; m.close()
fa = O.forward_all(arch, P, x1.cpu().numpy())
for L in (11, 12, 13, 23, 22, 21):
    a, b = out["tile"][L], out["plain"][L]
    if a is None or b is None: continue
    if L < 20:
        print("pool", L - 10, "tile == plain bitwise:", np.array_equal(a, b), " tile == oracle bitwise:", np.array_equal(a, fa["pool%d" % (L - 10)]))
    else:
        d = np.abs(a - b); mx = np.abs(b).max()
        bad = np.argwhere(d > 1e-5 * mx)
        print("gpre", L - 20, "max |tile - plain| / max|plain| = %.2e" % (d.max() / mx), "elements off:", len(bad))
        for ix in bad[:6]:
            ix = tuple(ix); pre = fa["pre%d" % (L - 20)][ix]
            print("    ", ix, "tile %.6g plain %.6g ratio %.4f  oracle pre %.6g" % (a[ix], b[ix], a[ix] / b[ix] if b[ix] else float('nan'), pre))
if len(bad):
    L = 22
    a, b = out["tile"][L], out["plain"][L]
    d = np.abs(a - b); bad = np.argwhere(d > 1e-5 * np.abs(b).max())
    for ix in bad[:4]:
        ix = tuple(ix)
        print(ix, "pre", float(fa["pre2"][ix]).hex(), "act", float(fa["act2"][ix]).hex())
    # the oracle's own pre-activation gradient is not exported: compare weight gradients instead
    l_or, parts, g_or = O.loss_grad(arch, P, x1.cpu().numpy(), y1.cpu().numpy(), lam=0.0)
    def flat(m, which):
        t = torch.empty(m.numParameters, device="cuda")
        _lib.check(m._lib.cv_flat_copy(m._h, which, ctypes.c_void_p(t.data_ptr()), 0, None))
        torch.cuda.synchronize(); return t.cpu().numpy().copy()
    for name, impl in (("tile", 1), ("plain", 0)):
        m = mod.Clairvoyante(); m.setParameters(P); m.setOption("impl", impl); m.setOption("train_ksplit", 0)
        m.dropoutRateFC4Val = 0.0; m.setL2RegularizationLambda(0.0); m.setLearningRate(1e-3)
        m.train(x1, y1); g = flat(m, 1); m.close()
        off = 0; worst = {}
        for pn in O.PARAM_NAMES:
            sz = g_or[pn].size
            e = np.abs(g[off:off + sz] - g_or[pn].ravel()).max() / (np.abs(g_or[pn]).max() + 1e-30); off += sz
            if e > 2e-5: worst[pn] = "%.1e" % e
        print(name, "vs oracle:", worst or "ok")
