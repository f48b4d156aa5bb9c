"""Differential soak of cv_forward: seeded random passes (size log-uniform over 1 .. NMAX) of transformed synthetic pileups --
scaled by 1e-3 .. 1e3, position ranges or whole candidates zeroed, -0.0 sprinkled, a few candidates holding NaN / +-Inf /
denormals / 1e30 -- through the tile kernels (whatever launch shapes the size selects) and the plain one-thread-per-output
kernels; the 16 outputs per candidate must agree BITWISE (NaN = any NaN).  Zero-bias weights every third pass (exact-zero
pre-activations).  usage: gpu_infer_fuzz.py full|slim ROUNDS [NMAX] [seed0]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import common
from clairvoyante_amd import clairvoyante_v3_slim, clairvoyante_v3, synth
arch = sys.argv[1]; rounds = int(sys.argv[2]); nmax = int(sys.argv[3]) if len(sys.argv) > 3 else 70000
seed0 = int(sys.argv[4]) if len(sys.argv) > 4 else 0
mod = clairvoyante_v3_slim if arch == "slim" else clairvoyante_v3
P = common.bench_params(None, arch)
P0 = {k: (np.zeros_like(v) if "bias" in k else v) for k, v in P.items()}
M = {}
for name, impl in (("tile", 1), ("plain", 0)):
    m = mod.Clairvoyante(); m.setOption("impl", impl); M[name] = m
rng = np.random.RandomState(seed0)
bad = 0; total = 0; t0 = time.time(); cur = None
for r in range(rounds):
    n = int(np.exp(rng.uniform(0.0, np.log(float(nmax))))); seed = int(rng.randint(1, 1 << 30))
    x = synth.make_candidates(n, seed=seed, device="cuda")
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    kind = r % 5
    if kind == 1: x = x * float(10.0 ** rng.uniform(-3, 3))
    if kind == 2:
        keep = (torch.rand((n, 33, 1, 1), device="cuda", generator=g) > 0.4).float(); x = x * keep
        x[torch.rand(n, device="cuda", generator=g) < 0.1] = 0.0
    if kind == 3:
        x = torch.where(torch.rand(x.shape, device="cuda", generator=g) < 0.05, torch.full_like(x, -0.0), x)
    if kind == 4 and n >= 4:
        specials = torch.tensor([float("nan"), float("inf"), -float("inf"), 1e-40, -1e-40, 1e30, -1e30, 3.4e38], device="cuda")
        k = max(1, n // 50)
        ci = torch.randint(0, n, (k,), device="cuda", generator=g)
        pi = torch.randint(0, 33 * 16, (k,), device="cuda", generator=g)
        x.view(n, -1)[ci, pi] = specials[torch.randint(0, len(specials), (k,), device="cuda", generator=g)]
    x = x.contiguous()
    want = P0 if r % 3 == 2 else P
    if cur is not want:
        for m in M.values(): m.setParameters(want)
        cur = want
    out = {name: m.predict_device(x).clone() for name, m in M.items()}
    a, b = out["tile"], out["plain"]
    same = (a.view(torch.int32) == b.view(torch.int32)) | (torch.isnan(a) & torch.isnan(b))
    total += n
    if not bool(same.all()):
        bad += 1
        rows = (~same).any(dim=1).nonzero().flatten()
        print("DIFF %s n=%d seed=%d kind=%d zero_bias=%d: %d candidates differ, first %d: tile %s plain %s" % (
            arch, n, seed, kind, want is P0, rows.numel(), int(rows[0]), a[rows[0]].tolist(), b[rows[0]].tolist()), flush=True)
print("%s: %d passes, %d candidates, %d differ; %.0f s" % (arch, rounds, total, bad, time.time() - t0))
