"""Development check: one cv_forward call over millions of candidates (internally chunked) equals batch-by-batch calls."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common
from oracle import cv_oracle as O
from clairvoyante_amd import clairvoyante_v3, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000000
m = clairvoyante_v3.Clairvoyante()
m.setParameters(common.bench_params(O, "full"))
x = torch.cat([synth.make_candidates(500000, seed=s, device="cuda") for s in range((n + 499999) // 500000)])[:n].contiguous()
torch.cuda.synchronize()
t0 = time.time()
a = m.predict_device(x)
torch.cuda.synchronize()
dt = time.time() - t0
b = torch.cat([m.predict_device(x[s:s + 65536]) for s in range(0, n, 65536)])
print("n=%d one call %.3f s = %.2f M cand/s; equal to per-batch calls: %s" % (n, dt, n / dt / 1e6, bool(torch.equal(a, b))))
