"""Training-step throughput (development tool): train.py's batch of 10 000 (param.trainBatchSize)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from clairvoyante_amd import clairvoyante_v3, clairvoyante_v3_slim, synth
for arch in ("full", "slim"):
    m = clairvoyante_v3.Clairvoyante() if arch == "full" else clairvoyante_v3_slim.Clairvoyante()
    m.init()
    n = 10000
    xt, cls, rf, alt, il = synth.make_candidates(n, seed=3, device="cuda", return_class=True)
    y = synth.make_labels(cls, rf, alt, il)
    for _ in range(2):
        m.train(xt, y)
    torch.cuda.synchronize(); t0 = time.time()
    reps = 5
    for _ in range(reps):
        loss, s = m.train(xt, y)
    torch.cuda.synchronize(); dt = (time.time() - t0) / reps
    t0 = time.time()
    for _ in range(reps):
        m.getLoss(xt, y)
    torch.cuda.synchronize(); dl = (time.time() - t0) / reps
    print(arch, "train step %.1f ms = %.0f cand/s ; getLoss %.1f ms = %.0f cand/s ; loss %.1f" % (dt * 1e3, n / dt, dl * 1e3, n / dl, loss), flush=True)
    m.close()
