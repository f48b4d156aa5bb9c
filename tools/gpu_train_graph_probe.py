"""Development probe: would a captured graph of the training step help?  Captures ONE optimizer step (cv_grad_async on
two streams + Adam + loss accumulation) with torch.cuda.graph and replays it; timing only -- the dropout step counter
and the Adam bias correction are baked into the capture, so this is not a way to train.
usage: gpu_train_graph_probe.py [batch ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from clairvoyante_amd import clairvoyante_v3, synth

batches = [int(a) for a in sys.argv[1:]] or [1250, 2500, 10000]
for n in batches:
    m = clairvoyante_v3.Clairvoyante()
    m._seed_rng.seed(1); m.init()
    xt, cls, rf, alt, il = synth.make_candidates(n, seed=3, device="cuda", return_class=True)
    y = synth.make_labels(cls, rf, alt, il)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(5):
            m.trainDeferred(xt, y)
    torch.cuda.synchronize()
    reps = 200
    with torch.cuda.stream(s):
        t0 = time.perf_counter()
        for _ in range(reps):
            m.trainDeferred(xt, y)
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / reps
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            m.trainDeferred(xt, y)
        torch.cuda.synchronize()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / reps
        print("batch %6d: stream launches %.3f ms / step, graph replay %.3f ms / step" % (n, eager * 1e3, graph * 1e3), flush=True)
    except Exception as e:
        print("batch %6d: stream launches %.3f ms / step, capture failed: %s" % (n, eager * 1e3, str(e)[:300]), flush=True)
    m.close()
