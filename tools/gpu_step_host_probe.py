"""Is the optimizer step bound by the host that enqueues it?  (development tool)  For each batch: seconds the host needs
to ENQUEUE a step (trainDeferred returns without synchronising) against the seconds the device needs to run it.
python tools/gpu_step_host_probe.py [batches...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from clairvoyante_amd import clairvoyante_v3, clairvoyante_v3_slim, synth
for arch in ("full", "slim"):
    for n in [int(a) for a in sys.argv[1:]] or [1250, 2500, 10000]:
        m = clairvoyante_v3.Clairvoyante() if arch == "full" else clairvoyante_v3_slim.Clairvoyante()
        m._seed_rng.seed(1234); m.init()
        xt, cls, rf, alt, il = synth.make_candidates(n, seed=3, device="cuda", return_class=True)
        y = synth.make_labels(cls, rf, alt, il)
        for _ in range(5):
            m.trainDeferred(xt, y)
        torch.cuda.synchronize()
        K = 40
        t0 = time.perf_counter()
        for _ in range(K):
            m.trainDeferred(xt, y)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        # the same steps one at a time: device time of a step with an idle queue in front of it
        ts = []
        for _ in range(10):
            torch.cuda.synchronize(); a = time.perf_counter(); m.trainDeferred(xt, y); b = time.perf_counter(); torch.cuda.synchronize(); c = time.perf_counter()
            ts.append((b - a, c - a))
        print("%s batch %5d: host enqueue %.3f ms per step, %d steps back to back %.3f ms per step (host %s the device); "
              "one step alone: enqueue %.3f ms, done after %.3f ms" % (
                  arch, n, (t1 - t0) / K * 1e3, K, (t2 - t0) / K * 1e3, "BOUNDS" if (t1 - t0) > 0.9 * (t2 - t0) else "runs ahead of",
                  sorted(t[0] for t in ts)[5] * 1e3, sorted(t[1] for t in ts)[5] * 1e3))
        m.readLosses(); m.close()
