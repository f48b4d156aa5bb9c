"""Development probe: latency / throughput of cv_forward for small batches (the reference's default is 1 000)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common
from oracle import cv_oracle as O
from clairvoyante_amd import clairvoyante_v3, synth

m = clairvoyante_v3.Clairvoyante()
m.setParameters(common.bench_params(O, "full"))
x = synth.make_candidates(65536, seed=1, device="cuda")
out = torch.empty((65536, 16), device="cuda")
import ctypes
from clairvoyante_amd import _lib
variant = int(sys.argv[1]) if len(sys.argv) > 1 else None
if variant is not None:
    m.setOption("variant", variant)
print("variant", variant)
for n in (16, 256, 1000, 2560, 4096, 16384, 65536):
    xs = x[:n].contiguous(); os_ = out[:n]
    for _ in range(20):
        m.predict_device(xs, os_)
    torch.cuda.synchronize()
    reps = 300 if n <= 4096 else 60
    t0 = time.perf_counter()
    for _ in range(reps):
        m.predict_device(xs, os_)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    xh = xs.cpu().numpy()
    t0 = time.perf_counter()
    for _ in range(20):
        m.predict(xh)
    dh = (time.perf_counter() - t0) / 20
    print("n=%6d: device call %8.1f us -> %6.2f M cand/s | predict(numpy) %8.1f us -> %6.2f M cand/s" % (
        n, dt * 1e6, n / dt / 1e6, dh * 1e6, n / dh / 1e6))
    m.setOption("profile", 1)
    for _ in range(20):
        m.predict_device(xs, os_)
    ms = (ctypes.c_double * 6)(); cnt = (ctypes.c_int64 * 6)()
    _lib.check(m._lib.cv_kernel_times(m._h, ms, cnt))
    m.setOption("profile", 0)
    print("          per kernel (us): " + "  ".join("%s %.1f" % (k, ms[i] / max(cnt[i], 1) * 1e3) for i, k in
                                                  enumerate(("conv1", "conv2", "conv3", "fc4", "fc5", "heads")) if cnt[i]))
