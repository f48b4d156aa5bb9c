"""Development probe: cv_forward over a ladder of batch sizes, device-resident input -- looks for steps in the time per
call where the library changes kernels by size (options infer_small_groups / infer_fc4_small_groups / infer_slab_groups).
usage: gpu_infer_size_sweep.py [full|slim] [option=value ...] [sizes=n1,n2,...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from clairvoyante_amd import clairvoyante_v3, clairvoyante_v3_slim, synth

arch = sys.argv[1] if len(sys.argv) > 1 else "full"
m = (clairvoyante_v3 if arch == "full" else clairvoyante_v3_slim).Clairvoyante()
m.init()
sizes = (1000, 1600, 2000, 2560, 2576, 3200, 4096, 4112, 5120, 6400, 8192, 10000, 12288, 16384, 24576, 32768, 32784, 40000, 49152, 65536)
opts = []
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    if k == "sizes":
        sizes = tuple(int(t) for t in v.split(","))
    else:
        m.setOption(k, int(v)); opts.append(kv)
x = synth.make_candidates(65536, seed=1, device="cuda")
out = torch.empty((65536, 16), device="cuda")
print("%s %s" % (arch, " ".join(opts) or "default"))
for n in sizes:
    xs = x[:n].contiguous(); os_ = out[:n]
    for _ in range(10):
        m.predict_device(xs, os_)
    torch.cuda.synchronize()
    reps = 200 if n <= 8192 else 50
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            m.predict_device(xs, os_)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    print("n=%6d (%5d groups): %8.1f us per call  %6.2f M cand/s  %6.1f ns per candidate" % (n, (n + 15) // 16, best * 1e6, n / best / 1e6, best / n * 1e9))
