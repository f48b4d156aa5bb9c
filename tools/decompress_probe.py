"""Development probe: time of utils_v2.DecompressArray (X + Y) per 10 000-item batch, the host side of train.py's loop."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from clairvoyante_amd import _lib, utils_v2

rng = np.random.RandomState(0)
total = 40000
X = rng.randint(-30, 60, size=(total, 33, 4, 4)).astype(np.float32)
X[rng.rand(*X.shape) < 0.7] = 0
Y = np.zeros((total, 16)); Y[:, 0] = 1
XC = [utils_v2.pack_array(X[s:s + 500]) for s in range(0, total + 1, 500)]
YC = [utils_v2.pack_array(Y[s:s + 500]) for s in range(0, total + 1, 500)]
print("usable cores", _lib.usable_cores(), "compressed %.1f of %.1f MB" % (sum(len(b) for b in XC) / 1e6, X.nbytes / 1e6))
for threads in (1, 4, 16):
    _lib.load().cv_set_host_threads(threads)
    best = 1e9
    for rep in range(4):
        t = time.time()
        for p in range(0, 30000, 10000):
            a, n, e = utils_v2.DecompressArray(XC, p, 10000, total)
            b, n2, e2 = utils_v2.DecompressArray(YC, p, 10000, total)
        best = min(best, (time.time() - t) / 3)
    print("threads %2d: %.1f ms per 10 000-item batch -> %.2f M candidates/s" % (threads, best * 1e3, 10000 / best / 1e6))
