"""Development probe: rows/s of cv_parse_tensor_text (utils_v2.GetTensor's per-row work) by host thread count."""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from clairvoyante_amd import _lib

lib = _lib.load()
n = 40000
vals = np.random.RandomState(1).randint(0, 60, size=(528,)).astype(np.float32)
row = ("chr1 12345 " + "ACGT" * 8 + "A " + " ".join("%0.1f" % v for v in vals) + "\n").encode()
text = row * n
x = np.empty((n, 528), np.float32); meta = np.zeros((n, 6), np.int64)
c = ctypes.c_int64(); r = ctypes.c_int64(); b = ctypes.c_int64()
print("usable cores", _lib.usable_cores())
for T in (1, 2, 4, 8, 16):
    lib.cv_set_host_threads(T)
    best = 1e9
    for _ in range(3):
        t = time.time()
        lib.cv_parse_tensor_text(text, len(text), n, x.ctypes.data_as(ctypes.c_void_p), meta.ctypes.data_as(ctypes.c_void_p),
                                 ctypes.byref(c), ctypes.byref(r), ctypes.byref(b))
        best = min(best, time.time() - t)
    print("threads %2d: %8.0f rows/s  %6.0f MB/s" % (T, r.value / best, len(text) / best / 1e6))
