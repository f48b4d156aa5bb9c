"""records/s of the VCF formatter: cv_format_vcf (native, host threads) against the CPython loop it replaces
(callVar._format_record); same inputs, identical output checked.  usage: python tools/vcf_format_probe.py [n] [threads]"""
import os
import sys
import time
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clairvoyante_amd import callVar, _lib  # noqa: E402
from clairvoyante_amd.utils_v2 import PosBatch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
lib = _lib.load()
threads = int(sys.argv[2]) if len(sys.argv) > 2 else min(_lib.usable_cores(), 16)
lib.cv_set_host_threads(threads)
rng = np.random.RandomState(1)
X = rng.randint(0, 40, size=(n, 33, 4, 4)).astype(np.float32)
call = np.zeros((n, 8), np.int32)
call[:, 0] = rng.choice([1, 1, 1, 2, 3], n); call[:, 1] = rng.randint(0, 2, n); call[:, 2] = rng.randint(0, 6, n)
call[:, 3] = rng.randint(0, 4, n); call[:, 4] = (call[:, 3] + 1) % 4
q = np.zeros((n, 4), np.float32); q[:, 0] = 0.9; q[:, 1] = rng.rand(n) * 0.5; q[:, 2] = callVar._depth(X)
seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=(n, 33))
pos = PosBatch.from_columns("chr20", rng.randint(1, 60000000, n), [bytes(r) for r in seq])
args = types.SimpleNamespace(showRef=False, qual=None)
callVar.format_records(args, 4096, X, pos, call, q)
t0 = time.perf_counter(); text = callVar.format_records(args, n, X, pos, call, q); dt = time.perf_counter() - t0
m = min(n, 20000)
t0 = time.perf_counter()
lines = [callVar._format_record(args, X[i], pos[i], int(call[i, 0]), int(call[i, 1]), int(call[i, 2]), int(call[i, 3]),
                                int(call[i, 4]), callVar._qual(q[i, 0], q[i, 1]), q[i, 2]) for i in range(m)]
dp = time.perf_counter() - t0
same = text.decode().splitlines()[:m] == lines
print("cv_format_vcf: %d records in %.3f s = %.2f M records/s on %d threads (%.0f MB/s of text); CPython loop: %.0f records/s "
      "(%d records); speed-up %.0fx; identical text: %s" % (n, dt, n / dt / 1e6, threads, len(text) / dt / 1e6, m / dp, m,
                                                           (n / dt) / (m / dp), same))
