"""Development probe: cv_forward over a ladder of batch sizes with the per-stage HIP-event times of option "profile"
(cv_kernel_times) next to the wall time per call -- shows WHICH kernel's launch shape makes the time of a call step
between two sizes, and how far each size is from time proportional to the largest one.
usage: gpu_infer_stage_ladder.py [full|slim] [option=value ...] [sizes=n1,n2,...]"""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from clairvoyante_amd import _lib, clairvoyante_v3, clairvoyante_v3_slim, synth

NUM_STAGES = 6      # CV_NUM_STAGES of include/clairvoyante_amd.h
LADDER = (1000, 1600, 2000, 2560, 2576, 3200, 4096, 4112, 5120, 6400, 8192, 10000, 12288, 16384, 24576, 32768, 32784,
          40000, 49152, 65536)


def main():
    arch = sys.argv[1] if len(sys.argv) > 1 else "full"
    m = (clairvoyante_v3 if arch == "full" else clairvoyante_v3_slim).Clairvoyante()
    m.init()
    sizes, opts = LADDER, []
    for kv in sys.argv[2:]:
        k, v = kv.split("=")
        if k == "sizes":
            sizes = tuple(int(t) for t in v.split(","))
        else:
            m.setOption(k, int(v)); opts.append(kv)
    x = synth.make_candidates(65536, seed=1, device="cuda")
    out = torch.empty((65536, 16), device="cuda")
    print("%s %s" % (arch, " ".join(opts) or "default"))
    rows = []
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
        m.setOption("profile", 1)
        ms = (ctypes.c_double * NUM_STAGES)(); cnt = (ctypes.c_int64 * NUM_STAGES)()
        _lib.check(m._lib.cv_kernel_times(m._h, ms, cnt))      # reset
        for _ in range(20):
            m.predict_device(xs, os_)
        _lib.check(m._lib.cv_kernel_times(m._h, ms, cnt))
        m.setOption("profile", 0)
        st = []
        for s in range(NUM_STAGES):
            kn = ctypes.c_char_p()
            _lib.check(m._lib.cv_kernel_name(m._h, s, ctypes.byref(kn)))
            if cnt[s]:
                st.append("%s %.1f" % ((kn.value or b"?").decode(), ms[s] / cnt[s] * 1e3))
        rows.append((n, best))
        print("n=%6d (%5d groups): %8.1f us per call  %6.1f ns per candidate | %s" % (n, (n + 15) // 16, best * 1e6, best / n * 1e9, " | ".join(st)))
    # the two size-robustness figures: time against the proportional share of the largest size, and adjacent pairs
    nl, tl = rows[-1]
    print("-- against time proportional to n = %d (%.1f us):" % (nl, tl * 1e6))
    worst_pair = 0.0
    for i, (n, t) in enumerate(rows):
        prop = t / (tl * n / nl)
        pair = ""
        if i > 0:
            n0, t0 = rows[i - 1]
            r = (t / t0) / (n / n0)
            worst_pair = max(worst_pair, r)
            pair = "  step from the size before / work ratio %.3f" % r
        print("n=%6d  time / proportional %.3f%s" % (n, prop, pair))
    print("worst adjacent pair %.3f; worst time / proportional for n >= 16384: %.3f" % (
        worst_pair, max(t / (tl * n / nl) for n, t in rows if n >= 16384)))


if __name__ == "__main__":
    main()
