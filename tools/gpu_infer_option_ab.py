"""Development probe: cv_forward at a list of batch sizes (in GROUPS of 16 candidates) under several option settings --
per-stage HIP-event times (option "profile") and the bits of the 16 outputs against the FIRST setting.
usage: gpu_infer_option_ab.py full|slim sizes=g1,g2,... "infer_flat=0" "infer_flat=2" "infer_flat=1,dense_rag=-1" ..."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from clairvoyante_amd import _lib, clairvoyante_v3, clairvoyante_v3_slim, synth


def stages(m, xs, out, reps=20):
    ms = (ctypes.c_double * 6)(); cnt = (ctypes.c_int64 * 6)()
    for _ in range(3):
        m.predict_device(xs, out)
    m.setOption("profile", 1)
    _lib.check(m._lib.cv_kernel_times(m._h, ms, cnt))
    for _ in range(reps):
        m.predict_device(xs, out)
    _lib.check(m._lib.cv_kernel_times(m._h, ms, cnt))
    m.setOption("profile", 0)
    res = []
    for s in range(6):
        kn = ctypes.c_char_p()
        _lib.check(m._lib.cv_kernel_name(m._h, s, ctypes.byref(kn)))
        if cnt[s]:
            res.append(((kn.value or b"?").decode().split("<")[0], ms[s] / cnt[s] * 1e3))
    return res


def main():
    arch = sys.argv[1]
    groups = (257, 320, 400, 512, 625, 768, 1024, 1536, 2048, 2049, 2500, 3072, 4096)
    settings = []
    for a in sys.argv[2:]:
        if a.startswith("sizes="):
            groups = tuple(int(t) for t in a[6:].split(","))
        else:
            settings.append([(kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",")])
    m = (clairvoyante_v3 if arch == "full" else clairvoyante_v3_slim).Clairvoyante()
    m.init()
    defaults = {}
    for st in settings:
        for k, _v in st:
            if k not in defaults:
                v = ctypes.c_int64()
                _lib.check(m._lib.cv_get_option(m._h, k.encode(), ctypes.byref(v)))
                defaults[k] = int(v.value)
    x = synth.make_candidates(65536, seed=1, device="cuda")
    for G in groups:
        n = G * 16 - 5
        xs = x[:n].contiguous()
        ref = None
        for st in settings:
            for k, v in defaults.items():
                m.setOption(k, v)
            for k, v in st:
                m.setOption(k, v)
            out = torch.zeros((n, 16), device="cuda")
            res = stages(m, xs, out)
            if ref is None:
                ref = out; tag = "(reference)"
            else:
                d = (ref.view(torch.int32) != out.view(torch.int32)).any(dim=1)
                tag = "bits same" if not bool(d.any()) else "DIFFER in %d rows, first %d" % (int(d.sum()), int(d.nonzero()[0]))
            print("G=%5d %-40s total %7.1f | %s | %s" % (G, ",".join("%s=%d" % kv for kv in st), sum(t for _k, t in res),
                                                      " ".join("%s %.1f" % kt for kt in res), tag))
            sys.stdout.flush()


if __name__ == "__main__":
    main()
