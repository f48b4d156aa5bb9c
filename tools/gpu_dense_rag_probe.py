"""Development probe / calibration of dense_rag (fc4's three-slab form on ragged waves, csrc/cv_kernels_mfma.hip): HIP-event
time of the fc4 stage of cv_forward over a ladder of batch sizes, for the round-5 kernel (option dense_rag -1), every forced
shape s = 4..14 and the shape the launcher's formula picks (0); the 16 outputs of every run must equal the round-5
kernel's bit for bit.  usage: gpu_dense_rag_probe.py [sizes=g1,g2,...] [shapes=s1,s2,...]   (sizes in GROUPS of 16)"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from clairvoyante_amd import _lib, clairvoyante_v3, synth


def stage_us(m, xs, out, reps=20):
    ms = (ctypes.c_double * 6)(); cnt = (ctypes.c_int64 * 6)()
    for _ in range(3):
        m.predict_device(xs, out)
    m.setOption("profile", 1)
    _lib.check(m._lib.cv_kernel_times(m._h, ms, cnt))
    for _ in range(reps):
        m.predict_device(xs, out)
    _lib.check(m._lib.cv_kernel_times(m._h, ms, cnt))
    m.setOption("profile", 0)
    kn = ctypes.c_char_p()
    _lib.check(m._lib.cv_kernel_name(m._h, 3, ctypes.byref(kn)))
    return ms[3] / cnt[3] * 1e3, (kn.value or b"?").decode()


def main():
    groups = (289, 320, 400, 512, 625, 681, 682, 768, 1024, 1365, 1536, 2048, 2049, 2500, 3072, 3585, 4096)
    shapes = tuple(range(4, 15))
    for kv in sys.argv[1:]:
        k, v = kv.split("=")
        if k == "sizes":
            groups = tuple(int(t) for t in v.split(","))
        if k == "shapes":
            shapes = tuple(int(t) for t in v.split(","))
    m = clairvoyante_v3.Clairvoyante()
    m.init()
    m.setOption("infer_slab_groups", 65536)
    x = synth.make_candidates(65536, seed=1, device="cuda")
    for G in groups:
        n = G * 16 - 5            # ragged last group
        xs = x[:n].contiguous()
        ref = torch.empty((n, 16), device="cuda"); out = torch.empty((n, 16), device="cuda")
        m.setOption("dense_rag", -1)
        t_old, k_old = stage_us(m, xs, ref)
        m.setOption("dense_rag", 0)
        t_f, k_f = stage_us(m, xs, out)
        bad = []
        if not torch.equal(ref.view(torch.int32), out.view(torch.int32)):
            bad.append("formula:%d" % int((ref.view(torch.int32) != out.view(torch.int32)).any(dim=1).sum()))
        row = []
        for s in shapes:
            m.setOption("dense_rag", s)
            out.zero_()
            t, _k = stage_us(m, xs, out, reps=10)
            if not torch.equal(ref.view(torch.int32), out.view(torch.int32)):
                d = (ref.view(torch.int32) != out.view(torch.int32)).any(dim=1)
                bad.append("s=%d:%d rows, first %d" % (s, int(d.sum()), int(d.nonzero()[0])))
            wgs = 3 * ((7 * G + 4 * s - 1) // (4 * s))
            row.append("s=%d:%.0f(%dr)" % (s, t, (wgs + 255) // 256))
        print("G=%5d  %s %.1f us | formula %s %.1f us | %s | bits %s" % (G, k_old, t_old, k_f, t_f, " ".join(row), "same" if not bad else "DIFFER " + "; ".join(bad)))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
