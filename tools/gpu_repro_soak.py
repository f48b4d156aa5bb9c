"""Development check: the inference kernels give the same bits on every one of many repeated launches (full and slim,
large and small passes) -- a race in the barrier / LDS-DMA choreography of front2_tm, conv3fc4_slim or dense_tm would
show as an occasional differing output."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common
from oracle import cv_oracle as O
from clairvoyante_amd import clairvoyante_v3, clairvoyante_v3_slim, synth

for arch, cls in (("full", clairvoyante_v3), ("slim", clairvoyante_v3_slim)):
    m = cls.Clairvoyante()
    m.setParameters(common.bench_params(O, arch))
    for n, reps in ((65536, 400), (40010, 300), (1000, 2000), (2560, 1000), (17, 2000)):
        x = synth.make_candidates(n, seed=n, device="cuda")
        ref = m.predict_device(x).clone()
        bad = 0
        for _ in range(reps):
            out = m.predict_device(x)
            bad += 0 if torch.equal(out, ref) else 1
        print("%s n=%6d: %d launches, %d differ" % (arch, n, reps, bad), flush=True)
        assert bad == 0
    m.close()
print("REPRO SOAK OK")
