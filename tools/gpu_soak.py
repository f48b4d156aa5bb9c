"""Development check: repeated inference / training / pileup passes; device memory must not grow."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common
from oracle import cv_oracle as O
from clairvoyante_amd import clairvoyante_v3, synth, synth_pileup
from clairvoyante_amd._lib import check
from clairvoyante_amd.pileup import Pileup


def free_mb():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / 1e6


m = clairvoyante_v3.Clairvoyante()
m.setParameters(common.bench_params(O, "full"))
x = synth.make_candidates(65536, seed=1, device="cuda")
xt, cls, rf, alt, il = synth.make_candidates(10000, seed=2, device="cuda", return_class=True)
y = synth.make_labels(cls, rf, alt, il)
ref, text = synth_pileup.fast_alignments(300000, 1500000)
pl = Pileup(evc=True, retain=True, contig="ctgA")
pl.set_reference(ref, 0)
pl.add_sam(text)
pl.extract_candidates(0.06, 4); pl.adopt_candidates(); pl.finish(subtract=True)
out0 = m.predict_device(x).clone()
for phase in range(3):
    f0 = free_mb(); t0 = time.time()
    for _ in range(1500):
        out = m.predict_device(x)
    assert torch.equal(out, out0)
    for _ in range(600):
        loss, _s = m.train(xt, y)
    assert np.isfinite(loss)
    for _ in range(150):
        check(pl.lib.cv_pileup_recount(pl.h, pl._stream()))
        pl.extract_candidates(0.06, 4); pl.adopt_candidates(); t, d, u = pl.finish(subtract=True)
    m2 = clairvoyante_v3.Clairvoyante(); m2.init(); m2.predict_device(x[:1000]); m2.close()
    p2 = Pileup(); p2.set_reference(ref, 0); p2.set_candidates([100, 200]); p2.add_sam(text[:1 << 20]); p2.finish(); p2.close()
    print("phase %d: %.1f s, free memory %.0f -> %.0f MB, loss %.1f" % (phase, time.time() - t0, f0, free_mb(), loss), flush=True)
    out0 = m.predict_device(x).clone()      # weights changed by training
print("SOAK OK")
