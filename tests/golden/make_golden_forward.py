"""Writes tests/golden/forward_{full,slim}.npz: seeded inputs and the float64 outputs of the
independent torch formulation (tests/torch_ref.py) -- the committed pin of the oracle's
arithmetic (the reference has no golden vectors: SURVEY.md 4).  Weights are regenerated
from the seed (numpy RandomState is stable), so the fixture stays small.
    python tests/golden/make_golden_forward.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import common
import torch_ref
from oracle import cv_oracle as O

for arch in ("full", "slim"):
    seed = 7
    P = common.bench_params(O, arch, seed=seed)
    x = common.inputs(28, seed=21, stress=4)
    r = torch_ref.forward(arch, P, x)
    np.savez_compressed(os.path.join(HERE, "forward_%s.npz" % arch), seed=seed, x=x.astype(np.float32),
                        out64=r["out"].numpy(), pool3_64=r["pool3"].numpy()[:4], fc5_64=r["fc5"].numpy())
    print(arch, "ok", r["out"].shape)
