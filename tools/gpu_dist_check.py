"""N=1 exercise of the RCCL code paths (development tool): bench barrier/all-reduce and one
data-parallel training step with CV_FORCE_DIST=1 under torchrun --nproc-per-node 1."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, torch.distributed as dist
from clairvoyante_amd import parallel, clairvoyante_v3_slim, synth
rank, ws, local = parallel.init_from_env()
print("dist initialised:", dist.is_initialized(), dist.get_backend(), rank, ws, flush=True)
m = clairvoyante_v3_slim.Clairvoyante(); m.init()
parallel.broadcast_parameters(m)
xt, cls, rf, alt, il = synth.make_candidates(512, seed=3, return_class=True)
y = synth.make_labels(cls, rf, alt, il).numpy(); x = xt.numpy()
l0 = float(m.getLoss(x, y))
for _ in range(5):
    loss, s = m.train(x, y)
print("loss", l0, "->", float(m.getLoss(x, y)), "allreduce scalar", parallel.allreduce_scalar(1.5, m), flush=True)
dist.barrier(); dist.destroy_process_group(); print("DIST OK")
