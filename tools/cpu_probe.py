"""Host-core probe for the cpu_baseline leg (development tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, common
from oracle import cv_oracle as O
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(f): print(f, open(f).read().strip())
os.system("lscpu | grep -E 'Model name|Socket|Core|Thread|NUMA node\\(s\\)' | head -8")
P = common.bench_params(O, "full"); x = common.inputs(16384, seed=3)
O.predict("full", P, x[:1024], nthreads=8)
for nt in (8, 16, 32, 64, 128, 256):
    t = time.time(); O.predict("full", P, x, nthreads=nt); dt = time.time() - t
    print("threads %3d: %.0f cand/s" % (nt, x.shape[0] / dt), flush=True)
