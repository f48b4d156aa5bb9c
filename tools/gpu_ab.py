"""Within-process interleaved A/B timing of kernel variants (development tool).
usage: python tools/gpu_ab.py [arch] ; writes gpurun_out/ab_<arch>.json"""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import common
from oracle import cv_oracle as O
from clairvoyante_amd import clairvoyante_v3, clairvoyante_v3_slim, _lib, synth

arch = sys.argv[1] if len(sys.argv) > 1 else "full"
configs = [dict(variant=v, chunk=65536) for v in (15, 47)]
m = clairvoyante_v3.Clairvoyante() if arch == "full" else clairvoyante_v3_slim.Clairvoyante()
P = common.bench_params(O, arch)
m.setParameters(P)
x = common.inputs(2000, stress=48)
ref = O.predict(arch, P, x)
xd = torch.from_numpy(x).cuda()
res = {"arch": arch, "configs": []}
for cfg in configs:
    for k, v in cfg.items():
        m.setOption(k, v)
    out = m.predict_device(xd).cpu().numpy()
    ok = bool(np.array_equal(out.view(np.uint32), ref.view(np.uint32)))
    print(cfg, "bitwise", ok, "maxabs", float(np.abs(out - ref).max()), flush=True)
    cfg["bitwise"] = ok
N = 65536
xb = [synth.make_candidates(N, seed=100 + i, device="cuda") for i in range(4)]
ob = torch.empty((N, 16), device="cuda")
stats = [dict(cfg=c, wall=[], stages=[]) for c in configs]
for rnd in range(6):
    for i, cfg in enumerate(configs):
        for k, v in cfg.items():
            if k != "bitwise":
                m.setOption(k, v)
        m.predict_device(xb[0], ob); torch.cuda.synchronize()
        m.setOption("profile", 1)
        t0 = time.perf_counter()
        for b in xb:
            m.predict_device(b, ob)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        m.setOption("profile", 0)
        ms = (ctypes.c_double * 6)(); cnt = (ctypes.c_int64 * 6)()
        _lib.check(m._lib.cv_kernel_times(m._h, ms, cnt))
        stats[i]["wall"].append(len(xb) * N / dt)
        stats[i]["stages"].append([ms[s] / max(cnt[s], 1) for s in range(6)])
for st in stats:
    w = np.array(st["wall"]); sg = np.median(np.array(st["stages"]), axis=0)
    st["cand_per_s_median"] = float(np.median(w)); st["cand_per_s_max"] = float(w.max()); st["stage_ms_median"] = sg.tolist()
    print(st["cfg"], "median %.2fM max %.2fM" % (np.median(w) / 1e6, w.max() / 1e6), "stages(ms/launch)", np.round(sg, 4).tolist(), flush=True)
    del st["wall"], st["stages"]
res["configs"] = stats
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "ab_%s.json" % arch), "w"), indent=1)
