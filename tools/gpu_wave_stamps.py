"""When and where the waves of the instrumented training kernels run (development tool; needs a library built with
CV_EXTRA_FLAGS=-DCV_WG_STAMP).  A few training steps with the streams serialized (or N side streams), then, per kernel id,
the stamps of its newest launch: start spread, end spread, wave lifetimes, waves per SIMD.
python tools/gpu_wave_stamps.py [batch] [side streams, 0 = serialized] [npz to write]"""
import collections, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from clairvoyante_amd import clairvoyante_v3, synth, _lib
NAMES = ["wgrad_conv_cm conv3", "conv3_rot train", "conv_tm dgrad conv3", "dense_tm fc4 train", "dense_dgrad_unpool", "wgrad_dense_cm fc4", "conv_tm conv2 fwd", "-"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
sides = int(sys.argv[2]) if len(sys.argv) > 2 else 0
m = clairvoyante_v3.Clairvoyante(); m.init()
m.setOption("train_side_streams", sides) if sides else m.setOption("train_overlap", 0)
xt, cls, rf, alt, il = synth.make_candidates(n, seed=3, device="cuda", return_class=True)
y = synth.make_labels(cls, rf, alt, il)
for _ in range(4):
    m.train(xt, y)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros(8 * 4096 * 4, dtype=np.uint64); cnt = np.zeros(8, dtype=np.uint32)
assert lib.cv_debug_wg_stamps(buf.ctypes.data_as(ctypes.c_void_p), cnt.ctypes.data_as(ctypes.c_void_p)) == 0
allst = buf.reshape(8, 4096, 4)
if len(sys.argv) > 3:
    np.savez_compressed(sys.argv[3], stamps=allst, counts=cnt)
us = lambda v: v / 100.0
print("batch %d  side streams %d" % (n, sides))
for kid in range(8):
    if cnt[kid] == 0: continue
    s = allst[kid]; s = s[s[:, 1] > 0]
    order = np.argsort(s[:, 0]); s = s[order]
    t0 = s[:, 0].astype(np.int64)
    gaps = np.nonzero(np.diff(t0) > 50000)[0]             # launches of one kernel are a step (> 0.5 ms) apart
    s = s[gaps[-1] + 1:] if len(gaps) else s
    t0 = s[:, 0].astype(np.int64); t1 = s[:, 1].astype(np.int64); cyc = s[:, 2].astype(np.int64)
    hw = (s[:, 3] & np.uint64(0xffffffff)).astype(np.int64); xcc = (s[:, 3] >> np.uint64(32)).astype(np.int64) & 0xf
    base = t0.min(); life = us(t1 - t0)
    print("== %s: waves %d" % (NAMES[kid], len(s)))
    print("   start  median %.1f  p90 %.1f  last %.1f us | end  first %.1f  median %.1f  p90 %.1f  last %.1f us" % (
        tuple(us(np.percentile(t0 - base, q)) for q in (50, 90, 100)) + tuple(us(np.percentile(t1 - base, q)) for q in (0, 50, 90, 100))))
    print("   life   min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f us ; shader clock %.3f GHz" % (
        tuple(np.percentile(life, q) for q in (0, 10, 50, 90, 100)) + (np.median(cyc / (life * 1e3)),)))
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    key = xcc * 100000 + se * 10000 + sh * 1000 + cu * 10 + simd
    # waves resident together: count the other waves of the same SIMD whose lifetime covers this wave's midpoint
    mid = (t0 + t1) // 2
    per = collections.defaultdict(list)
    for i, k in enumerate(key.tolist()): per[k].append(i)
    co = np.zeros(len(s), dtype=np.int64)
    for k, idx in per.items():
        for i in idx:
            co[i] = sum(1 for j in idx if t0[j] <= mid[i] <= t1[j])
    print("   SIMDs used %d, CUs used %d ; waves sharing their SIMD at mid-life: %s" % (len(per), len(set((key // 10).tolist())), dict(sorted(collections.Counter(co.tolist()).items()))))
    late = t0 - base > 500
    if late.any():
        print("   waves starting > 5 us after the first: %d (start median %.1f us, life median %.1f us)" % (late.sum(), us(np.median(t0[late] - base)), np.median(life[late])))
m.close()
