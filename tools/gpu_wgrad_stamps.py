"""When and where the waves of the conv3 weight-gradient kernel run (development tool; needs a library built with
CV_EXTRA_FLAGS=-DCV_WG_STAMP).  One training step at train.py's batch with the streams serialized, then the stamps of
the last launch: start spread, end spread, wave lifetimes, waves per SIMD."""
import collections, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from clairvoyante_amd import clairvoyante_v3, synth, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
sides = int(sys.argv[2]) if len(sys.argv) > 2 else 0
m = clairvoyante_v3.Clairvoyante(); m.init()
m.setOption("train_side_streams", sides) if sides else m.setOption("train_overlap", 0)
xt, cls, rf, alt, il = synth.make_candidates(n, seed=3, device="cuda", return_class=True)
y = synth.make_labels(cls, rf, alt, il)
for _ in range(4):
    m.train(xt, y)
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros(4096 * 4, dtype=np.uint64)
rc = lib.cv_debug_wg_stamps(buf.ctypes.data_as(ctypes.c_void_p))
assert rc == 0
s = buf.reshape(4096, 4)
if len(sys.argv) > 3:
    np.save(sys.argv[3], s)
s = s[s[:, 1] > 0]
t0 = s[:, 0].astype(np.int64); t1 = s[:, 1].astype(np.int64); cyc = s[:, 2].astype(np.int64)
hw = (s[:, 3] & np.uint64(0xffffffff)).astype(np.int64); xcc = (s[:, 3] >> np.uint64(32)).astype(np.int64) & 0xf
base = t0.min()
us = lambda v: v / 100.0
print("waves %d  side streams %d" % (len(s), sides))
print("start  first %.2f us  median %.2f  p90 %.2f  last %.2f" % tuple(us(np.percentile(t0 - base, q)) for q in (0, 50, 90, 100)))
print("end    first %.2f us  median %.2f  p90 %.2f  last %.2f" % tuple(us(np.percentile(t1 - base, q)) for q in (0, 50, 90, 100)))
life = us(t1 - t0)
print("life   min %.2f us  p10 %.2f  median %.2f  p90 %.2f  max %.2f" % tuple(np.percentile(life, q) for q in (0, 10, 50, 90, 100)))
print("shader clock over the wave lifetimes: median %.3f GHz" % np.median(cyc / (life * 1e3)))
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = xcc * 100000 + se * 10000 + sh * 1000 + cu * 10 + simd
per = collections.Counter(key.tolist())
hist = collections.Counter(per.values())
print("SIMDs used %d ; waves per SIMD histogram %s" % (len(per), dict(sorted(hist.items()))))
cus = collections.Counter((key // 10).tolist())
print("CUs used %d ; waves per CU histogram %s" % (len(cus), dict(sorted(collections.Counter(cus.values()).items()))))
for k in sorted(hist):
    sel = np.array([per[x] == k for x in key.tolist()])
    print("  SIMDs holding %d wave(s): wave life median %.2f us, start median %.2f us, end median %.2f us" % (
        k, np.median(life[sel]), us(np.median(t0[sel] - base)), us(np.median(t1[sel] - base))))
clk = cyc / (life * 1e3)
for x in range(8):
    sel = xcc == x
    print("  XCD %d: waves %3d  life median %.1f max %.1f us  shader clock median %.3f GHz" % (x, sel.sum(), np.median(life[sel]), life[sel].max(), np.median(clk[sel])))
m.close()
