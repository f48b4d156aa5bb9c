"""Where the cycles of dense_dgrad_unpool's row loop go, per wave (development tool; needs a library built with
-DCV_WG_STAMP -DCV_ROW_PHASES, e.g. python tools/build_variant_lib.py phases -DCV_WG_STAMP -DCV_ROW_PHASES and
CV_HIP_LIB=.../libclairvoyante_hip_phases.so).  python tools/gpu_row_phases.py [batch]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from clairvoyante_amd import clairvoyante_v3, synth, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
m = clairvoyante_v3.Clairvoyante(); m.init(); m.setOption("train_overlap", 0)
xt, cls, rf, alt, il = synth.make_candidates(n, seed=3, device="cuda", return_class=True)
y = synth.make_labels(cls, rf, alt, il)
for _ in range(3):
    m.train(xt, y)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros(8 * 4096 * 4, dtype=np.uint64); cnt = np.zeros(8, dtype=np.uint32)
assert lib.cv_debug_wg_stamps(buf.ctypes.data_as(ctypes.c_void_p), cnt.ctypes.data_as(ctypes.c_void_p)) == 0
s = buf.reshape(8, 4096, 4)[7][:min(int(cnt[7]), 4096)].astype(np.float64)
print("batch %d: %d wave records of dense_dgrad_unpool (newest 4096)" % (n, len(s)))
tot = s[:, 0] + s[:, 1] + s[:, 2]
for name, sel in (("all waves", np.ones(len(s), bool)), ("waves 0-3 (short head)", s[:, 3] < 4), ("waves 4-7 (long head)", s[:, 3] >= 4)):
    if sel.any():
        t = tot[sel]
        print("  %-24s loop cycles median %.0f ; issue %.3f  own-memory wait %.3f  barrier wait %.3f" % (
            name, np.median(t), (s[sel, 0] / t).mean(), (s[sel, 1] / t).mean(), (s[sel, 2] / t).mean()))
m.close()
