"""Where the cycles of front2_tm (conv1+pool1+conv2+pool2 of the full topology) go, per wave (development tool; needs a
library built with -DCV_WG_STAMP -DCV_ROW_PHASES: python tools/build_variant_lib.py phases -DCV_WG_STAMP -DCV_ROW_PHASES,
CV_HIP_LIB=clairvoyante_amd/csrc/libclairvoyante_hip_phases.so).  python tools/gpu_front2_phases.py [batch]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from clairvoyante_amd import clairvoyante_v3, synth, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
m = clairvoyante_v3.Clairvoyante(); m.setParameters(synth.bench_params("full"))
x = synth.make_candidates(n, seed=synth.BASE_SEED, device="cuda")
for _ in range(3):
    m.predict_device(x)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros(8 * 4096 * 4, dtype=np.uint64); cnt = np.zeros(8, dtype=np.uint32)
assert lib.cv_debug_wg_stamps(buf.ctypes.data_as(ctypes.c_void_p), cnt.ctypes.data_as(ctypes.c_void_p)) == 0
r = buf.reshape(8, 4096, 4)[7][:min(int(cnt[7]), 4096)]
wave = (r[:, 3] & np.uint64(0xff)).astype(np.int64)
ph = np.stack([r[:, 0], r[:, 1], r[:, 2], r[:, 3] >> np.uint64(8)], 1).astype(np.float64)
tot = ph.sum(1)
print("batch %d: %d wave records of front2_tm<6> (newest 4096 of %d)" % (n, len(r), int(cnt[7])))
print("  cycles per wave (s_memtime): median %.0f  p10 %.0f  p90 %.0f" % (np.median(tot), np.percentile(tot, 10), np.percentile(tot, 90)))
names = ("producer half (3 first-layer rows: 84 MFMA + pool + 48 SELU + LDS writes)", "conv2 over the chunk (6 x 96 MFMA + pool + 96 SELU + stores)",
         "the two barriers of a chunk", "raw X rows: wait for the strided loads")
for i, nm in enumerate(names):
    print("  %-82s %6.3f of the wave's cycles   (median %.0f cycles)" % (nm, (ph[:, i] / tot).mean(), np.median(ph[:, i])))
for w in range(4):
    sel = wave == w
    if sel.any():
        print("  wave %d (group %d, tile %d): producer %.3f  conv2 %.3f  barriers %.3f  X wait %.3f" % (
            w, w >> 1, w & 1, *[(ph[sel, i] / tot[sel]).mean() for i in range(4)]))
# the ideal: MFMA issue of one wave (5 chunks x 84 + 28 x 96 + 48 first + 48 last-row MFMAs ~ 3 200) x 32 cycles x 2 waves per SIMD
print("  MFMA issue of the two waves of a SIMD: ~%.0f cycles (3 156 MFMAs x 32 x 2)" % (3156 * 32 * 2))
m.close()
