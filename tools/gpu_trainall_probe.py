"""Development probe: throughput of the real train.TrainAll loop (DecompressArray on the host overlapped with the
training step on the GPU) on a synthetic .bin -- candidates/s per epoch vs the bare step rate of bench.py --mode train."""
import logging
import os
import pickle
import sys
import tempfile
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from clairvoyante_amd import clairvoyante_v3, param, synth, train, utils_v2
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
    xt, cls, rf, alt, il = synth.make_candidates(total, seed=5, device="cuda", return_class=True)
    X = xt.cpu().numpy()
    Y = synth.make_labels(cls, rf, alt, il).cpu().numpy().astype(np.float64)
    t0 = time.time()
    XC = [utils_v2.pack_array(X[s:s + 500]) for s in range(0, total + 1, 500)]
    YC = [utils_v2.pack_array(Y[s:s + 500]) for s in range(0, total + 1, 500)]
    fn = os.path.join(tempfile.mkdtemp(prefix="cv_tr_"), "t.bin")
    with open(fn, "wb") as fh:
        pickle.dump(total, fh); pickle.dump(XC, fh); pickle.dump(YC, fh); pickle.dump([], fh)
    print("packed %d items in %.1f s (%.0f MB)" % (total, time.time() - t0, os.path.getsize(fn) / 1e6))
    param.maxEpoch = 4                                     # 3 epochs
    m = clairvoyante_v3.Clairvoyante()
    m.init()
    args = types.SimpleNamespace(bin_fn=fn, tensor_fn=None, var_fn=None, bed_fn=None, chkpnt_fn=None, learning_rate=1e-3,
                                 lambd=1e-3, ochk_prefix=None, olog_dir=None, v2=False, v3=True, slim=False)
    times = []

    class H(logging.Handler):
        def emit(self, rec):
            msg = rec.getMessage()
            if msg.startswith("Epoch time elapsed"):
                times.append(float(msg.split(":")[1].split()[0]))
    logging.getLogger().addHandler(H())
    logging.getLogger().setLevel(logging.INFO)
    train.TrainAll(args, m, utils_v2)
    for k, t in enumerate(times):
        print("epoch %d: %.2f s -> %.2f M candidates/s (training part is 90 %% of the set)" % (k + 1, t, total / t / 1e6))
    m.close()


if __name__ == "__main__":
    main()
