"""Development probe: two training runs of N steps from the same seeds must end in the same weights bit for bit
(every weight gradient is reduced in a fixed order); a race in the LDS staging of the weight-gradient kernels
would show up here.  With `deferred` the steps are enqueued without any host synchronisation (model.trainDeferred: many
steps in flight, main stream + up to three side streams), the way train.run_epoch drives them.
usage: python tools/gpu_train_repro_soak.py [steps] [arch] [deferred]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from clairvoyante_amd import _lib, clairvoyante_v3, clairvoyante_v3_slim, synth
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    arch = sys.argv[2] if len(sys.argv) > 2 else "full"
    deferred = len(sys.argv) > 3 and sys.argv[3] == "deferred"
    batches = []
    for b, n in enumerate((10000, 9991, 4096, 777, 1250, 2560, 2576)):
        xt, cls, rf, alt, il = synth.make_candidates(n, seed=40 + b, device="cuda", return_class=True)
        batches.append((xt, synth.make_labels(cls, rf, alt, il)))

    def run():
        m = (clairvoyante_v3 if arch == "full" else clairvoyante_v3_slim).Clairvoyante()
        m._seed_rng.seed(5); m._dropout_seed = 99
        m.init(); m.setLearningRate(1e-4)
        loss = 0.0
        for s in range(steps):
            x, y = batches[s % len(batches)]
            if deferred:
                m.trainDeferred(x, y)
            else:
                loss, _ = m.train(x, y)
        if deferred:
            loss = m.readLosses()[0][5]
        w = torch.empty(m.numParameters, device="cuda")
        _lib.check(m._lib.cv_flat_copy(m._h, 0, ctypes.c_void_p(w.data_ptr()), 0, None))
        w = w.cpu().numpy().copy()
        m.close()
        return float(loss), w
    l1, w1 = run()
    l2, w2 = run()
    same = np.array_equal(w1.view(np.uint32), w2.view(np.uint32))
    print("%s%s: %d steps x 2 runs, last loss %.17g / %.17g (equal: %s), weights finite %s, bitwise equal %s" % (
        arch, " (deferred)" if deferred else "", steps, l1, l2, l1 == l2, bool(np.isfinite(w1).all()), same))
    sys.exit(0 if same and l1 == l2 and np.isfinite(w1).all() else 1)


if __name__ == "__main__":
    main()
