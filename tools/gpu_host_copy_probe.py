"""Development probe for model._predict_host: how a caller-owned numpy batch [n,33,4,4] gets to the device fastest.
  a  torch.from_numpy(x).to(device): the runtime's staged pageable copy (what rounds 1-5 did)
  b  hipHostRegister on the caller's buffer, one async copy, hipHostUnregister
  c  pageable copies of parts (host-synchronous each) with the kernels of the part before running meanwhile
  d  parts staged by host threads into page-locked buffers (numpy copies release the GIL), async copies
Prints the time of each for 65 536 and 16 384 candidates."""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from clairvoyante_amd import clairvoyante_v3, synth


def best(fn, reps=8):
    fn(); torch.cuda.synchronize()
    b = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); b = min(b, time.perf_counter() - t0)
    return b * 1e3


def main():
    m = clairvoyante_v3.Clairvoyante(); m.init()
    rt = torch.cuda.cudart()
    pool = ThreadPoolExecutor(8)
    for n in (65536, 16384):
        x = synth.make_candidates(n, seed=1).numpy()
        nbytes = x.nbytes
        dev = torch.empty((n, 33, 4, 4), device="cuda")
        out = torch.empty((n, 16), device="cuda")
        print("n=%d (%.1f MB)" % (n, nbytes / 1e6))
        print("  a pageable .to(device):          %.3f ms" % best(lambda: torch.from_numpy(x).to("cuda")))
        print("  pass on device-resident input:   %.3f ms" % best(lambda: m.predict_device(dev, out)))

        def reg():
            assert int(rt.cudaHostRegister(x.ctypes.data, nbytes, 0)) == 0
        def unreg():
            assert int(rt.cudaHostUnregister(x.ctypes.data)) == 0
        t0 = time.perf_counter(); reg(); t1 = time.perf_counter()
        xt = torch.from_numpy(x)
        tc = best(lambda: dev.copy_(xt, non_blocking=True))
        t2 = time.perf_counter(); unreg(); t3 = time.perf_counter()
        print("  b register %.3f ms, async copy %.3f ms (%.1f GB/s), unregister %.3f ms" % ((t1 - t0) * 1e3, tc, nbytes / tc / 1e6, (t3 - t2) * 1e3))
        for _ in range(3):
            t0 = time.perf_counter(); reg(); t1 = time.perf_counter(); unreg(); t2 = time.perf_counter()
            print("    again: register %.3f ms unregister %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))

        cs = torch.cuda.Stream()
        main = torch.cuda.current_stream()
        for parts in (2, 4, 8):
            cut = [n * i // parts // 16 * 16 for i in range(parts)] + [n]
            evs = [torch.cuda.Event() for _ in range(parts)]
            def run_c():
                cs.wait_stream(main)                  # the buffer is free again
                for i in range(parts):
                    lo, hi = cut[i], cut[i + 1]
                    with torch.cuda.stream(cs):
                        dev[lo:hi].copy_(torch.from_numpy(x[lo:hi]))      # host-synchronous staged copy, on the copy stream
                        evs[i].record(cs)
                    main.wait_event(evs[i])
                    m.predict_device(dev[lo:hi], out[lo:hi])
                return out.cpu()
            print("  c %d pageable parts on a copy stream, pass of part k under copy k+1, + D2H: %.3f ms" % (parts, best(run_c)))

        pin = torch.empty((n, 33, 4, 4), pin_memory=True)
        pn = pin.numpy()
        for parts in (4, 8, 16):
            cut = [n * i // parts // 16 * 16 for i in range(parts)] + [n]
            evs = [torch.cuda.Event() for _ in range(parts)]
            def stage(i):
                np.copyto(pn[cut[i]:cut[i + 1]], x[cut[i]:cut[i + 1]])
                return i
            def run_d():
                cs.wait_stream(main)
                futs = [pool.submit(stage, i) for i in range(parts)]
                for i, f in enumerate(futs):
                    f.result()
                    lo, hi = cut[i], cut[i + 1]
                    with torch.cuda.stream(cs):
                        dev[lo:hi].copy_(pin[lo:hi], non_blocking=True)
                        evs[i].record(cs)
                    main.wait_event(evs[i])
                    m.predict_device(dev[lo:hi], out[lo:hi])
                return out.cpu()
            print("  d %d parts staged by 8 host threads into pinned memory, async copies on a copy stream, + D2H: %.3f ms" % (parts, best(run_d)))
        def run_a():
            o = m.predict_device(torch.from_numpy(x).to("cuda"))
            return o.cpu()
        print("  a whole (rounds 1-5 predict): %.3f ms" % best(run_a))
        print("  m.predict(x) as shipped: %.3f ms" % best(lambda: m.predict(x)))


if __name__ == "__main__":
    main()
