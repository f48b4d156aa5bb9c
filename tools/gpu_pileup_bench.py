"""Development probe: throughput of the pileup front end on synthetic short-read alignments
(host SAM parse rate, scatter / finalize kernel time from cv_pileup_stats).
    python tools/gpu_pileup_bench.py [n_reads] [contig_len] [candidate_spacing]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    import torch
    from clairvoyante_amd.pileup import Pileup
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 10000000
    spacing = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    t0 = time.time()
    from clairvoyante_amd.synth_pileup import fast_alignments
    ref, text = fast_alignments(n_reads, L)
    rng = np.random.RandomState(9)
    centers = np.unique(rng.randint(20, L - 20, L // spacing)).astype(np.int64)
    gen_s = time.time() - t0
    fused = os.environ.get("CV_PILEUP_FUSED", "0") == "1"
    pl = Pileup(evc=fused, retain=fused, contig="ctgA")
    pl.set_reference(ref, 0)
    if not fused:
        pl.set_candidates(centers)
    torch.cuda.synchronize()
    t0 = time.time()
    step = 64 << 20
    for s in range(0, len(text), step):
        pl.add_sam(text[s:s + step])
    if fused:       # candidates from the alignments themselves (substitution rate 1 % -> threshold picks the noisy sites)
        pl.extract_candidates(float(os.environ.get("CV_PILEUP_THR", "0.06")), 4)
        centers = pl.adopt_candidates()
    t, d, u = pl.finish(subtract=True)
    torch.cuda.synchronize()
    wall = time.time() - t0
    st = pl.stats()
    out = {"fused": fused, "reads": n_reads, "contig": L, "candidates": int(len(centers)), "sam_bytes": len(text), "gen_s": gen_s,
           "wall_s": wall, "host_MBps": len(text) / wall / 1e6, "stats": st,
           "scatter_Gcol_per_s": st["columns"] / (st["scatter_ms"] * 1e-3) / 1e9 if st["scatter_ms"] else None,
           "finalize_GBps": len(centers) * (33 * 9 * 4 + 33 * 64) / (st["finalize_ms"] * 1e-3) / 1e9,
           "touched": int(u.sum().item()), "mean_depth": float(d.float().mean().item())}
    print(json.dumps(out))
    pl.close()


if __name__ == "__main__":
    main()
