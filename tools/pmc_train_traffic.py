"""HBM traffic of ONE optimizer step from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate passes, guide
section "rocprofv3 PMC slots") over `bench.py --mode train --batch B`:
bytes per step = sum over the kernels of the step of (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (counters in KiB; gfx950
reports half of a wide read, see tools/pmc_traffic.py), steps = number of adam_kernel launches in the pass.
Writes the section "train" of profiles/pmc_traffic.json, key "<arch>_<candidates per rank>"; bench.py attaches an entry
to a training line only at the same arch and per-rank batch.
    python tools/pmc_train_traffic.py --arch full --batch 10000 --fetch F.csv --write W.csv --head GIT --out profiles/pmc_traffic.json"""
import argparse
import collections
import csv
import json
import os

from pmc_traffic import short_name


def totals(path, counter):
    tot = collections.defaultdict(float)
    cnt = collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        n = short_name(r["Kernel_Name"])
        if n.startswith(("at::", "rocprim", "elementwise_kernel", "void at::")):      # torch's own kernels (data generation)
            continue
        tot[n] += float(r["Counter_Value"]); cnt[n] += 1
    return tot, cnt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="full")
    ap.add_argument("--batch", type=int, required=True)
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--head", default="")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    f, fc = totals(a.fetch, "FETCH_SIZE")
    w, wc = totals(a.write, "WRITE_SIZE")
    steps = fc.get("adam_kernel", 0)
    if steps == 0 or wc.get("adam_kernel", 0) != steps:
        raise SystemExit("the two passes do not hold the same number of optimizer steps (%d / %d)" % (steps, wc.get("adam_kernel", 0)))
    per = {}
    for k in sorted(set(f) | set(w)):
        b = (2.0 * f.get(k, 0.0) + w.get(k, 0.0)) * 1024.0 / steps
        per[k] = {"hbm_bytes_per_step": int(round(b)), "launches_per_step": fc.get(k, 0) / float(steps)}
    total = sum(v["hbm_bytes_per_step"] for v in per.values())
    doc = json.load(open(a.out)) if os.path.exists(a.out) else {}
    tr = doc.setdefault("train", {})
    tr["_note"] = ("HBM bytes of one optimizer step = sum over its kernels of (2 x FETCH_SIZE + WRITE_SIZE) x 1024, separate "
                   "rocprofv3 --pmc passes over `bench.py --mode train` (tools/pmc_train_traffic.py)")
    tr["_git_head"] = a.head
    tr["%s_%d" % (a.arch, a.batch)] = {"hbm_bytes_per_step": int(total), "steps_averaged": steps,
                                         "bytes_per_candidate": total / float(a.batch),
                                         "per_kernel": dict(sorted(per.items(), key=lambda kv: -kv[1]["hbm_bytes_per_step"])[:16])}
    json.dump(doc, open(a.out, "w"), indent=1)
    print("%s batch %d: %.1f MB per step (%.1f KB per candidate) over %d steps" % (a.arch, a.batch, total / 1e6, total / a.batch / 1e3, steps))
    for k, v in list(sorted(per.items(), key=lambda kv: -kv[1]["hbm_bytes_per_step"]))[:10]:
        print("   %-60s %8.1f MB" % (k[:60], v["hbm_bytes_per_step"] / 1e6))


if __name__ == "__main__":
    main()
