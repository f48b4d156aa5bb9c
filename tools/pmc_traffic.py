"""HBM traffic per kernel launch from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass:
MI355X_MICROARCH.md "rocprofv3 PMC slots") over `bench.py --steps 8 --no-cpu`, written as the JSON that bench.py
matches against the kernels it runs.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d fetch -- python bench.py --steps 8 --warmup 2 --no-cpu
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d write -- python bench.py --steps 8 --warmup 2 --no-cpu
    python tools/pmc_traffic.py --arch full --fetch fetch/.../*counter_collection.csv --write write/.../*counter_collection.csv \
           --bench-json bench.json --head $(git rev-parse --short HEAD) --out profiles/pmc_traffic.json

bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: the counters are in KiB, and on gfx950 FETCH_SIZE reports half
of the bytes of a wide read (guide, section HBM); WRITE_SIZE is taken as is (checked here against the algorithmic
bytes each kernel must write: the ratio is printed).  Only launches over the bench batch are averaged (Grid_Size of
the most frequent launch of each kernel)."""
import argparse
import collections
import csv
import json
import os
import re


def short_name(kernel_name):
    """'void (anonymous namespace)::conv_tm<2, 1, ...>(float ...)' -> 'conv_tm<2, 1, ...>'"""
    n = kernel_name.replace("void ", "").replace("(anonymous namespace)::", "")
    depth, out = 0, []
    for ch in n:
        if ch == "(" and depth == 0:
            break
        depth += ch == "<"
        depth -= ch == ">"
        out.append(ch)
    return "".join(out).strip()


def per_kernel(path, counter):
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        rows[short_name(r["Kernel_Name"])].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    out = {}
    for k, v in rows.items():
        grid = collections.Counter(g for g, _ in v).most_common(1)[0][0]
        vals = [c for g, c in v if g == grid]
        out[k] = (sum(vals) / len(vals), len(vals), grid)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="full")
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--bench-json", required=True, help="JSON line of the same bench command (kernel names per stage)")
    ap.add_argument("--head", default="")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    fetch = per_kernel(a.fetch, "FETCH_SIZE")
    write = per_kernel(a.write, "WRITE_SIZE")
    bench = json.loads(open(a.bench_json).read().strip().splitlines()[-1])
    per_launch = bench["roofline"]["candidates_per_launch"]
    doc = json.load(open(a.out)) if os.path.exists(a.out) else {}
    doc["_note"] = ("HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 from separate rocprofv3 --pmc FETCH_SIZE / "
                    "--pmc WRITE_SIZE passes over `bench.py --steps 8 --no-cpu` (gfx950: FETCH_SIZE reports half of a wide "
                    "coalesced read, MI355X_MICROARCH.md section HBM); made by tools/pmc_traffic.py; bench.py uses an entry "
                    "only if its kernel name is the one the running binary launches for that stage")
    ent = {}
    total = 0.0
    for st in bench["kernels"]:
        name = st.get("kernel_name")
        if not name or name not in fetch or name not in write:
            raise SystemExit("no counter rows for stage %r kernel %r; have %s" % (st["kernel"], name, sorted(fetch)))
        f, nf, grid = fetch[name]
        w, nw, _ = write[name]
        b = (2.0 * f + w) * 1024.0
        total += b
        ent[st["kernel"]] = {"kernel_name": name, "candidates_per_launch": per_launch, "hbm_bytes_per_launch": int(round(b)),
                             "fetch_kib": f, "write_kib": w, "launches_averaged": min(nf, nw), "grid_size": grid}
    ent["_whole_path"] = {"hbm_bytes_per_launch": int(round(total)), "compulsory_bytes_per_launch": 2176 * per_launch,
                          "ratio_to_compulsory": total / (2176.0 * per_launch)}
    ent["_git_head"] = a.head
    doc[a.arch] = ent
    json.dump(doc, open(a.out, "w"), indent=1)
    for k, v in ent.items():
        if isinstance(v, dict) and "kernel_name" in v:
            print("%-36s %-34s %8.1f MB / launch" % (k, v["kernel_name"], v["hbm_bytes_per_launch"] / 1e6))
    print("whole path %.1f MB / launch = %.1fx the compulsory %.1f MB" % (total / 1e6, ent["_whole_path"]["ratio_to_compulsory"],
                                                                      2176 * per_launch / 1e6))


if __name__ == "__main__":
    main()
