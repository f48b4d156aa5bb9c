#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) + MFMA busy / VALU instruction counters of the inference
# kernels of the shipped binary -> gpurun_out/<tag>/ ; tools/pmc_traffic.py turns the first two into profiles/pmc_traffic.json
set -u
TAG=${1:-pmc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
[ -f $OUT/pmc_traffic.json ] || cp profiles/pmc_traffic.json $OUT/pmc_traffic.json    # keep the other sections (pileup, train)
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
HEAD=${2:-unknown}
for arch in full slim; do
  python bench.py --steps 8 --warmup 2 --no-cpu --no-extras --arch $arch > $OUT/bench_$arch.json 2> $OUT/bench_$arch.err
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/p_${arch}_$c -o p -- python bench.py --steps 8 --warmup 2 --no-cpu --no-extras --arch $arch > /dev/null 2> $OUT/p_${arch}_$c.err
    f=$(find $OUT/p_${arch}_$c -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && cp "$f" $OUT/${arch}_${c}_counter_collection.csv
    rm -rf $OUT/p_${arch}_$c
  done
  python tools/pmc_traffic.py --arch $arch --fetch $OUT/${arch}_FETCH_SIZE_counter_collection.csv --write $OUT/${arch}_WRITE_SIZE_counter_collection.csv \
      --bench-json $OUT/bench_$arch.json --head $HEAD --out $OUT/pmc_traffic.json
done
# MFMA-busy and VALU instruction counters (own pass)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p_sq -o p -- python bench.py --steps 8 --warmup 2 --no-cpu --no-extras > /dev/null 2> $OUT/p_sq.err
f=$(find $OUT/p_sq -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python - "$f" > $OUT/sq_summary.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and m.get("GRBM_GUI_ACTIVE"):
        # busy cycles are summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs (wall = GRBM / 8)
        print("%-40s launches %3d  SQ_INSTS_VALU %.4g  MFMA busy %.3f" % (k[:40], len(v["SQ_INSTS_VALU"]), m.get("SQ_INSTS_VALU", 0),
              m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] * 128.0)))
PY
rm -rf $OUT/p_sq
cat $OUT/sq_summary.txt 2>/dev/null
