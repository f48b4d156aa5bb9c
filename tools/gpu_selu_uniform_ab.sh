#!/bin/bash
# A/B of the wave-uniform SELU fast path (build flag CV_SELU_UNIFORM): parity + kernel times + SQ counters, both builds
set -u
OUT=gpurun_out/${1:-r03selu}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for mode in default uniform; do
  if [ $mode = uniform ]; then export CV_EXTRA_FLAGS="-DCV_SELU_UNIFORM"; else unset CV_EXTRA_FLAGS; fi
  python -c "from clairvoyante_amd import build; build.build(force=True)" > $OUT/build_$mode.log 2>&1
  python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $OUT/pytest_$mode.log 2>&1; echo "$mode parity rc=$?" >> $OUT/status.txt
  for i in 1 2 3; do python bench.py --no-cpu --no-extras >> $OUT/bench_$mode.jsonl 2>> $OUT/bench.err; done
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p_$mode -o p -- python bench.py --steps 8 --warmup 2 --no-cpu --no-extras > /dev/null 2> $OUT/p_$mode.err
  f=$(find $OUT/p_$mode -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" > $OUT/sq_$mode.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and m.get("GRBM_GUI_ACTIVE"):
        print("%-40s launches %3d  SQ_INSTS_VALU %.4g  MFMA busy %.3f" % (k[:40], len(v["SQ_INSTS_VALU"]), m.get("SQ_INSTS_VALU", 0),
              m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] * 128.0)))
PY
  rm -rf $OUT/p_$mode
  python - $OUT/bench_$mode.jsonl $mode <<'PY'
import json, sys
for l in open(sys.argv[1]):
    r = json.loads(l)
    print(sys.argv[2], "%.3f M/s" % (r["value"] / 1e6), " ".join("%s %.4f" % (k["kernel_name"].split("<")[0], k["avg_ms"]) for k in r["kernels"]))
PY
  cat $OUT/sq_$mode.txt
done
cat $OUT/status.txt
