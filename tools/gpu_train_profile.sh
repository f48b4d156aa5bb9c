#!/bin/bash
# serialized per-kernel profile of the training step: bash tools/gpu_train_profile.sh TAG BATCH "dbg settings ('-' = none)"
set -u
OUT=gpurun_out/${1:-r03p}
B=${2:-10000}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for d in ${3:-"-"}; do
  dd=$d; [ "$d" = "-" ] && dd=""
  tag=$(echo "$d" | tr '=,' '__')
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$tag -o t -- python bench.py --mode train --batch $B --steps 20 --warmup 3 --overlap 0 --dbg "$dd" ${4:-} > $OUT/train_${B}_$tag.json 2> $OUT/prof_$tag.err
  f=$(find $OUT/prof_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/train_${B}_${tag}_kernel_stats.csv; rm -rf $OUT/prof_$tag
  python - $OUT/train_${B}_${tag}_kernel_stats.csv <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if int(r["Calls"]) in (23, 46, 69)]
print(sys.argv[1], "sum per step %.1f us" % (sum(float(r["TotalDurationNs"]) for r in rows) / 23e3))
for r in rows[:22]:
    print("  %-100s %3s x %8.1f us" % (r["Name"].replace("(anonymous namespace)::", "").replace("float __vector(4)", "f4")[:100], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
