#!/bin/bash
# where the waves of the training kernels spend their cycles (stream order): parked on s_waitcnt / barrier (WAIT_ANY),
# issue-stalled (WAIT_INST_ANY, of which LDS), issuing (ACTIVE_INST_ANY), LDS bank conflicts
set -u
OUT=gpurun_out/${1:-pmcwait}; B=${2:-10000}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/p -o p -- python bench.py --mode train --batch $B --steps 8 --warmup 2 --overlap 0 > /dev/null 2> $OUT/p.err
f=$(find $OUT/p -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python - "$f" > $OUT/train_${B}_wait_summary.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, v in acc.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0)
    if wc > 0 and not k.startswith(("at::", "rocprim", "__amd")):
        rows.append((wc, "%-40s wave-cycles %.3g  parked %.2f  issue-stall %.2f (LDS %.2f)  issuing %.2f  LDS bank-conflict/active %.2f" % (
            k[:40], wc, m.get("SQ_WAIT_ANY", 0) / wc, m.get("SQ_WAIT_INST_ANY", 0) / wc, m.get("SQ_WAIT_INST_LDS", 0) / wc,
            m.get("SQ_ACTIVE_INST_ANY", 0) / wc, m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 0), 1))))
for _, l in sorted(rows, reverse=True)[:16]: print(l)
PY
rm -rf $OUT/p
cat $OUT/train_${B}_wait_summary.txt; tail -3 $OUT/p.err
