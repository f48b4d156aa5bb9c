#!/bin/bash
# MFMA-busy / VALU-instruction counters of the training step's kernels (stream order, so kernels do not share the chip)
set -u
OUT=gpurun_out/${1:-pmcsq}; B=${2:-10000}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p_sq -o p -- python bench.py --mode train --batch $B --steps 8 --warmup 2 --overlap 0 > /dev/null 2> $OUT/p_sq.err
f=$(find $OUT/p_sq -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python - "$f" > $OUT/train_${B}_sq_summary.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, v in acc.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and m.get("GRBM_GUI_ACTIVE") and not k.startswith(("at::", "rocprim")):
        rows.append((m["GRBM_GUI_ACTIVE"], "%-44s launches %3d  GRBM_GUI_ACTIVE %.4g  SQ_INSTS_VALU %.4g  MFMA busy %.3f" % (
            k[:44], len(v["SQ_INSTS_VALU"]), m["GRBM_GUI_ACTIVE"], m.get("SQ_INSTS_VALU", 0), m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] * 128.0))))
for _, l in sorted(rows, reverse=True): print(l)
PY
rm -rf $OUT/p_sq
cat $OUT/train_${B}_sq_summary.txt | head -24
