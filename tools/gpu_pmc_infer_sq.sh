#!/bin/bash
# SQ counters of the inference kernels (one pass of 8 SQ slots each): where front2_tm's wave cycles go.
#   bash tools/gpu_pmc_infer_sq.sh TAG
set -u
OUT=gpurun_out/${1:-pmcsq}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
pass() {  # name, counters
  rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/p_$1 -o p -- python bench.py --steps 6 --warmup 2 --no-extras --no-cpu > /dev/null 2> $OUT/p_$1.err
  f=$(find $OUT/p_$1 -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/infer_sq_$1.csv
  rm -rf $OUT/p_$1
}
pass a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE"
pass b "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
python - $OUT > $OUT/infer_sq_summary.txt <<'PY'
import csv, sys, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in sorted(glob.glob(sys.argv[1] + "/infer_sq_*.csv")):
    for r in csv.DictReader(open(fn)):
        n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        if n.startswith(("at::", "rocprim", "__amd")): continue
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    m = {c: sum(x[len(x) // 4:]) / len(x[len(x) // 4:]) for c, x in v.items()}
    print(k)
    wc = m.get("SQ_WAVE_CYCLES", 0)
    for c in sorted(m):
        extra = ""
        if wc and c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
            extra = "  = %.3f of SQ_WAVE_CYCLES" % (m[c] / wc)
        if c == "SQ_VALU_MFMA_BUSY_CYCLES" and m.get("GRBM_GUI_ACTIVE"):
            extra = "  MFMA busy %.3f" % (m[c] / (m["GRBM_GUI_ACTIVE"] * 128.0))
        print("    %-28s %.5g%s" % (c, m[c], extra))
PY
cat $OUT/infer_sq_summary.txt
