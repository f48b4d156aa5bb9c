#!/bin/bash
# Effective shader clock of the hot kernels under load: GRBM_GUI_ACTIVE (per XCD) / kernel wall time of the SAME run,
# for the inference pass and the training step, next to the pure-MFMA probe (tools/mfma_peak.hip) on constant and on
# random operands.  bash tools/gpu_clock_probe.sh TAG
set -u
OUT=gpurun_out/${1:-clock}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/mfma_peak.hip -o $OUT/mfma_peak 2> $OUT/build.err && $OUT/mfma_peak > $OUT/mfma_peak.txt 2>&1
rm -f $OUT/mfma_peak
cat $OUT/mfma_peak.txt
run() {  # name, bench args
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/p_$1 -o p -- python bench.py $2 > /dev/null 2> $OUT/p_$1.err
  c=$(find $OUT/p_$1 -name "*counter_collection.csv" | head -1); t=$(find $OUT/p_$1 -name "*kernel_trace.csv" | head -1)
  python - "$c" "$t" > $OUT/clock_$1.txt <<'PY'
import csv, sys, collections
dur = {}
for r in csv.DictReader(open(sys.argv[2])):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    acc[(n, r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
per = collections.defaultdict(list)
for (n, d), m in acc.items():
    if d in dur and m.get("GRBM_GUI_ACTIVE") and not n.startswith(("at::", "rocprim", "__amd")):
        per[n].append((dur[d], m["GRBM_GUI_ACTIVE"] / 8.0, m.get("SQ_WAVE_CYCLES", 0)))
rows = []
for n, v in per.items():
    v = v[len(v) // 4:]                      # skip the warm-up launches
    ns = sum(x[0] for x in v) / len(v); cyc = sum(x[1] for x in v) / len(v)
    rows.append((ns, "%-44s launches %3d  wall %8.1f us  GRBM_GUI_ACTIVE/8 %.4g  => %.3f GHz" % (n[:44], len(v), ns / 1e3, cyc, cyc / ns)))
for _, l in sorted(rows, reverse=True)[:14]: print(l)
PY
  rm -rf $OUT/p_$1
  echo "== $1"; cat $OUT/clock_$1.txt
}
run infer "--steps 6 --warmup 2 --no-extras"
run train "--mode train --batch 10000 --steps 8 --warmup 2 --overlap 0"
