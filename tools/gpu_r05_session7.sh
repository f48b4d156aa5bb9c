#!/bin/bash
# Round-5 session 7: eight-wave tail kernel, chained join alone (bit 7) / late fc4 weight gradient (bit 8), 8 or 16 k ranges
set -u
TAG=${1:-r05g}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests/test_gpu_dp.py tests/test_gpu_train_parity.py -m gpu -q -x --durations=5 -k "tiny or alpha_dropout or 1250 or reproducible or backward_kernel_variants or every_gradient" > $OUT/pytest_step.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_step.log
A=$PWD/clairvoyante_amd/csrc/libclairvoyante_hip_base.so
run() {  # label, batch, lib ('' = in-tree), bench flags
  local label=$1 b=$2 lib=$3; shift 3
  if [ -n "$lib" ]; then export CV_HIP_LIB=$lib; else unset CV_HIP_LIB; fi
  python bench.py --mode train --batch $b --steps 40 --warmup 4 "$@" 2>> $OUT/err.txt | LABEL="$label" python -c "
import json,sys,os
r=json.loads(sys.stdin.read()); print('%s batch %5d %-44s %.3f ms' % (r['config']['arch'], r['config']['global_batch'], os.environ['LABEL'], r['ms_per_step']))" >> $OUT/step_ab.txt
  unset CV_HIP_LIB
}
for round in 1 2 3; do
  for b in 1250 2500; do
    run "round-4 library" $b $A
    run "in-tree, sched 255 (default)" $b ""
    run "in-tree, sched 127 (join: one wait per stream)" $b "" --sched 127
    run "in-tree, sched 511 (+ late fc4 wgrad)" $b "" --sched 511
    run "in-tree, sched 255, 16 k ranges" $b "" --kranges 16
    run "in-tree, sched 127, 16 k ranges" $b "" --sched 127 --kranges 16
  done
done
sort $OUT/step_ab.txt; grep -i "error\|Traceback" $OUT/err.txt | head -5
timeout 300 bash tools/gpu_train_profile.sh $TAG 1250 - "--kranges 16" > /dev/null 2>&1
python - <<'PY'
import csv
rows=[r for r in csv.DictReader(open('gpurun_out/r05g/train_1250_-_kernel_stats.csv')) if int(r["Calls"]) in (23,46,69)]
for r in rows[:12]: print("%-70s %3s x %7.1f us" % (r["Name"].replace("(anonymous namespace)::","").replace("float __vector(4)","f4").replace("void ","")[:70], r["Calls"], float(r["AverageNs"])/1e3))
PY
