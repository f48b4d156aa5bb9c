#!/bin/bash
# Round-5 session 6: the quicker training heads (packed [k][12] head weights, labels through LDS), fc4's weight gradient at
# conv3's marker + chained join at tiny batches (train_sched bit 7): tests, then A/B on one box.
set -u
TAG=${1:-r05f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests/test_gpu_dp.py tests/test_gpu_train_parity.py tests/test_gpu_pipeline.py -m gpu -q -x --durations=5 -k "not ranks and not rank and not data_parallel and not empty_shards and not 70001 and not several_slices" > $OUT/pytest_step.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest_step.log
A=$PWD/clairvoyante_amd/csrc/libclairvoyante_hip_base.so
run() {  # label, batch, lib ('' = in-tree), bench flags
  local label=$1 b=$2 lib=$3; shift 3
  if [ -n "$lib" ]; then export CV_HIP_LIB=$lib; else unset CV_HIP_LIB; fi
  python bench.py --mode train --batch $b --steps 40 --warmup 4 "$@" 2>> $OUT/err.txt | LABEL="$label" python -c "
import json,sys,os
r=json.loads(sys.stdin.read()); print('%s batch %5d %-40s %.3f ms' % (r['config']['arch'], r['config']['global_batch'], os.environ['LABEL'], r['ms_per_step']))" >> $OUT/step_ab.txt
  unset CV_HIP_LIB
}
for round in 1 2 3; do
  for b in 1250 2500; do
    run "round-4 library" $b $A
    run "in-tree (sched 255)" $b ""
    run "in-tree, sched 127 (no late fc4 wgrad / chain)" $b "" --sched 127
  done
  run "round-4 library" 10000 $A
  run "in-tree (sched 255)" 10000 ""
  run "in-tree (sched 255)" 1250 "" --arch slim
  run "round-4 library" 1250 $A --arch slim
done
sort $OUT/step_ab.txt; grep -i "error\|Traceback" $OUT/err.txt | head -5
timeout 300 bash tools/gpu_train_profile.sh $TAG 1250 - > /dev/null 2>&1
python - <<'PY'
import csv
rows=[r for r in csv.DictReader(open('gpurun_out/%s/train_1250_-_kernel_stats.csv' % 'r05f')) if int(r["Calls"]) in (23,46,69)]
for r in rows: print("%-70s %3s x %7.1f us" % (r["Name"].replace("(anonymous namespace)::","").replace("float __vector(4)","f4").replace("void ","")[:70], r["Calls"], float(r["AverageNs"])/1e3))
PY
