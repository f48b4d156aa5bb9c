#!/bin/bash
# Round-5 session 9: fused first-layer kernel with block prefetch, bias reduce in one walk, train_sched 767 as default
set -u
TAG=${1:-r05i}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests/test_gpu_dp.py tests/test_gpu_train_parity.py -m gpu -q --maxfail=5 --durations=5 -k "not ranks and not rank and not data_parallel and not empty_shards and not bench_line and not several_slices" > $OUT/pytest_step.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_step.log
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
  for b in 1250 2500 5000 10000; do
    run "round-4 library" $b $A
    run "in-tree (train_sched 767)" $b ""
  done
  run "round-4 library" 10000 $A --arch slim
  run "in-tree (train_sched 767)" 10000 "" --arch slim
  run "in-tree, train_sched 255" 10000 "" --arch slim --sched 255
  run "round-4 library" 1250 $A --arch slim
  run "in-tree (train_sched 767)" 1250 "" --arch slim
done
sort $OUT/step_ab.txt; grep -i "error\|Traceback" $OUT/err.txt | head -5
timeout 300 bash tools/gpu_train_profile.sh $TAG 1250 - > /dev/null 2>&1
timeout 300 bash tools/gpu_train_profile.sh $TAG 10000 - > /dev/null 2>&1
python - <<'PY'
import csv
for b in (1250, 10000):
    rows=[r for r in csv.DictReader(open('gpurun_out/r05i/train_%d_-_kernel_stats.csv' % b)) if int(r["Calls"]) in (23,46,69)]
    print("== serial kernels at", b, " sum %.1f us" % (sum(float(r["TotalDurationNs"]) for r in rows)/23e3))
    for r in rows:
        if any(k in r["Name"] for k in ("conv1", "train_tail", "heads_train", "adam")): print("%-70s %3s x %7.1f us" % (r["Name"].replace("(anonymous namespace)::","").replace("float __vector(4)","f4").replace("void ","")[:70], r["Calls"], float(r["AverageNs"])/1e3))
PY
