#!/bin/bash
# Round-5 session 5: the full GPU suite on the current tree, then round-4 library against it on one box (training step
# at 1 250 / 2 500 / 5 000 / 10 000, slim, and the inference pass), a serial per-kernel profile of the 1 250 step.
#   gpurun --timeout 1800 -- 'bash tools/gpu_r05_session5.sh r05e'
set -u
TAG=${1:-r05e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests -m gpu -q --maxfail=10 --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -14 $OUT/pytest_gpu.log
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
    run "in-tree" $b ""
  done
  run "in-tree, no fused head dgrad (sched 95)" 10000 "" --sched 95
  run "in-tree, memset (sched 63)" 1250 "" --sched 63
  run "round-4 library" 10000 $A --arch slim
  run "in-tree" 10000 "" --arch slim
  run "round-4 library" 1250 $A --arch slim
  run "in-tree" 1250 "" --arch slim
done
sort $OUT/step_ab.txt; grep -i "error\|Traceback" $OUT/err.txt | head -5
for which in A B; do
  if [ $which = A ]; then export CV_HIP_LIB=$A; else unset CV_HIP_LIB; fi
  python bench.py --no-cpu --no-extras --steps 16 --warmup 3 2>> $OUT/err.txt | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$which infer %.3f M/s' % (r['value']/1e6), ' '.join('%.4f' % k['avg_ms'] for k in r['kernels']))" >> $OUT/infer_ab.txt
done; unset CV_HIP_LIB; cat $OUT/infer_ab.txt
timeout 300 bash tools/gpu_train_profile.sh $TAG 1250 - > /dev/null 2>&1; ls $OUT | head -30
