#!/bin/bash
# Round-5 session 4: which bit of the re-cut schedule costs the 10 000 step time (train_sched bits), same box, alternating.
#   gpurun --timeout 1200 -- 'bash tools/gpu_r05_session4.sh r05d'
set -u
TAG=${1:-r05d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
A=$PWD/clairvoyante_amd/csrc/libclairvoyante_hip_base.so
run() {  # label, batch, lib ('' = in-tree), bench flags
  local label=$1 b=$2 lib=$3; shift 3
  if [ -n "$lib" ]; then export CV_HIP_LIB=$lib; else unset CV_HIP_LIB; fi
  python bench.py --mode train --batch $b --steps 40 --warmup 4 "$@" 2>> $OUT/err.txt | LABEL="$label" python -c "
import json,sys,os
r=json.loads(sys.stdin.read()); print('batch %5d %-52s %.3f ms' % (r['config']['global_batch'], os.environ['LABEL'], r['ms_per_step']))" >> $OUT/step_ab.txt
  unset CV_HIP_LIB
}
for round in 1 2 3; do
  run "round-4 library" 10000 $A
  run "in-tree, sched 31 (default)" 10000 ""
  run "in-tree, sched 0 (round-4 schedule)" 10000 "" --sched 0
  run "in-tree, sched 4 (one fork marker only)" 10000 "" --sched 4
  run "in-tree, sched 8 (shared site markers only)" 10000 "" --sched 8
  run "in-tree, sched 16 (per-layout packing only)" 10000 "" --sched 16
  run "in-tree, sched 31, slim" 10000 "" --arch slim
  run "round-4 library, slim" 10000 $A --arch slim
  run "round-4 library" 2500 $A
  run "in-tree, sched 31" 2500 ""
  run "round-4 library" 5000 $A
  run "in-tree, sched 31" 5000 ""
done
sort $OUT/step_ab.txt; grep -i "error\|Traceback" $OUT/err.txt | head -5
