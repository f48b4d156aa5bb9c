#!/bin/bash
# Training-step A/B on ONE box, alternating (box-to-box differences are as large as the effects measured): the base
# library (csrc/libclairvoyante_hip_base.so, built from that commit with clairvoyante_amd/build.py's flags) against the
# in-tree build, plus any number of in-tree settings given as "label|bench flags".  How profiles/r05/step_ab_*.txt were made.
#   bash tools/gpu_step_ab.sh TAG "1250 10000" 3 "sched 255|--sched 255" "two kernels for conv1|--dbg 4=4"
set -u
TAG=${1:-stepab}; BATCHES=${2:-"1250 10000"}; R=${3:-3}; shift 3 || true
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
A=$PWD/clairvoyante_amd/csrc/libclairvoyante_hip_base.so
run() {  # label, batch, lib ('' = in-tree), bench flags
  local label=$1 b=$2 lib=$3; shift 3
  if [ -n "$lib" ]; then export CV_HIP_LIB=$lib; else unset CV_HIP_LIB; fi
  python bench.py --mode train --batch $b --steps 40 --warmup 4 "$@" 2>> $OUT/err.txt | LABEL="$label" python -c "
import json,sys,os
r=json.loads(sys.stdin.read()); print('%s batch %5d %-52s %.3f ms' % (r['config']['arch'], r['config']['global_batch'], os.environ['LABEL'], r['ms_per_step']))" >> $OUT/step_ab.txt
  unset CV_HIP_LIB
}
for round in $(seq $R); do
  for b in $BATCHES; do
    [ -f $A ] && run "base library" $b $A
    run "in-tree" $b ""
    for spec in "$@"; do run "in-tree, ${spec%%|*}" $b "" ${spec#*|}; done
    [ -f $A ] && run "base library" $b $A --arch slim
    run "in-tree" $b "" --arch slim
  done
done
sort $OUT/step_ab.txt; grep -i "error\|Traceback" $OUT/err.txt | head -5
