#!/bin/bash
# Training-step A/B of two builds of the library on ONE box, alternating, over a list of batch sizes:
#   bash tools/gpu_lib_step_ab.sh TAG "320 640 1250" 3        (ARCH=slim for the slim topology)
# A = clairvoyante_amd/csrc/libclairvoyante_hip_base.so (tools/build_variant_lib.py base, from the commit to compare
# with), B = the in-tree build.
set -u
TAG=${1:-libstep}; BATCHES=${2:-"1250 10000"}; R=${3:-3}
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
A=$PWD/clairvoyante_amd/csrc/libclairvoyante_hip_base.so
for round in $(seq $R); do
  for b in $BATCHES; do
    for which in A B; do
      if [ $which = A ]; then export CV_HIP_LIB=$A; else unset CV_HIP_LIB; fi
      python bench.py --mode train --arch ${ARCH:-full} --batch $b --steps 40 --warmup 4 ${OPT:+--opt $OPT} 2>> $OUT/err.txt | W=$which python -c "
import json,sys,os
r=json.loads(sys.stdin.read()); print('%s batch %5d %s %.3f ms' % (r['config']['arch'], r['config']['global_batch'], os.environ['W'], r['ms_per_step']))" >> $OUT/ab.txt
    done
  done
done
sort $OUT/ab.txt; grep -i "error\|Traceback" $OUT/err.txt | head -3
