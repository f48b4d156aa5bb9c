#!/bin/bash
# Training-step A/B of library options on ONE box, alternating (box-to-box differences are as large as the effects):
#   bash tools/gpu_step_opt_ab.sh TAG "10000 1250" 3 "" "dense_rag=13" "dense_rag=14"      ("" = defaults)
set -u
TAG=${1:-optab}; BATCHES=${2:-"10000"}; R=${3:-3}; shift 3 || true
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for round in $(seq $R); do
  for b in $BATCHES; do
    for spec in "$@"; do
      python bench.py --mode train --arch ${ARCH:-full} --batch $b --steps 40 --warmup 4 ${spec:+--opt $spec} 2>> $OUT/err.txt | LABEL="${spec:-defaults}" python -c "
import json,sys,os
r=json.loads(sys.stdin.read()); print('%s batch %5d %-40s %.3f ms' % (r['config']['arch'], r['config']['global_batch'], os.environ['LABEL'], r['ms_per_step']))" >> $OUT/ab.txt
    done
  done
done
sort $OUT/ab.txt; grep -i "error\|Traceback" $OUT/err.txt | head -3
