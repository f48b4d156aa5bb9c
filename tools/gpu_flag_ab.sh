#!/bin/bash
# training-step A/B over bench.py flag sets on ONE box, alternating: bash tools/gpu_flag_ab.sh TAG "flags1|flags2|.." [rounds] [batches]
set -u
OUT=gpurun_out/${1:-flagab}; R=${3:-2}; BATCHES=${4:-"10000 1250"}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
IFS='|' read -ra SETS <<< "${2:-}"
for r in $(seq $R); do
  for d in "${SETS[@]}"; do
    for b in $BATCHES; do python bench.py --mode train --batch $b --steps 40 --warmup 4 $d 2>> $OUT/err.txt | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('full %5d %-24s %.3f ms' % (r['config']['global_batch'], '$d', r['ms_per_step']))" >> $OUT/ab.txt; done
    python bench.py --mode train --arch slim --batch 10000 --steps 40 --warmup 4 $d 2>> $OUT/err.txt | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('slim 10000 %-24s %.3f ms' % ('$d', r['ms_per_step']))" >> $OUT/ab.txt
  done
done
sort $OUT/ab.txt
