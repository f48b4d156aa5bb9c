#!/bin/bash
# timeline of one optimizer step (two streams): bash tools/gpu_train_timeline.sh TAG BATCH "dbg" [extra bench args]
set -u
OUT=gpurun_out/${1:-r03t}
B=${2:-1250}
dd=${3:-}; [ "$dd" = "-" ] && dd=""
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
tag=$(echo "${B}_${dd}" | tr '=,' '__')
rocprofv3 --kernel-trace --output-format csv -d $OUT/tl_$tag -o t -- python bench.py --mode train --batch $B --steps 12 --warmup 3 --dbg "$dd" ${4:-} > $OUT/tl_$tag.json 2> $OUT/tl_$tag.err
f=$(find $OUT/tl_$tag -name "*kernel_trace.csv" | head -1)
python tools/trace_timeline.py "$f" > $OUT/timeline_$tag.txt
rm -rf $OUT/tl_$tag
cat $OUT/timeline_$tag.txt
