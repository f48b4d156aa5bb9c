#!/bin/bash
# Round-5 session 2: the re-cut step (one fork marker, shared launch-site markers, chained join, early loss header, conv1's
# weight gradient on the main stream at tiny batches, row-segment unpool, per-layout packing) -- tests, then the round-4
# library against the in-tree one on ONE box, the exchange forms with one RCCL rank, a step timeline.
#   gpurun --timeout 1500 -- 'bash tools/gpu_r05_session2.sh r05b'
set -u
TAG=${1:-r05b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests/test_gpu_dp.py tests/test_gpu_train_parity.py tests/test_gpu_parity.py -m gpu -q -x --durations=6 -k "not ranks and not rank and not data_parallel and not empty_shards" > $OUT/pytest_step.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest_step.log
bash tools/gpu_lib_ab.sh ${TAG}_lib 3 train > $OUT/lib_ab_r04_head_vs_recut_step.txt 2>&1; cat $OUT/lib_ab_r04_head_vs_recut_step.txt
ex() {  # label, batch, env...
  local label=$1 b=$2; shift 2
  env "$@" python bench.py --mode train --batch $b --steps 40 --warmup 4 2>> $OUT/exchange.err | LABEL="$label" python -c "
import json,sys,os
r=json.loads(sys.stdin.read()); print('batch %5d %-30s step %.3f ms  compute %s  exchange alone %s  plan %s' % (r['config']['global_batch'], os.environ['LABEL'], r['ms_per_step'], r.get('compute_ms_per_step'), r.get('exchange_ms'), r.get('exchange_plan')))" >> $OUT/exchange_fixed_cost.txt
}
D="CV_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29411"
for round in 1 2; do
  for b in 1250 10000; do
    ex "no process group" $b CV_NOTHING=1
    ex "one collective" $b $D CV_EXCHANGE=one
    ex "split, sync on step streams" $b $D CV_EXCHANGE=split
    ex "split, async (round 4 form)" $b $D CV_EXCHANGE=split CV_EXCHANGE_ASYNC=1
  done
done
sort $OUT/exchange_fixed_cost.txt; grep -i "error\|Traceback" $OUT/exchange.err | head -5
timeout 300 bash tools/gpu_train_timeline.sh $TAG 1250 - > /dev/null 2>&1; cat $OUT/timeline_1250_.txt
timeout 300 bash tools/gpu_train_timeline.sh $TAG 10000 - > /dev/null 2>&1; tail -3 $OUT/timeline_10000_.txt
