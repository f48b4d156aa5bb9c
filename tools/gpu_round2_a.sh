#!/bin/bash
# round-2 GPU call A: new parity tests, DP tests, training-step A/B (side stream, deferred losses), kernel trace at the
# per-rank batch of config 4 (1 250) and at train.py's batch
set -u
OUT=gpurun_out/r02_a
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests -m gpu -q --maxfail=10 -x -k "dp or parity" > $OUT/pytest_new.log 2>&1
echo "pytest new rc=$?" >> $OUT/status.txt
for b in 1250 2500 10000; do
  for ov in 0 1; do
    python bench.py --mode train --batch $b --overlap $ov --steps 50 --warmup 5 >> $OUT/train_ab.jsonl 2>> $OUT/train_ab.err
    python bench.py --mode train --batch $b --overlap $ov --steps 50 --warmup 5 --sync-loss >> $OUT/train_ab.jsonl 2>> $OUT/train_ab.err
  done
done
echo "train ab done" >> $OUT/status.txt
for b in 1250 10000; do
  rocprofv3 --kernel-trace --stats -d $OUT/prof_train_$b -o t -- python bench.py --mode train --batch $b --steps 20 --warmup 3 > $OUT/prof_train_$b.json 2> $OUT/prof_train_$b.err
  f=$(find $OUT/prof_train_$b -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/train_${b}_kernel_stats.csv
  rm -rf $OUT/prof_train_$b
done
echo "prof done" >> $OUT/status.txt
python bench.py --steps 64 --warmup 4 > $OUT/bench.json 2> $OUT/bench.err
python -m pytest tests -m gpu -q --maxfail=10 > $OUT/pytest_gpu.log 2>&1
echo "pytest all rc=$?" >> $OUT/status.txt
tail -5 $OUT/pytest_new.log; tail -5 $OUT/pytest_gpu.log; cat $OUT/train_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print(d['config']['global_batch'], d['config']['weight_gradients'], d['config']['losses'], '%.3f ms' % d['ms_per_step'])
"
