#!/bin/bash
# round-2 GPU call D: tiny-batch training variants: parity, A/B, timeline
set -u
OUT=gpurun_out/${1:-r02_d}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests -m gpu -q --maxfail=10 -k "pipeline or dp" > $OUT/pytest_sel.log 2>&1
echo "pytest rc=$?" >> $OUT/status.txt
for b in 1250 2500 10000; do
  python bench.py --mode train --batch $b --steps 50 --warmup 5 >> $OUT/train_ab.jsonl 2>> $OUT/train_ab.err
  python bench.py --mode train --batch $b --steps 50 --warmup 5 --ksplit 0 >> $OUT/train_ab.jsonl 2>> $OUT/train_ab.err
  python bench.py --mode train --batch $b --steps 50 --warmup 5 --tiny 0 >> $OUT/train_ab.jsonl 2>> $OUT/train_ab.err
done
python bench.py --mode train --batch 1250 --steps 50 --warmup 5 --arch slim >> $OUT/train_ab.jsonl 2>> $OUT/train_ab.err
python bench.py --mode train --batch 10000 --steps 50 --warmup 5 --arch slim >> $OUT/train_ab.jsonl 2>> $OUT/train_ab.err
for b in 1250 10000; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train_$b -o t -- python bench.py --mode train --batch $b --steps 20 --warmup 3 > $OUT/prof_train_$b.json 2> $OUT/prof_train_$b.err
  f=$(find $OUT/prof_train_$b -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/train_${b}_kernel_stats.csv
  f=$(find $OUT/prof_train_$b -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python tools/trace_timeline.py "$f" > $OUT/train_${b}_timeline.txt 2>&1
  rm -rf $OUT/prof_train_$b
done
tail -4 $OUT/pytest_sel.log
python - <<PY
import json
for l in open("$OUT/train_ab.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    print(d['config']['arch'], d['config']['global_batch'], '%.3f ms' % d['ms_per_step'], '%.2f M/s' % (d['value'] / 1e6))
PY
