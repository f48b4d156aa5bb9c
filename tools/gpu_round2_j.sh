#!/bin/bash
set -u
OUT=gpurun_out/${1:-r02_j}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests -m gpu -q --maxfail=10 -k "pipeline or dp" > $OUT/pytest_sel.log 2>&1
echo "pytest rc=$?" >> $OUT/status.txt
for b in 1250 2500 5000 10000 20000 40000; do
  python bench.py --mode train --batch $b --steps 50 --warmup 5 >> $OUT/train_ab.jsonl 2>> $OUT/train_ab.err
  python bench.py --mode train --batch $b --steps 50 --warmup 5 --tiny 0 >> $OUT/train_ab.jsonl 2>> $OUT/train_ab.err
done
python bench.py --mode train --batch 10000 --steps 50 --warmup 5 --arch slim >> $OUT/train_ab.jsonl 2>> $OUT/train_ab.err
tail -4 $OUT/pytest_sel.log
python - <<PY
import json
for l in open("$OUT/train_ab.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    print(d['config']['arch'], d['config']['global_batch'], '%.3f ms' % d['ms_per_step'], '%.2f M/s' % (d['value'] / 1e6))
PY
