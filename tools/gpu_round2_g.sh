#!/bin/bash
set -u
OUT=gpurun_out/${1:-r02_g}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests -m gpu -q --maxfail=10 -k "parity or pipeline or dp" > $OUT/pytest_sel.log 2>&1
echo "pytest rc=$?" >> $OUT/status.txt
python tools/gpu_small_batch_probe.py 111 > $OUT/small_v111.txt 2>&1
python tools/gpu_small_batch_probe.py 239 > $OUT/small_v239.txt 2>&1
for b in 1250 2500; do
  python bench.py --mode train --batch $b --steps 50 --warmup 5 >> $OUT/train_ab.jsonl 2>> $OUT/train_ab.err
  python bench.py --mode train --batch $b --steps 50 --warmup 5 --ksplit 0 >> $OUT/train_ab.jsonl 2>> $OUT/train_ab.err
done
tail -4 $OUT/pytest_sel.log; cat $OUT/small_v111.txt $OUT/small_v239.txt
python - <<PY
import json
for l in open("$OUT/train_ab.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    print(d['config']['arch'], d['config']['global_batch'], '%.3f ms' % d['ms_per_step'], '%.2f M/s' % (d['value'] / 1e6))
PY
