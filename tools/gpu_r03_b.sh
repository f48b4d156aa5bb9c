#!/bin/bash
# round 3: GPU tests + training-step A/B of the development switches
#   bash tools/gpu_r03_b.sh TAG "dbg settings separated by spaces, e.g. '' 3=1"
set -u
OUT=gpurun_out/${1:-r03b}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_pileup.py --deselect tests/test_bam_native.py > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" > $OUT/status.txt
tail -4 $OUT/pytest_gpu.log
for d in ${2:-""}; do
  [ "$d" = "-" ] && d=""
  for b in 10000 1250; do
    for ov in 1 0; do
      python bench.py --mode train --batch $b --steps 30 --warmup 3 --overlap $ov --dbg "$d" >> $OUT/train_ab.jsonl 2>> $OUT/bench.err
    done
  done
  python bench.py --mode train --batch 10000 --arch slim --steps 30 --warmup 3 --dbg "$d" >> $OUT/train_ab.jsonl 2>> $OUT/bench.err
done
python - $OUT/train_ab.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    r = json.loads(l)
    print("%-5s %6d dbg %-8s %-12s %.3f ms" % (r["config"]["arch"], r["config"]["global_batch"], r["config"]["dbg"], r["config"]["weight_gradients"], r["ms_per_step"]))
PY
cat $OUT/status.txt
