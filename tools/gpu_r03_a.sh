#!/bin/bash
# round 3, first GPU run: new tests, bench line with the slim / train legs, serialized per-kernel profile of the
# training step (train_overlap 0: kernel durations are not inflated by the second stream), position-split A/B
set -u
OUT=gpurun_out/${1:-r03a}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py -m gpu -q -x -k "float64 or gradients or reproducible" > $OUT/pytest_subset.log 2>&1
echo "pytest rc=$?" > $OUT/status.txt
python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?" >> $OUT/status.txt
for b in 10000 1250; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_t$b -o t -- python bench.py --mode train --batch $b --steps 20 --warmup 3 --overlap 0 > $OUT/train_serial_$b.json 2> $OUT/prof_t$b.err
  f=$(find $OUT/prof_t$b -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/train_serial_${b}_kernel_stats.csv; rm -rf $OUT/prof_t$b
done
for d in "" "0=1" "0=2" "0=3" "0=4" "0=6" "0=8" "1=1" "1=2" "1=3" "1=4" "2=1" "2=2"; do
  for ov in 1 0; do
    python bench.py --mode train --batch 10000 --steps 30 --warmup 3 --overlap $ov --dbg "$d" >> $OUT/train_ab.jsonl 2>> $OUT/bench.err
  done
done
python - $OUT/train_ab.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    r = json.loads(l)
    print("%-8s %-12s %.3f ms" % (r["config"]["dbg"], r["config"]["weight_gradients"], r["ms_per_step"]))
PY
cat $OUT/status.txt; tail -3 $OUT/pytest_subset.log
