#!/bin/bash
# round-2 GPU call B: packed SELU + pool-before-activation kernels: parity, bench, kernel traces
set -u
OUT=gpurun_out/${1:-r02_b}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests -m gpu -q --maxfail=10 -k "parity or pipeline or dp" > $OUT/pytest_sel.log 2>&1
echo "pytest rc=$?" >> $OUT/status.txt
python bench.py --steps 64 --warmup 4 --no-cpu > $OUT/bench.json 2> $OUT/bench.err
python bench.py --steps 64 --warmup 4 --no-cpu --arch slim > $OUT/bench_slim.json 2>> $OUT/bench.err
for b in 1250 10000; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_train_$b -o t -- python bench.py --mode train --batch $b --steps 20 --warmup 3 > $OUT/prof_train_$b.json 2> $OUT/prof_train_$b.err
  f=$(find $OUT/prof_train_$b -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/train_${b}_kernel_stats.csv
  f=$(find $OUT/prof_train_$b -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python tools/trace_timeline.py "$f" > $OUT/train_${b}_timeline.txt 2>&1
  rm -rf $OUT/prof_train_$b
done
tail -3 $OUT/pytest_sel.log
for f in $OUT/bench.json $OUT/bench_slim.json; do python -c "
import json,sys; d=json.loads(open('$f').read()); print('%.0f cand/s' % d['value'], 'dominant frac %.3f' % d['roofline']['frac'], 'whole %.3f' % d['roofline']['whole_path_frac']); [print('  ', k['kernel'], round(k['avg_ms'],4), 'ms', round(k['tflops'],1), 'TF') for k in d['kernels']]"; done
