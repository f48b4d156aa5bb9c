#!/bin/bash
set -u
OUT=gpurun_out/${1:-r02_h}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests -m gpu -q --maxfail=10 -k "parity" > $OUT/pytest_sel.log 2>&1
echo "pytest rc=$?" >> $OUT/status.txt
for v in 495 239; do
python bench.py --steps 64 --warmup 4 --no-cpu --arch slim --variant $v > $OUT/bench_slim_v$v.json 2>> $OUT/bench.err
done
tail -4 $OUT/pytest_sel.log
for f in $OUT/bench_slim_v495.json $OUT/bench_slim_v239.json; do python -c "
import json,sys; d=json.loads(open('$f').read()); print('%.0f cand/s' % d['value'], 'dominant frac %.3f' % d['roofline']['frac'], 'whole %.3f' % d['roofline']['whole_path_frac']); [print('  ', k['kernel'], k['kernel_name'], round(k['avg_ms'],4), 'ms', round(k['tflops'],1), 'TF') for k in d['kernels']]"; done
