#!/bin/bash
# A/B of a build flag on the training step: bash tools/gpu_build_flag_ab.sh TAG -DCV_SOMETHING [batch]
# builds the library twice on the box (default, with the flag), runs the training-variant tests and a serialized profile
set -u
OUT=gpurun_out/${1:-flagab}; FLAG=${2:-}; B=${3:-10000}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for mode in default flag; do
  if [ $mode = flag ]; then export CV_EXTRA_FLAGS="$FLAG"; else unset CV_EXTRA_FLAGS; fi
  python -c "from clairvoyante_amd import build; build.build(force=True)" > $OUT/build_$mode.log 2>&1
  python -m pytest tests/test_gpu_pipeline.py -m gpu -q -x -k "gradients or reproducible" > $OUT/pytest_$mode.log 2>&1; echo "$mode tests rc=$?"
  for b in $B 1250; do python bench.py --mode train --batch $b --steps 40 --warmup 4 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$mode', r['config']['global_batch'], '%.3f ms' % r['ms_per_step'])"; done
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$mode -o t -- python bench.py --mode train --batch $B --steps 20 --warmup 3 --overlap 0 > /dev/null 2> $OUT/prof_$mode.err
  f=$(find $OUT/prof_$mode -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/${mode}_kernel_stats.csv; rm -rf $OUT/prof_$mode
  grep -E "wgrad_conv_cm|wgrad_dense_cm<21" $OUT/${mode}_kernel_stats.csv | python -c "
import sys,csv
for r in csv.reader(sys.stdin): print('   $mode', r[0][:60], r[3])"
done
