#!/bin/bash
# Development probe: the training step over a ladder of batch sizes (one GPU, 40 steps each), full and slim -- looks for
# steps in the time per step where the library changes kernels or schedule by size.  bash tools/gpu_step_size_sweep.sh TAG
set -u
OUT=gpurun_out/${1:-stepsizes}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for arch in ${ARCHS:-full slim}; do
  for b in ${SIZES:-320 640 960 1250 1280 1296 1600 2000 2500 2561 3200 4000 5000 6000 6400 6416 7000 8000 10000 12288 16384 24576 32768 49152 65536}; do
    python bench.py --mode train --arch $arch --batch $b --steps ${STEPS:-40} --warmup 4 2>> $OUT/err.txt | python -c "
import json,sys
r=json.loads(sys.stdin.read()); b=r['config']['global_batch']; print('%s batch %5d (%4d groups)  %.3f ms per step  %6.1f ns per candidate' % (r['config']['arch'], b, (b+15)//16, r['ms_per_step'], r['ms_per_step']*1e6/b))" >> $OUT/step_sizes.txt
  done
done
cat $OUT/step_sizes.txt
