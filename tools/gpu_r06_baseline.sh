#!/bin/bash
# Round-6 baseline on one box: per-stage inference ladder (full, slim), the training step at large batches (the
# asymptote of the per-candidate step time), per-kernel stats of the serial step at 10 000 and 32 768.
set -u
OUT=gpurun_out/${1:-r06_base}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python tools/gpu_infer_stage_ladder.py full > $OUT/infer_stage_ladder_full.txt 2> $OUT/err.txt
python tools/gpu_infer_stage_ladder.py slim > $OUT/infer_stage_ladder_slim.txt 2>> $OUT/err.txt
SIZES="8000 10000 12288 16384 24576 32768 49152 65536" STEPS=30 bash tools/gpu_step_size_sweep.sh $(basename $OUT) > /dev/null 2>&1
for b in 10000 32768; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_t$b -o t -- python bench.py --mode train --batch $b --steps 20 --warmup 3 --overlap 0 > $OUT/train_serial_$b.json 2> $OUT/prof_t$b.err
  f=$(find $OUT/prof_t$b -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/train_serial_${b}_kernel_stats.csv; rm -rf $OUT/prof_t$b
done
tail -3 $OUT/infer_stage_ladder_full.txt; cat $OUT/step_sizes.txt
