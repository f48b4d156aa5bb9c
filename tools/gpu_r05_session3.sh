#!/bin/bash
# Round-5 session 3: the fused tail of the tiny-batch forward (tests + A/B), and which of the step's changes costs the
# 10 000 step its 60 us (r04 library / new default / late loss header / every layout packed), alternating on one box.
#   gpurun --timeout 1500 -- 'bash tools/gpu_r05_session3.sh r05c'
set -u
TAG=${1:-r05c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests/test_gpu_dp.py tests/test_gpu_train_parity.py -m gpu -q -x --durations=6 -k "tiny or alpha_dropout or 1250 or reproducible or single_chain or packed_layouts or backward_kernel_variants or side_stream" > $OUT/pytest_tail.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest_tail.log
A=$PWD/clairvoyante_amd/csrc/libclairvoyante_hip_base.so
run() {  # label, batch, lib ('' = in-tree), bench flags
  local label=$1 b=$2 lib=$3; shift 3
  if [ -n "$lib" ]; then export CV_HIP_LIB=$lib; else unset CV_HIP_LIB; fi
  python bench.py --mode train --batch $b --steps 40 --warmup 4 "$@" 2>> $OUT/err.txt | LABEL="$label" python -c "
import json,sys,os
r=json.loads(sys.stdin.read()); print('batch %5d %-44s %.3f ms' % (r['config']['global_batch'], os.environ['LABEL'], r['ms_per_step']))" >> $OUT/step_ab.txt
  unset CV_HIP_LIB
}
for round in 1 2 3; do
  run "round-4 library" 10000 $A
  run "in-tree, default" 10000 ""
  run "in-tree, loss header at the tail (dbg5=3)" 10000 "" --dbg 5=3
  run "in-tree, every layout packed (dbg5=4)" 10000 "" --dbg 5=4
  run "round-4 library" 1250 $A
  run "in-tree, default (fused tail)" 1250 ""
  run "in-tree, tail as three kernels (dbg2=5)" 1250 "" --dbg 2=5
  run "in-tree, conv1 wgrad on a side stream (dbg5=2)" 1250 "" --dbg 5=2
  run "in-tree, thread-per-row unpool (dbg2=4)" 1250 "" --dbg 2=4
done
sort $OUT/step_ab.txt; grep -i "error\|Traceback" $OUT/err.txt | head -5
