#!/bin/bash
set -u
OUT=gpurun_out/r05k; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests/test_gpu_dp.py -m gpu -q -x -k "backward_kernel_variants or every_gradient or tiny_batch or side_stream" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
bash tools/gpu_step_ab.sh r05k_ab "1250 5000 10000" 3 "side work behind conv1 (sched 3839)|--sched 3839" > $OUT/step_ab.txt 2>&1; cat $OUT/step_ab.txt
