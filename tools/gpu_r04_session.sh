#!/bin/bash
# Round-4 measurement session on one box: GPU tests, training-step A/B (fc4 forward ring: dbg3 = 2 is the three-slot
# ring), base library (round-3 head) against the in-tree build, the cheaper-activation A/B, the PMC traffic passes.
#   gpurun --timeout 2400 -- 'bash tools/gpu_r04_session.sh TAG GITHEAD'
set -u
TAG=${1:-r04s}; HEAD=${2:-unknown}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests -m gpu -q --maxfail=15 --durations=12 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/status.txt
tail -4 $OUT/pytest_gpu.log
bash tools/gpu_flag_ab.sh ${TAG}_ring "--dbg 3=2|--dbg 3=0" 3 "10000" > $OUT/train_ab_fc4_forward_ring.txt 2>&1
cat $OUT/train_ab_fc4_forward_ring.txt
bash tools/gpu_lib_ab.sh ${TAG}_lib 2 > $OUT/lib_ab_r03_head_vs_now.txt 2>&1
cat $OUT/lib_ab_r03_head_vs_now.txt
bash tools/gpu_fast_selu_ab.sh ${TAG}_selu > $OUT/fast_selu_ab.txt 2>&1
grep -v "^$" $OUT/fast_selu_ab.txt | tail -40
bash tools/gpu_pmc.sh ${TAG}_pmc $HEAD > $OUT/pmc.log 2>&1
cp gpurun_out/${TAG}_pmc/pmc_traffic.json $OUT/pmc_traffic.json 2>/dev/null
tail -12 $OUT/pmc.log
