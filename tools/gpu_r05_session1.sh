#!/bin/bash
# Round-5 measurement session 1 on one box: the MFMA-shape probe, front2_tm's phase probe + SQ counters, host-boundness of
# the optimizer step, the fixed cost of the gradient exchange by form (one RCCL rank: no byte moves), a step timeline, GPU tests.
#   gpurun --timeout 1500 -- 'bash tools/gpu_r05_session1.sh r05a'
set -u
TAG=${1:-r05a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Wno-unused-value -Rpass-analysis=kernel-resource-usage tools/mfma_feed.hip -o $OUT/mfma_feed 2> $OUT/mfma_feed_build.txt
{ grep -E "Function Name|VGPRs:|Spill" $OUT/mfma_feed_build.txt | sed 's/^.*remark: *//' | paste - - - - ; timeout 600 $OUT/mfma_feed; } > $OUT/mfma_shape_feed.txt 2>&1
rm -f $OUT/mfma_feed; cat $OUT/mfma_shape_feed.txt
CV_HIP_LIB=$PWD/clairvoyante_amd/csrc/libclairvoyante_hip_phases.so timeout 300 python tools/gpu_front2_phases.py > $OUT/front2_phases.txt 2> $OUT/front2_phases.err; cat $OUT/front2_phases.txt; tail -3 $OUT/front2_phases.err
timeout 600 bash tools/gpu_pmc_infer_sq.sh ${TAG}_sq > /dev/null 2>&1; cp gpurun_out/${TAG}_sq/infer_sq_summary.txt $OUT/ 2>/dev/null; cat $OUT/infer_sq_summary.txt
timeout 300 python tools/gpu_step_host_probe.py > $OUT/step_host_probe.txt 2> $OUT/step_host_probe.err; cat $OUT/step_host_probe.txt
# exchange forms with ONE RCCL rank (a sum over one rank moves no byte: what is measured is the machinery)
ex() {  # label, batch, env...
  local label=$1 b=$2; shift 2
  env "$@" python bench.py --mode train --batch $b --steps 40 --warmup 4 2>> $OUT/exchange.err | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('batch %5d %-28s step %.3f ms  compute %s  exchange alone %s  plan %s' % (r['config']['global_batch'], '$label', r['ms_per_step'], r.get('compute_ms_per_step'), r.get('exchange_ms'), r.get('exchange_plan')))" >> $OUT/exchange_fixed_cost.txt
}
D="CV_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29411"
for round in 1 2; do
  for b in 1250 10000; do
    ex "no process group" $b CV_NOTHING=1
    ex "one collective" $b $D CV_EXCHANGE=one
    ex "split, on the step's streams" $b $D CV_EXCHANGE=split
    ex "split, async (round 4)" $b $D CV_EXCHANGE=split CV_EXCHANGE_ASYNC=1
  done
done
sort $OUT/exchange_fixed_cost.txt; tail -3 $OUT/exchange.err
timeout 300 bash tools/gpu_train_timeline.sh $TAG 1250 - > /dev/null 2>&1; cat $OUT/timeline_1250_.txt
python -m pytest tests -m gpu -q --maxfail=10 --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_gpu.log
