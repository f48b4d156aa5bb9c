#!/bin/bash
# Two builds of the library timed on ONE GPU box, alternating (box-to-box differences are as large as the effects
# being measured): bash tools/gpu_lib_ab.sh TAG [rounds] [train = training step only]
# A = clairvoyante_amd/csrc/libclairvoyante_hip_base.so (built from the commit to compare with), B = the in-tree build.
set -u
OUT=gpurun_out/${1:-libab}; R=${2:-3}; ONLY=${3:-all}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
A=$PWD/clairvoyante_amd/csrc/libclairvoyante_hip_base.so
for r in $(seq $R); do
  for which in A B; do
    if [ $which = A ]; then export CV_HIP_LIB=$A; else unset CV_HIP_LIB; fi
    [ $ONLY = train ] || python bench.py --no-cpu --no-extras --steps 12 --warmup 3 2>> $OUT/err.txt | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$which infer      %.3f M/s' % (r['value']/1e6), ' '.join('%.4f' % k['avg_ms'] for k in r['kernels']))" >> $OUT/ab.txt
    [ $ONLY = train ] || python bench.py --no-cpu --no-extras --arch slim --steps 12 --warmup 3 2>> $OUT/err.txt | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$which infer slim %.3f M/s' % (r['value']/1e6))" >> $OUT/ab.txt
    for b in 10000 1250; do python bench.py --mode train --batch $b --steps 40 --warmup 4 2>> $OUT/err.txt | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$which train %5d %.3f ms' % (r['config']['global_batch'], r['ms_per_step']))" >> $OUT/ab.txt; done
    python bench.py --mode train --arch slim --batch 10000 --steps 40 --warmup 4 2>> $OUT/err.txt | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$which train slim  %.3f ms' % r['ms_per_step'])" >> $OUT/ab.txt
  done
done
sort $OUT/ab.txt | awk '{k=$1" "$2" "$3; if ($2=="infer" && $3!="slim") k=$1" "$2; print}' 
tail -3 $OUT/err.txt
