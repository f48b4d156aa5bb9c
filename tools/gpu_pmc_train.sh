#!/bin/bash
# HBM traffic of the training step (FETCH_SIZE / WRITE_SIZE in separate passes) -> gpurun_out/TAG/pmc_traffic.json ("train")
set -u
TAG=${1:-pmctrain}; HEAD=${2:-unknown}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
for cfg in "full 10000" "full 1250" "slim 10000"; do
  set -- $cfg; arch=$1; b=$2
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/p_${arch}_${b}_$c -o p -- python bench.py --mode train --arch $arch --batch $b --steps 8 --warmup 2 > /dev/null 2> $OUT/p_${arch}_${b}_$c.err
    f=$(find $OUT/p_${arch}_${b}_$c -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && cp "$f" $OUT/train_${arch}_${b}_${c}_counter_collection.csv
    rm -rf $OUT/p_${arch}_${b}_$c
  done
  (cd tools && python pmc_train_traffic.py --arch $arch --batch $b --fetch ../$OUT/train_${arch}_${b}_FETCH_SIZE_counter_collection.csv \
      --write ../$OUT/train_${arch}_${b}_WRITE_SIZE_counter_collection.csv --head $HEAD --out ../$OUT/pmc_traffic.json)
done
