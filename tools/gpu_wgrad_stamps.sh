#!/bin/bash
# bash tools/gpu_wgrad_stamps.sh TAG   (library built beforehand with CV_EXTRA_FLAGS=-DCV_WG_STAMP)
set -u
OUT=gpurun_out/${1:-stamps}; mkdir -p $OUT
for sides in 0 3; do python tools/gpu_wgrad_stamps.py 10000 $sides $OUT/stamps_10000_sides$sides.npy >> $OUT/wgrad_conv3_wave_stamps.txt 2>> $OUT/err.txt; echo >> $OUT/wgrad_conv3_wave_stamps.txt; done
python tools/gpu_wgrad_stamps.py 1250 0 >> $OUT/wgrad_conv3_wave_stamps.txt 2>> $OUT/err.txt
cat $OUT/wgrad_conv3_wave_stamps.txt; tail -5 $OUT/err.txt
