#!/bin/bash
# bash tools/gpu_wave_stamps.sh TAG   (library built beforehand with CV_EXTRA_FLAGS=-DCV_WG_STAMP)
set -u
OUT=gpurun_out/${1:-stamps}; mkdir -p $OUT
python tools/gpu_wave_stamps.py 10000 0 $OUT/stamps_10000.npz > $OUT/wave_stamps_10000.txt 2>> $OUT/err.txt
python tools/gpu_wave_stamps.py 1250 0 > $OUT/wave_stamps_1250.txt 2>> $OUT/err.txt
cat $OUT/wave_stamps_10000.txt $OUT/wave_stamps_1250.txt; grep -v amdgpu.ids $OUT/err.txt | tail -5
