#!/bin/bash
set -u
OUT=gpurun_out/${1:-r02_full}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests -m gpu -q --maxfail=10 > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/status.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke rc=$?" >> $OUT/status.txt
bash tools/gpu_pmc.sh $(basename $OUT) cd734c3 > $OUT/pmc.log 2>&1
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --arch slim --no-cpu > $OUT/bench_slim.json 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o b -- python bench.py --no-cpu > $OUT/bench_under_rocprof.json 2> $OUT/prof_bench.err
f=$(find $OUT/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/bench_kernel_stats.csv; rm -rf $OUT/prof_bench
tail -3 $OUT/pytest_gpu.log; tail -3 $OUT/smoke.log; tail -12 $OUT/pmc.log
