#!/bin/bash
# Full validation on the GPU box (what profiles/r02/ was made with): every GPU test, smoke(), the PMC passes of the
# shipped kernels (tools/gpu_pmc.sh), bench full + slim, rocprofv3 kernel stats of the bench.
#   gpurun --timeout 1500 -- 'bash tools/gpu_validate.sh TAG GITHEAD'
set -u
OUT=gpurun_out/${1:-validate}
HEAD=${2:-unknown}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests -m gpu -q --maxfail=10 > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/status.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke rc=$?" >> $OUT/status.txt
bash tools/gpu_pmc.sh $(basename $OUT) $HEAD > $OUT/pmc.log 2>&1
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --arch slim --no-cpu > $OUT/bench_slim.json 2>> $OUT/bench.err
for b in 1250 10000; do python bench.py --mode train --batch $b --steps 50 --warmup 5 >> $OUT/bench_train.jsonl 2>> $OUT/bench.err; done
python bench.py --mode train --batch 10000 --steps 50 --warmup 5 --arch slim >> $OUT/bench_train.jsonl 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o b -- python bench.py --no-cpu > $OUT/bench_under_rocprof.json 2> $OUT/prof_bench.err
f=$(find $OUT/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/bench_kernel_stats.csv; rm -rf $OUT/prof_bench
tail -3 $OUT/pytest_gpu.log; tail -3 $OUT/smoke.log; grep -v "^at::\|rocprim\|rocclr\|elementwise" $OUT/pmc.log | tail -16
