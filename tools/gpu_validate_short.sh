#!/bin/bash
# Short validation after a change that does not touch the PMC figures: every GPU test, smoke(), the bench line, the slim
# line, the training lines.  bash tools/gpu_validate_short.sh TAG
set -u
OUT=gpurun_out/${1:-vshort}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests -m gpu -q --maxfail=10 > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/status.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke rc=$?" >> $OUT/status.txt
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --arch slim --no-cpu > $OUT/bench_slim.json 2>> $OUT/bench.err
for b in 1250 10000; do python bench.py --mode train --batch $b --steps 50 --warmup 5 >> $OUT/bench_train.jsonl 2>> $OUT/bench.err; done
python bench.py --mode train --batch 10000 --steps 50 --warmup 5 --arch slim >> $OUT/bench_train.jsonl 2>> $OUT/bench.err
python bench.py --mode train --batch 1250 --steps 50 --warmup 5 --arch slim >> $OUT/bench_train.jsonl 2>> $OUT/bench.err
bash tools/gpu_train_profile.sh $(basename $OUT)_slim 10000 - "--arch slim" > $OUT/slim_profile.txt 2>&1
cp gpurun_out/$(basename $OUT)_slim/train_10000_-_kernel_stats.csv $OUT/train_serial_slim_10000_kernel_stats.csv 2>/dev/null
tail -3 $OUT/pytest_gpu.log; tail -2 $OUT/smoke.log; cat $OUT/status.txt
python - $OUT <<'PY'
import json, sys, os
o = sys.argv[1]
b = json.load(open(os.path.join(o, "bench.json")))
print("bench: %.2f M cand/s, dominant %.3f, whole path %.3f; slim %.2f M/s; train %s" % (
    b["value"] / 1e6, b["roofline"]["frac"], b["roofline"]["whole_path_frac"], b["slim"]["value"] / 1e6,
    {k: "%.3f ms (%.3f)" % (v["ms_per_step"], v["roofline"]["frac"]) for k, v in b["train"].items()}))
for l in open(os.path.join(o, "bench_train.jsonl")):
    r = json.loads(l)
    print("train %-4s %6d: %.3f ms, frac %.3f" % (r["config"]["arch"], r["config"]["global_batch"], r["ms_per_step"], r["roofline"]["frac"]))
PY
