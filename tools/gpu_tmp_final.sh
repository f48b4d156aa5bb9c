set -u
OUT=gpurun_out/r05z; HEAD=${1:-unknown}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests -m gpu -q --maxfail=10 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/status.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/status.txt
bash tools/gpu_pmc_train.sh r05z_t $HEAD > $OUT/pmc_train.log 2>&1
cp gpurun_out/r05z_t/pmc_traffic.json profiles/pmc_traffic.json 2>/dev/null; cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
cp gpurun_out/r05z_t/*counter_collection.csv $OUT/ 2>/dev/null
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --arch slim --no-cpu > $OUT/bench_slim.json 2>> $OUT/bench.err
for b in 1250 2500 5000 10000; do python bench.py --mode train --batch $b --steps 50 --warmup 5 >> $OUT/bench_train.jsonl 2>> $OUT/bench.err; done
for b in 1250 5000 10000; do python bench.py --mode train --batch $b --steps 50 --warmup 5 --arch slim >> $OUT/bench_train.jsonl 2>> $OUT/bench.err; done
bash tools/gpu_step_ab.sh r05z_ab "1250 2500 5000 10000" 2 > $OUT/step_ab.txt 2>&1
bash tools/gpu_train_timeline.sh r05z 1250 "" > /dev/null 2>&1
bash tools/gpu_train_timeline.sh r05z 5000 "" > /dev/null 2>&1
run() { local label=$1 b=$2; shift 2
  python bench.py --mode train --batch $b --steps 40 --warmup 4 "$@" 2>> $OUT/err.txt | LABEL="$label" python -c "
import json,sys,os
r=json.loads(sys.stdin.read()); print('%s batch %5d %-40s %.3f ms' % (r['config']['arch'], r['config']['global_batch'], os.environ['LABEL'], r['ms_per_step']))" >> $OUT/parts_ab.txt
}
for round in 1 2; do for b in 1250 1600 2000 2500; do run "in-tree (position parts)" $b; run "flat ranges" $b --dbg 0=7,1=7; done; done
sort $OUT/parts_ab.txt
cat $OUT/status.txt; tail -2 $OUT/pytest_gpu.log; tail -2 $OUT/smoke.log; grep -v slim $OUT/step_ab.txt | tail -20
python -c "
import json
r=json.load(open('$OUT/bench.json')); print(r['value'], r['roofline']['frac'], {k:(v.get('ms_per_step'), (v.get('roofline') or {}).get('traffic')) for k,v in r['train'].items() if isinstance(v,dict) and 'ms_per_step' in v})"
