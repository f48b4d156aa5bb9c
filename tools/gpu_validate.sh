#!/bin/bash
# Full validation on the GPU box (what profiles/rNN/ is made with): every GPU test, smoke(), the PMC passes of the
# shipped kernels (inference: tools/gpu_pmc.sh; training step: tools/gpu_pmc_train.sh), the bench line, slim / training
# lines, rocprofv3 kernel stats of the bench and of the training step, the host-side probes.
#   gpurun --timeout 2400 -- 'bash tools/gpu_validate.sh TAG GITHEAD [host]'
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
cp $OUT/pmc_traffic.json profiles/pmc_traffic.json 2>/dev/null     # the bench below reads it (this copy stays on the box)
bash tools/gpu_pmc_train.sh $(basename $OUT)_t $HEAD > $OUT/pmc_train.log 2>&1
cp gpurun_out/$(basename $OUT)_t/pmc_traffic.json profiles/pmc_traffic.json 2>/dev/null
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
# the previous round's library (clairvoyante_amd/csrc/libclairvoyante_hip_base.so, built from that commit with its own flags)
# against the in-tree one on this box, alternating (only if the base library was built and shipped)
[ -f clairvoyante_amd/csrc/libclairvoyante_hip_base.so ] && bash tools/gpu_step_ab.sh $(basename $OUT)_ab "320 1250 5000 10000" 2 > $OUT/step_ab_base_vs_final.txt 2>&1
# cv_forward over the ladder of pass sizes (full, slim), per stage
python tools/gpu_infer_stage_ladder.py full > $OUT/infer_stage_ladder_full.txt 2>> $OUT/bench.err
python tools/gpu_infer_stage_ladder.py slim > $OUT/infer_stage_ladder_slim.txt 2>> $OUT/bench.err
SIZES="320 640 1250 2500 5000 8000 10000 12288 16384 32768" STEPS=30 bash tools/gpu_step_size_sweep.sh $(basename $OUT)_steps > /dev/null 2>&1
cp gpurun_out/$(basename $OUT)_steps/step_sizes.txt $OUT/step_size_ladder.txt 2>/dev/null
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --arch slim --no-cpu > $OUT/bench_slim.json 2>> $OUT/bench.err
for b in 1250 10000; do python bench.py --mode train --batch $b --steps 50 --warmup 5 >> $OUT/bench_train.jsonl 2>> $OUT/bench.err; done
python bench.py --mode train --batch 10000 --steps 50 --warmup 5 --arch slim >> $OUT/bench_train.jsonl 2>> $OUT/bench.err
python bench.py --mode train --batch 1250 --steps 50 --warmup 5 --arch slim >> $OUT/bench_train.jsonl 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o b -- python bench.py --no-cpu --no-extras > $OUT/bench_under_rocprof.json 2> $OUT/prof_bench.err
f=$(find $OUT/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/bench_kernel_stats.csv; rm -rf $OUT/prof_bench
for b in 10000 1250; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_t$b -o t -- python bench.py --mode train --batch $b --steps 20 --warmup 3 --overlap 0 > $OUT/train_serial_$b.json 2> $OUT/prof_t$b.err
  f=$(find $OUT/prof_t$b -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/train_serial_${b}_kernel_stats.csv; rm -rf $OUT/prof_t$b
  rocprofv3 --kernel-trace --output-format csv -d $OUT/tl_$b -o t -- python bench.py --mode train --batch $b --steps 12 --warmup 3 > /dev/null 2> $OUT/tl_$b.err
  f=$(find $OUT/tl_$b -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/trace_timeline.py "$f" > $OUT/train_${b}_timeline.txt; rm -rf $OUT/tl_$b
done
if [ "${3:-}" = "host" ]; then      # the loops around the kernels (unchanged code: not re-measured every round)
  python tools/vcf_format_probe.py 2000000 16 > $OUT/vcf_format_probe.txt 2>&1
  python tools/vcf_format_probe.py 2000000 1 >> $OUT/vcf_format_probe.txt 2>&1
  timeout 900 python tools/gpu_callvar_text_probe.py > $OUT/callvar_text_probe.txt 2>&1
  timeout 600 python tools/gpu_e2e_bam.py > $OUT/e2e_bam.txt 2>&1
  timeout 300 python tools/gpu_small_batch_probe.py > $OUT/small_batch.txt 2>&1
fi
timeout 300 python tools/gpu_step_host_probe.py 1250 10000 > $OUT/step_host_probe.txt 2>&1
# the RCCL branch with ONE rank (a sum over one rank moves no byte: what is measured is the machinery): the training lines
# with their exchange keys, the exchange alone (--mode exchange), and the step by exchange form
D="CV_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517"
for b in 10000 1250; do env $D python bench.py --mode train --batch $b --steps 40 --warmup 4 >> $OUT/bench_train_rccl_one_rank.jsonl 2>> $OUT/bench.err; done
env $D python bench.py --mode exchange > $OUT/bench_exchange_rccl_one_rank.json 2>> $OUT/bench.err
ex() {  # label, batch, env...
  local label=$1 b=$2; shift 2
  env "$@" python bench.py --mode train --batch $b --steps 40 --warmup 4 2>> $OUT/bench.err | LABEL="$label" python -c "
import json,sys,os
r=json.loads(sys.stdin.read()); print('batch %5d %-30s step %.3f ms  compute %s  exchange alone %s  plan %s' % (r['config']['global_batch'], os.environ['LABEL'], r['ms_per_step'], r.get('compute_ms_per_step'), r.get('exchange_ms'), r.get('exchange_plan')))" >> $OUT/exchange_fixed_cost.txt
}
for round in 1 2; do
  for b in 1250 10000; do
    ex "no process group" $b CV_NOTHING=1
    ex "one collective" $b $D CV_EXCHANGE=one
    ex "split, sync, both on the comm stream" $b $D CV_EXCHANGE=split
    ex "split, async (round 4 form)" $b $D CV_EXCHANGE=split CV_EXCHANGE_ASYNC=1
  done
done
sort -o $OUT/exchange_fixed_cost.txt $OUT/exchange_fixed_cost.txt
tail -3 $OUT/pytest_gpu.log; tail -3 $OUT/smoke.log; cat $OUT/status.txt
python - $OUT <<'PY'
import json, sys, os
o = sys.argv[1]
b = json.load(open(os.path.join(o, "bench.json")))
print("bench: %.2f M cand/s, dominant %.3f, whole path %.3f; slim %.2f M/s; train %s" % (
    b["value"] / 1e6, b["roofline"]["frac"], b["roofline"]["whole_path_frac"], b["slim"]["value"] / 1e6,
    {k: "%.3f ms (%.3f)" % (v["ms_per_step"], v["roofline"]["frac"]) for k, v in b["train"].items() if k != "parity"}))
print("parity:", b.get("parity"))
for l in open(os.path.join(o, "bench_train.jsonl")):
    r = json.loads(l)
    print("train %-4s %6d: %.3f ms, frac %.3f, traffic %s" % (r["config"]["arch"], r["config"]["global_batch"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"]["traffic"]))
PY
