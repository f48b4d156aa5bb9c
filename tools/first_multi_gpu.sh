#!/bin/bash
# The FIRST run on a node with more than one MI355X (nothing of this build has run over xGMI yet): every stage under its
# own timeout, RCCL's own log kept next to each line, so that whatever stops is visible and whatever runs is read in order --
#   1 dry        the ranks start, count themselves (one all-reduce), name their devices            bench.py --dry
#   2 exchange   only the all-reduce of the step's 6.5 MB bucket and of its pieces: time, algorithm / bus bandwidth per
#                piece, RCCL's algorithm / protocol choices                                       bench.py --mode exchange
#   3 train      optimizer steps on train.py's batch split over the ranks, under BOTH exchange plans (one collective behind
#                the step / the dense 95 % under the backward pass): step time, exchange alone, compute alone, hidden fraction,
#                exchange_busbw_GBps of the step's own exchange                                   bench.py --mode train
#   4 train at 10 000 per rank (the efficient operating point)                                    bench.py --mode train --batch
#   5 infer      the headline line with its slim / training legs                                  bench.py
# usage: bash tools/first_multi_gpu.sh [N=8] [OUT=gpurun_out/first_multi_gpu]      (one node; run where the GPUs are)
set -u
N=${1:-8}
OUT=${2:-gpurun_out/first_multi_gpu}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
export HSA_ENABLE_IPC_MODE_LEGACY=0 NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,COLL,TUNING
rocm-smi --showtopo > $OUT/topology.txt 2>&1 || true
stage() {   # name, timeout seconds, bench flags...
  local name=$1 limit=$2; shift 2
  echo "== $name: python bench.py --gpus $N $*" | tee -a $OUT/status.txt
  NCCL_DEBUG_FILE=$OUT/$name.rccl.%h.%p.log timeout $limit python bench.py --gpus $N "$@" > $OUT/$name.json 2> $OUT/$name.err
  local rc=$?
  echo "   rc=$rc" | tee -a $OUT/status.txt
  if [ $rc -ne 0 ]; then tail -5 $OUT/$name.err | tee -a $OUT/status.txt; fi
  return $rc
}
stage 1_dry 120 --dry || { echo "the ranks do not start: stop here" | tee -a $OUT/status.txt; exit 1; }
stage 2_exchange 300 --mode exchange
CV_EXCHANGE=one stage 3_train_one_collective 600 --mode train --steps 40 --warmup 4
CV_EXCHANGE=split stage 3_train_split 600 --mode train --steps 40 --warmup 4
stage 3_train_planned 600 --mode train --steps 40 --warmup 4
stage 4_train_10000_per_rank 600 --mode train --batch $((10000 * N)) --steps 40 --warmup 4
stage 5_infer 1200 --steps 20 --warmup 3 --no-cpu
python - $OUT <<'PY'
import glob, json, os, sys
o = sys.argv[1]
for f in sorted(glob.glob(os.path.join(o, "*.json"))):
    try:
        b = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print("%-32s no line (%s)" % (os.path.basename(f), e)); continue
    keys = ("n_gpus", "rccl_ranks", "counted_ranks", "distinct_devices", "backend", "value", "unit", "ms_per_step", "exchange_plan", "exchange_ms",
            "compute_ms_per_step", "exchange_hidden_frac", "exchange_busbw_GBps")
    print("%-32s %s" % (os.path.basename(f), {k: (round(b[k], 4) if isinstance(b[k], float) else b[k]) for k in keys if k in b}))
    if "other_plan" in b:
        print("%-32s other plan: %s" % ("", b["other_plan"]))
    if "pieces" in b:
        for p in b["pieces"]:
            print("%-32s piece %s" % ("", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in p.items()}))
    if "per_rank_ms" in b:
        print("%-32s per rank %s" % ("", b["per_rank_ms"]))
PY
