#!/bin/bash
# GPU tests (training subset or all) + A/B over bench flags: bash tools/gpu_train_ab.sh TAG "flagset1|flagset2|..." [all]
set -u
OUT=gpurun_out/${1:-r03f}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
if [ "${3:-}" = "all" ]; then SEL=""; else SEL="--deselect tests/test_gpu_pileup.py --deselect tests/test_bam_native.py"; fi
timeout 1200 python -m pytest tests -m gpu -q -x $SEL > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" > $OUT/status.txt
tail -4 $OUT/pytest_gpu.log
IFS='|' read -ra SETS <<< "${2:-}"
for fl in "${SETS[@]}"; do
  for b in 10000 1250; do
    python bench.py --mode train --batch $b --steps 40 --warmup 4 $fl >> $OUT/train_ab.jsonl 2>> $OUT/bench.err
  done
  python bench.py --mode train --arch slim --batch 10000 --steps 40 --warmup 4 $fl >> $OUT/train_ab.jsonl 2>> $OUT/bench.err
done
python - $OUT/train_ab.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    r = json.loads(l)
    print("%-5s %6d %-28s %-12s %.3f ms" % (r["config"]["arch"], r["config"]["global_batch"], r["config"]["dbg"], r["config"]["weight_gradients"], r["ms_per_step"]))
PY
cat $OUT/status.txt
