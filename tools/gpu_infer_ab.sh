#!/bin/bash
# parity tests + inference bench A/B over kernel variants: bash tools/gpu_infer_ab.sh TAG "v1 v2 ..."
set -u
OUT=gpurun_out/${1:-r03o}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $OUT/pytest_parity.log 2>&1
echo "pytest rc=$?" > $OUT/status.txt
tail -5 $OUT/pytest_parity.log
for v in ${2:-"495"}; do
  for i in 1 2; do python bench.py --no-cpu --no-extras --variant $v >> $OUT/bench_$v.jsonl 2>> $OUT/bench.err; done
  python - $OUT/bench_$v.jsonl $v <<'PY'
import json, sys
for l in open(sys.argv[1]):
    r = json.loads(l)
    print("variant", sys.argv[2], "%.3f M/s  whole %.4f " % (r["value"] / 1e6, r["roofline"]["whole_path_frac"]), " ".join("%s %.4f" % (k["kernel_name"].split("(")[0], k["avg_ms"]) for k in r["kernels"]))
PY
done
cat $OUT/status.txt
