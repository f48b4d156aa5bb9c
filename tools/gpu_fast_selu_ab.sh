#!/bin/bash
# A/B of a cheaper activation (build flag CV_FAST_SELU: negative SELU branch = v_exp_f32 + one fma) against the exact
# fixed-sequence SELU: kernel times, SQ_INSTS_VALU / MFMA busy, and what it does to the outputs (vs the CPU checker, vs
# float64, argmax on the timed set and on the stress set).  bash tools/gpu_fast_selu_ab.sh TAG
set -u
OUT=gpurun_out/${1:-r04selu}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for mode in exact fast exact2 fast2; do
  # the fast build is made beforehand: python tools/build_variant_lib.py fast -DCV_FAST_SELU (it travels with the snapshot)
  if [ ${mode:0:4} = fast ]; then export CV_HIP_LIB=$PWD/clairvoyante_amd/csrc/libclairvoyante_hip_fast.so; else unset CV_HIP_LIB; fi
  for i in 1 2 3; do python bench.py --no-cpu --no-extras >> $OUT/bench_$mode.jsonl 2>> $OUT/bench.err; done
  python bench.py --no-cpu --no-extras --arch slim >> $OUT/bench_slim_$mode.jsonl 2>> $OUT/bench.err
  python - $OUT/bench_$mode.jsonl $OUT/bench_slim_$mode.jsonl $mode <<'PY'
import json, sys
for fn in sys.argv[1:3]:
    for l in open(fn):
        r = json.loads(l)
        print(sys.argv[3], r["config"]["arch"], "%.3f M/s" % (r["value"] / 1e6), " ".join("%s %.4f" % (k["kernel_name"].split("<")[0], k["avg_ms"]) for k in r["kernels"]))
PY
  [ ${#mode} -gt 5 ] && continue          # second round: timings only
  python bench.py --no-extras > $OUT/bench_parity_$mode.json 2>> $OUT/bench.err
  python - $OUT/bench_parity_$mode.json $mode <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))
print(sys.argv[2], "parity on the timed set:", r["parity"])
PY
  python - $mode <<'PY'
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, common
from oracle import cv_oracle as O
from clairvoyante_amd import clairvoyante_v3, clairvoyante_v3_slim
for arch, cls in (("full", clairvoyante_v3), ("slim", clairvoyante_v3_slim)):
    P = common.bench_params(O, arch)
    x = common.inputs(8192, seed=91, stress=8192)
    m = cls.Clairvoyante(); m.setParameters(P)
    got = np.concatenate(m.predict(x), axis=1); m.close()
    want = O.predict(arch, P, x)
    d = np.abs(got - want)
    print(sys.argv[1], arch, "8192 synthetic + 8192 stress: max|dp| vs exact %.3g, argmax match %s, bitwise %.4f, rows with any argmax flip %d" % (
        d.max(), common.argmax_match(got, want), common.bitwise_frac(got, want),
        int(sum(np.any([np.argmax(got[:, lo:hi], 1) != np.argmax(want[:, lo:hi], 1) for lo, hi in common.HEADS], axis=0)))))
PY
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p_$mode -o p -- python bench.py --steps 8 --warmup 2 --no-cpu --no-extras > /dev/null 2> $OUT/p_$mode.err
  f=$(find $OUT/p_$mode -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" > $OUT/sq_$mode.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and m.get("GRBM_GUI_ACTIVE"):
        print("%-40s launches %3d  SQ_INSTS_VALU %.4g  MFMA busy %.3f" % (k[:40], len(v["SQ_INSTS_VALU"]), m.get("SQ_INSTS_VALU", 0),
              m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] * 128.0)))
PY
  rm -rf $OUT/p_$mode
  cat $OUT/sq_$mode.txt
done
unset CV_HIP_LIB
