#!/bin/bash
# round 3: every GPU test + the host-side probes of the VCF path
set -u
OUT=gpurun_out/${1:-r03d}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1200 python -m pytest tests -m gpu -q --maxfail=8 > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" > $OUT/status.txt
tail -8 $OUT/pytest_gpu.log
python tools/vcf_format_probe.py 2000000 16 > $OUT/vcf_format_probe.txt 2>&1
python tools/vcf_format_probe.py 2000000 1 >> $OUT/vcf_format_probe.txt 2>&1
python tools/vcf_format_probe.py 65536 16 >> $OUT/vcf_format_probe.txt 2>&1
cat $OUT/vcf_format_probe.txt
timeout 600 python tools/gpu_callvar_text_probe.py 200000 > $OUT/callvar_text_probe.txt 2>&1
grep -E "rows/s|wrote" $OUT/callvar_text_probe.txt
timeout 600 python tools/gpu_e2e_bam.py > $OUT/e2e_bam.txt 2>&1
grep -E "^callVarBam|prefetch" $OUT/e2e_bam.txt
cat $OUT/status.txt
