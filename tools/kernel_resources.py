"""Registers / spills / occupancy of the kernels whose mangled name contains one of the given substrings:
python tools/kernel_resources.py clairvoyante_amd/csrc/cv_kernels_mfma.hip conv3_rot wgrad_conv_cm   (development tool)"""
import re, subprocess, sys
src, pats = sys.argv[1], sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-value",
       "-Wno-pass-failed", "--cuda-device-only", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None; rows = {}
for l in err.splitlines():
    if " error" in l: print(l)
    m = re.search(r"remark: (?:Function Name: (\S+)|\s+([A-Za-z ]+(?:\[[^\]]*\])?[A-Za-z ]*): (\S+))", l)
    if not m: continue
    if m.group(1): cur = m.group(1); rows[cur] = {}
    elif cur: rows[cur][m.group(2).strip()] = m.group(3)
for k, v in rows.items():
    if not pats or any(p in k for p in pats):
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0]
        print("%-70s VGPR %3s AGPR %3s spillV %s spillS %s occ %s LDS %s" % ((name if len(name) > 6 else k)[-70:], v.get("VGPRs"), v.get("AGPRs"), v.get("VGPRs Spill"),
              v.get("SGPRs Spill"), v.get("Occupancy [waves/SIMD]"), v.get("LDS Size [bytes/block]")))
