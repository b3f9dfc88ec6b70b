#!/bin/bash
# VGPRs / spills / LDS / occupancy of every kernel of one source file (hipcc's resource remarks):
#   tools/kernel_resources.sh hybridbackend_amd/csrc/lookup_bwd.hip [extra hipcc flags]
src=$1; shift
d=$(dirname $src)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$d -Iinclude -I/opt/rocm/include \
  -fno-fast-math -ffp-contract=off "$@" -c $src -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
python3 -c '
import re, sys, subprocess
rows, cur = [], None
for line in sys.stdin:
  m = re.search(r"remark: [^ ]* *(Function Name|VGPRs|AGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
  if not m: continue
  k, v = m.groups()
  if k == "Function Name":
    cur = {"name": v}; rows.append(cur)
  elif cur is not None:
    cur[k] = v
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.split("\n")
print("%-70s %5s %6s %6s %7s %4s %6s" % ("kernel", "VGPR", "vspill", "sspill", "scratch", "occ", "LDS"))
for r, n in zip(rows, names):
  n = re.sub(r"hbk::\(anonymous namespace\)::", "", n)
  n = re.sub(r"\(.*", "", n)
  print("%-70s %5s %6s %6s %7s %4s %6s" % (n[:70], r.get("VGPRs"), r.get("VGPRs Spill"), r.get("SGPRs Spill"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
'
