#!/bin/bash
# TCC request counters of the large-column grouping kernels (ragged case through the C ABI), with
# the scatter tiles dealt to the XCDs contiguously (shipped) and round robin (HBK_BWD_XCD=3)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for x in 1 3; do
  rm -rf $R/gpurun_out/pmc_sc_$x
  HBK_BWD_XCD=$x rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $R/gpurun_out/pmc_sc_$x -o p -- $R/tools/bin/bench_ops r > /dev/null 2>&1
  echo "== bwd_xcd=$x"
  python $R/tools/prof_summary.py pmc $R/gpurun_out/pmc_sc_$x | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in sorted(d.items()):
  if 'scatter' in k or 'hist' in k or 'segof' in k: print(k[:40].ljust(40), {c:round(x['mean']) for c,x in v.items()})"
done
