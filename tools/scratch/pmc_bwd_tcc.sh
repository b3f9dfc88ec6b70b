#!/bin/bash
# L2 <-> memory request counters of the backward kernels (one --pmc pass per group of four, kernel-trace only)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
source <(sed -n '/^prof()/,/^}/p' tools/gpu_r2.sh)
prof pmc_bwd_rd "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" -- $R/tools/bin/bench_ops ${OPS_MODE:-b}
prof pmc_bwd_wr "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" -- $R/tools/bin/bench_ops ${OPS_MODE:-b}
python - <<PY
import json
for f in ("pmc_bwd_rd", "pmc_bwd_wr"):
  d = json.load(open("$O/%s.json" % f))
  for k, v in sorted(d.items()):
    if 'bwd_' in k:
      print(k[:70].ljust(70), {c: round(x['mean']) for c, x in v.items()})
PY
