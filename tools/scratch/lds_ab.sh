#!/bin/bash
# in-box A/B of the grouping kernels' residency (HBK_BWD_LDS_PAD KB of unused LDS = the old footprint)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for pad in 0 14 0 14; do
  for w in R r; do HBK_BWD_LDS_PAD=$pad timeout 300 tools/bin/bench_ops $w 2>&1 | grep group_lookup_bwd | sed "s/^/lds_pad=$pad  /"; done
done
source tools/gpu_r5.sh "none" > /dev/null 2>&1
ab "bwd_lds_pad:0,14,0,14" h
