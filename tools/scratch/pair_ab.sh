#!/bin/bash
# whole-line stores of the row-sorted reduce (HBK_RS_PAIR_STORES): shipped vs the probe build without them
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for v in shipped v_nopair; do
  if [ $v = shipped ]; then L=$PWD/hybridbackend_amd/lib; else L=$PWD/tools/bin/$v; fi
  for w in R; do LD_LIBRARY_PATH=$L timeout 100 tools/bin/bench_ops $w 2>&1 | grep -E "group_lookup_bwd|rror" | sed "s|^|$v  |"; done
done; done
source tools/gpu_r5.sh none > /dev/null 2>&1
export HBK_BENCH_ITERS=2
for v in shipped v_nopair; do
  if [ $v = shipped ]; then L=$R/hybridbackend_amd/lib; else L=$R/tools/bin/$v; fi
  LD_LIBRARY_PATH=$L prof pmc_$v "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" -- $R/tools/bin/bench_ops R
  echo "== $v"; tail -1 $O/pmc_$v.log; pmc_table $O/pmc_$v.json bwd_rowsort_kernel; trim pmc_$v
done
unset HBK_BENCH_ITERS
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "backward" 2>&1 | tail -2
