#!/bin/bash
# WIDE walk width 6 (shipped) vs 5 (probe build): config 4 and the config-5 mix; then the ragged cases with the wider row-sorted ratio
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in shipped v_ww5 shipped v_ww5; do
  if [ $v = shipped ]; then L=$PWD/hybridbackend_amd/lib; else L=$PWD/tools/bin/$v; fi
  echo "== $v"; HBK_LIBRARY=$L/libhbk_core.so timeout 300 python tools/sweep.py --big --cases d,h 2>/dev/null | grep "^{" | grep bwd | python -c "
import sys,json
for l in sys.stdin:
  d=json.loads(l); print('  ',d['case'][:70].ljust(70), d['us'])"
done
timeout 100 tools/bin/bench_ops r 2>&1 | grep group_lookup_bwd
