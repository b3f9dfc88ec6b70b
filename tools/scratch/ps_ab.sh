#!/bin/bash
# whole-line stores also in the SGD / Adagrad instantiations of the row-sorted reduce (probe build -DHBK_RS_PAIR_STEPS=0: without them)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for v in shipped v_ps0; do
  if [ $v = shipped ]; then L=$PWD/hybridbackend_amd/lib; else L=$PWD/tools/bin/$v; fi
  LD_LIBRARY_PATH=$L timeout 100 tools/bin/bench_ops s 2>&1 | grep -E "group_lookup_bwd" | sed "s|^|$v  |"
  LD_LIBRARY_PATH=$L timeout 100 tools/bin/bench_ops r 2>&1 | grep -E "group_lookup" | sed "s|^|$v  |"; LD_LIBRARY_PATH=$L timeout 100 tools/bin/bench_ops R 2>&1 | grep -E "group_lookup" | sed "s|^|$v  |"
done; done
for v in shipped v_ps0 shipped v_ps0; do
  if [ $v = shipped ]; then L=$PWD/hybridbackend_amd/lib; else L=$PWD/tools/bin/$v; fi
  echo "== $v"; HBK_LIBRARY=$L/libhbk_core.so timeout 300 python tools/sweep.py --cases h 2>/dev/null | grep "^{" | grep bwd | python -c "
import sys,json
for l in sys.stdin:
  d=json.loads(l); print('  ',d['case'][:70].ljust(70), d['us'])"
done
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "backward" 2>&1 | tail -2
