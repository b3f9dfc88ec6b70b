#!/bin/bash
# rebuild lookup_bwd.hip with different reduce-kernel shapes on the GPU box and time them
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R/hybridbackend_amd/csrc
for cfg in "512 4 4 4" "512 4 4 8" "512 4 4 16" "512 4 3 16"; do
  set -- $cfg
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I/opt/rocm/include -fno-fast-math -ffp-contract=off -DHBK_BWD_CP=$1 -DHBK_BWD_UA=$2 -DHBK_BWD_WAVES=$3 -DHBK_BWD_UH=$4 -c lookup_bwd.hip -o ../lib/obj/lookup_bwd.hip.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libhbk_core.so ../lib/obj/*.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
  echo "=== CP=$1 UA=$2 WAVES=$3 UH=$4"
  (cd $R && python tools/sweep.py --cases c,f 2>&1 | grep case | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('   %-55s %9.1f us' % (d['case'], d['us']))")
done
