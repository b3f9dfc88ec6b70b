#!/bin/bash
# rebuild lookup_bwd.hip with different reduce-kernel shapes on the GPU box and time them
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R/hybridbackend_amd/csrc
for cfg in ${VARIANTS:-"512,4,4,8" "512,4,5,8" "512,2,5,8" "512,8,5,8" "512,4,6,8" "256,4,8,8"}; do
  IFS=, read CP UA WAVES UH <<< "$cfg"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I/opt/rocm/include -fno-fast-math -ffp-contract=off -DHBK_BWD_CP=$CP -DHBK_BWD_UA=$UA -DHBK_BWD_WAVES=$WAVES -DHBK_BWD_UH=$UH -c lookup_bwd.hip -o ../lib/obj/lookup_bwd.hip.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libhbk_core.so ../lib/obj/*.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
  echo "=== CP=$CP UA=$UA WAVES=$WAVES UH=$UH"
  (cd $R && tools/bin/bench_ops 2>&1 | grep group_lookup_bwd)
done
