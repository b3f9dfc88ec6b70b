#!/bin/bash
# row-sorted job: cache policy bits of the output-row stores (probe builds tools/bin/v_*), ragged case
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for v in shipped v_sc1 v_sc01 v_ntsc1 v_ntsc01 v_sc0; do
  if [ $v = shipped ]; then L=hybridbackend_amd/lib; else L=tools/bin/$v; fi
  LD_LIBRARY_PATH=$PWD/$L timeout 60 tools/bin/bench_ops R 2>&1 | grep -E "group_lookup_bwd|rror|fault" | head -3 | sed "s|^|$v  |"
done; done
