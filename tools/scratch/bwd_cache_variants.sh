#!/bin/bash
# Probe builds of the backward (cache policies of its loads / stores, tile sizes):
#   tools/scratch/bwd_cache_variants.sh      (build box: writes tools/bin/v_*/libhbk_core.so)
#   VARIANT_CASES="b s r" tools/gpu_r3.sh variants          (GPU box)
# Round 3: the non-temporal hint on the gradient loads cost 9 % of the config-2 backward (a
# 128-byte line holds two 64-byte rows; ragged columns re-read their rows) -> plain is shipped.
set -e
cd "$(dirname "$0")/../.."
CS=hybridbackend_amd/csrc
OBJ=hybridbackend_amd/lib/obj
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -I/opt/rocm/include -fno-fast-math -ffp-contract=off"
build() {  # build <name> <defines...>
  local name=$1; shift
  mkdir -p tools/bin/v_$name
  /opt/rocm/bin/hipcc $FLAGS "$@" -c $CS/lookup_bwd.hip -o tools/bin/v_$name/lookup_bwd.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/v_$name/libhbk_core.so \
    tools/bin/v_$name/lookup_bwd.o $(ls $OBJ/*.o | grep -v lookup_bwd) -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
  rm tools/bin/v_$name/lookup_bwd.o
}
rm -rf tools/bin/v_*
for spec in "$@"; do   # e.g.  "shipped" "grad_nt -DHBK_BWD_GRAD_NT=1" "tile4096 -DHBK_BWD_TILE=4096"
  set -- $spec
  build "$@" &
done
wait
ls -la tools/bin/v_*/libhbk_core.so
