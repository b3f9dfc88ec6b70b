#!/bin/bash
# Probe builds of the library with single features of the dense backward turned off:
#   tools/scratch/bwd_dense_variants.sh            (on the build box: writes tools/bin/v_*/libhbk_core.so)
#   for v in tools/bin/v_*; do LD_LIBRARY_PATH=$v tools/bin/bench_ops b; done     (on the GPU box)
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
build all_on &
build no_fold -DHBK_DENSE_FOLD=0 &
build late_claim -DHBK_DENSE_EARLY_CLAIM=0 &
build no_sort -DHBK_DENSE_SORT=0 &
wait
build none -DHBK_DENSE_FOLD=0 -DHBK_DENSE_EARLY_CLAIM=0 -DHBK_DENSE_SORT=0 &
build w4 '-DHBK_BWD_DENSE_WAVES(S)=4' &
build stamps_fold0 -DHBK_BWD_STAMPS -DHBK_DENSE_FOLD=0 &
wait
ls -la tools/bin/v_*/libhbk_core.so
