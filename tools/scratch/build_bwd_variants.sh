#!/bin/bash
# Builds variants of the library that differ only in compile-time knobs of lookup_bwd.hip into
# tools/bin/var/<name>/libhbk_core.so (here, on the CPU box); run them on the GPU box with
#   for v in tools/bin/var/*; do echo == $v; LD_LIBRARY_PATH=$v tools/bin/bench_ops bwd; done
R=$(cd $(dirname $0)/../.. && pwd); cd $R/hybridbackend_amd/csrc
rm -rf $R/tools/bin/var
while read name flags; do
  [ -z "$name" ] && continue
  mkdir -p $R/tools/bin/var/$name
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I/opt/rocm/include -fno-fast-math -ffp-contract=off $flags -c lookup_bwd.hip -o /tmp/var_$name.o || exit 1
  objs=$(ls ../lib/obj/*.o | grep -v lookup_bwd)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/bin/var/$name/libhbk_core.so $objs /tmp/var_$name.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib &
done
wait
ls $R/tools/bin/var
