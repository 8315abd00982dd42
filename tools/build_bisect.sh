#!/bin/bash
# builds product-library variants that keep one of the two debug hooks of kernels_filter.hip as a RUNTIME-null pointer
# (bisecting which compile-time removal exposes the nondeterministic essential-matrix filter): regard3d_amd/libr3dm_bisect_{dbg,trace,vmwait}.so
set -e
cd "$(dirname "$0")/.."
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fopenmp -Wall -Wno-unused-result -Iinclude"
mkdir -p build/bisect
for v in dbg trace vmwait; do
  D=-DR3DM_BISECT_$(echo $v | tr a-z A-Z)
  $HIPCC $FLAGS $D -x hip -c regard3d_amd/csrc/kernels_filter.hip -o build/bisect/kernels_filter_$v.o &
done
wait
for v in dbg trace vmwait; do
  objs=$(ls build/product/*.o | grep -v kernels_filter.o)
  $HIPCC --offload-arch=gfx950 -fPIC -fopenmp -shared $objs build/bisect/kernels_filter_$v.o -o regard3d_amd/libr3dm_bisect_$v.so
  echo built regard3d_amd/libr3dm_bisect_$v.so
done
