#!/bin/bash
# Product-library variants for bisecting the nondeterministic essential-matrix filter (kernels_filter.hip): each keeps ONE of
# the debug hooks that the product build compiles out as a runtime-null test (dbg = the FCHECKs, t1..t4 = the four trace
# sites), or adds a full vmcnt wait before the barrier behind the solves (vmwait), or poisons the solver's LDS workspace on
# top of the passing developer configuration (nanfill).  -> regard3d_amd/libr3dm_bisect_<v>.so, linked from the current
# product objects (run build.sh first).
set -e
cd "$(dirname "$0")/.."
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fopenmp -Wall -Wno-unused-result -Iinclude"
mkdir -p build/bisect
VARS="ldsflag"
# ldsflag = the sample handed over in LDS + SGPR spills to memory (the combination that passed in profiles/r02_f_efilter_variants.txt
# as "nosgprvgpr"); the product is the by-reference call + SGPR spills to memory.  Earlier variants: git history.
for v in $VARS; do
  $HIPCC $FLAGS -DR3DM_E_SAMPLE_VIA_LDS=1 -mllvm -amdgpu-spill-sgpr-to-vgpr=0 -x hip -c regard3d_amd/csrc/kernels_filter_e.hip -o build/bisect/kernels_filter_e_$v.o &
done
wait
for v in $VARS; do
  objs=$(ls build/product/*.o | grep -v kernels_filter_e.o)
  $HIPCC --offload-arch=gfx950 -fPIC -fopenmp -shared $objs build/bisect/kernels_filter_e_$v.o -o regard3d_amd/libr3dm_bisect_$v.so
done
echo "built regard3d_amd/libr3dm_bisect_{$(echo $VARS | tr ' ' ',')}.so"
