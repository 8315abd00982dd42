#!/bin/bash
# Throwaway variants of the AC-RANSAC filter translation units, linked against the current product objects (run build.sh first)
#   -> regard3d_amd/libr3dm_bisect_<v>.so, loaded with api.use_library():
#   timing  -DR3DM_E_TIMING: the report's (threshold, NFA) fields carry wave-0 cycles of the solve / evaluation phases
#           (all three models; tools/filter_phase_split.py)
# Earlier variants (the E filter's nondeterminism bisect, profiles/r02_d..f): git history.
set -e
cd "$(dirname "$0")/.."
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fopenmp -Wall -Wno-unused-result -Iinclude"
mkdir -p build/bisect
$HIPCC $FLAGS -DR3DM_E_TIMING -x hip -c regard3d_amd/csrc/kernels_filter_e.hip -o build/bisect/kernels_filter_e_timing.o &
$HIPCC $FLAGS -DR3DM_E_TIMING -x hip -c regard3d_amd/csrc/kernels_filter.hip -o build/bisect/kernels_filter_timing.o &
wait
objs=$(ls build/product/*.o | grep -v "kernels_filter_e.o\|kernels_filter.o")
$HIPCC --offload-arch=gfx950 -fPIC -fopenmp -shared $objs build/bisect/kernels_filter_e_timing.o build/bisect/kernels_filter_timing.o -o regard3d_amd/libr3dm_bisect_timing.so
echo "built regard3d_amd/libr3dm_bisect_timing.so"
