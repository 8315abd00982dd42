#!/bin/bash
# Builds regard3d_amd/libr3dm.so (HIP kernels + C ABI) for gfx950.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
SRC="regard3d_amd/csrc/kernels_match.hip regard3d_amd/csrc/kernels_filter.hip regard3d_amd/csrc/kernels_liop.hip regard3d_amd/csrc/kernels_ann.hip regard3d_amd/csrc/kernels_akaze.hip regard3d_amd/csrc/api_core.cpp regard3d_amd/csrc/api_match.cpp regard3d_amd/csrc/api_filter.cpp regard3d_amd/csrc/api_features.cpp regard3d_amd/csrc/compute_matches.cpp"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fopenmp -Wall -Wno-unused-result -Iinclude"
$HIPCC $FLAGS -x hip -shared $SRC -o regard3d_amd/libr3dm.so
echo "built regard3d_amd/libr3dm.so"
