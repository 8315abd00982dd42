#!/bin/bash
# Builds the HIP kernels + C ABI for gfx950 (hipcc cross-compiles without a GPU):
#   regard3d_amd/libr3dm.so      the product library -- no developer knobs, never reads the environment
#   regard3d_amd/libr3dm_dev.so  the same sources with -DR3DM_DEVTOOLS + tools/devtools/dev_knobs.cpp: A/B kernel variants,
#                                traces, invariant checks and test hooks for tools/ and the fallback-path tests
# Usage: build.sh [product|dev|all]   (default: all).  Objects are cached under build/ and rebuilt when a source or header changes.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
WHAT=${1:-all}
SRC="kernels_match.hip kernels_match_16bit.hip kernels_match_hamming.hip kernels_match_exact.hip kernels_filter.hip kernels_filter_e.hip kernels_filter_coop.hip kernels_liop.hip kernels_ann.hip kernels_hnsw.hip kernels_mrpt.hip kernels_akaze.hip kernels_graph.hip api_core.cpp api_match.cpp api_hnsw.cpp api_mrpt.cpp api_filter.cpp api_features.cpp api_multi.cpp api_comm.cpp compute_matches.cpp"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fopenmp -Wall -Wno-unused-result -Iinclude"
HDRS="regard3d_amd/csrc/*.hpp regard3d_amd/csrc/*.inc include/*.h include/*.hpp"
# kernels_filter.hip (wg_fence) orders global-memory exchanges INSIDE a workgroup with a workgroup-scope fence: valid only while the
# waves of a workgroup share a CU and its L1, i.e. never in threadgroup-split mode
case " $HIPCC $FLAGS $HIPCC_FLAGS $CXXFLAGS " in *tgsplit*) echo "build.sh: -mtgsplit is not supported (kernels_filter.hip: wg_fence)"; exit 1;; esac

build_one() {   # $1 = variant dir, $2 = extra flags, $3 = output, $4 = extra sources
  local dir=build/$1; mkdir -p $dir
  local objs="" pids=""
  for f in $SRC; do
    local o=$dir/${f%.*}.o
    objs="$objs $o"
    local stale=0
    if [ ! -f $o ]; then stale=1; else
      for d in regard3d_amd/csrc/$f $HDRS build.sh; do [ $d -nt $o ] && stale=1; done
    fi
    local extra=""
    # (no per-file compiler options: the essential-matrix kernel no longer needs -amdgpu-spill-sgpr-to-vgpr=0)
    case $f in kernels_filter_e.hip|kernels_filter_coop.hip)
      [ regard3d_amd/csrc/kernels_filter.hip -nt $o ] && stale=1; [ regard3d_amd/csrc/kernels_filter_coop.hip -nt $o ] && stale=1;; esac
    if [ $stale = 1 ]; then ( $HIPCC $FLAGS $2 $extra -x hip -c regard3d_amd/csrc/$f -o $o ) & pids="$pids $!"; fi
  done
  for f in $4; do
    local o=$dir/$(basename ${f%.*}).o
    objs="$objs $o"
    ( $HIPCC $FLAGS $2 -x hip -c $f -o $o ) & pids="$pids $!"
  done
  for p in $pids; do wait $p; done
  $HIPCC --offload-arch=gfx950 -fPIC -fopenmp -shared $objs -ldl -o $3
  echo "built $3"
}

if [ "$WHAT" = product ] || [ "$WHAT" = all ]; then build_one product "" regard3d_amd/libr3dm.so ""; fi
if [ "$WHAT" = dev ] || [ "$WHAT" = all ]; then build_one dev "-DR3DM_DEVTOOLS" regard3d_amd/libr3dm_dev.so "tools/devtools/dev_knobs.cpp"; fi
