// kernels_filter_e.hip -- the essential-matrix instantiation of the AC-RANSAC kernel (acransac_kernel<2>), compiled as its own
// translation unit with -mllvm -amdgpu-spill-sgpr-to-vgpr=0 (build.sh): see launch_filter_E in kernels_filter.hip.
#define R3DM_FILTER_ONLY_E 1
#include "kernels_filter.hip"
