// kernels_filter_e.hip -- the essential-matrix instantiations of the AC-RANSAC kernel (acransac_kernel<2, *>) as their own
// translation unit, so that they compile in parallel with the F / H instantiations of kernels_filter.hip (the 5-point solver is
// the slowest thing to compile in the library).  No special compiler options (build.sh); see launch_filter_E in kernels_filter.hip.
#define R3DM_FILTER_ONLY_E 1
#include "kernels_filter.hip"
