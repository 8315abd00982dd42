// kernels_filter_coop_e.hip -- the essential-matrix instantiation of the cooperative AC-RANSAC kernel (acransac_coop_kernel<2>) as
// its own translation unit, to compile in parallel with the F / H instantiations of kernels_filter_coop.hip.
#define R3DM_FILTER_COOP_ONLY_E 1
#include "kernels_filter_coop.hip"
