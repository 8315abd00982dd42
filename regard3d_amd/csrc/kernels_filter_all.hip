// kernels_filter_all.hip -- the AC-RANSAC filters of one putative graph in one launch (acransac_all_kernel, r3dm_filter_FEH): the
// three model kinds instantiated into one kernel, compiled as its own translation unit beside kernels_filter.hip (F, H) and
// kernels_filter_e.hip (E).
#define R3DM_FILTER_ALL 1
#include "kernels_filter.hip"
