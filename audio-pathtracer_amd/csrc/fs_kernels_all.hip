// fs_kernels_all.hip — every kernel translation unit as ONE unit: for the diagnostic builds (-DFS_WAVE_TIMELINE,
// -DFS_TRAV_STATS, tools/build_variant.sh), whose device-side debug symbols are shared by all kernels.  The product
// builds the four units separately (Makefile).
#include "fs_walk.hip"
#include "fs_connect.hip"
#include "fs_frame.hip"
#include "fs_aux_kernels.hip"
