// fs_frame_wide.hip — the second flavour of the fused frame kernel (see the top of fs_frame.hip): no register limit, the
// tree's worst-case stack rows in LDS, no deep-store logic.  launch_frame (fs_frame.hip) picks it for small launches.
#define FS_FRAME_WIDE 1
#include "fs_frame.hip"
