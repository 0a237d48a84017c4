// fs_frame_wide_ext.hip — the fused frame kernel (fs_frame.hip) whose walk parts ignore the actor they start from
// (AddIgnoredActor, AudioRayTracingSubsystem.cpp:322-327), wide flavour (no register limit, worst-case LDS stack rows).
#define FS_FRAME_WIDE 1
#define FS_FRAME_EXT 1
#include "fs_frame.hip"
