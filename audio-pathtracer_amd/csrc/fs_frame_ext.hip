// fs_frame_ext.hip — the fused frame kernel (fs_frame.hip) whose walk parts ignore the actor they start from
// (AddIgnoredActor, AudioRayTracingSubsystem.cpp:322-327), narrow flavour.
#define FS_FRAME_EXT 1
#include "fs_frame.hip"
