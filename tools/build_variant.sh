#!/bin/bash
# Build an experimental libfrequensee.so into tools/tmp/<name>/ (git-ignored; travels to the GPU box):
#   bash tools/build_variant.sh n1 "-DFS_CHILD_ORDER=1"
# then on the GPU box: bash tools/ab_builds.sh n1 ...
# Diagnostic builds whose device-side debug symbols are shared by all kernels (-DFS_WAVE_TIMELINE, -DFS_TRAV_STATS) are
# compiled as ONE unit (fs_kernels_all.hip); everything else goes through the Makefile (objects side by side).
set -e
name=$1; shift
mkdir -p tools/tmp/$name
flags="$*"
if [[ "$flags" == *FS_WAVE_TIMELINE* || "$flags" == *FS_TRAV_STATS* ]]; then
  cd audio-pathtracer_amd/csrc
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -munsafe-fp-atomics --offload-arch=gfx950 \
    -Wall -Wextra -Wno-unused-parameter "$@" -shared -o ../../tools/tmp/$name/libfrequensee.so -x hip \
    fs_capi_context.cpp fs_capi_scene.cpp fs_capi_frame.cpp fs_capi_pipeline.cpp fs_capi_publish.cpp fs_capi_ir.cpp fs_capi_comm.cpp fs_capi_aux.cpp fs_bvh.cpp \
    fs_kernels_all.hip fs_frame_wide.hip fs_frame_ext.hip fs_frame_wide_ext.hip fs_fft.hip fs_refit.hip fs_build.hip fs_oneshot.hip
else
  make -s -C audio-pathtracer_amd/csrc -j8 lib OBJ=build_$name OUT=../../tools/tmp/$name EXTRA="$flags"
fi
echo built tools/tmp/$name
