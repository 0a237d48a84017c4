#!/bin/bash
# Build an experimental libfrequensee.so into tools/tmp/<name>/ (git-ignored; travels to the GPU box):
#   bash tools/build_variant.sh n1 "-DFS_CHILD_ORDER=1"
# then on the GPU box: bash tools/ab_builds.sh n1 ...
set -e
name=$1; shift
mkdir -p tools/tmp/$name
cd audio-pathtracer_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -munsafe-fp-atomics --offload-arch=gfx950 \
  -Wall -Wextra -Wno-unused-parameter "$@" -shared -o ../../tools/tmp/$name/libfrequensee.so -x hip \
  fs_capi.cpp fs_bvh.cpp fs_kernels.hip fs_fft.hip fs_refit.hip fs_build.hip
echo built tools/tmp/$name
