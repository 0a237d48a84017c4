#!/bin/bash
set -o pipefail
out=gpurun_out; mkdir -p $out; export TMPDIR=/tmp
AB_TEST=1 bash tools/ab_builds.sh p1 t96 t256 p1t96 2>&1 | tee $out/r02_ab5.log
