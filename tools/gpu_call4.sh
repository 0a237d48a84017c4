#!/bin/bash
set -o pipefail
out=gpurun_out; mkdir -p $out; export TMPDIR=/tmp
AB_TEST=1 bash tools/ab_builds.sh e64 2>&1 | tee $out/r02_ab4.log
