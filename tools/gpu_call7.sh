#!/bin/bash
set -o pipefail
out=gpurun_out; mkdir -p $out; export TMPDIR=/tmp
bash tools/ab_builds.sh x0 xm xmp p1 base 2>&1 | tee -a $out/r02_ab7.log
