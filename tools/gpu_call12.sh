#!/bin/bash
set -o pipefail
out=gpurun_out; mkdir -p $out; export TMPDIR=/tmp
bash tools/ab_builds.sh p1 m1 base 2>&1 | tee $out/r02_ab12.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
