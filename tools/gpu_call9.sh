#!/bin/bash
set -o pipefail
out=gpurun_out; mkdir -p $out; export TMPDIR=/tmp
bash tools/ab_builds.sh dv20 dv40 ds20 base 2>&1 | tee -a $out/r02_ab9.log
