#!/bin/bash
set -o pipefail
out=gpurun_out; mkdir -p $out; export TMPDIR=/tmp
for rows in 29 21; do
  echo "== FS_UNSAFE_STACK_ROWS=$rows"
  FS_UNSAFE_STACK_ROWS=$rows AB_TEST=1 bash tools/ab_builds.sh x0 xm xmp 2>&1 | tee -a $out/r02_ab10.log
done
