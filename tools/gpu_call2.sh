#!/bin/bash
set -o pipefail
out=gpurun_out; mkdir -p $out; export TMPDIR=/tmp
hipcc -O3 --offload-arch=gfx950 -o /tmp/valu_issue_bench tools/valu_issue_bench.hip 2>/dev/null && timeout -k 5 120 /tmp/valu_issue_bench > $out/r02_valu_issue.jsonl 2>&1
cat $out/r02_valu_issue.jsonl
echo "== A/B"
AB_TEST=1 bash tools/ab_builds.sh m1 n1 a2 a1 m1a2 m1n1a2 2>&1 | tee $out/r02_ab2.log
