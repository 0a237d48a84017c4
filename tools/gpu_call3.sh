#!/bin/bash
set -o pipefail
out=gpurun_out; mkdir -p $out; export TMPDIR=/tmp
hipcc -O3 --offload-arch=gfx950 -o /tmp/valu_issue_bench tools/valu_issue_bench.hip 2>/dev/null && timeout -k 5 120 /tmp/valu_issue_bench > $out/r02_valu_issue.jsonl 2>&1
grep -i "cndmask\|v_cmp\|v_fma_f32\|v_mov" $out/r02_valu_issue.jsonl
