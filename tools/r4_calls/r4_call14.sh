#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
for b in 0 1; do
  SG2IM_GCN_PERSIST_BWD=$b SG2IM_MARKS=1 timeout 600 python bench.py --dtype bf16 --steps 40 --warmup 10 --cpu_baseline_steps 0 --no_roofline 2> gpurun_out/r4_call14_bf16_bwd$b.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 persistent backward=$b', d['ms_per_step'], d['value'], 'host issue', d['host_issue_ms_per_step'])"
  grep -h "mark\|gcn-stamps" gpurun_out/r4_call14_bf16_bwd$b.err | awk '{printf "%s %s | ", $2, $3} END{print ""}'
done
