#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
for b in 1 0; do
  SG2IM_GCN_PERSIST_BWD=$b SG2IM_MARKS=1 timeout 600 python bench.py --steps 40 --warmup 10 --cpu_baseline_steps 0 --no_roofline > gpurun_out/r4_call5_bench_bwd$b.json 2> gpurun_out/r4_call5_bench_bwd$b.err
  python -c "
import json; d=json.loads(open('gpurun_out/r4_call5_bench_bwd$b.json').read().strip().splitlines()[-1]); print('persistent backward=$b', d['ms_per_step'], d['value'], d['roofline'], d['host_issue_ms_per_step'])"
  grep -h "gcn-stamps\|g_bwd_done\|wgrad_lane_done\|crn_bwd_done\|adam_done\|gcn_layers_done" gpurun_out/r4_call5_bench_bwd$b.err
done
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r4_call5_pytest_gpu.log 2>&1
tail -n 15 gpurun_out/r4_call5_pytest_gpu.log
