#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r4_pytest_gpu_final.log 2>&1
echo "rc=$?" >> gpurun_out/r4_pytest_gpu_final.log
grep -v "^  File" gpurun_out/r4_pytest_gpu_final.log | tail -8
timeout 300 python __graft_entry__.py smoke > gpurun_out/r4_smoke_final.log 2>&1; tail -2 gpurun_out/r4_smoke_final.log
python bench.py --steps 40 --warmup 10 --cpu_baseline_steps 0 --no_roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d['value'])"
