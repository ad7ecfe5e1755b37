#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r4_pytest_gpu_final.log 2>&1
echo "rc=$?" >> gpurun_out/r4_pytest_gpu_final.log
grep -v "^  File" gpurun_out/r4_pytest_gpu_final.log | tail -12
timeout 600 python __graft_entry__.py smoke > gpurun_out/r4_smoke_final.log 2>&1; tail -3 gpurun_out/r4_smoke_final.log
