#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/grad_parity.log
timeout 3000 python -m pytest tests -m gpu -q > gpurun_out/r4_pytest_gpu.log 2>&1
tail -n 8 gpurun_out/r4_pytest_gpu.log
