#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/grad_parity.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -k "bf16_training_step" > gpurun_out/r4_call6_bf16.log 2>&1
tail -n 8 gpurun_out/r4_call6_bf16.log; cat gpurun_out/grad_parity.log
