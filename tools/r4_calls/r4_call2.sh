#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/gcn_stack_probe.py > gpurun_out/r4_gcn_probe.log 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "stack_persistent" > gpurun_out/r4_call2_tests.log 2>&1
cat gpurun_out/r4_gcn_probe.log; tail -n 12 gpurun_out/r4_call2_tests.log
