#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4_call36.log
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "persistent_launch" > $L 2>&1
grep -v "^  File" $L | tail -60
