#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/bench_layout.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_layout_kernels.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "level_gradients or layout" 2>&1 | tail -3
