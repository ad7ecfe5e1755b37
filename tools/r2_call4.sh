#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/clock_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c4_clock.log
timeout 600 python tools/gemm_skeleton.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c4_skeleton.log
