#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python tools/gemm_skeleton3.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c5_skeleton3.log
