#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 python tools/conv_v2.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c10_conv_v2.log
