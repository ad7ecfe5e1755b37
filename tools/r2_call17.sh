#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for m in 1 2 4; do
  echo "== SG2IM_WGRAD_SIDE_SMALL=$m"
  SG2IM_WGRAD_SIDE_SMALL=$m timeout 300 python -X faulthandler bench.py --steps 8 --warmup 2 --no_roofline --cpu_baseline_steps 0 2>&1 | grep -v amdgpu.ids | grep -v "^Extension modules" | tail -30
done > gpurun_out/c17_modes.log 2>&1
tail -100 gpurun_out/c17_modes.log
