#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
for e in 0 1 0 1; do
  SG2IM_D_EARLY=$e SG2IM_MARKS=1 timeout 600 python bench.py --steps 40 --warmup 10 --cpu_baseline_steps 0 --no_roofline > gpurun_out/r4_call7_early$e.json 2> gpurun_out/r4_call7_early$e.err
  python -c "
import json; d=json.loads(open('gpurun_out/r4_call7_early$e.json').read().strip().splitlines()[-1]); print('D early=$e', d['ms_per_step'], d['value'])"
  grep -h "g_fwd_done\|g_losses_done\|crn_bwd_start\|crn_bwd_done\|d_img_start\|d_obj_done\|wgrad_lane_done\|adam_done" gpurun_out/r4_call7_early$e.err | tr '\n' ' '; echo
done
