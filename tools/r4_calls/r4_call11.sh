#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
for e in 0 1 0 1; do
  SG2IM_LATE_HEADS=$e SG2IM_MARKS=1 timeout 600 python bench.py --steps 40 --warmup 10 --cpu_baseline_steps 0 --no_roofline 2> gpurun_out/r4_call11_late$e.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('late heads=$e', d['ms_per_step'], d['value'])"
  grep -h "crn_fwd_start\|g_fwd_done\|g_losses_done\|crn_bwd_start\|crn_bwd_done\|adam_done" gpurun_out/r4_call11_late$e.err | tr '\n' ' '; echo
done
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "trainer_two_steps or graph_replay_matches or bit_reproducible or vg_style" 2>&1 | tail -3
