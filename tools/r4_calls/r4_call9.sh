#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "level_gradients or trainer_two_steps or padded_batch_step or other_baseline or layout" > gpurun_out/r4_call9_tests.log 2>&1
tail -n 6 gpurun_out/r4_call9_tests.log
for i in 1 2; do
SG2IM_MARKS=1 timeout 600 python bench.py --steps 40 --warmup 10 --cpu_baseline_steps 0 --no_roofline 2>gpurun_out/r4_call9_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lazy layout grad', d['ms_per_step'], d['value'])"
grep -h "crn_bwd_done\|g_bwd_done\|wgrad_lane_done\|adam_done" gpurun_out/r4_call9_bench.err | tr '\n' ' '; echo
done
